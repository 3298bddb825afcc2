"""CU-time budget of one streaming hop: see tools/hop_cu_time.sh.
usage: hop_cu_time.py trace.csv raw_counters.csv bench_graph.json"""
import collections
import csv
import json
import re
import sys

N_CU, N_XCD, SIMD_PER_CU = 256, 8, 4
# per-stream columns of a hop (320 samples) at the stage kernels' widths, and the output columns of a tile (resblock_cfg.h: NCOL - HALO)
T_OF_C = {64: 320, 128: 160, 256: 40, 512: 8, 768: 8, 384: 40, 192: 160, 96: 320}


def short(name):
    k = name.replace("(anonymous namespace)::", "").replace("hilc::", "").replace("void ", "")
    m = re.search(r"(\w+_kernel)(<[^(]*>)?\(", k)
    return (m.group(1) + (m.group(2) or "")).replace(", ", ",") if m else k.split("(")[0][:70]


def tiles_per_wg(name, wgs, streams):
    m = re.match(r"resblock_kernel<(\d+),(true|false),(true|false),(\d+),(true|false),(-?\d+)", name)
    if not m or m.group(2) != "true":
        return None
    c, scarry = int(m.group(1)), m.group(3) == "true"
    ncol = 128 if c <= 192 else (32 if (c >= 512 or scarry) else 64)       # (C = 256 / 384 in the carry form: 32-column tiles)
    halo = 0 if (scarry or c >= 512) else 8
    to = ncol - halo
    tiles = -(-streams * T_OF_C[c] // to)
    return tiles / wgs


def main():
    trace, raw, benchf = sys.argv[1:4]
    streams = 1024
    try:
        line = [l for l in open(benchf) if l.startswith("{")][-1]
        bj = json.loads(line)
        streams = int(bj.get("config", {}).get("streams", streams))
        hop_ms = bj.get("ms_per_step")
    except Exception:
        bj, hop_ms = {}, None
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
             int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
            for r in csv.DictReader(open(trace))]
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "spec_block_kernel<64" in r[2] or "resblock_kernel<64" in r[2] and "true>" in r[2]]
    # a hop starts at its first kernel: keep the first of each hop (kernels of that name more than 1 ms apart)
    hop_starts = []
    for i in starts:
        if not hop_starts or rows[i][0] - rows[hop_starts[-1]][0] > 1_000_000:
            hop_starts.append(i)
    hop_starts = hop_starts[-20:]
    nh = len(hop_starts) - 1
    period_us = (rows[hop_starts[-1]][0] - rows[hop_starts[0]][0]) / nh / 1e3
    wall = collections.defaultdict(float); launches = collections.Counter(); geom = {}
    first_seen = {}
    for a, b in zip(hop_starts[:-1], hop_starts[1:]):
        for s, e, k, wgs, wsz, lds, regs in rows[a:b]:
            n = short(k)
            wall[n] += (e - s) / 1e3; launches[n] += 1; geom[n] = (wgs, wsz, lds, regs)
            first_seen.setdefault(n, s - rows[a][0])
    # counters: averaged per launch of a kernel name (eager hop loop, same launches)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    hdr = None
    for r in csv.reader(open(raw)):
        if "Kernel_Name" in r:
            hdr = r; continue
        if hdr is None or len(r) != len(hdr):
            continue
        d = dict(zip(hdr, r))
        n = short(d["Kernel_Name"])
        agg[n][d["Counter_Name"]] += float(d["Counter_Value"]); cnt[(n, d["Counter_Name"])] += 1

    def per(n, c):
        return agg[n].get(c, 0.0) / (cnt[(n, c)] or 1)

    print(f"# CU-time budget of one streaming hop ({streams} streams x 320 samples); graph replay period {period_us:.1f} us"
          + (f" (bench line: {hop_ms:.3f} ms/hop)" if hop_ms else ""))
    print("# wall      = the kernel's duration inside the replayed graph (kernel trace; side-branch kernels overlap the chain)")
    print("# WGs       = workgroups per launch; life = mean wave lifetime (SQ_WAVE_CYCLES x 4 / SQ_WAVES, shader cycles, eager loop with counters)")
    print("# CU-time   = WGs x life / 256 CUs at the hop's clock (life measured alone; 2.4 GHz assumed) — what the launch occupies if CUs were never shared")
    print("# occ       = CU-time / (the kernel's duration in the counter pass): mean workgroups resident per CU while it runs")
    print("# mfma      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8")
    print("# mfma-min  = 64 cycles x MFMA instructions / 1024 SIMDs at 2.4 GHz: the time the launch's matrix work needs on the whole chip")
    print("# issue / stall / parked = shares of wave-cycles issuing, waiting to issue (WAIT_INST_ANY), at waitcnt / barrier (WAIT_ANY)")
    print(f"{'kernel':66s} {'n':>2s} {'wall us':>8s} {'WGs':>5s} {'thr':>4s} {'LDS KB':>6s} {'regs':>4s} {'life kcyc':>9s} {'CU-time us':>10s} {'occ':>5s} {'mfma':>5s} {'mfma-min us':>11s} {'tiles/WG':>8s} {'issue':>5s} {'stall':>5s} {'parked':>6s}")
    tot_wall = tot_cu = tot_min = 0.0
    for n in sorted(wall, key=lambda n: first_seen[n]):
        wgs, wsz, lds, regs = geom[n]
        nl = launches[n] / nh
        w = wall[n] / nh
        wc, wv = per(n, "SQ_WAVE_CYCLES"), per(n, "SQ_WAVES")
        life = wc * 4 / wv if wv else float("nan")
        cu_us = wgs * life / N_CU / 2400.0 * nl
        kcyc = per(n, "GRBM_GUI_ACTIVE") / N_XCD
        occ = (wgs * life / N_CU) / kcyc if kcyc else float("nan")
        mfma = per(n, "SQ_VALU_MFMA_BUSY_CYCLES") / (N_CU * SIMD_PER_CU * kcyc) if kcyc else float("nan")
        mmin = per(n, "SQ_INSTS_MFMA") * 64 / (N_CU * SIMD_PER_CU) / 2400.0 * nl
        tpw = tiles_per_wg(n, wgs, streams)
        issue = per(n, "SQ_ACTIVE_INST_ANY") / wc if wc else float("nan")
        stall = per(n, "SQ_WAIT_INST_ANY") / wc if wc else float("nan")
        parked = per(n, "SQ_WAIT_ANY") / wc if wc else float("nan")
        tot_wall += w; tot_cu += cu_us if cu_us == cu_us else 0.0; tot_min += mmin
        print(f"{n[:66]:66s} {nl:2.0f} {w:8.1f} {wgs:5d} {wsz:4d} {lds / 1024:6.1f} {regs:4d} {life / 1e3:9.1f} {cu_us:10.1f} {occ:5.2f} {mfma:5.2f} {mmin:11.1f} "
              f"{(f'{tpw:8.2f}' if tpw else '       -')} {issue:5.2f} {stall:5.2f} {parked:6.2f}")
    print(f"{'sum':66s}    {tot_wall:8.1f} {'':5s} {'':4s} {'':6s} {'':4s} {'':9s} {tot_cu:10.1f} {'':5s} {'':5s} {tot_min:11.1f}")
    print(f"# period {period_us:.1f} us = {tot_min:.1f} us of matrix work at the peak ({tot_min / period_us:.3f}) + {period_us - tot_min:.1f} us not covered by it;")
    print(f"# the kernels' wall times add up to {tot_wall:.1f} us (side-branch overlap {tot_wall - period_us:.1f} us).")


if __name__ == "__main__":
    main()
