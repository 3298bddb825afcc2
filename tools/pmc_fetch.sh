#!/bin/bash
# HBM-side bytes per launch of every kernel of the offline step (PMC passes only; FETCH_SIZE and WRITE_SIZE in separate passes).
#   [HILC_LIB=...] bash tools/pmc_fetch.sh <tag>
# FETCH_SIZE x 1024 x 2 (the guide's gfx950 correction for wide coalesced reads), WRITE_SIZE x 1024.
TAG=${1:-pmc_fetch}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/raw.csv
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --output-format csv -d $O/p_$CNT -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-clock-probe \
    --no-launch-timing --no-other-configs > $O/p_$CNT.log 2>&1
  f=$(find $O/p_$CNT -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cat "$f" >> $O/raw.csv
  rm -rf $O/p_$CNT
done
python - $O/raw.csv <<'PY' > $O/hbm_per_kernel.txt
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
hdr = None
for r in csv.reader(open(sys.argv[1])):
    if "Kernel_Name" in r:
        hdr = r; continue
    if hdr is None or len(r) != len(hdr): continue
    d = dict(zip(hdr, r))
    k = d["Kernel_Name"].replace("(anonymous namespace)::", "").replace("hilc::", "")
    m = re.search(r"(\w+_kernel)(<[^(]*>)?\(", k)
    k = (m.group(1) + (m.group(2) or "")).replace(", ", ",") if m else k[:60]
    agg[k][d["Counter_Name"]] += float(d["Counter_Value"]); n[(k, d["Counter_Name"])] += 1
tot_r = tot_w = 0.0
print(f"{'kernel':64s} {'launches/step':>13s} {'read GB/launch':>15s} {'write GB/launch':>16s}")
for k in sorted(agg, key=lambda k: -(agg[k].get('FETCH_SIZE', 0) * 2 + agg[k].get('WRITE_SIZE', 0))):
    l = n[(k, "FETCH_SIZE")] or 1
    rd = agg[k].get("FETCH_SIZE", 0) * 1024 * 2 / l / 1e9; wr = agg[k].get("WRITE_SIZE", 0) * 1024 / (n[(k, "WRITE_SIZE")] or 1) / 1e9
    tot_r += rd * l / 2; tot_w += wr * l / 2          # 2 steps (1 warm-up + 1 timed) were profiled
    print(f"{k[:64]:64s} {l / 2:13.1f} {rd:15.3f} {wr:16.3f}")
print(f"step total: read {tot_r:.2f} GB, written {tot_w:.2f} GB")
PY
cat $O/hbm_per_kernel.txt | head -30
