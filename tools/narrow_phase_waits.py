#!/usr/bin/env python
"""Per-WAVE wait accounting of the narrow offline stage launches (C = 96 + conv_post, C = 64 + conv_pre / SpecBlock; optionally others):
every wave stamps every barrier of every tile (stamped copy of the library, -DHILC_DEBUG_WAVE_STAMPS: s_memtime before its
`s_waitcnt lgkmcnt(0)`, after it, and after `s_barrier`), so a phase's time splits into
   issue   = from the previous barrier's exit to the end of the wave's own instruction issue (VALU / MFMA / address arithmetic, incl.
             every stall inside the phase: waiting for the shared vector / matrix issue port, for operands of its own loads),
   lds     = waiting for its own outstanding LDS operations (s_waitcnt lgkmcnt(0)),
   barrier = waiting for the workgroup's other waves (s_barrier).
Reported per phase as the mean over all tiles and waves of the launch (cycles = s_memtime ticks = shader cycles), for all workgroups
and per dispatch class (blockIdx / CUs: co-resident workgroups are not served equally), next to what the phase's instruction count
would need alone.
  python tools/narrow_phase_waits.py [--widths 96,64] > gpurun_out/narrow_phase_waits.txt"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = os.path.join(ROOT, "gpurun_out", "libhilcodec_amd_wavestamps.so")
os.makedirs(os.path.dirname(DBG), exist_ok=True)
import __graft_entry__ as G
G.compile_library(DBG, defines=("HILC_DEBUG_STAMPS", "HILC_DEBUG_WAVE_STAMPS"), only=("resblock.hip", "resblock_chain.hip"))
os.environ["HILC_LIB"] = DBG
import torch
import hilcodec_amd
from hilcodec_amd import ops, synth
from hilcodec_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--widths", default="96,64")
args = ap.parse_args()
want = tuple("C%s " % w for w in args.widths.split(","))
dev = torch.device("cuda:0")
mk = synth.model_kwargs("hil_speech")
model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
model.load_state_dict(synth.synth_state_dict("hil_speech", 7), strict=False)
for l in model.quantizer.layers:
    l.initted = True
x = synth.synth_clips(args.batch, 24000).to(dev)
MAXB = 48
WAVES = {64: 4, 96: 4, 128: 8, 192: 8}
TILES = {64: 188, 96: 188, 128: 94, 192: 94}          # 128-column tiles per clip at the stage's T
NAMES = {
    96: ["U build h0", "U GEMM h0", "U build h1", "U GEMM h1", "U acc->tile", "U +bias->x"]
        + [f"b{b} {p}" for b in range(3) for p in ("P0 ELU->tile", "P1 GEMM1", "P2 acc->tile", "P3 dw1+ELU", "P4 GEMM2", "P5 acc->tile", "P6 dw2+res")]
        + ["Q taps->PR"],
    64: ["S segment", "S DFT+logmag", "S 1x1 GEMM", "S acc->tile", "S +conv_pre->x"]
        + [f"b{b} {p}" for b in range(2) for p in ("P0 ELU->tile", "P1 GEMM1", "P2 acc->tile", "P3 dw1+ELU", "P4 GEMM2", "P5 acc->tile", "P6 dw2+res")]
        + ["D0 ELU->tile", "D GEMM x2", "D acc->tile h0", "D conv h0", "D acc->tile h1", "D conv h1"],
    192: ["U build h0", "U GEMM h0", "U build h1", "U GEMM h1", "U acc->tile", "U +bias->x"]
         + [f"b{b} {p}" for b in range(3) for p in ("P0 ELU->tile", "P1 GEMM1", "P2 acc->tile", "P3 dw1+ELU", "P4 GEMM2", "P5 acc->tile", "P6 dw2+res")],
    128: [f"b{b} {p}" for b in range(2) for p in ("P0 ELU->tile", "P1 GEMM1", "P2 acc->tile", "P3 dw1+ELU", "P4 GEMM2", "P5 acc->tile", "P6 dw2+res")]
         + ["D0 ELU->tile", "D GEMM x2", "D acc->tile h0", "D conv h0", "D acc->tile h1", "D conv h1"],
}
buf = None
results = []
set_one, set_chain = lib.hilc_debug_set_stamp_buffer, lib.hilc_debug_set_chain_stamp_buffer


class Stamped(ops._timed):
    def __enter__(self):
        global buf
        self.on = self.kind == "resblock" and self.tag.startswith(want)
        if self.on:
            self.C = int(self.tag.split()[0][1:])
            self.nw, tiles = WAVES[self.C], TILES[self.C] * args.batch
            buf = torch.zeros(tiles * self.nw * MAXB * 3, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            p = ctypes.c_void_p(buf.data_ptr())
            set_one(p); set_chain(p)
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        global buf
        if self.on:
            self.e1.record()
            torch.cuda.synchronize()
            set_one(None); set_chain(None)
            ms = self.e0.elapsed_time(self.e1)
            b = buf.view(-1, self.nw, MAXB, 3)
            meta = b[:, :, MAXB - 1]                                     # (blockIdx, barriers, tile start)
            nbar = int(meta[:, 0, 1].max())
            live = (meta[:, :, 1] == nbar).all(dim=1) & (b[:, :, 0, 0] > 0).all(dim=1)
            bb, mm = b[live][:, :, :nbar].double(), meta[live].double()
            start = torch.cat([mm[:, :, 2:3], bb[:, :, :-1, 2]], dim=2)      # phase start = tile start / previous barrier's exit
            issue, ldsw, barw = bb[..., 0] - start, bb[..., 1] - bb[..., 0], bb[..., 2] - bb[..., 1]
            blk = mm[:, 0, 0]
            results.append((self.tag, self.C, ms, int(live.sum()), nbar, issue, ldsw, barw, blk, self.work / ms / 1e9))
            buf = None
        return False


ops._timed = Stamped
with torch.no_grad():
    for it in range(2):
        results.clear()
        z = model.encoder(x); q, _, _, idx = model.quantizer(z, None, return_indices=True); model.decoder(q)
print("# Per-wave barrier stamps of the narrow offline stage launches (stamped library: three s_memtime per barrier and wave, ~+10 - 15 % per tile).")
print("# issue = previous barrier's exit -> end of the wave's own issue; lds = its s_waitcnt lgkmcnt(0); barrier = s_barrier.  Mean cycles per tile and wave.")
for tag, C, ms, n, nbar, issue, ldsw, barw, blk, tf in results:
    names = NAMES.get(C, [])
    names = names + [f"phase {k}" for k in range(len(names), nbar)]
    ncls = int(blk.max().item()) // 256 + 1
    print(f"\n{tag}: {ms:.3f} ms ({tf:.1f} TF as stamped), {n} tiles, {nbar} barriers per tile, {ncls} workgroup(s) per CU")
    tot = (issue + ldsw + barw).sum(dim=2).mean().item()
    hdr = f"{'phase':18s} {'issue':>8s} {'lds':>7s} {'barrier':>8s} {'total':>8s} {'% tile':>7s}"
    for c in range(ncls):
        hdr += f" | class {c}: {'issue':>7s} {'lds':>6s} {'barrier':>7s}"
    print(hdr)
    sums = [0.0, 0.0, 0.0]
    groups = {}
    for k in range(nbar):
        i, l, w = issue[:, :, k].mean().item(), ldsw[:, :, k].mean().item(), barw[:, :, k].mean().item()
        sums = [sums[0] + i, sums[1] + l, sums[2] + w]
        line = f"{names[k]:18s} {i:8.0f} {l:7.0f} {w:8.0f} {i + l + w:8.0f} {100 * (i + l + w) / tot:6.1f}%"
        for c in range(ncls):
            sel = (blk >= 256 * c) & (blk < 256 * (c + 1))
            line += f" |          {issue[sel][:, :, k].mean().item():7.0f} {ldsw[sel][:, :, k].mean().item():6.0f} {barw[sel][:, :, k].mean().item():7.0f}"
        print(line)
        key = "GEMM" if "GEMM" in names[k] else "element-wise"
        g = groups.setdefault(key, [0.0, 0.0, 0.0])
        g[0] += i; g[1] += l; g[2] += w
    print(f"{'sum':18s} {sums[0]:8.0f} {sums[1]:7.0f} {sums[2]:8.0f} {sum(sums):8.0f}   (tile {tot:.0f} cycles per wave: issue {100 * sums[0] / tot:.1f} %, own LDS {100 * sums[1] / tot:.1f} %, barrier {100 * sums[2] / tot:.1f} %)")
    for key, g in groups.items():
        print(f"   {key:13s}: issue {g[0]:8.0f}  lds {g[1]:7.0f}  barrier {g[2]:8.0f}  = {100 * sum(g) / tot:.1f} % of the tile")
    for c in range(ncls):
        sel = (blk >= 256 * c) & (blk < 256 * (c + 1))
        t = (issue[sel] + ldsw[sel] + barw[sel]).sum(dim=2).mean().item()
        print(f"   dispatch class {c}: {int(sel.sum())} tiles, {t:.0f} cycles per tile (issue {issue[sel].sum(dim=2).mean().item():.0f}, lds {ldsw[sel].sum(dim=2).mean().item():.0f}, barrier {barw[sel].sum(dim=2).mean().item():.0f})")
    # spread between the waves of a workgroup at each barrier = what the barrier wait is made of
    slow = (issue + ldsw).max(dim=1).values - (issue + ldsw).min(dim=1).values
    print(f"   slowest - fastest wave of a workgroup, per phase, mean over phases: {slow.mean().item():.0f} cycles (max phase mean {slow.mean(dim=0).max().item():.0f})")
