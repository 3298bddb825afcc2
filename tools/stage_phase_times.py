#!/usr/bin/env python
"""s_memtime stamps of the STAGE launches as they run in the offline step (debug aid, stamped copy of the library).
Per stage launch: tile-time quantiles (stamp 0 = tile start, stamp 7 = end of the LAST block's P6), tiles in flight per CU,
and the last block's phases (stamps 1..7 are overwritten per block: the last block's survive).
  python tools/stage_phase_times.py [--mode streaming] > gpurun_out/stage_phase_times.txt"""
import argparse, ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = os.path.join(ROOT, "gpurun_out", "libhilcodec_amd_stamps.so")
if os.environ.get("HILC_STAMP_LIB"):
    DBG = os.path.abspath(os.environ["HILC_STAMP_LIB"])
else:
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    import __graft_entry__ as G
    G.compile_library(DBG, defines=("HILC_DEBUG_STAMPS",), only=("resblock.hip", "resblock_chain.hip"))
os.environ["HILC_LIB"] = DBG
import torch
import hilcodec_amd
from hilcodec_amd import ops, synth
from hilcodec_amd._lib import lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--model", default="hil_speech")
args = ap.parse_args()
dev = torch.device("cuda:0")
mk = synth.model_kwargs(args.model)
model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
model.load_state_dict(synth.synth_state_dict(args.model, 7), strict=False)
for l in model.quantizer.layers:
    l.initted = True
x = synth.synth_clips(args.batch, 24000).to(dev)
NT = 1 << 20
buf = torch.zeros(NT, 8, dtype=torch.int64, device=dev)
rows = []
set_one = lib.hilc_debug_set_stamp_buffer
set_chain = lib.hilc_debug_set_chain_stamp_buffer


class Stamped(ops._timed):
    def __enter__(self):
        if self.kind == "resblock":
            buf.zero_()
            torch.cuda.synchronize()
            p = ctypes.c_void_p(buf.data_ptr())
            set_one(p); set_chain(p)
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.kind == "resblock":
            self.e1.record()
            torch.cuda.synchronize()
            set_one(None); set_chain(None)
            ms = self.e0.elapsed_time(self.e1)
            live = (buf[:, 0] > 0) & (buf[:, 7] > buf[:, 0])
            b = buf[live]
            tt = (b[:, 7] - b[:, 0]).double()
            qs = torch.quantile(tt[:400000], torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64, device=dev)).tolist()
            d = (b[:, 1:] - b[:, :-1]).double().median(dim=0).values.tolist()
            k = 1.0      # an s_memtime tick is a shader cycle (guide, "s_memtime tick vs SQ PMC units")
            span = ms * 1e-3 * 2.39e9                      # (s_memtime bases differ between XCDs: the launch time at the sustained clock instead)
            rows.append((self.tag, ms, int(live.sum()), tt.mean().item() * k, [q * k for q in qs], tt.sum().item() * k / span / 256, [v * k for v in d], self.work / ms / 1e9))
        return False


ops._timed = Stamped
with torch.no_grad():
    z = model.encoder(x); q, _, _, idx = model.quantizer(z, None, return_indices=True); model.decoder(q)   # warm-up (stamped too: discarded)
    rows.clear()
    z = model.encoder(x); q, _, _, idx = model.quantizer(z, None, return_indices=True); model.decoder(q)
print("# per stage launch of one offline step (stamped library: ~+10 % per tile); cycles = s_memtime ticks")
for tag, ms, n, mean, qs, infl, d, tf in rows:
    print(f"{tag:44s} {ms:7.3f} ms {tf:6.1f} TF  {n:6d} tiles  mean {mean:8.0f} cyc  q5/25/50/75/95 = " + "/".join(f"{q:.0f}" for q in qs)
          + f"  tiles in flight per CU {infl:.2f}")
    print(f"{'':44s} last block: [U + earlier blocks + P0]={d[0]:.0f} G1={d[1]:.0f} P2={d[2]:.0f} P3={d[3]:.0f} G2={d[4]:.0f} P5={d[5]:.0f} P6={d[6]:.0f}")
