#!/usr/bin/env python
"""Why does a streaming stage launch cost more per tile than the offline one?  The C = 192 decoder stage of a hop (1024 streams x 160
samples: 5 tiles per workgroup) launched back to back (weights stay in L2) against the same launch after 200 MB of unrelated traffic
(what the rest of a hop does to L2) — and with 512 / 2048 streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
for C, r, Tin in ((192, 4, 40), (96, 2, 160)):
    n = 3
    blocks = []
    for j in range(n):
        w1, w2 = rnd(C, C) / C ** 0.5, rnd(C, C) / C ** 0.5
        blocks.append((ops.resblock_chain_pack(w1), rnd(C, 5) * 0.5, rnd(C) * 0.2, ops.resblock_chain_pack(w2), rnd(C, 5) * 0.5, rnd(C) * 0.2, 1.0, 0.4))
    tw, wu, bu = rnd(2 * C, 2 * r) * 0.3, rnd(2 * C, C) / (2 * C) ** 0.5, rnd(C) * 0.1
    up = (tw, ops.resblock_chain_pack(wu[:C].contiguous()), ops.resblock_chain_pack(wu[C:].contiguous()), bu, 0.7071, r)
    junk = torch.empty(64 * 1024 * 1024, device=dev)
    for B in (512, 1024, 2048):
        ca = [[rnd(B, C, 4), rnd(B, C, 4)] for _ in range(n)]
        ua = rnd(B, 2 * C, 1)
        xin = rnd(B, 2 * C, Tin)
        run = lambda: ops.decoder_stage(xin, up, blocks, ca, ua)
        for mode in ("back to back", "after 256 MB of other traffic"):
            ts = []
            for _ in range(12):
                if mode != "back to back":
                    junk.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            print(f"C={C} streams={B:5d}  {mode:32s} median {ts[len(ts) // 2] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us", flush=True)
