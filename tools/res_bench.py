#!/usr/bin/env python
"""Isolated timing of hilc_resblock vs two hilc_dws_conv launches per channel width."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", type=int, default=0)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda:0")
B = args.batch
for C, T in [(64, 24000), (96, 24000), (128, 12000), (192, 12000)]:
    if args.only and C != args.only:
        continue
    x = torch.randn(B, C, T, device=dev)
    w1 = torch.randn(C, C, device=dev) / C ** 0.5
    w2 = torch.randn(C, C, device=dev) / C ** 0.5
    d1 = torch.randn(C, 5, device=dev); b1 = torch.randn(C, device=dev)
    d2 = torch.randn(C, 5, device=dev); b2 = torch.randn(C, device=dev)
    p1, p2 = ops.resblock_pack(w1), ops.resblock_pack(w2)

    def fused():
        return ops.resblock(x, p1, d1, b1, p2, d2, b2, 0.9, 0.5)
    def two():
        g = ops.dws_conv(x, w1, d1, b1, in_scale=0.9, in_elu=True, out_elu=True)
        return ops.dws_conv(g, w2, d2, b2, res=x, out_scale=0.5)
    res = []
    for fn in (fused, two):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort(); res.append(ts[len(ts) // 2])
    fl = 4.0 * B * T * C * C
    print(f"C={C:4d} T={T:6d}  fused {res[0]:7.3f} ms ({fl / res[0] / 1e9:6.1f} TF)   two-launch {res[1]:7.3f} ms ({fl / res[1] / 1e9:6.1f} TF)")
