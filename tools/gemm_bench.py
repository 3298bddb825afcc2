#!/usr/bin/env python
"""Micro-benchmark of the GEMM-core kernels on isolated layer shapes (HIP events, median of reps)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hilcodec_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--only", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
B = args.batch
shapes = [  # (K, M, T)
    (64, 64, 24000), (96, 96, 24000), (128, 128, 12000), (192, 192, 12000), (256, 256, 3000), (384, 384, 3000),
    (512, 512, 600), (768, 768, 600), (1536, 768, 600), (768, 384, 3000), (384, 192, 12000), (192, 96, 24000),
]
if args.only:
    want = [tuple(int(v) for v in s.split("x")) for s in args.only.split(",")]
    shapes = [s for s in shapes if s in want]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


print(f"{'K':>5} {'M':>5} {'T':>6} | {'pw':>8} {'pw+elu':>8} {'dws':>8} {'dws+elu':>8}  (TFLOP/s, fp32 MFMA peak 157.3)")
for K, M, T in shapes:
    x = torch.randn(B, K, T, device=dev)
    wt = torch.randn(K, M, device=dev) / K ** 0.5
    dw = torch.randn(M, 5, device=dev)
    db = torch.randn(M, device=dev)
    y = torch.empty(B, M, T, device=dev)
    fl = 2.0 * B * T * K * M
    r = []
    r.append(fl / timeit(lambda: ops.pw_conv(x, wt)) / 1e12)
    r.append(fl / timeit(lambda: ops.pw_conv(x, wt, in_scale=0.9, in_elu=True)) / 1e12)
    r.append(fl / timeit(lambda: ops.dws_conv(x, wt, dw, db)) / 1e12)
    r.append(fl / timeit(lambda: ops.dws_conv(x, wt, dw, db, in_scale=0.9, in_elu=True, out_elu=True)) / 1e12)
    print(f"{K:5d} {M:5d} {T:6d} | " + " ".join(f"{v:8.1f}" for v in r))
    del x, y
