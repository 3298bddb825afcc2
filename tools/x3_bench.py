#!/usr/bin/env python
"""Isolated timing of the EXPERIMENTAL bf16x3 layer kernels against the fp32 kernels they mirror (decoder shapes, B = 256)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--only", type=int, default=0, help="only the depthwise-separable layer with this K")
args = ap.parse_args()
dev = torch.device("cuda:0")
B = args.batch


def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for K, M, T in [(768, 768, 600), (384, 384, 3000), (192, 192, 12000)]:
    if args.only and K != args.only:
        continue
    x = torch.randn(B, K, T, device=dev)
    wt = torch.randn(K, M, device=dev) / K ** 0.5
    ws = ops.x3_split(wt)
    dw = torch.randn(M, 5, device=dev) * 0.4; db = torch.randn(M, device=dev) * 0.2
    res = torch.randn(B, M, T, device=dev)
    fl = 2.0 * B * T * K * M
    for name, kw in (("ELU in/out", dict(in_scale=0.9, in_elu=True, out_elu=True)), ("shortcut", dict(res=res, out_scale=0.5)),
                     ("plain", dict())):
        a = timeit(lambda: ops.dws_conv(x, wt, dw, db, **kw))
        b = timeit(lambda: ops.dws_conv_x3(x, ws, dw, db, **kw))
        print(f"dws K{K} M{M} T{T} {name:10s}  fp32 {a:6.3f} ms ({fl / a / 1e9:6.1f} TF)   bf16x3 {b:6.3f} ms ({fl / b / 1e9:6.1f} TF-eq)   x{a / b:.2f}")
for K, M, Tin, r in [(1536, 768, 75, 8), (768, 384, 600, 5), (384, 192, 3000, 4), (192, 96, 12000, 2)]:
    if args.only:
        break
    x = torch.randn(B, K, Tin, device=dev)
    tw = torch.randn(K, 2 * r, device=dev)
    wt = torch.randn(K, M, device=dev) / K ** 0.5
    ws = ops.x3_split(wt)
    bias = torch.randn(M, device=dev) * 0.1
    taps = ops.up_conv_taps(tw, r)
    fl = 2.0 * B * Tin * r * K * M
    a = timeit(lambda: ops.up_conv(x, tw, wt, bias, r, in_scale=0.7, in_elu=True, taps=taps))
    b = timeit(lambda: ops.up_conv_x3(x, tw, ws, bias, r, in_scale=0.7, taps=taps))
    print(f"up  K{K} M{M} Tin{Tin} r{r}            fp32 {a:6.3f} ms ({fl / a / 1e9:6.1f} TF)   bf16x3 {b:6.3f} ms ({fl / b / 1e9:6.1f} TF-eq)   x{a / b:.2f}")
