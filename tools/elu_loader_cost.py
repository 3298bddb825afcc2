#!/usr/bin/env python
"""What the ELU in the B-operand loaders costs: the same launches with in_elu on / off (timing only; used to decide whether
producers should emit the activated tensor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops
dev = torch.device("cuda:0")
B = 256


def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for K, M, T in [(768, 768, 600), (384, 384, 3000), (256, 256, 3000), (512, 512, 600)]:
    x = torch.randn(B, K, T, device=dev)
    w = torch.randn(K, M, device=dev) / K ** 0.5
    dw = torch.randn(M, 5, device=dev); db = torch.randn(M, device=dev)
    a = timeit(lambda: ops.dws_conv(x, w, dw, db, in_scale=0.9, in_elu=True, out_elu=True))
    b = timeit(lambda: ops.dws_conv(x, w, dw, db, in_scale=1.0, in_elu=False, out_elu=True))
    print(f"dws K{K} M{M} T{T}: ELU loader {a:.3f} ms, plain loader {b:.3f} ms  ({(a - b) / a * 100:.1f} %)")
for K, M, Tin, r in [(1536, 768, 75, 8), (768, 384, 600, 5), (384, 192, 3000, 4), (192, 96, 12000, 2)]:
    x = torch.randn(B, K, Tin, device=dev)
    tr = torch.randn(K, 2 * r, device=dev)
    w = torch.randn(K, M, device=dev) / K ** 0.5
    bias = torch.randn(M, device=dev)
    taps = ops.up_conv_taps(tr, r)
    a = timeit(lambda: ops.up_conv(x, tr, w, bias, r, in_scale=0.9, in_elu=True, taps=taps))
    b = timeit(lambda: ops.up_conv(x, tr, w, bias, r, in_scale=1.0, in_elu=False, taps=taps))
    print(f"up K{K} M{M} Tin{Tin} r{r}: ELU loader {a:.3f} ms, plain loader {b:.3f} ms  ({(a - b) / a * 100:.1f} %)")
