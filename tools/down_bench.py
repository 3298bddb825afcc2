#!/usr/bin/env python
"""Isolated timing of the encoder's down-sampling layers (pointwise conv + strided depthwise conv, hilc_dws_conv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops
dev = torch.device("cuda:0")
B = 256
for K, M, T, r in [(64, 128, 24000, 2), (128, 256, 12000, 4), (256, 512, 3000, 5), (512, 1024, 600, 8)]:
    x = torch.randn(B, K, T, device=dev); wt = torch.randn(K, M, device=dev) / K ** 0.5
    dw = torch.randn(M, 2 * r, device=dev) * 0.3; db = torch.randn(M, device=dev) * 0.1
    fn = lambda: ops.dws_conv(x, wt, dw, db, stride=r, in_scale=0.9, in_elu=True)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); ms = ts[len(ts) // 2]
    print(f"down K{K} M{M} T{T} k{2 * r} s{r}: {ms:6.3f} ms  {2.0 * B * T * K * M / ms / 1e9:6.1f} TF")
