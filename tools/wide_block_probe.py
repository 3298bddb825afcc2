#!/usr/bin/env python
"""The wide residual blocks of the OFFLINE step (C = 256 / 512 encoder, 768 / 384 decoder) — per stage: two hilc_dws_conv launches per
block (rounds 1-3), one carry-form launch per block (hilc_resblock, round 4), one launch per stage (hilc_resblock_chain, where the
carry slots fit LDS).  Same arithmetic (bit-identical, asserted), time per stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B = int(os.environ.get("B", "256"))
for C, T, n in ((256, 3000, 2), (512, 600, 2), (768, 600, 3), (384, 3000, 3)):
    x = torch.randn(B, C, T, generator=g).to(dev)
    blocks = []
    for j in range(n):
        w1, w2 = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev), (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
        d1, b1 = (torch.randn(C, 5, generator=g) * 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
        d2, b2 = (torch.randn(C, 5, generator=g) * 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
        blocks.append((w1, d1, b1, w2, d2, b2, ops.resblock_pack(w1), ops.resblock_pack(w2), (1 + j / 3) ** -0.5, 0.4))

    def two():
        y = x
        for w1, d1, b1, w2, d2, b2, _, _, pre, post in blocks:
            h = ops.dws_conv(y, w1, d1, b1, in_scale=pre, in_elu=True, out_elu=True)
            y = ops.dws_conv(h, w2, d2, b2, res=y, out_scale=post)
        return y

    def one():
        y = x
        for w1, d1, b1, w2, d2, b2, p1, p2, pre, post in blocks:
            y = ops.resblock(y, p1, d1, b1, p2, d2, b2, pre, post)
        return y

    def chain():
        return ops.resblock_chain(x, [(p1, d1, b1, p2, d2, b2, pre, post) for _, d1, b1, _, d2, b2, p1, p2, pre, post in blocks])

    forms = [("two dws_conv launches per block", two), ("one launch per block", one)]
    if ops.resblock_chain_supported(C, T, n, B, streaming=False):
        forms.append(("one launch per stage", chain))
    ref = two()
    for name, fn in forms[1:]:
        assert torch.equal(ref, fn()), name
    for name, fn in forms + forms:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"C={C} T={T} B={B} x{n}  {name:34s} {ms:7.3f} ms  {n * 4.0 * B * T * C * C / ms / 1e9:6.1f} TF", flush=True)

# the widest decoder stage: up-sampling layer (1536 -> 768, r = 8) + first block: two launches / one (hilc_decoder_stage, offline)
C, r, Tin = 768, 8, 75
xin = torch.randn(B, 2 * C, Tin, generator=g).to(dev)
tw = (torch.randn(2 * C, 2 * r, generator=g) * 0.3).to(dev)
wu = (torch.randn(2 * C, C, generator=g) / (2 * C) ** 0.5).to(dev)
bu = (torch.randn(C, generator=g) * 0.1).to(dev)
w1, w2 = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev), (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
d1, b1 = (torch.randn(C, 5, generator=g) * 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
d2, b2 = (torch.randn(C, 5, generator=g) * 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
p1, p2 = ops.resblock_pack(w1), ops.resblock_pack(w2)
up = (tw, ops.resblock_chain_pack(wu[:C].contiguous(), False), ops.resblock_chain_pack(wu[C:].contiguous(), False), bu, 0.7071, r)


def sep():
    return ops.resblock(ops.up_conv(xin, tw, wu, bu, r, in_scale=0.7071, in_elu=True), p1, d1, b1, p2, d2, b2, 1.0, 0.4)


def fused():
    return ops.decoder_stage(xin, up, [(p1, d1, b1, p2, d2, b2, 1.0, 0.4)])


assert torch.equal(sep(), fused())
for name, fn in (("up_conv + block", sep), ("decoder_stage (up + block)", fused)) * 2:
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"C=768 r=8 Tin=75 B={B}  {name:34s} {ms:7.3f} ms  {(4.0 + 4.0) * B * Tin * r * C * C / ms / 1e9:6.1f} TF", flush=True)
