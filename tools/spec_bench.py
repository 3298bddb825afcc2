#!/usr/bin/env python
"""Isolated timing of the one-launch SpecBlocks (n_fft 64 with the first conv, 128, 256) at the offline shapes."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--only", type=int, default=0)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, T = args.batch, 24000
wav = (synth.synth_clips(B, T, seed=1234)).to(dev)
for n_fft, hop in [(64, 1), (128, 2), (256, 8)]:
    if args.only and n_fft != args.only:
        continue
    C, Tf = n_fft, T // hop
    basis = synth.stft_basis(n_fft)[:, 0, :]              # [n_fft+2, n_fft]: cos rows, then sin rows
    nb = n_fft // 2 + 1
    m_pad = (n_fft + 2 + 31) // 32 * 32
    bt = torch.zeros(n_fft, m_pad)
    bt[:, 0:n_fft + 2:2] = basis[:nb].t()                 # hilc_stft_logmag's layout: (cos_k, sin_k) interleaved
    bt[:, 1:n_fft + 2:2] = basis[nb:].t()
    wt = torch.randn(n_fft // 2 + 1, C) / (n_fft // 2 + 1) ** 0.5
    dft, nyq, pw = ops.spec_block_tables(bt.to(dev), wt.to(dev), n_fft)
    x = torch.randn(B, C, Tf, device=dev)
    pre_w, pre_b = torch.randn(64, 5, device=dev), torch.randn(64, device=dev)

    def run():
        if n_fft == 64:
            return ops.spec_block_conv_pre(wav, dft, nyq, pw, None, pre_w, pre_b, 8.9, 64, 1, -4.5, 2.8, True, 0.5)
        return ops.spec_block(wav, dft, nyq, pw, None, x, n_fft, hop, -4.3, 2.8, True, 0.5)

    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    fl = 2.0 * B * Tf * (n_fft * (n_fft + 2) + (n_fft // 2 + 1) * C)
    print(f"spec_block N{n_fft} hop{hop}: {ts[len(ts) // 2]:.3f} ms ({fl / ts[len(ts) // 2] / 1e9:.1f} TF)")
