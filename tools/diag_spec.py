import sys, torch
sys.path.insert(0, '.')
from hilcodec_amd import ops, fold, synth
dev = torch.device('cuda:0')
for n_fft, hop in ((64, 1), (128, 2)):
    nb = n_fft // 2 + 1
    bt = fold.stft_basis_layout(synth.stft_basis(n_fft)).to(dev)
    wt = torch.zeros(nb, n_fft, device=dev)
    wt[torch.arange(nb), torch.arange(nb)] = 1.0
    wav = synth.synth_clips(2, 1000 * hop, seed=3).to(dev)
    Tf = 1000
    x = torch.zeros(2, n_fft, Tf, device=dev)
    dft_p, nyq, pw_p = ops.spec_block_tables(bt, wt, n_fft)
    y = ops.spec_block(wav, dft_p, nyq, pw_p, None, x, n_fft, hop, -4.0, 2.8, 2, 1.0)   # plain magnitude
    s = ops.stft_logmag(wav, bt, n_fft, hop, -4.0, 2.8, 2)
    d = (y[:, :nb] - s).abs()
    per_bin = d.amax(dim=(0, 2))
    print(n_fft, 'bins differing:', (per_bin > 0).nonzero().flatten().tolist(), 'max', per_bin.max().item(), 'rel', (d / s.abs().clamp_min(1e-9)).max().item())
    y = ops.spec_block(wav, dft_p, nyq, pw_p, None, x, n_fft, hop, -4.0, 2.8, 1, 1.0)
    s = ops.stft_logmag(wav, bt, n_fft, hop, -4.0, 2.8, 1)
    d = (y[:, :nb] - s).abs()
    per_bin = d.amax(dim=(0, 2))
    print(n_fft, 'normalised: bins differing:', (per_bin > 0).nonzero().flatten().tolist(), 'max', per_bin.max().item())
print("---- dense conv weight")
for n_fft, hop in ((64, 1),):
    nb = n_fft // 2 + 1
    bt = fold.stft_basis_layout(synth.stft_basis(n_fft)).to(dev)
    g = torch.Generator().manual_seed(1)
    wt = (torch.randn(nb, n_fft, generator=g) / nb ** 0.5).to(dev)
    wav = synth.synth_clips(2, 1000 * hop, seed=3).to(dev)
    x = torch.zeros(2, n_fft, 1000, device=dev)
    dft_p, nyq, pw_p = ops.spec_block_tables(bt, wt, n_fft)
    s = ops.stft_logmag(wav, bt, n_fft, hop, -4.0, 2.8, 1)
    y = ops.spec_block(wav, dft_p, nyq, pw_p, None, x, n_fft, hop, -4.0, 2.8, 1, 1.0)
    y2 = ops.pw_conv(s, wt, None, res=x)
    y3 = ops.pw_conv(s, wt, None)
    ref = torch.einsum('km,bkt->bmt', wt.double(), s.double())
    # an explicit fp32 fmaf chain over k = 0..nb-1 on the host
    sc, wc = s.cpu(), wt.cpu()
    chain = torch.zeros(2, n_fft, 1000)
    import numpy as np
    acc = np.zeros((2, n_fft, 1000), dtype=np.float32)
    for k in range(nb):
        prod = (wc[k].numpy()[None, :, None].astype(np.float64) * sc[:, k].numpy()[:, None, :].astype(np.float64)) + acc.astype(np.float64)
        acc = prod.astype(np.float32)       # one rounding per k: fmaf
    chain = torch.from_numpy(acc)
    print('fused vs unfused(res)', (y - y2).abs().max().item(), ' unfused(res) vs unfused(nores)', (y2 - y3).abs().max().item())
    print('fused vs chain', (y.cpu() - chain).abs().max().item(), ' unfused vs chain', (y2.cpu() - chain).abs().max().item(), ' unfused(nores) vs chain', (y3.cpu() - chain).abs().max().item())
    print('fused vs fp64', (y.double() - ref).abs().max().item(), ' unfused vs fp64', (y2.double() - ref).abs().max().item())
print("---- epilogue variants")
bias = (torch.randn(64, generator=torch.Generator().manual_seed(2)) * 0.1).to(dev)
xr = torch.randn(2, 64, 1000, generator=torch.Generator().manual_seed(3)).to(dev)
for name, b_, xx, sc in (("bias", bias, x, 1.0), ("scale", None, x, 0.37), ("res", None, xr, 1.0), ("scale+res", None, xr, 0.37), ("all", bias, xr, 0.37)):
    y = ops.spec_block(wav, dft_p, nyq, pw_p, b_, xx, n_fft, hop, -4.0, 2.8, 1, sc)
    y2 = ops.pw_conv(s, wt, b_, res=xx, out_scale=sc)
    acc = ops.pw_conv(s, wt, None)
    t = acc
    if b_ is not None:
        t = t + b_.view(1, -1, 1)
    t = t * sc
    t = t + xx
    print(name, 'fused vs unfused', (y - y2).abs().max().item(), ' fused vs torch-fp32-steps', (y - t).abs().max().item(), ' unfused vs torch', (y2 - t).abs().max().item())
