"""GPU: the EXPERIMENTAL bf16x3 numerics mode (csrc/gemm_x3.h; opt-in, offline decoder only, never the default).
Layer level: the split-operand GEMM against the exact fp32 kernels it mirrors (same prologues, same epilogues), on the
decoder's shapes.  Path level: with the decoder's `exec_options.decoder_gemm = "bf16x3"` every index is unchanged (the encoder and the RVQ
are not touched), the decoded waveform stays within 5e-5 of the fp32 product path and within the codec's 1e-4 bar of
the real reference's golden waveform; the default mode is restored and bit-identical afterwards."""
import numpy as np
import pytest
import torch

from hilcodec_amd import synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def rnd(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def test_split_is_exact_to_16_bits():
    from hilcodec_amd import ops
    w = (rnd(1, 96, 64) * 0.3).to(DEV)
    s = ops.x3_split(w)
    assert s.shape == (2, 96, 64) and s.dtype == torch.int16
    parts = (s.to(torch.int32) << 16).view(torch.float32)            # bf16 bit patterns -> fp32
    hi, lo = parts[0], parts[1]
    assert torch.equal(hi, w.to(torch.bfloat16).float())             # round to nearest even, like torch
    assert torch.equal(lo, (w - hi).to(torch.bfloat16).float())
    assert ((hi + lo) - w).abs().max() <= w.abs().max() * 2.0 ** -16


@pytest.mark.parametrize("K,M,Tn,B", [(768, 768, 600, 3), (384, 384, 1000, 2), (192, 192, 372, 2), (96, 96, 128, 3),
                                       (64, 72, 40, 1), (1536, 768, 76, 2)])
def test_dws_conv_x3_vs_fp32(K, M, Tn, B):
    from hilcodec_amd import ops, fold
    x = rnd(K + Tn, B, K, Tn).to(DEV)
    w = rnd(K + M, M, K, 1) / K ** 0.5
    wt = fold.pointwise_layout(w).to(DEV)
    dw = (rnd(3, M, 5) * 0.4).to(DEV)
    db = (rnd(4, M) * 0.2).to(DEV)
    res = rnd(5, B, M, Tn).to(DEV)
    ws = ops.x3_split(wt)
    for kw in (dict(in_scale=0.9, in_elu=True, out_elu=True), dict(res=res, out_scale=0.5)):
        ref = ops.dws_conv(x, wt, dw, db, **kw)
        y = ops.dws_conv_x3(x, ws, dw, db, **kw)
        err = (y - ref).abs().max().item()
        assert err <= 4e-5 * max(1.0, ref.abs().max().item()), (kw.keys(), err)
        assert err > 0.0                                             # it IS a different arithmetic: not silently the fp32 kernel
    assert ops.x3_supported(768, 768, 600) and not ops.x3_supported(100, 768, 600) and not ops.x3_supported(768, 100, 600)
    with pytest.raises(RuntimeError):
        ops.dws_conv_x3(x[:, : K - 16].contiguous(), ops.x3_split(wt[: K - 16].contiguous()), dw, db)   # K % 32 != 0: no silent fallback


@pytest.mark.parametrize("K,M,Tin,r,B", [(1536, 768, 75, 8, 2), (768, 384, 60, 5, 2), (384, 192, 300, 4, 2), (192, 96, 1000, 2, 2)])
def test_up_conv_x3_vs_fp32(K, M, Tin, r, B):
    from hilcodec_amd import ops, fold
    x = rnd(K + Tin, B, K, Tin).to(DEV)
    tw = rnd(K + r, K, 2 * r).to(DEV)
    w = rnd(K + M, M, K, 1) / K ** 0.5
    wt = fold.pointwise_layout(w).to(DEV)
    b = (rnd(M, M) * 0.1).to(DEV)
    ref = ops.up_conv(x, tw, wt, b, r, in_scale=0.7071, in_elu=True)
    y = ops.up_conv_x3(x, tw, ops.x3_split(wt), b, r, in_scale=0.7071)
    # streaming form: the frame before the hop from the cache, new cache = the activated last frame (as the fp32 op)
    hist = torch.nn.functional.elu(rnd(9, B, K, 1)).to(DEV)
    ref_s, c_ref = ops.up_conv(x, tw, wt, b, r, in_scale=0.7071, in_elu=True, hist=hist, want_hist=True)
    y_s, c_x3 = ops.up_conv_x3(x, tw, ops.x3_split(wt), b, r, in_scale=0.7071, hist=hist, want_hist=True)
    assert torch.equal(c_x3, c_ref) and (y_s - ref_s).abs().max().item() <= 4e-5 * max(1.0, ref_s.abs().max().item())
    err = (y - ref).abs().max().item()
    assert 0.0 < err <= 4e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("C,Tn,B", [(96, 1000, 2), (192, 600, 2), (96, 120, 1), (192, 124, 3)])
def test_fused_resblock_x3_vs_fp32(C, Tn, B):
    """the fused residual block with bf16x3 GEMM phases against the same block in fp32 (same kernel, same LDS tile, same
    depthwise / ELU phases; only the two matrix products change)"""
    from hilcodec_amd import ops, fold
    x = rnd(C + Tn, B, C, Tn).to(DEV)
    w1 = fold.pointwise_layout(rnd(C, C, C, 1) / C ** 0.5).to(DEV)
    w2 = fold.pointwise_layout(rnd(C + 1, C, C, 1) / C ** 0.5).to(DEV)
    d1, b1 = (rnd(3, C, 5) * 0.4).to(DEV), (rnd(4, C) * 0.2).to(DEV)
    d2, b2 = (rnd(5, C, 5) * 0.4).to(DEV), (rnd(6, C) * 0.2).to(DEV)
    ref = ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5)
    y = ops.resblock_x3(x, ops.resblock_x3_pack(w1), d1, b1, ops.resblock_x3_pack(w2), d2, b2, 0.9, 0.5)
    err = (y - ref).abs().max().item()
    assert 0.0 < err <= 4e-5 * max(1.0, ref.abs().max().item()), err
    assert ops.resblock_x3_supported(96, 120) and not ops.resblock_x3_supported(64, 120)
    with pytest.raises(RuntimeError):
        ops.resblock_x3(x[:, :64].contiguous(), ops.resblock_pack(w1[:64, :64].contiguous()), d1[:64], b1[:64],
                        ops.resblock_pack(w2[:64, :64].contiguous()), d2[:64], b2[:64], 0.9, 0.5)     # C = 64: no such kernel


@pytest.mark.parametrize("name", ["hil_speech", "hil_music"])
def test_decoder_in_bf16x3_mode(golden, name):
    import hilcodec_amd
    from hilcodec_amd import engine
    g = golden(f"offline_{name}")
    mk = synth.model_kwargs(name)
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    model.load_state_dict(synth.synth_state_dict(name, seed=7), strict=False)
    for l in model.quantizer.layers:
        l.initted = True
    n = g["z"].shape[0]
    x = synth.synth_clips(max(n, 8), 24000, seed=int(g["clip_seed"])).to(DEV)

    def run():
        with torch.no_grad():
            z = model.encoder(x)
            q, _, _, idx = model.quantizer(z, None, return_indices=True)
            return idx, model.decoder(q)

    idx0, wav0 = run()
    opts = model.decoder.exec_options
    assert opts.decoder_gemm == "fp32" and engine.ExecOptions().decoder_gemm == "fp32"
    # a second model in the same process stays in the reference's arithmetic while this one is switched (per-model state)
    other = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    other.load_state_dict(synth.synth_state_dict(name, seed=7), strict=False)
    opts.decoder_gemm = "bf16x3"
    try:
        idx1, wav1 = run()
        with torch.no_grad():
            wav_other = other.decoder(model.quantizer(model.encoder(x), None, return_indices=True)[0])
        assert other.decoder.exec_options.decoder_gemm == "fp32" and torch.equal(wav_other, wav0)
    finally:
        opts.decoder_gemm = "fp32"
    idx2, wav2 = run()
    assert torch.equal(idx0, idx1) and torch.equal(idx1[:n].cpu(), torch.from_numpy(np.asarray(g["indices"])).long())
    d = (wav1 - wav0).abs().max().item()
    assert 0.0 < d < 5e-5, d
    assert (wav1[:n].cpu() - torch.from_numpy(np.asarray(g["wav"]))).abs().max() < 1e-4      # the reference's golden waveform
    assert torch.equal(wav2, wav0) and torch.equal(idx2, idx0)                                # the default is untouched
    opts.decoder_gemm = "tf32"
    try:
        with pytest.raises(RuntimeError):
            run()
    finally:
        opts.decoder_gemm = "fp32"


@pytest.mark.parametrize("B,hop_frames", [(3, 1), (40, 1), (2, 3)])
def test_streaming_decoder_in_bf16x3_mode(B, hop_frames):
    """streaming hops with the decoder's GEMMs in bf16x3 (cache-carrying forms of the same kernels): indices identical (the
    encoder and the RVQ are not touched), decoded audio and every decoder cache within 5e-5 of the fp32 hops over several
    hops; the pipelined graph captured in this mode replays the eager bf16x3 hops bit for bit."""
    from hilcodec_amd import engine
    from hilcodec_amd.graph_step import PipelinedHop
    from tests.test_gpu_streaming import build_streaming
    model, mk, sd = build_streaming()
    L, hops = 320 * hop_frames, 4
    x = synth.synth_clips(B, L * hops, seed=31).to(DEV)

    def run():
        ce, cd = model.initialize_cache(x)
        outs = []
        with torch.no_grad():
            for h in range(hops):
                z, ce = model.encoder(x[:, :, L * h: L * (h + 1)].contiguous(), *ce)
                idx = model.quantizer(z, 8)
                wav, cd = model.decoder(model.dequantizer(idx, 8), *cd)
                outs.append((idx.clone(), wav.clone()))
        return outs, [c.clone() for c in cd]

    ref, cd_ref = run()
    model.decoder.exec_options.decoder_gemm = "bf16x3"
    try:
        got, cd_got = run()
        for h in range(hops):
            assert torch.equal(got[h][0], ref[h][0])
            d = (got[h][1] - ref[h][1]).abs().max().item()
            assert 0.0 < d < 5e-5, (h, d)
        for a, b in zip(cd_got, cd_ref):
            assert (a - b).abs().max().item() <= 5e-5 * max(1.0, b.abs().max().item())
        if hop_frames == 1:
            g = PipelinedHop(model, B, 320, 8, DEV)
            for h in range(hops):
                idx, wav = g.step(x[:, :, 320 * h: 320 * (h + 1)])
                assert torch.equal(idx, got[h][0])
                if h > 0:
                    assert torch.equal(wav, got[h - 1][1])
            assert torch.equal(g.flush(), got[hops - 1][1])
    finally:
        model.decoder.exec_options.decoder_gemm = "fp32"
    again, _ = run()
    assert all(torch.equal(a[1], b[1]) for a, b in zip(again, ref))
