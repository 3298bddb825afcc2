"""GPU parity, per kernel: every C-ABI entry point (through hilcodec_amd.ops / the module classes)
against (a) the golden known-answer vectors produced by the REAL reference (tests/golden/ops.npz)
and (b) the CPU oracle on seeded inputs at larger, awkward shapes.

Tolerances (fp32, written per op): a conv differs from the oracle only by fp32 summation order
(the reference uses oneDNN on CPU, the kernels an fmaf chain in k order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hilcodec_amd import synth

pytestmark = pytest.mark.gpu

RS = 0.5773502691896258


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def env():
    from hilcodec_amd import fold, ops
    from oracle import hilcodec_oracle as O
    return ops, fold, O, torch.device("cuda:0")


def rnd(seed, *shape):
    return torch.from_numpy(synth.normalish(seed, int(np.prod(shape)))).view(*shape)


def close(a, b, atol, what=""):
    a = a.detach().cpu()
    d = (a - b).abs().max().item()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert d <= atol, f"{what}: max abs diff {d:.3e} > {atol:.1e}"


def sub(g, tag):
    pre = tag + "."
    return {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}


def test_golden_ops(env, golden):
    ops, fold, O, dev = env
    g = golden("ops")
    d = sub(g, "pw")
    w, b = O.conv_weight(d, "conv.conv")
    y = ops.pw_conv(d["x"].to(dev), fold.pointwise_layout(w).to(dev), b.to(dev), in_elu=True)
    close(y, d["y"], 2e-6, "pw")
    d = sub(g, "dw5")
    w, b = O.conv_weight(d, "conv.conv")
    close(ops.dw_conv(d["x"].to(dev), fold.depthwise_layout(w).to(dev), b.to(dev)), d["y"], 2e-6, "dw5")
    for r in (2, 4, 5, 8):
        d = sub(g, f"dws{r}")
        w, b = O.conv_weight(d, "conv.conv")
        close(ops.dw_conv(d["x"].to(dev), fold.depthwise_layout(w).to(dev), b.to(dev), stride=r), d["y"], 2e-6, f"dws{r}")
        d = sub(g, f"dwt{r}")
        w, _ = O.conv_weight(d, "convtr.convtr")
        close(ops.dw_convtr(d["x"].to(dev), fold.depthwise_layout(w).to(dev), r), d["y"], 2e-6, f"dwt{r}")
    d = sub(g, "pre")
    w, b = O.conv_weight(d, "conv.conv")
    close(ops.conv_pre(d["x"].to(dev), w[:, 0, :].contiguous().to(dev), b.to(dev)), d["y"], 2e-6, "pre")
    d = sub(g, "post")
    w, b = O.conv_weight(d, "conv.conv")
    close(ops.conv_post(d["x"].to(dev), w[0].contiguous().to(dev), b.to(dev), in_elu=False, do_tanh=False),
          d["y"], 4e-6, "post")
    for n_fft, hop in ((16, 1), (32, 4)):
        d = sub(g, f"spec{n_fft}")
        bt = fold.stft_basis_layout(d["spec.weight"]).to(dev)
        mag = ops.stft_logmag(d["wav"].to(dev), bt, n_fft, hop, normalize=2)
        close(mag, d["mag"], 2e-6, "stft mag")
        w, _ = O.conv_weight(d, "layer.conv.conv")
        s = ops.stft_logmag(d["wav"].to(dev), bt, n_fft, hop, -4.0, 2.8, True)
        scale = float((d["scale_param"] * RS)[0])
        y = ops.pw_conv(s, fold.pointwise_layout(w).to(dev), None, res=d["x"].to(dev), out_scale=scale)
        # log of a tiny magnitude amplifies its fp32 rounding: compare where the reference mag is not tiny
        close(y, d["y"], 2e-4, "specblock")
    d = sub(g, "l2")
    close(ops.l2norm(d["x"].to(dev), 1e-12, 128 ** 0.5), d["y"], 1e-6, "l2norm")
    close(ops.l2norm(d["x"].to(dev), 1e-12, 128 ** 0.5, channel_last_out=True), d["y"].transpose(1, 2), 1e-6, "l2norm cl")
    d = sub(g, "cconv")
    y, c = ops.dw_conv(d["x"].to(dev), d["w"][:, 0].contiguous().to(dev), d["b"].to(dev), stride=5,
                       hist=d["cache"].to(dev), want_hist=True)
    close(y, d["y"], 2e-6, "cconv")
    assert torch.equal(c.cpu(), d["cache_out"])
    d = sub(g, "cconvtr")
    y, c = ops.dw_convtr(d["x"].to(dev), d["w"][:, 0].contiguous().to(dev), 5, hist=d["cache"].to(dev), want_hist=True)
    close(y, d["y"], 2e-6, "cconvtr")
    assert torch.equal(c.cpu(), d["cache_out"])


def test_golden_resblock_modules(env, golden):
    """The module classes (reference names / state-dict keys) against the reference's own outputs."""
    ops, fold, O, dev = env
    from hilcodec_amd.models.hilcodec.modules import SEANetResnetBlock, SpecBlock, SConv1d
    g = golden("ops")
    for idx in (0, 1, 2):
        d = sub(g, f"res{idx}")
        m = SEANetResnetBlock(16, kernel_size=5, dilations=[1, 1], norm="weight_norm", causal=True,
                              skip="identity", res_scale=RS, idx=idx, zero_init=True)
        sd = {k: v for k, v in d.items() if k not in ("x", "y")}
        m.load_state_dict(sd)
        close(m(d["x"].to(dev)), d["y"], 5e-6, f"resblock idx{idx}")
    for n_fft, hop in ((16, 1), (32, 4)):
        d = sub(g, f"spec{n_fft}")
        m = SpecBlock("stft", "log", n_fft, 8, hop, "weight_norm", {}, bias=False, pad_mode="constant",
                      learnable=False, causal=True, mean=-4.0, std=2.8, res_scale=RS)
        m.load_state_dict({k: v for k, v in d.items() if k not in ("x", "y", "wav", "mag")})
        close(m(d["x"].to(dev), d["wav"].to(dev)), d["y"], 2e-4, "SpecBlock module")
    d = sub(g, "dw5")
    m = SConv1d(24, 24, 5, groups=24, causal=True, norm="weight_norm", bias=True)
    m.load_state_dict({k: v for k, v in d.items() if k not in ("x", "y")})
    close(m(d["x"].to(dev)), d["y"], 2e-6, "SConv1d dw")
    d = sub(g, "ws")
    m = SConv1d(6, 10, 3, groups=1, causal=True, norm="weight_standardization", norm_kwargs={"scale": 1.7}, bias=True)
    with torch.no_grad():
        m.conv.conv.weight_v.copy_(d["v"]); m.conv.conv.weight_g.copy_(d["g"])
    assert torch.equal(m.conv.conv.effective_weight(), d["w"])


@pytest.mark.parametrize("B,K,M,Tn", [(3, 64, 64, 1000), (2, 96, 96, 601), (2, 33, 64, 75), (1, 1024, 128, 75),
                                      (2, 128, 1536, 75), (2, 257, 512, 40), (1, 768, 384, 360), (5, 192, 192, 128),
                                      # linear-addressing core (T % 4 == 0): ragged K tails with every row-tile height
                                      (2, 33, 64, 1000), (1, 65, 96, 400), (4, 129, 160, 128), (2, 40, 192, 64),
                                      (1, 16, 32, 4), (3, 17, 128, 2048), (1, 513, 1024, 76),
                                      # single-frame layers of a streaming hop (csrc/frame1.hip): ragged K / M / B
                                      (70, 513, 1024, 1), (1024, 1024, 128, 1), (33, 128, 1536, 1), (5, 7, 20, 1),
                                      (97, 130, 36, 1), (1, 2, 4, 1)])
def test_pw_conv_vs_oracle(env, B, K, M, Tn):
    ops, fold, O, dev = env
    x = rnd(B * 7 + K, B, K, Tn)
    w = rnd(K + M, M, K, 1) / K ** 0.5
    b = rnd(M, M) * 0.1
    r = rnd(3, B, M, Tn)
    ref = (F.conv1d(F.elu(x * 0.77), w, b) * 0.61 + r)
    y = ops.pw_conv(x.to(dev), fold.pointwise_layout(w).to(dev), b.to(dev), res=r.to(dev), in_scale=0.77,
                    in_elu=True, out_scale=0.61)
    close(y, ref, 1e-5, "pw_conv")
    # residual add (SpecBlock usage), no bias, no prologue; the op is functional: the residual operand is untouched
    r2 = r.clone().to(dev)
    y2 = ops.pw_conv(x.to(dev), fold.pointwise_layout(w).to(dev), None, res=r2, out_scale=0.5)
    close(y2, F.conv1d(x, w) * 0.5 + r, 1e-5, "pw_conv residual")
    assert torch.equal(r2.cpu(), r) and y2.data_ptr() != r2.data_ptr()


def test_pw_conv_transpose_detect(env):
    """Asymmetric weights / identity check (a swapped C layout would pass a symmetric test)."""
    ops, fold, O, dev = env
    K = M = 64
    x = rnd(1, 1, K, 256)
    w = torch.zeros(M, K, 1)
    for m in range(M):
        w[m, (m * 7 + 3) % K, 0] = 1.0 + m
    y = ops.pw_conv(x.to(dev), fold.pointwise_layout(w).to(dev))
    assert torch.equal(y.cpu(), F.conv1d(x, w))


@pytest.mark.parametrize("C,Tn,k,s", [(64, 1000, 5, 1), (24, 37, 5, 1), (128, 999, 4, 2), (256, 1203, 8, 4),
                                      (512, 77, 10, 5), (1024, 600, 16, 8), (96, 24, 5, 1)])
def test_dw_conv_vs_oracle(env, C, Tn, k, s):
    ops, fold, O, dev = env
    x = rnd(C + Tn, 2, C, Tn)
    w = rnd(C + k, C, 1, k)
    b = rnd(C, C) * 0.1
    ref = O.sconv1d(x, w, b, stride=s, groups=C)
    close(ops.dw_conv(x.to(dev), w[:, 0].contiguous().to(dev), b.to(dev), stride=s), ref, 5e-6, "dw")
    if s == 1:
        r = rnd(5, 2, C, Tn)
        ref2 = F.elu(O.sconv1d(F.elu(x * 0.9), w, b, groups=C) * 0.4 + r)
        y = ops.dw_conv(x.to(dev), w[:, 0].contiguous().to(dev), b.to(dev), res=r.to(dev), in_scale=0.9,
                        in_elu=True, out_scale=0.4, out_elu=True)
        close(y, ref2, 5e-6, "dw fused")


@pytest.mark.parametrize("C,Tn,r", [(1536, 75, 8), (768, 600, 5), (384, 301, 4), (192, 1000, 2)])
def test_dw_convtr_vs_oracle(env, C, Tn, r):
    ops, fold, O, dev = env
    x = rnd(C + Tn, 2, C, Tn)
    w = rnd(C + r, C, 1, 2 * r)
    ref = O.sconvtr1d(F.elu(x * 0.7), w, None, stride=r, groups=C)
    close(ops.dw_convtr(x.to(dev), w[:, 0].contiguous().to(dev), r, in_scale=0.7, in_elu=True), ref, 5e-6, "convtr")


@pytest.mark.parametrize("n_fft,hop,Tn", [(64, 1, 1000), (128, 2, 2001), (256, 8, 4000), (512, 40, 4800), (1024, 320, 4800)])
def test_stft_vs_oracle(env, n_fft, hop, Tn):
    ops, fold, O, dev = env
    wav = synth.synth_clips(2, Tn, seed=n_fft)
    basis = synth.stft_basis(n_fft)
    mag = O.causal_stft_mag(wav, basis, hop, True, True)
    y = ops.stft_logmag(wav.to(dev), fold.stft_basis_layout(basis).to(dev), n_fft, hop, normalize=2)
    # |re|,|im| are sums of n_fft products of O(0.1) terms: absolute error ~ 1e-7 * sqrt(n_fft) * 0.1 * few
    close(y, mag, 2e-5, "stft magnitude")
    ref = (mag.clamp_min(1e-5).log() - (-4.0)) / 2.8
    y = ops.stft_logmag(wav.to(dev), fold.stft_basis_layout(basis).to(dev), n_fft, hop, -4.0, 2.8, True).cpu()
    ok = mag > 1e-2                      # log amplifies relative error of tiny magnitudes
    assert (y - ref)[ok].abs().max() < 2e-4
    # history == explicit left context
    hist = synth.synth_clips(2, n_fft - 1 + 5, seed=77)
    full = torch.cat([hist, wav], dim=2)
    m2 = O.causal_stft_mag(full[:, :, 5:], basis, hop, False, False)
    y2 = ops.stft_logmag(wav.to(dev), fold.stft_basis_layout(basis).to(dev), n_fft, hop, normalize=2, hist=hist.to(dev))
    close(y2, m2, 2e-5, "stft with history")


def test_conv_pre_post_vs_oracle(env):
    ops, fold, O, dev = env
    wav = synth.synth_clips(3, 1001, seed=5)
    w = rnd(1, 64, 1, 5); b = rnd(2, 64) * 0.1
    ref = O.sconv1d(wav * (1 / 0.1122080159), w, b)
    close(ops.conv_pre(wav.to(dev), w[:, 0].contiguous().to(dev), b.to(dev), in_scale=1 / 0.1122080159), ref, 2e-5, "conv_pre")
    x = rnd(9, 3, 96, 1001)
    w = rnd(3, 1, 96, 5) / 20; b = rnd(4, 1) * 0.1
    ref = torch.tanh(O.sconv1d(F.elu(x * 0.707), w, b) * 0.1122080159)
    y = ops.conv_post(x.to(dev), w[0].contiguous().to(dev), b.to(dev), in_scale=0.707, in_elu=True,
                      out_scale=0.1122080159, do_tanh=True)
    close(y, ref, 2e-6, "conv_post")


def test_tail(env):
    ops, fold, O, dev = env
    x = rnd(1, 2, 3, 7); h = rnd(2, 2, 3, 10)
    for pad in (4, 7, 9, 10):
        ref = torch.cat([h, x], dim=2)[:, :, -pad:]
        assert torch.equal(ops.tail(x.to(dev), h.to(dev), pad).cpu(), ref)
    assert torch.equal(ops.tail(x.to(dev), None, 9).cpu(), torch.cat([torch.zeros(2, 3, 2), x], 2))


def test_errors(env):
    ops, fold, O, dev = env
    x = torch.zeros(1, 8, 16, device=dev)
    with pytest.raises(RuntimeError):
        ops.pw_conv(torch.zeros(1, 8, 16), torch.zeros(8, 8, device=dev))           # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.pw_conv(x, torch.zeros(8, 6, device=dev))                                # M % 4 != 0 -> unsupported
    with pytest.raises(RuntimeError):
        ops.dw_conv(x, torch.zeros(8, 40, device=dev))                               # k > 16
    with pytest.raises(RuntimeError):
        ops.pw_conv(x.double(), torch.zeros(8, 8, device=dev))


@pytest.mark.parametrize("B,K,M,Tn", [(2, 64, 64, 1000), (2, 96, 96, 372), (1, 192, 192, 601), (2, 128, 1536, 75),
                                      (1, 768, 768, 600), (3, 256, 256, 124), (2, 384, 384, 125),
                                      # linear core with ragged K and 1 / 2 / 3-block row tiles
                                      (2, 33, 96, 248), (1, 70, 160, 128), (3, 129, 64, 8), (1, 16, 32, 4)])
def test_dws_conv_k5_vs_oracle(env, B, K, M, Tn):
    """fused pointwise -> depthwise k5 (both residual-block halves) against the two-step oracle"""
    ops, fold, O, dev = env
    x = rnd(B * 3 + K + Tn, B, K, Tn)
    w = rnd(K + M, M, K, 1) / K ** 0.5
    dw = rnd(M + 5, M, 1, 5)
    db = rnd(M, M) * 0.1
    r = rnd(11, B, M, Tn)
    h = F.conv1d(F.elu(x * 0.86), w)
    ref1 = F.elu(O.sconv1d(h, dw, db, groups=M))
    y1 = ops.dws_conv(x.to(dev), fold.pointwise_layout(w).to(dev), dw[:, 0].contiguous().to(dev), db.to(dev),
                      in_scale=0.86, in_elu=True, out_elu=True)
    close(y1, ref1, 2e-5, "dws first half")
    ref2 = O.sconv1d(F.conv1d(x, w), dw, db, groups=M) * 0.37 + r
    r2 = r.clone().to(dev)
    y2 = ops.dws_conv(x.to(dev), fold.pointwise_layout(w).to(dev), dw[:, 0].contiguous().to(dev), db.to(dev), res=r2,
                      out_scale=0.37)
    close(y2, ref2, 2e-5, "dws second half (residual)")
    assert torch.equal(r2.cpu(), r)


@pytest.mark.parametrize("K,M,Tn,r", [(64, 128, 1000, 2), (128, 256, 1203, 4), (256, 512, 600, 5), (512, 1024, 77, 8),
                                      (64, 128, 124, 2), (96, 96, 250, 5)])
def test_dws_conv_strided_vs_oracle(env, K, M, Tn, r):
    ops, fold, O, dev = env
    x = rnd(K + Tn, 2, K, Tn)
    w = rnd(K + M, M, K, 1) / K ** 0.5
    dw = rnd(M + r, M, 1, 2 * r)
    db = rnd(M, M) * 0.1
    ref = O.sconv1d(F.conv1d(F.elu(x * 0.77), w), dw, db, stride=r, groups=M)
    y = ops.dws_conv(x.to(dev), fold.pointwise_layout(w).to(dev), dw[:, 0].contiguous().to(dev), db.to(dev),
                     stride=r, in_scale=0.77, in_elu=True)
    close(y, ref, 2e-5, "dws strided")


@pytest.mark.parametrize("C,Tn,B", [(64, 1000, 2), (96, 360, 3), (128, 124, 2), (192, 600, 1), (96, 120, 1), (64, 8, 2),
                                     (256, 600, 2), (384, 300, 3), (512, 124, 2), (768, 76, 2)])
def test_fused_resblock_vs_oracle(env, C, Tn, B):
    """hilc_resblock (whole residual block in one launch) against the oracle's resblock()"""
    ops, fold, O, dev = env
    assert ops.resblock_supported(C, Tn)
    sd = {
        "p.block.1.conv.conv.weight": rnd(1, C, C, 1) / C ** 0.5,
        "p.block.2.conv.conv.weight": rnd(2, C, 1, 5) * 0.5, "p.block.2.conv.conv.bias": rnd(3, C) * 0.2,
        "p.block.4.conv.conv.weight": rnd(4, C, C, 1) / C ** 0.5,
        "p.block.5.conv.conv.weight": rnd(5, C, 1, 5) * 0.5, "p.block.5.conv.conv.bias": rnd(6, C) * 0.2,
        "p.res_scale_param": torch.tensor([0.8]),
    }
    x = rnd(C + Tn, B, C, Tn)
    for idx in (0, 2):
        ref = O.resblock(sd, "p", x, RS, idx)
        y = ops.resblock(x.to(dev), fold.pointwise_layout(sd["p.block.1.conv.conv.weight"]).to(dev),
                         sd["p.block.2.conv.conv.weight"][:, 0].contiguous().to(dev), sd["p.block.2.conv.conv.bias"].to(dev),
                         fold.pointwise_layout(sd["p.block.4.conv.conv.weight"]).to(dev),
                         sd["p.block.5.conv.conv.weight"][:, 0].contiguous().to(dev), sd["p.block.5.conv.conv.bias"].to(dev),
                         (1 + idx * RS ** 2) ** -0.5, float((RS * sd["p.res_scale_param"])[0]))
        close(y, ref, 2e-5, f"fused resblock C{C} idx{idx}")
    assert not ops.resblock_supported(80, Tn) and not ops.resblock_supported(C, 75)


@pytest.mark.parametrize("C,hop,hops,B", [(64, 320, 3, 2), (96, 320, 2, 1), (128, 160, 3, 2), (192, 160, 2, 3), (64, 8, 5, 2),
                                          (96, 4, 4, 1),
                                          # stream counts whose hop splits into equal whole-stream runs: the carry form of the
                                          # streaming block (runs of 5 tiles = 4 / 2 streams, no halo), one run and several
                                          (192, 160, 2, 4), (192, 160, 2, 1100), (96, 320, 2, 2), (96, 320, 2, 1028), (192, 32, 3, 12)])
def test_fused_resblock_streaming_equals_offline(env, C, hop, hops, B):
    """hilc_resblock_stream hop by hop (caches = last 4 pointwise outputs of each depthwise conv,
    causal_layers.py:147-167) must reproduce the offline block on the concatenated signal bit for bit, and its
    caches must equal the oracle's streaming caches."""
    ops, fold, O, dev = env
    w1, w2 = (rnd(1, C, C) / C ** 0.5).to(dev), (rnd(4, C, C) / C ** 0.5).to(dev)
    d1, b1 = (rnd(2, C, 5) * 0.5).to(dev), (rnd(3, C) * 0.2).to(dev)
    d2, b2 = (rnd(5, C, 5) * 0.5).to(dev), (rnd(6, C) * 0.2).to(dev)
    x = rnd(C + hop, B, C, hop * hops).to(dev)
    full = ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.4)
    caches = [torch.zeros(B, C, 4, device=dev), torch.zeros(B, C, 4, device=dev)]
    outs = []
    for h in range(hops):
        y, caches = ops.resblock(x[:, :, h * hop:(h + 1) * hop].contiguous(), w1, d1, b1, w2, d2, b2, 0.9, 0.4, hist=caches)
        outs.append(y)
    assert torch.equal(torch.cat(outs, dim=2), full)
    # caches against the un-fused streaming ops (pointwise GEMM + depthwise conv with history)
    h1 = ops.pw_conv(x, w1, in_scale=0.9, in_elu=True)
    g = ops.dw_conv(h1, d1, b1)
    h2 = ops.pw_conv(g, w2, in_elu=True)
    close(caches[0], h1[:, :, -4:].cpu(), 2e-5, "cache 1")
    close(caches[1], h2[:, :, -4:].cpu(), 2e-5, "cache 2")


@pytest.mark.parametrize("C,T,B", [(768, 8, 37), (768, 8, 4), (512, 8, 21), (768, 4, 9), (512, 32, 3), (384, 40, 19), (384, 40, 1),
                                   (256, 40, 23), (256, 8, 50), (384, 4, 7), (256, 64, 5), (384, 128, 2)])
def test_wide_stream_block_equals_two_launches(env, C, T, B):
    """The wide blocks of a streaming hop as ONE launch (hilc_resblock_stream, NARROW shapes: 32-column tiles of whole streams
    for C = 512 / 768, 64-column tiles on the flat column space for C = 256 / 384) against the two hilc_dws_conv_stream
    launches they replace (`streaming.py:195-276`): output and both new caches bit for bit, over three hops, with ragged
    stream counts."""
    ops, fold, O, dev = env
    assert ops.resblock_supported(C, T, B, streaming=True)
    w1, w2 = (rnd(1, C, C) / C ** 0.5).to(dev), (rnd(4, C, C) / C ** 0.5).to(dev)
    d1, b1 = (rnd(2, C, 5) * 0.5).to(dev), (rnd(3, C) * 0.2).to(dev)
    d2, b2 = (rnd(5, C, 5) * 0.5).to(dev), (rnd(6, C) * 0.2).to(dev)
    w1p, w2p = ops.resblock_pack(w1), ops.resblock_pack(w2)
    ca = [(rnd(7, B, C, 4) * 0.7).to(dev), (rnd(8, B, C, 4) * 0.7).to(dev)]
    cb = [c.clone() for c in ca]
    for h in range(3):
        x = rnd(C + T + h, B, C, T).to(dev)
        y, ca = ops.resblock(x, w1p, d1, b1, w2p, d2, b2, 0.9, 0.4, hist=ca)
        g, c0 = ops.dws_conv_stream(x, w1, d1, b1, cb[0], in_scale=0.9, in_elu=True, out_elu=True)
        y2, c1 = ops.dws_conv_stream(g, w2, d2, b2, cb[1], res=x, out_scale=0.4)
        cb = [c0, c1]
        assert torch.equal(y, y2), (h, float((y - y2).abs().max()))
        assert torch.equal(ca[0], cb[0]) and torch.equal(ca[1], cb[1]), h
    # zero history (NULL caches) and caller-provided cache outputs
    x = rnd(99, B, C, T).to(dev)
    o = [torch.empty(B, C, 4, device=dev), torch.empty(B, C, 4, device=dev)]
    z = [torch.zeros(B, C, 4, device=dev), torch.zeros(B, C, 4, device=dev)]
    y, cs = ops.resblock(x, w1p, d1, b1, w2p, d2, b2, 0.9, 0.4, hist=z, hist_out=o)
    g, c0 = ops.dws_conv_stream(x, w1, d1, b1, z[0], in_scale=0.9, in_elu=True, out_elu=True)
    y2, c1 = ops.dws_conv_stream(g, w2, d2, b2, z[1], res=x, out_scale=0.4)
    assert torch.equal(y, y2) and torch.equal(o[0], c0) and torch.equal(o[1], c1)


@pytest.mark.parametrize("C,T,B,n", [(64, 320, 5, 2), (64, 320, 1024, 2), (96, 320, 3, 3), (96, 320, 1024, 3), (128, 160, 7, 2), (128, 160, 1024, 2),
                                      (192, 160, 6, 3), (192, 160, 1024, 3), (512, 8, 21, 2), (512, 8, 1024, 2), (768, 8, 37, 3),
                                      (768, 8, 1024, 3), (96, 640, 2, 3), (64, 960, 3, 2), (192, 480, 2, 3), (128, 4, 9, 2), (96, 12, 70, 2),
                                      (768, 8, 5, 2), (192, 160, 9, 2)])
def test_resblock_chain_equals_block_by_block(env, C, T, B, n):
    """The residual blocks of one stage of a streaming hop in ONE launch (hilc_resblock_chain: `streaming.py:497-503,633-639`
    runs them one after the other) against the same blocks launched one by one (hilc_resblock_stream): output and all 2n new
    caches bit for bit, over three hops — ragged stream counts (short last run), the full 1024 streams (256 equal runs),
    multi-frame hops, hops shorter than a tile, zero history and caller-owned cache outputs."""
    ops, fold, O, dev = env
    from hilcodec_amd._lib import lib
    assert ops.resblock_chain_supported(C, T, n, B) and lib.hilc_resblock_chain_supported(C, T, n, 1) == 1
    assert lib.hilc_resblock_chain_row_classes(C) == ops.resblock_chain_row_classes(C)
    blocks = []
    for j in range(n):
        w1, w2 = (rnd(10 * j + 1, C, C) / C ** 0.5).to(dev), (rnd(10 * j + 4, C, C) / C ** 0.5).to(dev)
        d1, b1 = (rnd(10 * j + 2, C, 5) * 0.5).to(dev), (rnd(10 * j + 3, C) * 0.2).to(dev)
        d2, b2 = (rnd(10 * j + 5, C, 5) * 0.5).to(dev), (rnd(10 * j + 6, C) * 0.2).to(dev)
        blocks.append(dict(single=(ops.resblock_pack(w1), d1, b1, ops.resblock_pack(w2), d2, b2),
                           chain=(ops.resblock_chain_pack(w1), d1, b1, ops.resblock_chain_pack(w2), d2, b2),
                           pre=(1.0 + j / 3.0) ** -0.5, post=0.4 + 0.1 * j))
    ca = [[(rnd(7 + j, B, C, 4) * 0.7).to(dev), (rnd(8 + j, B, C, 4) * 0.7).to(dev)] for j in range(n)]
    cb = [[c.clone() for c in pair] for pair in ca]
    for h in range(3):
        x = rnd(C + T + h, B, C, T).to(dev)
        y, flat = ops.resblock_chain(x, [blk["chain"] + (blk["pre"], blk["post"]) for blk in blocks], ca)
        ca = [flat[2 * j:2 * j + 2] for j in range(n)]
        y2 = x
        for j, blk in enumerate(blocks):
            y2, cb[j] = ops.resblock(y2, *blk["single"], blk["pre"], blk["post"], hist=cb[j])
        assert torch.equal(y, y2), (h, float((y - y2).abs().max()))
        for j in range(n):
            assert torch.equal(ca[j][0], cb[j][0]) and torch.equal(ca[j][1], cb[j][1]), (h, j)
    # zero history, caller-provided cache outputs
    x = rnd(99, B, C, T).to(dev)
    zeros = [[torch.zeros(B, C, 4, device=dev), torch.zeros(B, C, 4, device=dev)] for _ in range(n)]
    outs = [[torch.full((B, C, 4), 7.0, device=dev), torch.full((B, C, 4), 7.0, device=dev)] for _ in range(n)]
    y, flat = ops.resblock_chain(x, [blk["chain"] + (blk["pre"], blk["post"]) for blk in blocks], zeros, outs)
    y2 = x
    for j, blk in enumerate(blocks):
        y2, cs = ops.resblock(y2, *blk["single"], blk["pre"], blk["post"], hist=zeros[j])
        assert torch.equal(outs[j][0], cs[0]) and torch.equal(outs[j][1], cs[1]) and flat[2 * j] is outs[j][0]
    assert torch.equal(y, y2)


@pytest.mark.parametrize("K,M,T,r,B", [(64, 128, 320, 2, 9), (128, 256, 160, 4, 33), (256, 512, 40, 5, 7), (512, 1024, 8, 8, 21)])
def test_down_conv_stream_adds_the_next_spec_branch(env, K, M, T, r, B):
    """`res` of the streaming down-sampling layer (flat strided epilogue for T > 128, whole-stream tiles below): the next stage's
    SpecBlock branch is added by the epilogue, fadd(down, branch) — the same bits as the in-line `x.add_(branch)` one launch
    later (`streaming.py:497-511`)."""
    ops, fold, O, dev = env
    wt = (rnd(1, K, M) / K ** 0.5).to(dev)
    dw, db = (rnd(2, M, 2 * r) * 0.4).to(dev), (rnd(3, M) * 0.2).to(dev)
    x = rnd(4, B, K, T).to(dev)
    cache = (rnd(5, B, M, r) * 0.5).to(dev)
    branch = rnd(6, B, M, T // r).to(dev)
    y0, c0 = ops.dws_conv_stream(x, wt, dw, db, cache, stride=r, in_scale=0.77, in_elu=True)
    y1, c1 = ops.dws_conv_stream(x, wt, dw, db, cache, res=branch, stride=r, in_scale=0.77, in_elu=True)
    assert torch.equal(y1, y0 + branch) and torch.equal(c0, c1)


@pytest.mark.parametrize("n_fft,hop,B,T", [(128, 2, 5, 320), (64, 1, 2, 640), (256, 8, 3, 2048)])
def test_spec_block_branch_alone(env, n_fft, hop, B, T):
    """hilc_spec_block with x = NULL: the branch alone (what a streaming hop computes beside the previous stage); branch + x in a
    separate add equals the one-launch block bit for bit."""
    ops, fold, O, dev = env
    from hilcodec_amd.models.hilcodec.modules.conv import CausalSTFT
    st = CausalSTFT(n_fft, hop)
    basis = fold.stft_basis_layout(st.weight).to(dev)
    C = n_fft
    w = (rnd(1, n_fft // 2 + 1, C) / n_fft ** 0.5).to(dev)
    bias = (rnd(2, C) * 0.1).to(dev)
    tables = ops.spec_block_tables(basis, w, n_fft)
    wav = (rnd(3, B, 1, T) * 0.1).to(dev)
    hist = (rnd(4, B, 1, n_fft - 1) * 0.1).to(dev)
    x = rnd(5, B, C, (T - 1) // hop + 1).to(dev)
    full = ops.spec_block(wav, tables[0], tables[1], tables[2], bias, x, n_fft, hop, -4.0, 2.8, True, 0.45, hist=hist)
    alone = ops.spec_block(wav, tables[0], tables[1], tables[2], bias, None, n_fft, hop, -4.0, 2.8, True, 0.45, hist=hist)
    assert torch.equal(alone + x, full)


@pytest.mark.parametrize("C,T,B,n", [(64, 1000, 3, 2), (96, 24000, 2, 3), (128, 124, 5, 2), (192, 600, 2, 3), (96, 120, 70, 2), (64, 24000, 40, 2),
                                      (192, 12000, 24, 3), (256, 3000, 3, 2), (384, 3000, 5, 3), (512, 600, 7, 2), (384, 124, 300, 3), (256, 600, 70, 2),
                                      (384, 3000, 2, 2)])
def test_resblock_chain_offline_equals_block_by_block(env, C, T, B, n):
    """hilc_resblock_chain with streaming = 0: the blocks of a stage of the OFFLINE causal model in one launch (contiguous runs
    with one carry per block; a run that starts inside a clip warms up on the tile in front) == hilc_resblock block by block."""
    ops, fold, O, dev = env
    assert ops.resblock_chain_supported(C, T, n, B, streaming=False)
    blocks = []
    for j in range(n):
        w1, w2 = (rnd(10 * j + 1, C, C) / C ** 0.5).to(dev), (rnd(10 * j + 4, C, C) / C ** 0.5).to(dev)
        d1, b1 = (rnd(10 * j + 2, C, 5) * 0.5).to(dev), (rnd(10 * j + 3, C) * 0.2).to(dev)
        d2, b2 = (rnd(10 * j + 5, C, 5) * 0.5).to(dev), (rnd(10 * j + 6, C) * 0.2).to(dev)
        blocks.append(((ops.resblock_pack(w1), d1, b1, ops.resblock_pack(w2), d2, b2),
                       (ops.resblock_chain_pack(w1, False), d1, b1, ops.resblock_chain_pack(w2, False), d2, b2),
                       (1.0 + j / 3.0) ** -0.5, 0.4 + 0.1 * j))
    x = rnd(C + T, B, C, T).to(dev)
    y = ops.resblock_chain(x, [c + (pre, post) for _, c, pre, post in blocks])
    y2 = x
    for single, _, pre, post in blocks:
        y2 = ops.resblock(y2, *single, pre, post)
    assert torch.equal(y, y2), float((y - y2).abs().max())


@pytest.mark.parametrize("C,T,B", [(256, 3000, 3), (384, 3000, 2), (512, 600, 5), (768, 600, 3), (384, 3000, 40), (768, 600, 300), (256, 44, 700),
                                   (512, 4, 9)])
def test_wide_resblock_offline_equals_two_launches(env, C, T, B):
    """The wide blocks of the OFFLINE model (C = 256 ... 768) as ONE carry-form launch (round 4: 32- / 64-column tiles, the eight waves
    split the rows; runs of one workgroup per CU, warm-up tile where a run starts inside a clip) == the two hilc_dws_conv launches
    they replace, bit for bit — few clips (every run starts inside a clip), many clips, clips shorter than a tile."""
    ops, fold, O, dev = env
    w1, w2 = (rnd(1, C, C) / C ** 0.5).to(dev), (rnd(4, C, C) / C ** 0.5).to(dev)
    d1, b1 = (rnd(2, C, 5) * 0.5).to(dev), (rnd(3, C) * 0.2).to(dev)
    d2, b2 = (rnd(5, C, 5) * 0.5).to(dev), (rnd(6, C) * 0.2).to(dev)
    x = rnd(C + T, B, C, T).to(dev)
    assert ops.resblock_supported(C, T, B)
    y = ops.resblock(x, ops.resblock_pack(w1), d1, b1, ops.resblock_pack(w2), d2, b2, 0.9, 0.4)
    g = ops.dws_conv(x, w1, d1, b1, in_scale=0.9, in_elu=True, out_elu=True)
    y2 = ops.dws_conv(g, w2, d2, b2, res=x, out_scale=0.4)
    assert torch.equal(y, y2), float((y - y2).abs().max())


def _stage_params(ops, dev, C, r, n, streaming):
    blocks, singles = [], []
    for j in range(n):
        w1, w2 = (rnd(10 * j + 1, C, C) / C ** 0.5).to(dev), (rnd(10 * j + 4, C, C) / C ** 0.5).to(dev)
        d1, b1 = (rnd(10 * j + 2, C, 5) * 0.5).to(dev), (rnd(10 * j + 3, C) * 0.2).to(dev)
        d2, b2 = (rnd(10 * j + 5, C, 5) * 0.5).to(dev), (rnd(10 * j + 6, C) * 0.2).to(dev)
        pre, post = (1.0 + j / 3.0) ** -0.5, 0.4 + 0.1 * j
        singles.append(((ops.resblock_pack(w1), d1, b1, ops.resblock_pack(w2), d2, b2), pre, post))
        blocks.append((ops.resblock_chain_pack(w1, streaming), d1, b1, ops.resblock_chain_pack(w2, streaming), d2, b2, pre, post))
    wd = (rnd(91, C, 2 * C) / C ** 0.5).to(dev)                       # k-major [C][2C]
    dw, db = (rnd(92, 2 * C, 2 * r) * 0.4).to(dev), (rnd(93, 2 * C) * 0.2).to(dev)
    down = (ops.resblock_chain_pack(wd[:, :C].contiguous(), streaming), ops.resblock_chain_pack(wd[:, C:].contiguous(), streaming),
            dw, db, 0.7746, r)
    return blocks, singles, wd, dw, db, down


@pytest.mark.parametrize("C,r,T,B,n", [(64, 2, 1000, 3, 2), (128, 4, 12000, 2, 2), (64, 2, 24000, 24, 2), (128, 4, 124, 9, 2), (128, 4, 600, 40, 1),
                                        (64, 2, 8, 5, 2), (256, 5, 3000, 3, 2), (256, 5, 3000, 40, 2), (512, 8, 600, 5, 2), (512, 8, 600, 300, 2),
                                        (256, 5, 1160, 5, 2), (512, 8, 232, 7, 2), (256, 5, 20, 70, 2), (512, 8, 8, 33, 1), (256, 5, 60, 1, 1),
                                        (512, 8, 40, 2, 2)])
def test_encoder_stage_offline_equals_blocks_then_down(env, C, r, T, B, n):
    """hilc_encoder_stage, offline: the stage's residual blocks and its down-sampling layer (`seanet.py:316-339`) in ONE launch ==
    hilc_resblock per block followed by hilc_dws_conv (stride r), bit for bit — with and without `res`.  The wide stages (C = 256 /
    r = 5, C = 512 / r = 8: outputs that do not align with 4-column lanes, windows that reach up to 9 columns into the previous
    tile) with few clips (runs start inside clips: warm-up tiles), many, and clips shorter than a tile."""
    ops, fold, O, dev = env
    assert ops.encoder_stage_supported(C, T, n, r, B, streaming=False)
    blocks, singles, wd, dw, db, down = _stage_params(ops, dev, C, r, n, False)
    x = rnd(C + T, B, C, T).to(dev)
    res = rnd(5, B, 2 * C, T // r).to(dev)
    y2 = x
    for single, pre, post in singles:
        y2 = ops.resblock(y2, *single, pre, post)
    ref = ops.dws_conv(y2, wd, dw, db, stride=r, in_scale=0.7746, in_elu=True)
    y = ops.encoder_stage(x, blocks, down)
    assert torch.equal(y, ref), float((y - ref).abs().max())
    yr = ops.encoder_stage(x, blocks, down, res=res)
    assert torch.equal(yr, ref + res)


@pytest.mark.parametrize("C,r,T,B,n", [(64, 2, 320, 5, 2), (128, 4, 160, 7, 2), (64, 2, 320, 1024, 2), (128, 4, 160, 1024, 2), (64, 2, 640, 3, 2),
                                        (128, 4, 8, 21, 2), (64, 2, 12, 70, 1),
                                        # round 6: the hop's wide stages — C = 256 / r = 5 on 32-column carry tiles (runs of 4 whole streams),
                                        # C = 512 / r = 8 on whole-stream tiles; ragged stream counts, several frames per hop, a single block
                                        (256, 5, 40, 7, 2), (256, 5, 40, 1024, 2), (256, 5, 80, 3, 2), (256, 5, 40, 1, 1), (256, 5, 120, 6, 2), (256, 5, 40, 130, 2),
                                        (512, 8, 8, 21, 2), (512, 8, 8, 1024, 2), (512, 8, 16, 5, 2), (512, 8, 32, 3, 1), (512, 8, 8, 1, 2)])
def test_encoder_stage_streaming_equals_blocks_then_down(env, C, r, T, B, n):
    """hilc_encoder_stage, streaming hop (`streaming.py:497-511`): == the blocks one by one (hilc_resblock_stream) followed by
    hilc_dws_conv_stream with the layer's cache: output, the 2n block caches and the down-sampling cache, bit for bit over three
    hops; `res` = the next stage's SpecBlock branch added by the epilogue."""
    ops, fold, O, dev = env
    assert ops.encoder_stage_supported(C, T, n, r, B)
    blocks, singles, wd, dw, db, down = _stage_params(ops, dev, C, r, n, True)
    ca = [[(rnd(7 + j, B, C, 4) * 0.7).to(dev), (rnd(8 + j, B, C, 4) * 0.7).to(dev)] for j in range(n)]
    cb = [[c.clone() for c in pair] for pair in ca]
    da = (rnd(30, B, 2 * C, r) * 0.6).to(dev)
    db_ = da.clone()
    for h in range(3):
        x = rnd(C + T + h, B, C, T).to(dev)
        res = rnd(40 + h, B, 2 * C, T // r).to(dev) if h != 1 else None
        y, flat, da = ops.encoder_stage(x, blocks, down, hist=ca, down_hist=da, res=res)
        ca = [flat[2 * j:2 * j + 2] for j in range(n)]
        y2 = x
        for j, (single, pre, post) in enumerate(singles):
            y2, cb[j] = ops.resblock(y2, *single, pre, post, hist=cb[j])
        ref, db_ = ops.dws_conv_stream(y2, wd, dw, db, db_, res=res, stride=r, in_scale=0.7746, in_elu=True)
        assert torch.equal(y, ref), (h, float((y - ref).abs().max()))
        assert torch.equal(da, db_), h
        for j in range(n):
            assert torch.equal(ca[j][0], cb[j][0]) and torch.equal(ca[j][1], cb[j][1]), (h, j)


@pytest.mark.parametrize("C,r,Tin,B,n", [(768, 8, 1, 37, 3), (768, 8, 1, 1024, 3), (768, 8, 2, 9, 3), (768, 8, 4, 5, 2), (768, 8, 1, 3, 1),
                                          (192, 4, 40, 7, 3), (192, 4, 40, 1024, 3), (96, 2, 160, 5, 3), (96, 2, 160, 1024, 3), (192, 4, 80, 3, 2),
                                          (96, 2, 2, 70, 3), (192, 4, 1, 33, 3), (384, 5, 8, 19, 1), (384, 5, 8, 1024, 1), (384, 5, 8, 1, 1), (384, 5, 16, 5, 1),
                                          (384, 5, 4, 3, 1),
                                          # round 6: the whole C = 384 stage of a hop on 32-column carry tiles (runs of whole streams)
                                          (384, 5, 8, 19, 3), (384, 5, 8, 1024, 3), (384, 5, 8, 1, 3), (384, 5, 16, 5, 2), (384, 5, 4, 3, 3), (384, 5, 24, 6, 3),
                                          (384, 5, 8, 130, 3)])
def test_decoder_stage_streaming_equals_up_conv_then_blocks(env, C, r, Tin, B, n):
    """hilc_decoder_stage (a decoder stage of a streaming hop — `streaming.py:629-639` — in one launch: C = 768 / r = 8 on whole-stream
    tiles, C = 192 / r = 4 and C = 96 / r = 2 in the carry form, C = 384 / r = 5: 32-column carry tiles, runs of whole streams — rounds 4-5: the
    up-sampling layer + the first block on 64-column flat tiles with a halo) == hilc_up_conv_stream followed by the residual blocks (chain;
    C = 384: block by block, hilc_resblock_stream), bit for bit over three hops: output, the up-sampling cache and the 2n block caches."""
    ops, fold, O, dev = env
    T = Tin * r
    assert ops.decoder_stage_supported(C, T, n, r, B)
    blocks, raw = [], []
    for j in range(n):
        w1, w2 = (rnd(10 * j + 1, C, C) / C ** 0.5).to(dev), (rnd(10 * j + 4, C, C) / C ** 0.5).to(dev)
        d1, b1 = (rnd(10 * j + 2, C, 5) * 0.5).to(dev), (rnd(10 * j + 3, C) * 0.2).to(dev)
        d2, b2 = (rnd(10 * j + 5, C, 5) * 0.5).to(dev), (rnd(10 * j + 6, C) * 0.2).to(dev)
        blocks.append((ops.resblock_chain_pack(w1), d1, b1, ops.resblock_chain_pack(w2), d2, b2, 1.0, 0.4 + 0.1 * j))
        raw.append((w1, w2))
    tw = (rnd(80, 2 * C, 2 * r) * 0.3).to(dev)
    wu = (rnd(81, 2 * C, C) / (2 * C) ** 0.5).to(dev)                 # k-major [2C][C]
    bu = (rnd(82, C) * 0.1).to(dev)
    taps = ops.up_conv_taps(tw, r)                      # r = 5: the expanded table
    up = (tw if taps is None else taps, ops.resblock_chain_pack(wu[:C].contiguous()), ops.resblock_chain_pack(wu[C:].contiguous()), bu, 0.7071, r)
    ca = [[(rnd(7 + j, B, C, 4) * 0.7).to(dev), (rnd(8 + j, B, C, 4) * 0.7).to(dev)] for j in range(n)]
    cb = [[c.clone() for c in pair] for pair in ca]
    ua = (rnd(30, B, 2 * C, 1) * 0.6).to(dev)
    ub = ua.clone()
    for h in range(3):
        xin = rnd(100 + h, B, 2 * C, Tin).to(dev)
        y, flat, ua = ops.decoder_stage(xin, up, blocks, ca, ua)
        ca = [flat[2 * j:2 * j + 2] for j in range(n)]
        y2, ub = ops.up_conv(xin, tw, wu, bu, r, in_scale=0.7071, in_elu=True, hist=ub, want_hist=True)
        if n >= 2 and ops.resblock_chain_supported(C, T, n, B):
            y2, f2 = ops.resblock_chain(y2, blocks, cb)
            cb = [f2[2 * j:2 * j + 2] for j in range(n)]
        else:
            for j, blk in enumerate(blocks):
                y2, cb[j] = ops.resblock(y2, ops.resblock_pack(raw[j][0]), blk[1], blk[2], ops.resblock_pack(raw[j][1]), blk[4], blk[5],
                                         1.0, blk[7], hist=cb[j])
        assert torch.equal(y, y2), (h, float((y - y2).abs().max()))
        assert torch.equal(ua, ub), h
        for j in range(n):
            assert torch.equal(ca[j][0], cb[j][0]) and torch.equal(ca[j][1], cb[j][1]), (h, j)


@pytest.mark.parametrize("C,r,Tin,B", [(192, 4, 3000, 3), (96, 2, 12000, 2), (192, 4, 31, 40), (96, 2, 300, 70), (96, 2, 12000, 24),
                                       (768, 8, 75, 3), (768, 8, 75, 300), (768, 8, 2, 50), (768, 8, 301, 2),
                                       (384, 5, 600, 3), (384, 5, 600, 40), (384, 5, 4, 300), (384, 5, 232, 5), (384, 5, 8, 1)])
def test_decoder_stage_offline_equals_up_conv_then_blocks(env, C, r, Tin, B):
    """hilc_decoder_stage with streaming = 0 (`seanet.py:431-452`): the up-sampling layer and the three residual blocks of a narrow
    decoder stage of the OFFLINE model in one launch == hilc_up_conv followed by hilc_resblock block by block.  The widest stage
    (C = 768, r = 8): the up-sampling layer and the FIRST block (the carry slots of a second one do not fit LDS)."""
    ops, fold, O, dev = env
    n = 1 if C == 768 else 3
    assert ops.decoder_stage_supported(C, Tin * r, n, r, B, streaming=False)
    blocks, singles = [], []
    for j in range(n):
        w1, w2 = (rnd(10 * j + 1, C, C) / C ** 0.5).to(dev), (rnd(10 * j + 4, C, C) / C ** 0.5).to(dev)
        d1, b1 = (rnd(10 * j + 2, C, 5) * 0.5).to(dev), (rnd(10 * j + 3, C) * 0.2).to(dev)
        d2, b2 = (rnd(10 * j + 5, C, 5) * 0.5).to(dev), (rnd(10 * j + 6, C) * 0.2).to(dev)
        pre, post = (1.0 + j / 3.0) ** -0.5, 0.4 + 0.1 * j
        blocks.append((ops.resblock_chain_pack(w1, False), d1, b1, ops.resblock_chain_pack(w2, False), d2, b2, pre, post))
        singles.append(((ops.resblock_pack(w1), d1, b1, ops.resblock_pack(w2), d2, b2), pre, post))
    tw = (rnd(80, 2 * C, 2 * r) * 0.3).to(dev)
    wu = (rnd(81, 2 * C, C) / (2 * C) ** 0.5).to(dev)
    bu = (rnd(82, C) * 0.1).to(dev)
    taps = ops.up_conv_taps(tw, r)                       # r = 5: the expanded table
    up = (tw if taps is None else taps, ops.resblock_chain_pack(wu[:C].contiguous(), False), ops.resblock_chain_pack(wu[C:].contiguous(), False), bu, 0.7071, r)
    xin = rnd(100, B, 2 * C, Tin).to(dev)
    y = ops.decoder_stage(xin, up, blocks)
    y2 = ops.up_conv(xin, tw, wu, bu, r, in_scale=0.7071, in_elu=True)
    for single, pre, post in singles:
        y2 = ops.resblock(y2, *single, pre, post)
    assert torch.equal(y, y2), float((y - y2).abs().max())


def _oracle_blocks(O, C, n, seed0=0):
    """n residual blocks as reference state dicts (for O.resblock) and as the op's parameter tuples (k-major weights, not packed)"""
    sds, raw = [], []
    for j in range(n):
        sd = {
            "p.block.1.conv.conv.weight": rnd(seed0 + 10 * j + 1, C, C, 1) / C ** 0.5,
            "p.block.2.conv.conv.weight": rnd(seed0 + 10 * j + 2, C, 1, 5) * 0.5, "p.block.2.conv.conv.bias": rnd(seed0 + 10 * j + 3, C) * 0.2,
            "p.block.4.conv.conv.weight": rnd(seed0 + 10 * j + 4, C, C, 1) / C ** 0.5,
            "p.block.5.conv.conv.weight": rnd(seed0 + 10 * j + 5, C, 1, 5) * 0.5, "p.block.5.conv.conv.bias": rnd(seed0 + 10 * j + 6, C) * 0.2,
            "p.res_scale_param": torch.tensor([0.7 + 0.1 * j]),
        }
        sds.append(sd)
        raw.append((sd["p.block.1.conv.conv.weight"][:, :, 0].t().contiguous(), sd["p.block.2.conv.conv.weight"][:, 0].contiguous(),
                    sd["p.block.2.conv.conv.bias"], sd["p.block.4.conv.conv.weight"][:, :, 0].t().contiguous(),
                    sd["p.block.5.conv.conv.weight"][:, 0].contiguous(), sd["p.block.5.conv.conv.bias"],
                    (1 + j * RS ** 2) ** -0.5, float(RS * sd["p.res_scale_param"][0])))
    return sds, raw


def _chain_params(ops, raw, dev, streaming=False):
    return [(ops.resblock_chain_pack(w1.to(dev), streaming), d1.to(dev), b1.to(dev), ops.resblock_chain_pack(w2.to(dev), streaming), d2.to(dev),
             b2.to(dev), pre, post) for (w1, d1, b1, w2, d2, b2, pre, post) in raw]


@pytest.mark.parametrize("C,T,B,n", [(96, 120, 7, 3), (64, 1000, 3, 2), (384, 300, 2, 3), (512, 28, 5, 2), (192, 132, 300, 3)])
def test_resblock_chain_offline_vs_oracle_composition(env, C, T, B, n):
    """hilc_resblock_chain (offline) against the ORACLE's composition of the stage's blocks (`seanet.py:316-330`: O.resblock n times) — an
    op-level oracle leg for the stage launches, at shapes no model-level golden reaches (ragged batches, T below one tile)."""
    ops, fold, O, dev = env
    assert ops.resblock_chain_supported(C, T, n, B, streaming=False)
    sds, raw = _oracle_blocks(O, C, n)
    x = rnd(C + T, B, C, T)
    ref = x
    for j, sd in enumerate(sds):
        ref = O.resblock(sd, "p", ref, RS, j)
    y = ops.resblock_chain(x.to(dev), _chain_params(ops, raw, dev))
    close(y, ref, 5e-5, f"chain C{C} T{T} B{B}")


@pytest.mark.parametrize("C,r,T,B,n", [(64, 2, 124, 5, 2), (128, 4, 1000, 3, 2), (256, 5, 300, 2, 2), (512, 8, 72, 3, 2), (64, 2, 24, 300, 1)])
def test_encoder_stage_offline_vs_oracle_composition(env, C, r, T, B, n):
    """hilc_encoder_stage (offline) against the oracle: the stage's blocks, then `self.downsample[i]` = [Scale, ELU, 1x1 conv C -> 2C,
    depthwise conv k = 2r stride r] (`seanet.py:330-339`) and the next stage's SpecBlock branch added (`res`)."""
    ops, fold, O, dev = env
    assert ops.encoder_stage_supported(C, T, n, r, B, streaming=False)
    sds, raw = _oracle_blocks(O, C, n)
    x = rnd(C + T + r, B, C, T)
    ref = x
    for j, sd in enumerate(sds):
        ref = O.resblock(sd, "p", ref, RS, j)
    wd = rnd(70, 2 * C, C, 1) / C ** 0.5
    dw, db = rnd(71, 2 * C, 1, 2 * r) * 0.3, rnd(72, 2 * C) * 0.1
    res = rnd(73, B, 2 * C, T // r) * 0.5
    in_scale = (1 + n * RS ** 2) ** -0.5
    ref = O.sconv1d(F.conv1d(F.elu(ref * in_scale), wd), dw, db, stride=r, groups=2 * C) + res
    wt = wd[:, :, 0].t().contiguous().to(dev)                  # k-major [C, 2C]
    down = (ops.resblock_chain_pack(wt[:, :C].contiguous(), False), ops.resblock_chain_pack(wt[:, C:].contiguous(), False),
            dw[:, 0].contiguous().to(dev), db.to(dev), in_scale, r)
    y = ops.encoder_stage(x.to(dev), _chain_params(ops, raw, dev), down, res=res.to(dev))
    close(y, ref, 5e-5, f"encoder stage C{C} r{r} T{T} B{B}")


@pytest.mark.parametrize("C,r,Tin,B", [(192, 4, 31, 5), (96, 2, 300, 3), (768, 8, 9, 2), (384, 5, 24, 3), (96, 2, 14, 300)])
def test_decoder_stage_offline_vs_oracle_composition(env, C, r, Tin, B):
    """hilc_decoder_stage (offline) against the oracle: `[Scale, ELU, depthwise transposed conv k = 2r stride r, 1x1 conv 2C -> C + bias]`
    (`seanet.py:431-436`) followed by the stage's residual blocks (`:437-452`)."""
    ops, fold, O, dev = env
    n = 1 if C == 768 else 3
    assert ops.decoder_stage_supported(C, Tin * r, n, r, B, streaming=False)
    sds, raw = _oracle_blocks(O, C, n)
    tw = rnd(80, 2 * C, 1, 2 * r) * 0.3
    wu = rnd(81, C, 2 * C, 1) / (2 * C) ** 0.5
    bu = rnd(82, C) * 0.1
    xin = rnd(100 + C, B, 2 * C, Tin)
    ref = F.conv1d(O.sconvtr1d(F.elu(xin * 0.7071), tw, None, r, 2 * C), wu, bu)
    for j, sd in enumerate(sds):
        ref = O.resblock(sd, "p", ref, RS, j)
    twd = tw[:, 0].contiguous().to(dev)
    wt = wu[:, :, 0].t().contiguous().to(dev)                  # k-major [2C, C]
    taps = ops.up_conv_taps(twd, r)
    up = (twd if taps is None else taps, ops.resblock_chain_pack(wt[:C].contiguous(), False), ops.resblock_chain_pack(wt[C:].contiguous(), False),
          bu.to(dev), 0.7071, r)
    y = ops.decoder_stage(xin.to(dev), up, _chain_params(ops, raw, dev))
    close(y, ref, 5e-5, f"decoder stage C{C} r{r} Tin{Tin} B{B}")


@pytest.mark.parametrize("Tin,B", [(12000, 2), (300, 7), (14, 300), (62, 5), (6000, 24)])
def test_decoder_stage_post_equals_stage_then_conv_post_and_oracle(env, Tin, B):
    """hilc_decoder_stage_post: the offline decoder's LAST stage (C = 96, r = 2, three blocks) with the closing conv k = 5 C -> 1, the
    final scale and tanh (`seanet.py:453-476`) as the launch's closing phase == hilc_decoder_stage followed by hilc_conv_post, bit
    for bit (same row classes, same order of the partial sums) — ragged batches, T below a tile, runs that start inside a clip
    (warm-up tiles) — and, on the small shapes, the oracle's composition."""
    ops, fold, O, dev = env
    C, r, n = 96, 2, 3
    T = Tin * r
    assert ops.decoder_stage_post_supported(C, T, n, r, 5) and not ops.decoder_stage_post_supported(192, T, n, 4, 5)
    sds, raw = _oracle_blocks(O, C, n)
    blocks = _chain_params(ops, raw, dev)
    tw = rnd(80, 2 * C, 1, 2 * r) * 0.3
    wu = rnd(81, C, 2 * C, 1) / (2 * C) ** 0.5
    bu = rnd(82, C) * 0.1
    pw, pb = rnd(83, 1, C, 5) * 0.2, rnd(84, 1) * 0.1
    xin = rnd(100 + Tin, B, 2 * C, Tin)
    twd = tw[:, 0].contiguous().to(dev)
    wt = wu[:, :, 0].t().contiguous().to(dev)
    up = (twd, ops.resblock_chain_pack(wt[:C].contiguous(), False), ops.resblock_chain_pack(wt[C:].contiguous(), False), bu.to(dev), 0.7071, r)
    post = (pw[0].contiguous().to(dev), pb.to(dev), 0.5, 0.1122, True)
    wav = ops.decoder_stage_post(xin.to(dev), up, blocks, post)
    y = ops.decoder_stage(xin.to(dev), up, blocks)
    wav2 = ops.conv_post(y, post[0], post[1], in_scale=0.5, in_elu=True, out_scale=0.1122, do_tanh=True)
    assert wav.shape == (B, 1, T) and torch.equal(wav, wav2), float((wav - wav2).abs().max())
    if B * T <= 50000:
        ref = F.conv1d(O.sconvtr1d(F.elu(xin * 0.7071), tw, None, r, 2 * C), wu, bu)
        for j, sd in enumerate(sds):
            ref = O.resblock(sd, "p", ref, RS, j)
        ref = torch.tanh(O.sconv1d(F.elu(ref * 0.5), pw, pb) * 0.1122)
        close(wav, ref, 2e-5, f"decoder stage + conv_post Tin{Tin} B{B}")


@pytest.mark.parametrize("Tin,B", [(160, 5), (160, 1024), (160, 1), (2, 70), (80, 7), (320, 3), (6, 33)])
def test_decoder_stage_post_streaming_equals_stage_then_conv_post(env, Tin, B):
    """hilc_decoder_stage_post(streaming = 1), round 6: the LAST decoder stage of a hop (C = 96, r = 2, three blocks; `streaming.py:639-648`) with the
    closing conv as the closing phase of the launch == hilc_decoder_stage(streaming) followed by hilc_conv_post with the conv's cache, bit for
    bit over three hops: the waveform, the up-sampling cache, the six block caches and the closing conv's cache — ragged stream counts,
    hops shorter than a tile, several frames per hop, the full 1 024 streams."""
    ops, fold, O, dev = env
    C, r, n = 96, 2, 3
    T = Tin * r
    assert ops.decoder_stage_post_supported(C, T, n, r, 5) and ops.decoder_stage_supported(C, T, n, r, B)
    blocks = []
    for j in range(n):
        w1, w2 = (rnd(10 * j + 1, C, C) / C ** 0.5).to(dev), (rnd(10 * j + 4, C, C) / C ** 0.5).to(dev)
        d1, b1 = (rnd(10 * j + 2, C, 5) * 0.5).to(dev), (rnd(10 * j + 3, C) * 0.2).to(dev)
        d2, b2 = (rnd(10 * j + 5, C, 5) * 0.5).to(dev), (rnd(10 * j + 6, C) * 0.2).to(dev)
        blocks.append((ops.resblock_chain_pack(w1), d1, b1, ops.resblock_chain_pack(w2), d2, b2, 1.0, 0.4 + 0.1 * j))
    tw = (rnd(80, 2 * C, 2 * r) * 0.3).to(dev)
    wu = (rnd(81, 2 * C, C) / (2 * C) ** 0.5).to(dev)
    bu = (rnd(82, C) * 0.1).to(dev)
    up = (tw, ops.resblock_chain_pack(wu[:C].contiguous()), ops.resblock_chain_pack(wu[C:].contiguous()), bu, 0.7071, r)
    post = ((rnd(83, C, 5) * 0.2).to(dev), (rnd(84, 1) * 0.1).to(dev), 0.5, 0.1122, True)
    ca = [[(rnd(7 + j, B, C, 4) * 0.7).to(dev), (rnd(8 + j, B, C, 4) * 0.7).to(dev)] for j in range(n)]
    cb = [[c.clone() for c in pair] for pair in ca]
    ua = (rnd(30, B, 2 * C, 1) * 0.6).to(dev)
    ub = ua.clone()
    pa = (rnd(31, B, C, 4) * 0.6).to(dev)
    pb_ = pa.clone()
    for h in range(3):
        xin = rnd(100 + h, B, 2 * C, Tin).to(dev)
        if h == 1:      # caller-owned destinations (the ping-pong state block)
            outs = [[torch.full_like(c, 9.0) for c in pair] for pair in ca]
            uo, po = torch.full_like(ua, 9.0), torch.full_like(pa, 9.0)
            wav, flat, ua, pa = ops.decoder_stage_post(xin, up, blocks, post, ca, ua, pa, hist_out=outs, up_hist_out=uo, post_hist_out=po)
            assert ua is uo and pa is po and all(flat[2 * j] is outs[j][0] and flat[2 * j + 1] is outs[j][1] for j in range(n))
        else:
            wav, flat, ua, pa = ops.decoder_stage_post(xin, up, blocks, post, ca, ua, pa)
        ca = [flat[2 * j:2 * j + 2] for j in range(n)]
        y, f2, ub = ops.decoder_stage(xin, up, blocks, cb, ub)
        cb = [f2[2 * j:2 * j + 2] for j in range(n)]
        wav2, pb_ = ops.conv_post(y, post[0], post[1], in_scale=0.5, in_elu=True, out_scale=0.1122, do_tanh=True, hist=pb_, want_hist=True)
        assert wav.shape == (B, 1, T) and torch.equal(wav, wav2), (h, float((wav - wav2).abs().max()))
        assert torch.equal(ua, ub) and torch.equal(pa, pb_), h
        for j in range(n):
            assert torch.equal(ca[j][0], cb[j][0]) and torch.equal(ca[j][1], cb[j][1]), (h, j)


@pytest.mark.parametrize("T,B,n", [(24000, 3, 2), (1000, 5, 2), (124, 40, 2), (8, 300, 1), (9280, 24, 2)])
def test_encoder_stage0_equals_conv_pre_spec_then_stage(env, T, B, n):
    """hilc_encoder_stage0: the offline encoder's first conv and stage-0 SpecBlock (`seanet.py:280-286, 220-246`) as the opening phase of the
    C = 64 stage launch == hilc_spec_block_conv_pre followed by hilc_encoder_stage, bit for bit — clips shorter than a tile, ragged
    batches, runs that start inside a clip (warm-up tiles), with and without the biases and a shortcut behind the down-sampling layer."""
    ops, fold, O, dev = env
    C, r, n_fft = 64, 2, 64
    assert ops.encoder_stage0_supported(T, n, r, 64, 1, 5) and not ops.encoder_stage0_supported(T, n, 4, 64, 1, 5)
    basis = synth.stft_basis(n_fft)
    bt = fold.stft_basis_layout(basis).to(dev)
    w = rnd(n_fft + 1, C, n_fft // 2 + 1, 1) / (n_fft // 2 + 1) ** 0.5
    wt = fold.pointwise_layout(w).to(dev)
    bias = (rnd(n_fft + 2, C) * 0.1).to(dev)
    dft_p, nyq, pw_p = ops.spec_block_tables(bt, wt, n_fft)
    pre_w, pre_b = (rnd(71, 64, 5) * 0.5).to(dev), (rnd(72, 64) * 0.1).to(dev)
    sds, raw = _oracle_blocks(O, C, n)
    blocks = _chain_params(ops, raw, dev)
    wd = (rnd(70, C, 2 * C) / C ** 0.5).to(dev)                 # k-major [C, 2C]
    down = (ops.resblock_chain_pack(wd[:, :C].contiguous(), False), ops.resblock_chain_pack(wd[:, C:].contiguous(), False),
            (rnd(73, 2 * C, 2 * r) * 0.3).to(dev), (rnd(74, 2 * C) * 0.1).to(dev), (1 + n * RS ** 2) ** -0.5, r)
    wav = synth.synth_clips(B, T, seed=T + B).to(dev)
    res = (rnd(75, B, 2 * C, T // r) * 0.5).to(dev)
    for sb, pb, rs in ((bias, pre_b, None), (None, None, res), (bias, None, res)):
        spec = (dft_p, nyq, pw_p, sb, pre_w, pb, 1 / 0.1122080159, -4.0, 2.8, True, 0.37)
        y = ops.encoder_stage0(wav, spec, blocks, down, res=rs)
        x0 = ops.spec_block_conv_pre(wav, dft_p, nyq, pw_p, sb, pre_w, pb, 1 / 0.1122080159, n_fft, 1, -4.0, 2.8, True, 0.37)
        y2 = ops.encoder_stage(x0, blocks, down, res=rs)
        assert y.shape == (B, 2 * C, T // r) and torch.equal(y, y2), float((y - y2).abs().max())


@pytest.mark.parametrize("T,B,n", [(8, 300, 1), (372, 5, 2), (124, 33, 2)])
def test_encoder_stage0_vs_oracle_composition(env, T, B, n):
    """hilc_encoder_stage0 against the ORACLE at shapes no model-level golden reaches (clips shorter than a tile and ragged batches, a clip
    that ends inside a tile, runs that start inside a clip): first conv `Scale(1 / wav_std) -> SConv1d(1, 64, 5)` (`seanet.py:280-286`), stage 0's
    SpecBlock (`:220-246`), the stage's residual blocks (`:129-148`), `self.downsample[0]` (`:330-339`) and the next stage's branch."""
    ops, fold, O, dev = env
    C, r, n_fft = 64, 2, 64
    assert ops.encoder_stage0_supported(T, n, r, 64, 1, 5)
    basis = synth.stft_basis(n_fft)
    bt = fold.stft_basis_layout(basis).to(dev)
    w = rnd(n_fft + 1, C, n_fft // 2 + 1, 1) / (n_fft // 2 + 1) ** 0.5
    bias = rnd(n_fft + 2, C) * 0.1
    dft_p, nyq, pw_p = ops.spec_block_tables(bt, fold.pointwise_layout(w).to(dev), n_fft)
    pre_w, pre_b = rnd(71, 64, 1, 5) * 0.5, rnd(72, 64) * 0.1
    sds, raw = _oracle_blocks(O, C, n)
    wd = rnd(70, 2 * C, C, 1) / C ** 0.5
    dw, db = rnd(73, 2 * C, 1, 2 * r) * 0.3, rnd(74, 2 * C) * 0.1
    wav = synth.synth_clips(B, T, seed=T + B)
    res = rnd(75, B, 2 * C, T // r) * 0.5
    isc, in_scale = 1 / 0.1122080159, (1 + n * RS ** 2) ** -0.5
    # the oracle: conv_pre, x + scale * (W log|STFT| + b), blocks, [Scale, ELU, 1x1 conv, strided depthwise conv] + the next branch
    ref = O.sconv1d(wav * isc, pre_w, pre_b)
    mag = O.causal_stft_mag(wav, basis, 1, pad=True, clamp=True)
    ref = ref + (F.conv1d((torch.log(mag.clamp_min(1e-5)) - (-4.0)) / 2.8, w) + bias.view(1, -1, 1)) * 0.37
    for j, sd in enumerate(sds):
        ref = O.resblock(sd, "p", ref, RS, j)
    ref = O.sconv1d(F.conv1d(F.elu(ref * in_scale), wd), dw, db, stride=r, groups=2 * C) + res
    wt = wd[:, :, 0].t().contiguous().to(dev)
    down = (ops.resblock_chain_pack(wt[:, :C].contiguous(), False), ops.resblock_chain_pack(wt[:, C:].contiguous(), False),
            dw[:, 0].contiguous().to(dev), db.to(dev), in_scale, r)
    spec = (dft_p, nyq, pw_p, bias.to(dev), pre_w[:, 0].contiguous().to(dev), pre_b.to(dev), isc, -4.0, 2.8, True, 0.37)
    y = ops.encoder_stage0(wav.to(dev), spec, _chain_params(ops, raw, dev), down, res=res.to(dev))
    close(y, ref, 2e-4, f"encoder stage 0 (conv_pre + SpecBlock + blocks + down) T{T} B{B}")


@pytest.mark.parametrize("T,B,n", [(320, 5, 2), (320, 1024, 2), (320, 1, 2), (640, 3, 2), (128, 70, 1), (132, 9, 2), (960, 2, 2)])
def test_encoder_stage0_streaming_equals_conv_pre_spec_then_stage(env, T, B, n):
    """hilc_encoder_stage0(streaming = 1), round 6: a hop's first conv + stage-0 SpecBlock (`streaming.py:490-497`) as the opening phase of the C = 64
    stage launch == hilc_spec_block_conv_pre(hist) followed by hilc_encoder_stage(streaming), bit for bit over three hops with the waveform
    history carried from hop to hop: output, the block caches, the down-sampling cache — tiles that hold a stream's t = 0 (two waveform pieces),
    ragged stream counts, several frames per hop, hops that are no multiple of the tile, with and without biases / shortcut / history."""
    ops, fold, O, dev = env
    C, r, n_fft, H = 64, 2, 64, 1023
    assert ops.encoder_stage0_supported(T, n, r, 64, 1, 5, B, True) and not ops.encoder_stage0_supported(64, n, r, 64, 1, 5, B, True)
    basis = synth.stft_basis(n_fft)
    bt = fold.stft_basis_layout(basis).to(dev)
    w = rnd(n_fft + 1, C, n_fft // 2 + 1, 1) / (n_fft // 2 + 1) ** 0.5
    wt = fold.pointwise_layout(w).to(dev)
    bias = (rnd(n_fft + 2, C) * 0.1).to(dev)
    dft_p, nyq, pw_p = ops.spec_block_tables(bt, wt, n_fft)
    pre_w, pre_b = (rnd(71, 64, 5) * 0.5).to(dev), (rnd(72, 64) * 0.1).to(dev)
    blocks, singles, wd, dw, db, down = _stage_params(ops, dev, C, r, n, True)
    ca = [[(rnd(7 + j, B, C, 4) * 0.7).to(dev), (rnd(8 + j, B, C, 4) * 0.7).to(dev)] for j in range(n)]
    cb = [[c.clone() for c in pair] for pair in ca]
    da = (rnd(30, B, 2 * C, r) * 0.6).to(dev)
    db_ = da.clone()
    clip = synth.synth_clips(B, 3 * T + H, seed=T + B).to(dev)
    for h, (sb, pb, use_res, use_hist) in enumerate(((bias, pre_b, True, True), (None, None, False, True), (bias, None, True, False))):
        wav = clip[:, :, H + h * T: H + (h + 1) * T].contiguous()
        hist = clip[:, :, h * T: H + h * T].contiguous() if use_hist else None      # the 1023 samples in front of the hop (spec_post's window)
        res = (rnd(75 + h, B, 2 * C, T // r) * 0.5).to(dev) if use_res else None
        spec = (dft_p, nyq, pw_p, sb, pre_w, pb, 1 / 0.1122080159, -4.0, 2.8, True, 0.37)
        y, flat, da = ops.encoder_stage0(wav, spec, blocks, down, res=res, hist=ca, down_hist=da, wav_hist=hist)
        ca = [flat[2 * j:2 * j + 2] for j in range(n)]
        x0 = ops.spec_block_conv_pre(wav, dft_p, nyq, pw_p, sb, pre_w, pb, 1 / 0.1122080159, n_fft, 1, -4.0, 2.8, True, 0.37, hist=hist)
        y2, f2, db_ = ops.encoder_stage(x0, blocks, down, hist=cb, down_hist=db_, res=res)
        cb = [f2[2 * j:2 * j + 2] for j in range(n)]
        assert y.shape == (B, 2 * C, T // r) and torch.equal(y, y2), (h, float((y - y2).abs().max()))
        assert torch.equal(da, db_), h
        for j in range(n):
            assert torch.equal(ca[j][0], cb[j][0]) and torch.equal(ca[j][1], cb[j][1]), (h, j)


def test_resblock_chain_shapes_it_does_not_take(env):
    ops, fold, O, dev = env
    from hilcodec_amd._lib import lib
    assert lib.hilc_resblock_chain_supported(768, 40, 3, 1) == 0 and lib.hilc_resblock_chain_supported(384, 40, 3, 1) == 0 and lib.hilc_resblock_chain_supported(256, 40, 2, 1) == 0
    assert lib.hilc_resblock_chain_supported(96, 320, 1, 1) == 0 and lib.hilc_resblock_chain_supported(96, 320, 4, 1) == 0
    assert lib.hilc_resblock_chain_supported(96, 320, 3, 0) == 1 and lib.hilc_resblock_chain_supported(96, 322, 3, 1) == 0
    assert lib.hilc_resblock_chain_supported(768, 8, 3, 0) == 0 and lib.hilc_resblock_chain_supported(64, 320, 3, 0) == 0
    assert lib.hilc_resblock_chain_supported(64, 320, 3, 1) == 0 and lib.hilc_resblock_chain_supported(512, 8, 3, 1) == 0      # 2-block instantiations
    assert not ops.resblock_chain_supported(64, 320, 3, 2) and ops.resblock_chain_supported(64, 320, 2, 2)
    assert not ops.resblock_chain_supported(384, 40, 3, 8) and not ops.resblock_chain_supported(256, 40, 2, 8) and ops.resblock_chain_supported(256, 3000, 2, 8, streaming=False) and not ops.resblock_chain_supported(96, 320, 3, 40000)


def test_wide_stream_block_shapes_it_does_not_take(env):
    """C = 512 / 768 need whole streams per 32-column tile: other hop lengths are refused (the engine then runs two launches)"""
    ops, fold, O, dev = env
    from hilcodec_amd._lib import lib
    assert lib.hilc_resblock_stream_supported(768, 8) == 1 and lib.hilc_resblock_stream_supported(768, 40) == 0
    assert lib.hilc_resblock_stream_supported(384, 40) == 1 and lib.hilc_resblock_stream_supported(640, 8) == 0
    assert not ops.resblock_supported(768, 40, 4, streaming=True) and not ops.resblock_supported(1024, 8, 4, streaming=False)
    C, T, B = 768, 40, 2
    w = (rnd(1, C, C) / C ** 0.5).to(dev)
    d, b = (rnd(2, C, 5) * 0.5).to(dev), (rnd(3, C) * 0.2).to(dev)
    z = [torch.zeros(B, C, 4, device=dev), torch.zeros(B, C, 4, device=dev)]
    with pytest.raises(Exception):
        ops.resblock(rnd(5, B, C, T).to(dev), ops.resblock_pack(w), d, b, ops.resblock_pack(w), d, b, 0.9, 0.4, hist=z)


def test_elu_fast_error(env):
    """The hot-path ELU (2^(x log2 e) - 1 via v_exp_f32) against expm1 in fp64 on a dense grid:
    absolute error bounded by one fp32 ulp of an O(1) activation."""
    ops, fold, O, dev = env
    x = torch.cat([torch.linspace(-30, 0, 200001), torch.linspace(-1e-3, 1e-3, 20001), torch.linspace(0, 8, 1001),
                   -torch.logspace(-30, 1, 5001)]).float()
    n = x.numel() - x.numel() % 4
    x = x[:n].view(1, 1, n)
    y = ops.dw_conv(x.to(dev), torch.ones(1, 1, device=dev), in_elu=True).cpu().double()
    ref = torch.where(x.double() > 0, x.double(), torch.expm1(x.double()))
    err = (y - ref).abs().max().item()
    assert err <= 1.3e-7, err
    assert torch.equal(y[x.double() > 0].float(), x[x > 0])      # identity on the positive side


@pytest.mark.parametrize("K,M,Tin,r", [(1536, 768, 75, 8), (768, 384, 60, 5), (384, 192, 301, 4), (192, 96, 1000, 2),
                                       (40, 64, 24, 5), (24, 32, 16, 3), (33, 96, 8, 4),
                                       (64, 32, 4, 5)])
def test_up_conv_vs_oracle(env, K, M, Tin, r):
    """fused [Scale, ELU, depthwise ConvTranspose k=2r, 1x1 conv + bias] against the three-step oracle"""
    ops, fold, O, dev = env
    x = rnd(K + Tin, 2, K, Tin)
    tw = rnd(K + r, K, 1, 2 * r)
    w = rnd(K + M, M, K, 1) / K ** 0.5
    b = rnd(M, M) * 0.1
    ref = F.conv1d(O.sconvtr1d(F.elu(x * 0.7071), tw, None, stride=r, groups=K), w, b)
    y = ops.up_conv(x.to(dev), tw[:, 0].contiguous().to(dev), fold.pointwise_layout(w).to(dev), b.to(dev), r,
                    in_scale=0.7071, in_elu=True)
    close(y, ref, 3e-5, "up_conv")


@pytest.mark.parametrize("K,M,Tn,k,s,B", [(256, 256, 40, 5, 1, 7), (512, 512, 8, 5, 1, 33), (128, 1536, 1, 5, 1, 130),
                                           (256, 512, 40, 10, 5, 5), (512, 1024, 8, 16, 8, 20), (96, 64, 128, 4, 2, 3),
                                           (64, 128, 12, 5, 1, 11),
                                           # hops longer than one tile (the encoder's first two down-sampling layers)
                                           (64, 128, 320, 4, 2, 5), (128, 256, 160, 8, 4, 6), (96, 192, 132, 4, 2, 3),
                                           (64, 96, 640, 10, 5, 2),
                                           # the same on many streams: flat tiles straddle one or two stream boundaries, ragged last tile
                                           (128, 256, 160, 8, 4, 131), (64, 128, 320, 4, 2, 77), (64, 128, 136, 16, 8, 40), (128, 256, 160, 8, 4, 1)])
def test_dws_conv_stream_vs_unfused_and_offline(env, K, M, Tn, k, s, B):
    """hilc_dws_conv_stream (whole-clip tiles, or flat stream-major halo tiles for T > 128; cache-aware epilogues) over 3 hops: against the pointwise GEMM +
    cached depthwise conv (the already oracle-pinned streaming ops), and — concatenated — against the oracle's
    offline causal conv of the whole signal (causal_layers.py:147-165 cache semantics)."""
    ops, fold, O, dev = env
    hops = 3
    w = rnd(K + M, M, K, 1) / K ** 0.5
    dw = rnd(K + k, M, 1, k) * 0.4
    db = rnd(M, M) * 0.2
    x = rnd(K + Tn + B, B, K, Tn * hops)
    res = rnd(7, B, M, (Tn // s) * hops) if s == 1 else None
    wt, dww, dbd = fold.pointwise_layout(w).to(dev), dw[:, 0].contiguous().to(dev), db.to(dev)
    cache_f = torch.zeros(B, M, k - s, device=dev)
    cache_u = torch.zeros(B, M, k - s, device=dev)
    outs = []
    for h in range(hops):
        xh = x[:, :, h * Tn:(h + 1) * Tn].contiguous().to(dev)
        rh = res[:, :, h * Tn:(h + 1) * Tn].contiguous().to(dev) if res is not None else None
        y, cache_f = ops.dws_conv_stream(xh, wt, dww, dbd, cache_f, res=rh, stride=s, in_scale=0.9, in_elu=True,
                                         out_scale=0.5 if s == 1 else 1.0, out_elu=(s == 1))
        hp = ops.pw_conv(xh, wt, in_scale=0.9, in_elu=True)
        yu, cache_u = ops.dw_conv(hp, dww, dbd, res=rh, stride=s, hist=cache_u, want_hist=True,
                                  out_scale=0.5 if s == 1 else 1.0, out_elu=(s == 1))
        assert torch.equal(y, yu), f"hop {h}"
        assert torch.equal(cache_f, cache_u), f"cache after hop {h}"
        outs.append(y)
    full = O.sconv1d(F.conv1d(F.elu(x * 0.9), w), dw, db, stride=s, groups=M)
    if s == 1:
        full = F.elu(full * 0.5 + res)
    close(torch.cat(outs, dim=2), full, 3e-5, "streamed vs offline oracle")
    assert not ops.dws_conv_stream_supported(160, 5, 1) and not ops.dws_conv_stream_supported(40, 16, 16)
    assert ops.dws_conv_stream_supported(320, 4, 2) and not ops.dws_conv_stream_supported(322, 4, 2)


@pytest.mark.parametrize("K,M,Tin,r,B", [(1536, 768, 1, 8, 9), (768, 384, 8, 5, 4), (384, 192, 40, 4, 3), (192, 96, 160, 2, 2)])
def test_up_conv_stream_vs_unfused_and_offline(env, K, M, Tin, r, B):
    """hilc_up_conv_stream over 3 hops against hilc_dw_convtr (with cache) + hilc_pw_conv, and concatenated against
    the offline fused op on the whole signal (causal_layers.py:168-188: cache = last activated input frame)."""
    ops, fold, O, dev = env
    hops = 3
    x = rnd(K + Tin, B, K, Tin * hops).to(dev)
    tw = rnd(K + r, K, 2 * r).to(dev)
    wt = fold.pointwise_layout(rnd(K + M, M, K, 1) / K ** 0.5).to(dev)
    b = (rnd(M, M) * 0.1).to(dev)
    cf = torch.zeros(B, K, 1, device=dev)
    cu = torch.zeros(B, K, 1, device=dev)
    outs = []
    for h in range(hops):
        xh = x[:, :, h * Tin:(h + 1) * Tin].contiguous()
        y, cf = ops.up_conv(xh, tw, wt, b, r, in_scale=0.7071, in_elu=True, hist=cf, want_hist=True)
        u, cu = ops.dw_convtr(xh, tw, r, hist=cu, want_hist=True, in_scale=0.7071, in_elu=True)
        yu = ops.pw_conv(u, wt, b)
        assert torch.equal(cf, cu)
        close(y, yu.cpu(), 1e-6, f"hop {h}")          # same products; the fused loader adds them in one fmaf
        outs.append(y)
    if (Tin * hops * r) % 4 == 0:
        full = ops.up_conv(x, tw, wt, b, r, in_scale=0.7071, in_elu=True)
        assert torch.equal(torch.cat(outs, dim=2), full)


@pytest.mark.parametrize("n_fft,hop,B,T", [(64, 1, 2, 1000), (128, 2, 2, 2400), (256, 8, 3, 4000), (64, 1, 1, 128),
                                           (128, 2, 1, 24), (256, 8, 1, 24000),
                                           # round 6: short clips tile the FLAT frame space (a streaming hop: 40 / 160 / 320 frames per stream; tiles that
                                           # touch 2 - 5 streams, a last tile that ends inside the batch, several frames' worth of hop)
                                           (256, 8, 37, 320), (256, 8, 1024, 320), (128, 2, 33, 320), (128, 2, 1024, 320), (64, 1, 9, 320),
                                           (256, 8, 5, 1280), (128, 2, 3, 640), (256, 8, 2, 288), (256, 8, 70, 256)])
def test_fused_spec_block_equals_unfused_and_oracle(env, n_fft, hop, B, T):
    """hilc_spec_block (STFT -> log-magnitude -> normalise -> 1x1 conv -> += in one launch, exactly n_fft DFT rows)
    against hilc_stft_logmag + hilc_pw_conv bit for bit, and against the oracle's spec_block()."""
    ops, fold, O, dev = env
    C = n_fft
    nb = n_fft // 2 + 1
    basis = synth.stft_basis(n_fft)
    bt = fold.stft_basis_layout(basis).to(dev)
    w = rnd(n_fft + 1, C, nb, 1) / nb ** 0.5
    wt = fold.pointwise_layout(w).to(dev)
    bias = (rnd(n_fft + 2, C) * 0.1).to(dev)
    wav = synth.synth_clips(B, T, seed=n_fft + T)
    Tf = (T - 1) // hop + 1
    x = rnd(n_fft + 3, B, C, Tf)
    assert ops.spec_block_supported(n_fft, hop, C, T)
    dft_p, nyq, pw_p = ops.spec_block_tables(bt, wt, n_fft)
    for b_ in (bias, None):
        y = ops.spec_block(wav.to(dev), dft_p, nyq, pw_p, b_, x.to(dev), n_fft, hop, -4.0, 2.8, True, 0.37)
        s = ops.stft_logmag(wav.to(dev), bt, n_fft, hop, -4.0, 2.8, True)
        y2 = ops.pw_conv(s, wt, b_, res=x.to(dev), out_scale=0.37)
        assert torch.equal(y, y2), f"fused != unfused: {(y - y2).abs().max().item():.3e}"
    # un-normalised log-magnitude (streaming model's merged form) and plain magnitude
    for mode in (0, 2):
        y = ops.spec_block(wav.to(dev), dft_p, nyq, pw_p, None, x.to(dev), n_fft, hop, 0.0, 1.0, mode, 1.0)
        y2 = ops.pw_conv(ops.stft_logmag(wav.to(dev), bt, n_fft, hop, 0.0, 1.0, mode), wt, None, res=x.to(dev))
        assert torch.equal(y, y2)
    # oracle: log(max(|STFT|, 1e-5)), (. - mean) / std, 1x1 conv, x + scale * y
    mag = O.causal_stft_mag(wav, basis, hop, pad=True, clamp=True)
    spec = (torch.log(mag.clamp_min(1e-5)) - (-4.0)) / 2.8
    ref = x + (F.conv1d(spec, w) + bias.cpu().view(1, -1, 1)) * 0.37
    y = ops.spec_block(wav.to(dev), dft_p, nyq, pw_p, bias, x.to(dev), n_fft, hop, -4.0, 2.8, True, 0.37)
    close(y, ref, 2e-4, "fused SpecBlock vs oracle")
    assert not ops.spec_block_supported(512, 40, 512, T) and not ops.spec_block_supported(n_fft, hop, C, 4 * hop + 1)
    # streaming hop: the samples before t = 0 come from the waveform cache (1023 samples in the codec)
    hist = synth.synth_clips(B, 1023, seed=n_fft + 5).to(dev)
    y_h = ops.spec_block(wav.to(dev), dft_p, nyq, pw_p, bias, x.to(dev), n_fft, hop, -4.0, 2.8, 0, 0.37, hist=hist)
    s_h = ops.stft_logmag(wav.to(dev), bt, n_fft, hop, -4.0, 2.8, 0, hist=hist)
    assert torch.equal(y_h, ops.pw_conv(s_h, wt, bias, res=x.to(dev), out_scale=0.37))
    assert not torch.equal(y_h, ops.spec_block(wav.to(dev), dft_p, nyq, pw_p, bias, x.to(dev), n_fft, hop, -4.0, 2.8, 0, 0.37))
    if n_fft == 64:
        # the waveform cache holds RAW samples: both kernels scale them like the hop's own samples (the streaming model
        # merges 1/wav_std into the weights, `streaming.py:472-480`, so it calls with 1.0; 2.5 exercises the general case)
        for isc in (1.0, 2.5):
            x0h = ops.conv_pre(wav.to(dev), (rnd(71, 64, 5) * 0.5).to(dev), None, in_scale=isc, hist=hist)
            y_1 = ops.spec_block_conv_pre(wav.to(dev), dft_p, nyq, pw_p, bias, (rnd(71, 64, 5) * 0.5).to(dev), None, isc, n_fft,
                                          hop, -4.0, 2.8, 0, 0.37, hist=hist)
            assert torch.equal(y_1, ops.spec_block(wav.to(dev), dft_p, nyq, pw_p, bias, x0h, n_fft, hop, -4.0, 2.8, 0, 0.37, hist=hist))
        # and against the oracle's causal conv over [cache | hop]
        full = torch.cat([hist.cpu()[:, :, -4:], wav], dim=2) * 2.5
        ref0 = F.conv1d(full, (rnd(71, 64, 5) * 0.5).view(64, 1, 5))
        close(x0h, ref0, 2e-5, "conv_pre with a raw waveform cache")
    if n_fft == 64:
        # first encoder stage: conv_pre computed inside the SpecBlock launch == conv_pre launch + SpecBlock launch
        pw_, pb_ = (rnd(71, 64, 5) * 0.5).to(dev), (rnd(72, 64) * 0.1).to(dev)
        for pb in (pb_, None):
            x0 = ops.conv_pre(wav.to(dev), pw_, pb, in_scale=1 / 0.1122080159)
            y_two = ops.spec_block(wav.to(dev), dft_p, nyq, pw_p, bias, x0, n_fft, hop, -4.0, 2.8, True, 0.37)
            y_one = ops.spec_block_conv_pre(wav.to(dev), dft_p, nyq, pw_p, bias, pw_, pb, 1 / 0.1122080159, n_fft, hop,
                                            -4.0, 2.8, True, 0.37)
            assert torch.equal(y_one, y_two), f"{(y_one - y_two).abs().max().item():.3e}"


def test_non_finite_sample_stays_in_its_clip(env):
    """An Inf in the first samples of clip 0 must not reach any other clip: the zero padding in front of a clip's first tile is
    `0 * x` of a mapped dummy group, which used to be x[0, k, 0:4] for EVERY clip (0 * Inf = NaN in every clip's halo); the
    dummy is now the clip's own edge group, so — as in the reference — a bad clip only spoils itself."""
    ops, fold, O, dev = env
    B, K, M, Tn = 3, 128, 128, 248
    x = rnd(11, B, K, Tn).to(dev)
    w = (rnd(12, K, M) / K ** 0.5).to(dev)
    dw, db = (rnd(13, M, 5) * 0.5).to(dev), (rnd(14, M) * 0.2).to(dev)
    for kw in (dict(in_scale=1.0, in_elu=False), dict(in_scale=0.8, in_elu=True, out_elu=True)):
        good = ops.dws_conv(x, w, dw, db, **kw)
        bad_x = x.clone()
        bad_x[0, 5, 1] = float("inf")
        bad = ops.dws_conv(bad_x, w, dw, db, **kw)
        assert torch.equal(bad[1:], good[1:]) and torch.isfinite(bad[1:]).all()
        assert not torch.isfinite(bad[0]).all()                      # the bad clip itself is spoiled, as in the reference
    # strided (down-sampling) form: same loader
    dws, dbs = (rnd(15, M, 8) * 0.3).to(dev), (rnd(16, M) * 0.1).to(dev)
    good = ops.dws_conv(x, w, dws, dbs, stride=4, in_scale=0.9, in_elu=True)
    bad_x = x.clone()
    bad_x[0, 0, 0] = float("nan")
    bad = ops.dws_conv(bad_x, w, dws, dbs, stride=4, in_scale=0.9, in_elu=True)
    assert torch.equal(bad[1:], good[1:])


@pytest.mark.parametrize("K,M,Tn,B,in_elu,out_elu", [(128, 256, 496, 120, True, True), (96, 128, 500, 96, False, False),
                                                       (256, 384, 248, 240, True, False), (64, 128, 372, 160, False, True)])
def test_wave_row_form_equals_column_block_form(env, K, M, Tn, B, in_elu, out_elu):
    """The two tile forms of hilc_dws_conv (k5, stride 1, no shortcut): a launch big enough for 128-row tiles runs the wave-row
    form (accumulators as D[time][channel], depthwise taps in registers); the same clips one at a time run the column-block form
    with the LDS epilogue.  Same products in the same order: the outputs must be equal bit for bit — full tiles, a clip's ragged
    last tile, the zero padding in front of a clip, with and without the ELUs."""
    ops, fold, O, dev = env
    from hilcodec_amd._lib import lib
    assert lib.hilc_dws_conv_wave_row(B, M, Tn, 0) == 1 and lib.hilc_dws_conv_wave_row(1, M, Tn, 0) == 0
    assert lib.hilc_dws_conv_wave_row(B, M, Tn, 1) == 0 and lib.hilc_dws_conv_wave_row(B, M + 32, Tn, 0) == 0
    x = rnd(K + Tn, B, K, Tn).to(dev)
    w = (rnd(1, K, M) / K ** 0.5).to(dev)
    dw, db = (rnd(2, M, 5) * 0.5).to(dev), (rnd(3, M) * 0.2).to(dev)
    kw = dict(in_scale=0.83, in_elu=in_elu, out_scale=0.7, out_elu=out_elu)
    big = ops.dws_conv(x, w, dw, db, **kw)
    one = torch.cat([ops.dws_conv(x[b:b + 1].contiguous(), w, dw, db, **kw) for b in range(B)])
    assert torch.equal(big, one)
    ref = O.sconv1d(torch.nn.functional.conv1d(F.elu(x[:2].cpu() * 0.83) if in_elu else x[:2].cpu() * 0.83, w.cpu().t().unsqueeze(-1)),
                    dw.cpu().unsqueeze(1), db.cpu(), groups=M) * 0.7
    close(big[:2], F.elu(ref) if out_elu else ref, 2e-5, "wave-row dws vs oracle")


def test_offline_stage_launches_random_shapes():
    """tools/fuzz_stage_launches.py: 150 random shapes (clip counts and lengths around the tile widths, clips shorter than a tile, single clips)
    of the round-4 offline launches — wide one-launch blocks, chains, encoder stages, decoder stages — against the launches they replace,
    bit for bit.  (1 600 cases over four seeds ran clean on the final kernels.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_stage_launches.py"), "150", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches / errors: 0" in r.stdout


def test_streaming_stage_launches_random_shapes():
    """tools/fuzz_stream_launches.py: 100 random cases (1 ... 1100 streams, hop lengths around the tile widths) of a hop's chains, encoder
    stages and decoder stages against the launches they replace — outputs and every cache over two hops, bit for bit.  (850 cases over
    three seeds ran clean on the final kernels.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_stream_launches.py"), "100", "9"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches / errors: 0" in r.stdout
