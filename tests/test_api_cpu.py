"""CPU: host-side logic — the C-ABI library loads and exports every symbol of include/hilcodec_amd.h,
module trees carry the reference's state-dict keys, weight folding is bit-exact, and nothing falls
back to the CPU silently.  (No kernel is launched here.)"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from hilcodec_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from hilcodec_amd import _lib
    header = open(os.path.join(ROOT, "include", "hilcodec_amd.h")).read()
    declared = set(re.findall(r"\b(hilc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 14
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/hilcodec_amd.h but not exported"
    for name in _lib.SIGNATURES:
        assert name in declared, f"{name} bound in _lib.py but not declared in the header"
    assert _lib.lib.hilc_abi_version() == _lib.ABI_VERSION
    assert b"num_quantizers" in _lib.lib.hilc_error_string(-5)


def test_error_codes_without_gpu():
    """Argument validation happens before any launch, so it can be exercised on a CPU-only box."""
    from hilcodec_amd._lib import lib
    assert lib.hilc_pw_conv(None, None, None, None, None, 1, 8, 8, 8, 1.0, 0, 1.0, None) == -2      # NULL
    one = ctypes.c_void_p(16)
    assert lib.hilc_pw_conv(one, one, None, None, one, 0, 8, 8, 8, 1.0, 0, 1.0, None) == -1          # shape
    assert lib.hilc_pw_conv(one, one, None, None, one, 1, 8, 6, 8, 1.0, 0, 1.0, None) == -4          # unsupported
    assert lib.hilc_rvq_encode(one, one, one, one, one, None, None, 1, 128, 4, 1024, 8, 9, 0, 0, 0, None) == -5
    assert lib.hilc_rvq_encode(one, one, one, one, one, None, None, 1, 128, 4, 1024, 8, 0, 0, 0, 0, None) == -5
    # an unknown bit of `flags` (ABI 15) is refused, not ignored
    assert lib.hilc_rvq_encode(one, one, one, one, one, None, None, 1, 128, 4, 1024, 8, 8, 0, 0, 2, None) == -4
    assert lib.hilc_resblock_supported(96, 24000) == 1 and lib.hilc_resblock_supported(768, 600) == 1 and lib.hilc_resblock_supported(1024, 600) == 0
    from hilcodec_amd import ops
    for c in (32, 64, 80, 96, 128, 160, 192, 256, 768):          # the traceable Python mirror agrees with the library
        for t in (4, 75, 120, 600, 24000):
            assert ops.resblock_supported(c, t) == bool(lib.hilc_resblock_supported(c, t))
    for c in (32, 64, 96, 128, 192, 256, 320, 384, 512, 640, 768, 1024):       # ... and so does its streaming half
        for t in (1, 4, 8, 12, 16, 32, 40, 64, 160, 320):
            assert ops.resblock_supported(c, t, 4, streaming=True) == bool(lib.hilc_resblock_stream_supported(c, t)), (c, t)
    assert not ops.resblock_supported(768, 8, 1 << 18, streaming=True)          # flat 32-bit column space
    # the stage launches (round 4): chains, encoder / decoder stages, and the row split of their packed weights
    widths = (32, 64, 96, 128, 192, 256, 384, 512, 768, 1024)
    for c in widths:
        assert ops.resblock_chain_row_classes(c, True) == (lib.hilc_resblock_chain_row_classes(c) or 2), c
        assert ops.resblock_chain_row_classes(c, False) == (lib.hilc_resblock_chain_row_classes_offline(c) or 1), c
        for t in (4, 8, 20, 40, 75, 160, 320, 600, 3000):
            for n in (1, 2, 3, 4):
                for st in (0, 1):
                    assert ops.resblock_chain_supported(c, t, n, 2, streaming=bool(st)) == bool(lib.hilc_resblock_chain_supported(c, t, n, st)), (c, t, n, st)
                    for r in (2, 4, 5, 8):
                        assert ops.encoder_stage_supported(c, t, n, r, 2, streaming=bool(st)) == bool(lib.hilc_encoder_stage_supported(c, t, n, r, st)), (c, t, n, r, st)
                        assert ops.decoder_stage_supported(c, t, n, r, 2, streaming=bool(st)) == bool(lib.hilc_decoder_stage_supported(c, t, n, r, st)), (c, t, n, r, st)
    # ... and the round-5 forms that take the model's two ends into the first / last stage launch, plus their NULL / shape checks
    for c in (64, 96, 192):
        for t in (4, 75, 124, 24000):
            for n in (1, 2, 3):
                for r in (2, 4):
                    for k in (5, 7):
                        assert ops.decoder_stage_post_supported(c, t, n, r, k) == bool(lib.hilc_decoder_stage_post_supported(c, t, n, r, k)), (c, t, n, r, k)
                        for st in (0, 1):
                            assert ops.encoder_stage0_supported(t, n, r, c, 1, k, 1, bool(st)) == bool(lib.hilc_encoder_stage0_supported(t, n, r, c, 1, k, st)), (t, n, r, c, k, st)
    assert lib.hilc_decoder_stage_post(None, None, 3, None, 0, 1, 96, 8, None) == -2 and lib.hilc_encoder_stage0(None, None, 2, None, 0, 1, 8, None) == -2
    assert lib.hilc_resblock(one, one, one, one, one, one, one, one, 1, 96, 16, 1.0, 1.0, None) == -4  # y aliases x


def test_cpu_tensors_raise():
    import hilcodec_amd
    from hilcodec_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.pw_conv(torch.zeros(1, 8, 8), torch.zeros(8, 8))
    mk = synth.model_kwargs("hil_speech")
    m = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    with pytest.raises(RuntimeError):
        m.encoder(torch.zeros(1, 1, 640))


def test_offline_module_tree_matches_reference_keys():
    import hilcodec_amd
    for name in ("hil_speech", "hil_music"):
        mk = synth.model_kwargs(name)
        m = hilcodec_amd.HILCodec(24000, 1, **mk)
        own = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("_extra_state")}
        assert own == synth.offline_param_shapes(mk)          # same keys, same order, same shapes
        assert sum(p.numel() for p in m.parameters()) == 9577019
        assert m.encoder.hop_length == 320 and m.decoder.hop_length == 320 and m.sample_rate == 24000
        assert m.encoder.dimension == 128 and m.encoder.ratios == [2, 4, 5, 8] and m.decoder.ratios == [8, 5, 4, 2]
        # a freshly built model has kmeans_init codebooks marked un-initialised, like the reference
        assert not m.quantizer.layers[0].initted and m.quantizer.layers[0].get_extra_state() == {"initted": False}


def test_unsupported_options_fail_loudly():
    import hilcodec_amd
    mk = synth.model_kwargs("hil_speech")
    for bad in (dict(skip="1x1"), dict(causal=False), dict(dilation_base=2), dict(norm="spectral_norm"),
                dict(vq="ResidualGainShapeVQ"), dict(act_all=True)):
        with pytest.raises((NotImplementedError, ValueError, AssertionError)):
            hilcodec_amd.HILCodec(24000, 1, **{**mk, **bad})
    with pytest.raises(RuntimeError):
        hilcodec_amd.HILCodec(24000, 1, **{**mk, "expansion": 2, "groups": 4})


def test_weight_folds_bit_exact():
    from hilcodec_amd import fold
    from hilcodec_amd.models.hilcodec.modules import SConv1d
    from oracle import hilcodec_oracle as O
    v = torch.from_numpy(synth.normalish(1, 24 * 7 * 5)).view(24, 7, 5)
    g = torch.from_numpy(synth.uniform(2, 24, 0.5, 1.5)).view(24, 1, 1)
    assert torch.equal(fold.weight_norm_fold(v, g), torch._weight_norm(v, g, 0))
    assert torch.equal(fold.weight_standardization_fold(v, g, torch.tensor([1.3])),
                       O.fold_weight_standardization(v, g, torch.tensor([1.3])))
    m = SConv1d(7, 24, 1, norm="weight_norm", bias=True)
    with torch.no_grad():
        m.conv.conv.weight_v.copy_(v[:, :, :1]); m.conv.conv.weight_g.copy_(g)
    w0 = m.conv.conv.effective_weight()
    m.conv.conv.remove_reparameterization()
    assert "conv.conv.weight" in m.state_dict() and "conv.conv.weight_g" not in m.state_dict()
    assert torch.equal(m.conv.conv.effective_weight(), w0)
    basis = synth.stft_basis(64)
    bt = fold.stft_basis_layout(basis)
    assert bt.shape == (64, 96) and torch.equal(bt[:, 0], basis[0, 0]) and torch.equal(bt[:, 1], basis[33, 0])
    assert torch.equal(bt[:, 64], basis[32, 0]) and torch.equal(bt[:, 65], basis[65, 0]) and not bt[:, 66:].any()
    cb, cbt, norms = fold.codebook_tables([v[:, :, 0], v[:, :, 1]])
    assert cb.shape == (2, 24, 7) and torch.equal(cbt[1], v[:, :, 1].t()) and torch.equal(norms[0], v[:, :, 0].t().pow(2).sum(0))


def test_streaming_module_tree_and_mapping():
    from hilcodec_amd.models.hilcodec.streaming import HILCodec
    from oracle import hilcodec_oracle as O
    mk = dict(synth.model_kwargs("hil_speech"))
    full = synth.model_kwargs("hil_speech")
    for k in ("spec_learnable", "causal", "pad_mode"):
        mk.pop(k)
    m = HILCodec(24000, **mk).eval()
    x = torch.zeros(5, 1)
    ce, cd = m.initialize_cache(x)
    enc_shapes, dec_shapes = O.stream_cache_shapes(full)
    assert [tuple(c.shape) for c in ce] == [(5, c, l) for c, l in enc_shapes]
    assert [tuple(c.shape) for c in cd] == [(5, c, l) for c, l in dec_shapes]
    assert m.encoder.num_cache == 22 and not any(c.any() for c in ce + cd)
    sd = synth.synth_state_dict("hil_speech", seed=7)
    m.load_offline_state_dict(sd)
    assert m.decoder.blocks[0][2].pre_scale == 1.0 and m.encoder.blocks[0][1].pre_scale == (1 + 2 * 0.5773502691896258 ** 2) ** -0.5
    m.remove_weight_reparameterizations()
    p = O.stream_prepare(sd, full)
    # merge_scaling reproduces the reference's algebra bit for bit (oracle == reference, tests/test_oracle_*)
    assert torch.equal(m.encoder.conv_pre.weight.data, p["enc.conv_pre.weight"])
    assert torch.equal(m.encoder.spec_post.layer.weight.data, p["enc.spec_post.layer.weight"])
    assert torch.equal(m.encoder.spec_post.layer.bias.data, p["enc.spec_post.layer.bias"])
    assert torch.equal(m.encoder.blocks[2][1].block[1].depthwise.weight.data, p["enc.blocks.2.1.1.dw.weight"])
    assert torch.equal(m.decoder.blocks[3][2].block[1].depthwise.bias.data, p["dec.blocks.3.2.1.dw.bias"])
    assert torch.equal(m.decoder.conv_post.weight.data, p["dec.post.weight"])
    assert torch.equal(m.decoder.conv_post.bias.data, p["dec.post.bias"])
    assert torch.equal(m.dequantizer.layers[3].embed, sd["quantizer.layers.3.embed"])


def test_legacy_rvq_matches_reference_constructor_and_state_dict():
    """`modules/vector_quantize.py` (row a9): same ctor signatures, buffers and return arity as the reference; a real
    reference state_dict loads strictly when the checkout is present (build container), and the key/shape contract is
    checked against a literal everywhere else."""
    from hilcodec_amd.modules.vector_quantize import EuclideanCodebook, ResidualVQ, VectorQuantize
    kw = dict(num_quantizers=3, dropout=True, dropout_index=[1, 3], dim=16, codebook_size=32, kmeans_init=False,
              kmeans_iters=20, decay=0.9, eps=1e-7, ema_num_threshold=0.5, ema_num_initial=2.0, commitment=0.25,
              channel_last=False)
    own = ResidualVQ(**kw)
    keys = {k: tuple(v.shape) for k, v in own.state_dict().items()}
    want = {}
    for i in range(3):
        for name, shape in (("initted", (1,)), ("embed", (32, 16)), ("ema_embed", (32, 16)), ("ema_num", (32,))):
            want[f"layers.{i}._codebook.{name}"] = shape
    assert keys == want
    cbk = own.layers[0]._codebook
    assert torch.equal(cbk.ema_embed, cbk.embed * 2.0) and torch.equal(cbk.ema_num, torch.full((32,), 2.0))
    assert own.layers[0].commitment == 0.25 and own.layers[0].gradient_flow is False and own.dropout_index == [1, 3]
    # options this path does not implement are refused, unknown ones are a TypeError as in the reference
    with pytest.raises(NotImplementedError):
        VectorQuantize(use_shape_gain=True, dim=16, codebook_size=32)
    with pytest.raises(TypeError):
        VectorQuantize(dim=16, codebook_size=32, threshold_ema_dead_code=2.0)
    with pytest.raises(TypeError):
        EuclideanCodebook(16, 32, bogus=1)
    # un-initialised k-means codebooks never quantise against zeros silently
    lazy = ResidualVQ(num_quantizers=2, dim=16, codebook_size=32, kmeans_init=True).eval()
    assert not bool(lazy.layers[0]._codebook.initted)
    with pytest.raises(RuntimeError, match="not initialised"):
        lazy(torch.zeros(1, 16, 4))
    with pytest.raises(AssertionError):
        own.eval()(torch.zeros(1, 16, 4), 4)
    from oracle import refimport
    if refimport.reference_available():
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "_ref_legacy_vq", os.path.join(refimport.REFERENCE_ROOT, "modules", "vector_quantize.py"))
        ref_mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_mod)
        ref = ref_mod.ResidualVQ(**kw)
        res = own.load_state_dict(ref.state_dict(), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        assert torch.equal(own.layers[2]._codebook.embed, ref.layers[2]._codebook.embed)
        ref.load_state_dict(own.state_dict(), strict=True)
        import inspect
        for a, b in ((EuclideanCodebook, ref_mod.EuclideanCodebook), (VectorQuantize, ref_mod.VectorQuantize),
                     (ResidualVQ, ref_mod.ResidualVQ)):
            pa, pb = inspect.signature(a.__init__).parameters, inspect.signature(b.__init__).parameters
            assert list(pa) == list(pb), (a.__name__, list(pa), list(pb))
            assert all(pa[k].default == pb[k].default for k in pa if k not in ("self", "kwargs")), a.__name__
        assert list(inspect.signature(VectorQuantize.forward).parameters) == \
            list(inspect.signature(ref_mod.VectorQuantize.forward).parameters)


def test_scalar_n_accepts_any_integer_like():
    """`n` from config arithmetic (numpy ints, 0-dim arrays / tensors) is a scalar n, not a per-clip list."""
    from hilcodec_amd import ops
    for n in (3, np.int64(3), np.int32(3), np.array(3), torch.tensor(3), True + 2):
        assert ops.per_clip_n(n, 5, 8, torch.device("cpu")) == (3, None)
    rows, per = ops.per_clip_n([1, 2, 8, 2, 1], 5, 8, torch.device("cpu"))
    assert rows == 8 and per.dtype == torch.int32 and per.tolist() == [1, 2, 8, 2, 1]
    rows, per = ops.per_clip_n(np.array([4]), 1, 8, torch.device("cpu"))       # a 1-element LIST is per-clip (B = 1)
    assert rows == 4 and per.tolist() == [4]
    with pytest.raises(AssertionError):
        ops.per_clip_n([1, 9], 2, 8, torch.device("cpu"))
    with pytest.raises(RuntimeError):
        ops.per_clip_n([1, 2], 3, 8, torch.device("cpu"))


def test_dws_conv_stream_mirror_knows_which_long_hops_take_a_shortcut():
    """ops.dws_conv_stream_supported(..., has_res): hilc_dws_conv_stream adds `res` on long hops (T > 128) only in its flat-tile form —
    stride <= 8, fewer than 2^31 outputs; the engine asks the mirror before it hands the SpecBlock branch to the layer."""
    from hilcodec_amd import ops
    assert ops.dws_conv_stream_supported(320, 4, 2, True, 1024, 128) and ops.dws_conv_stream_supported(160, 8, 4, True, 1024, 256)
    assert ops.dws_conv_stream_supported(320, 32, 16) and not ops.dws_conv_stream_supported(320, 32, 16, True)       # stride 16: no shortcut
    assert ops.dws_conv_stream_supported(360, 18, 9) and not ops.dws_conv_stream_profitable(360, 18, 9, True, 4, 64)
    assert not ops.dws_conv_stream_supported(320, 4, 2, True, 1 << 24, 128)                                           # 2^31 outputs
    assert ops.dws_conv_stream_supported(40, 10, 5, True) and ops.dws_conv_stream_supported(8, 16, 8, True)         # short hops: whole-clip tiles


def test_exec_options_are_launch_structure_only():
    """engine.ExecOptions: four booleans that select launches, never arithmetic (the split-bf16 decoder mode and the rejected launch
    variants left the library in round 5); nothing process-global."""
    import dataclasses
    from hilcodec_amd import engine
    fields = dataclasses.fields(engine.ExecOptions)
    assert [f.name for f in fields] == ["stage_launches", "wide_blocks", "decoder_stage_narrow", "stream_defer_spec"]
    assert all(f.type in (bool, "bool") and f.default is True for f in fields)
    assert not hasattr(engine, "DECODER_GEMM") and not hasattr(engine, "SIDE_STREAM")     # no process-global switches
    with engine.exec_overrides(decoder_stage_narrow=False):
        assert engine._effective(engine.ExecOptions()).decoder_stage_narrow is False
    assert engine._effective(engine.ExecOptions()).decoder_stage_narrow is True
    with pytest.raises(TypeError):
        with engine.exec_overrides(decoder_gemm="bf16x3"):
            engine._effective(engine.ExecOptions())


def test_clip_chunks_keep_activations_below_32bit_offsets():
    """engine: an offline batch whose largest activation would reach 4 GiB runs as the fewest equal clip chunks below it"""
    from hilcodec_amd import engine
    per_clip = 96 * 24000
    assert engine._clip_chunks(256, per_clip) == [(0, 256)]
    assert engine._clip_chunks(512, per_clip) == [(0, 256), (256, 512)]
    for b in (467, 1000, 2048):
        ch = engine._clip_chunks(b, per_clip)
        assert ch[0][0] == 0 and ch[-1][1] == b and all(a[1] == c[0] for a, c in zip(ch, ch[1:]))
        assert all((hi - lo) * per_clip * 4 < 2 ** 32 for lo, hi in ch) and max(hi - lo for lo, hi in ch) - min(hi - lo for lo, hi in ch) <= len(ch)
    assert engine._clip_chunks(4, 2 ** 31) == [(0, 1), (1, 2), (2, 3), (3, 4)]          # a single clip above the limit still runs (generic cores)


def test_streaming_model_takes_a_weight_standardised_checkpoint(golden):
    """`load_offline_state_dict(sd, norm="weight_standardization", norm_kwargs=...)`: every conv folded to a plain weight
    with the reference's expression (`modules/weight_standardization.py:30-41`) — the streaming classes themselves know
    weight_norm only, like the reference's (`causal_layers.py:200-204`)."""
    import torch
    from hilcodec_amd import synth
    from hilcodec_amd.models.hilcodec.streaming import HILCodec as S
    g = golden("ws_hil_speech")
    mk = {k: v for k, v in synth.model_kwargs("hil_speech").items() if k not in ("spec_learnable", "causal", "pad_mode")}
    sd = synth.synth_state_dict("hil_speech", seed=int(g["weight_seed"]))
    with pytest.raises(ValueError):
        S(24000, norm="weight_standardization", **mk)
    m = S(24000, **mk).eval()
    with pytest.raises(ValueError):
        m.load_offline_state_dict(sd, norm="spectral_norm")
    m.load_offline_state_dict(sd, norm="weight_standardization", norm_kwargs={"eps": float(g["ws_eps"]), "scale": float(g["ws_scale"])})
    keys = m.state_dict()
    assert "decoder.upsample_depthwise.0.weight" in keys and "decoder.upsample_depthwise.0.weight_g" not in keys
    assert torch.equal(keys["decoder.upsample_depthwise.0.weight"], torch.from_numpy(g["fold_probe_convtr"]))
    assert torch.equal(keys["encoder.blocks.1.0.block.0.pointwise.1.weight"], torch.from_numpy(g["fold_probe_pw"]))
    m.remove_weight_reparameterizations()               # hooks are gone already; merge_scaling still runs
    assert "encoder.spec_post.layer.bias" in m.state_dict()
