"""GPU end-to-end parity of the offline path: HILCodec (reference API) on the gfx950 kernels vs
(a) golden vectors from the REAL reference, (b) the CPU oracle.

Bars (north_star): RVQ indices bit-exact, decoded waveform within 1e-4 max-abs.  An index may only
differ where the oracle's own fp64 best-vs-second distance gap is below 1e-4 (a near-tie that the
1e-6-level difference in z legitimately flips); such frames are counted and excluded from the
waveform comparison of that clip region."""
import numpy as np
import pytest
import torch

from hilcodec_amd import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def build(name, seed=7):
    import hilcodec_amd
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=seed)
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("_extra_state") for k in missing)
    for l in model.quantizer.layers:
        l.initted = True
    return model, mk, sd


def compare(model, mk, sd, x, z_ref, idx_ref, wav_ref, n=None, fixture=False):
    """`fixture`: the expected values are a committed golden from the real reference — every fp64 gap of those frames is
    data, no flip has ever been measured on them, so none is allowed (a regression that flips one index must not pass)."""
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    with torch.no_grad():
        z = model.encoder(x.to(dev))
        q, nr, loss, idx = model.quantizer(z, n, return_indices=True)
        wav = model.decoder(q)
    dz = (z.cpu() - z_ref).abs().max().item()
    assert dz < 1e-5, f"encoder output differs by {dz:.3e}"        # measured 3 - 6e-6 (profiles/r05_parity_census_final.json): a 2 x drift fails
    same = idx.cpu() == idx_ref
    flips = 0
    if not same.all():
        gaps = O.rvq_gaps_fp64(sd, z_ref, idx_ref)
        for b, t in {(b, t) for b, s, t in (~same).nonzero().tolist()}:
            s0 = int((~same[b, :, t]).nonzero()[0])
            assert gaps[b, s0, t] < 1e-4, f"genuine index mismatch b={b} s={s0} t={t} gap={gaps[b, s0, t]:.3e}"
            flips += 1
    assert flips <= (0 if fixture else 1), f"{flips} near-tie flips"
    # decoder parity on the SAME codes as the reference: feed the reference indices' q
    cb = model.quantizer.spec(dev).codebooks
    from hilcodec_amd import ops
    q_ref = ops.rvq_decode(idx_ref.to(dev), cb, idx_ref.shape[1], channel_last=False, stage_major=False)
    with torch.no_grad():
        wav2 = model.decoder(q_ref)
    dw = (wav2.cpu() - wav_ref).abs().max().item()
    assert dw < 1e-4, f"decoded waveform differs by {dw:.3e}"
    if flips == 0:
        assert (wav.cpu() - wav_ref).abs().max().item() < 1e-4
    return dz, dw, flips


@pytest.mark.parametrize("name", ["hil_speech", "hil_music"])
def test_offline_golden(golden, name):
    g = golden(f"offline_{name}")
    model, mk, sd = build(name, int(g["weight_seed"]))
    x = synth.synth_clips(g["z"].shape[0], 24000, seed=int(g["clip_seed"]))
    dz, dw, flips = compare(model, mk, sd, x, T(g["z"]), T(g["indices"]).long(), T(g["wav"]), fixture=True)
    print(f"{name}: |dz|={dz:.2e} |dwav|={dw:.2e} near-tie flips={flips}")
    # full forward contract (models.py:111-118)
    dev = torch.device("cuda:0")
    wav, num_replaces, loss = model(x.to(dev), None)
    assert wav.dtype == torch.float32 and wav.shape == (x.shape[0], 1, 24000)
    assert isinstance(num_replaces, np.ndarray) and num_replaces.dtype == np.int64 and not num_replaces.any()
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    # n < Nq
    n = int(g["partial_n"])
    with torch.no_grad():
        z = model.encoder(x.to(dev))
        qn, _, loss_n, idx_n = model.quantizer(z, n, return_indices=True)
    assert idx_n.shape[1] == n
    if flips == 0:
        assert torch.equal(idx_n.cpu(), T(g["indices_n"]).long())
    # ragged length (not a multiple of the hop): conv.py:61-68 "extra padding" semantics
    xr = x[:1, :, : int(g["ragged_len"])].contiguous()
    compare(model, mk, sd, xr, T(g["z_ragged"]), T(g["indices_ragged"]).long(), T(g["wav_ragged"]), fixture=True)


def test_offline_vs_oracle_other_seed():
    from oracle import hilcodec_oracle as O
    model, mk, sd = build("hil_speech", seed=21)
    x = torch.cat([synth.synth_clips(2, 7680, seed=99), synth.sweep_clip(7680)], dim=0)
    with torch.no_grad():
        wav_o, _, loss_o, aux = O.codec_forward(sd, x, mk)
    compare(model, mk, sd, x, aux["z"], aux["indices"], wav_o)


def test_remove_weight_reparameterizations_is_a_noop_numerically():
    model, mk, sd = build("hil_speech")
    dev = torch.device("cuda:0")
    x = synth.synth_clips(1, 3200, seed=3).to(dev)
    with torch.no_grad():
        z0 = model.encoder(x)
        model.remove_weight_reparameterizations()
        assert "encoder.conv_pre.1.conv.conv.weight" in model.state_dict()
        assert "decoder.model.4.convtr.convtr.weight_g" in model.state_dict()     # models.py:120-124 skips convtr
        z1 = model.encoder(x)
    assert torch.equal(z0, z1)


def test_cpu_input_fails_loudly():
    model, mk, sd = build("hil_speech")
    with pytest.raises(RuntimeError):
        model(synth.synth_clips(1, 640))


@pytest.mark.parametrize("gname", ["ws_hil_speech", "ws125_hil_speech"])      # weight_scale 0.8 and 1.25 (large activations through the stage kernels)
def test_whole_model_weight_standardization(golden, gname):
    """`HILCodec(norm="weight_standardization", norm_kwargs=...)` model-wide (`conv.py:36-37`,
    `modules/weight_standardization.py:30-41`) against the REAL reference's output for the same constructor call."""
    import hilcodec_amd
    g = golden(gname)
    kw = {"eps": float(g["ws_eps"]), "scale": float(g["ws_scale"])}
    mk = dict(synth.model_kwargs("hil_speech"), norm="weight_standardization", norm_kwargs=kw)
    sd = synth.synth_state_dict("hil_speech", seed=int(g["weight_seed"]))
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("_extra_state") or k.endswith(".weight_scale") for k in missing)
    assert "encoder.conv_pre.1.conv.conv.weight_scale" in model.state_dict()          # the reference's buffer name
    for l in model.quantizer.layers:
        l.initted = True
    assert torch.equal(model.decoder.model[4].convtr.convtr.effective_weight(), T(g["fold_probe_convtr"]))
    assert torch.equal(model.encoder.blocks[1][0].block[1].conv.conv.effective_weight(), T(g["fold_probe_pw"]))
    x = synth.synth_clips(2, int(g["samples"]), seed=int(g["clip_seed"]))
    from oracle import hilcodec_oracle as O
    sd_o = O.with_weight_standardization(sd, kw["scale"])
    dz, dw, flips = compare(model, mk, sd_o, x, T(g["z"]), T(g["indices"]).long(), T(g["wav"]), fixture=True)
    print(f"weight_standardization: |dz|={dz:.2e} |dwav|={dw:.2e}")
    # remove_weight_reparameterizations leaves a weight-standardised model alone (models.py:121: weight_norm only)
    model.remove_weight_reparameterizations()
    assert "encoder.conv_pre.1.conv.conv.weight_g" in model.state_dict()


def test_offline_launch_structure_options_change_no_bit():
    """`exec_options.stage_launches` (whole stages per launch, round 4) against one launch per residual block and per down- /
    up-sampling layer (round 3), `wide_blocks` (the C >= 256 blocks in the fused kernel) against two depthwise-separable launches per
    block, and `decoder_stage_narrow`: z, indices and wav bit for bit, on clips long enough for several tiles per run and ragged
    against the tile width."""
    import dataclasses
    from hilcodec_amd import engine, ops
    model, mk, sd = build("hil_speech")
    dev = torch.device("cuda:0")
    x = synth.synth_clips(5, 9280, seed=31).to(dev)
    assert [f.name for f in dataclasses.fields(engine.ExecOptions)] == ["stage_launches", "wide_blocks", "decoder_stage_narrow", "stream_defer_spec"]

    def run(stages, wide=True, narrow=True):
        for half in (model.encoder, model.decoder):
            half.exec_options.stage_launches = stages
            half.exec_options.wide_blocks = wide
            half.exec_options.decoder_stage_narrow = narrow
        with torch.no_grad(), ops.timed_launches() as t:
            z = model.encoder(x)
            q, _, _, idx = model.quantizer(z, None, return_indices=True)
            wav = model.decoder(q)
        return z, idx, wav, sum(r[0] == "resblock" for r in t.records), len(t.records)

    try:
        a = run(True)
        c = run(False)
        d = run(True, False)
        f = run(False, False)
        g = run(True, True, False)
    finally:
        for half in (model.encoder, model.decoder):
            half.exec_options.stage_launches = half.exec_options.wide_blocks = half.exec_options.decoder_stage_narrow = True
    # without the wide blocks' fused form (round 3 + the narrow stages of round 4) — fused-kernel launches: 4 stages / 2 * 2 + 2 * 3 blocks
    assert (d[3], f[3]) == (4, 10), (d[3:], f[3:])
    # with them: encoder 4 stages, decoder C = 768: (up-sampling layer +) first block, two blocks; C = 384 / 192 / 96: one launch each;
    # block by block: 4 * 2 + 4 * 3, eight down- / up-sampling launches, the first conv + SpecBlock and the closing conv (phases of the first / last stage by default) more; every wide block is one launch less than two
    assert (a[3], c[3]) == (10, 20) and c[4] - a[4] == 20 and f[4] - c[4] == 10, (a[3:], c[3:], f[3:])
    # the decoder's narrow / partial stage forms off: C = 768, 192 and 96 get their up-sampling launch back and the closing conv its own (C = 384 keeps its whole-stage launch)
    assert g[3] == 10 and g[4] - a[4] == 4, (a[3:], g[3:])
    for other in (c, d, f, g):
        assert torch.equal(a[0], other[0]) and torch.equal(a[1], other[1]) and torch.equal(a[2], other[2])
