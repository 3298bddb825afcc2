"""GPU parity on REALISTIC and ADVERSARIAL inputs (tests/golden/realistic.npz, produced by the real reference in
oracle/make_golden.py: realistic_golden): the reference's own 30.6 s speech recording through encoder -> RVQ -> decoder with
the reference's SHIPPED trained codebooks in the arg-min; digital silence, silence -> signal, a +-1 square wave, a lone
impulse, DC and a full-scale sine (every log-spectrogram bin at its clamp, the ELU's deep tails); RVQ encode of vectors
1e-6-close to sums of trained code words (near-ties on real tables); and a real `NNNNN.pth` checkpoint round trip through
`stream_driver.build_streaming_model(checkpoint=...)` (/root/reference/models/hilcodec/wrapper.py:428-444, key 'model').

Bars: indices equal to the REFERENCE's (a differing frame must be an fp64 near-tie < 1e-4 in the oracle, and at most one
per case), |dz| < 1e-5, decoded waveform within 1e-4."""
import numpy as np
import pytest
import torch

from hilcodec_amd import synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def real(golden):
    import hilcodec_amd
    from tests.test_oracle_golden import realistic_state_dict
    g = golden("realistic")
    mk = synth.model_kwargs("hil_speech")
    sd = realistic_state_dict(g)
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    model.load_state_dict(sd, strict=False)
    for l in model.quantizer.layers:
        l.initted = True
    return g, mk, sd, model


def oracle_and_compare(model, mk, sd, x, golden_idx=None):
    from oracle import hilcodec_oracle as O
    from tests.test_gpu_offline import compare
    with torch.no_grad():
        wav_o, _, _, aux = O.codec_forward(sd, x, mk)
    if golden_idx is not None:
        assert torch.equal(aux["indices"], golden_idx)          # the checker itself equals the real reference here
    # inputs that are committed data (the reference's recording, the adversarial clips): no flip has been measured, none is allowed
    return compare(model, mk, sd, x, aux["z"], aux["indices"], wav_o, fixture=True)


def test_real_speech_first_10_seconds(real):
    g, mk, sd, model = real
    x = T(g["pcm"][:240000].astype(np.float32) / 32768.0).view(1, 1, -1)
    dz, dw, flips = oracle_and_compare(model, mk, sd, x, T(g["speech10_indices"]).long())
    print(f"real speech 10 s: |dz|={dz:.2e} |dwav|={dw:.2e} near-tie flips={flips} of {8 * 750} argmins")


def test_real_speech_whole_recording(real):
    """all 734 720 samples the reference's driver encodes (len // 320 * 320, test_onnx.py:53) as ONE clip: 2 296 frames,
    18 368 argmins on trained tables"""
    g, mk, sd, model = real
    n = g["pcm"].shape[0] // 320 * 320
    assert n == 734720
    x = T(g["pcm"][:n].astype(np.float32) / 32768.0).view(1, 1, -1)
    dz, dw, flips = oracle_and_compare(model, mk, sd, x)
    print(f"real speech 30.6 s: |dz|={dz:.2e} |dwav|={dw:.2e} near-tie flips={flips} of {8 * 2296} argmins")


def test_adversarial_clips(real):
    g, mk, sd, model = real
    x = synth.adversarial_clips()
    dz, dw, flips = oracle_and_compare(model, mk, sd, x, T(g["adv_indices"]).long())
    with torch.no_grad():
        wav, _, _ = model(x.to(DEV), None)
    assert torch.isfinite(wav).all()
    print(f"adversarial: |dz|={dz:.2e} |dwav|={dw:.2e} near-tie flips={flips}")


def test_adversarial_clips_streaming(real):
    """the same six clips hop by hop through the streaming model (its SpecBlocks have the normalisation merged into the
    conv: another rounding order around the same clamps) against the oracle's streaming path"""
    from oracle import hilcodec_oracle as O
    from hilcodec_amd.models.hilcodec.streaming import HILCodec as StreamingHILCodec
    g, mk, sd, _ = real
    smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
    sm = StreamingHILCodec(24000, **smk).eval()
    sm.load_offline_state_dict(sd)
    sm.remove_weight_reparameterizations()
    x = synth.adversarial_clips(320 * 45)                # impulse at sample 12345: inside
    p = O.stream_prepare(sd, mk)
    ce_o, cd_o = O.stream_init_cache(mk, x.shape[0])
    ce, cd = sm.initialize_cache(x.to(DEV))
    flips = 0
    with torch.no_grad():
        for h in range(45):
            xin = x[:, :, 320 * h: 320 * (h + 1)].contiguous()
            z_o, ce_o = O.stream_encoder(p, mk, xin, ce_o)
            idx_o = O.stream_quantize(p, z_o, 8)
            w_o, cd_o = O.stream_decoder(p, mk, O.stream_dequantize(p, idx_o, 8), cd_o)
            z, ce = sm.encoder(xin.to(DEV), *ce)
            idx = sm.quantizer(z, 8)
            w, cd = sm.decoder(sm.dequantizer(idx_o.to(DEV), 8), *cd)          # decoder on the reference's own indices
            assert (z.cpu() - z_o).abs().max() < 1e-5 and (w.cpu() - w_o).abs().max() < 1e-4
            flips += int((idx.cpu() != idx_o).any(dim=0).sum())
    assert flips == 0          # fixed inputs (committed data): measured 0, a single flip is a regression


def test_real_speech_streaming(real):
    """1.6 s of the reference's speech recording (from 1.0 s in: speech, not the leading silence) hop by hop through the
    streaming model with the trained codebooks, four streams at different offsets in one batch, against the oracle's
    streaming path: z within 2e-5, every index equal (a differing frame must be an fp64 near-tie, at most one), audio
    decoded from the reference's own indices within 1e-4."""
    from oracle import hilcodec_oracle as O
    from hilcodec_amd.models.hilcodec.streaming import HILCodec as StreamingHILCodec
    g, mk, sd, _ = real
    smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
    sm = StreamingHILCodec(24000, **smk).eval()
    sm.load_offline_state_dict(sd)
    sm.remove_weight_reparameterizations()
    hops, B = 120, 4
    pcm = g["pcm"].astype(np.float32) / 32768.0
    x = torch.stack([T(pcm[24000 + 7000 * b: 24000 + 7000 * b + 320 * hops]) for b in range(B)]).view(B, 1, -1)
    p = O.stream_prepare(sd, mk)
    ce_o, cd_o = O.stream_init_cache(mk, B)
    ce, cd = sm.initialize_cache(x.to(DEV))
    flips, dz, dw = 0, 0.0, 0.0
    with torch.no_grad():
        for h in range(hops):
            xin = x[:, :, 320 * h: 320 * (h + 1)].contiguous()
            z_o, ce_o = O.stream_encoder(p, mk, xin, ce_o)
            idx_o = O.stream_quantize(p, z_o, 8)
            w_o, cd_o = O.stream_decoder(p, mk, O.stream_dequantize(p, idx_o, 8), cd_o)
            z, ce = sm.encoder(xin.to(DEV), *ce)
            idx = sm.quantizer(z, 8)
            w, cd = sm.decoder(sm.dequantizer(idx_o.to(DEV), 8), *cd)
            dz = max(dz, float((z.cpu() - z_o).abs().max()))
            dw = max(dw, float((w.cpu() - w_o).abs().max()))
            flips += int((idx.cpu() != idx_o).any(dim=0).sum())
    print(f"real speech, streaming: |dz|={dz:.2e} |dwav|={dw:.2e} flips={flips} of {8 * B * hops} argmins")
    assert dz < 2e-5 and dw < 1e-4 and flips == 0


def test_rvq_near_ties_on_trained_tables(real):
    """z = sum of the trained code words the reference's bitstream selects + 1e-6 noise: many stages sit a hair away from a
    code word.  Every index equals the reference's; a differing frame must be an fp64 near-tie (none has been seen)."""
    from oracle import hilcodec_oracle as O
    g, mk, sd, model = real
    z = T(g["near_z"])
    with torch.no_grad():
        q, _, _, idx = model.quantizer(z.to(DEV), None, return_indices=True)
    ref = T(g["near_indices"]).long()
    same = idx.cpu() == ref
    if not same.all():                          # committed fixture: 0 flips measured, none allowed — say which and how close
        gaps = O.rvq_gaps_fp64(sd, z, ref)
        bad = sorted({(b, t) for b, s, t in (~same).nonzero().tolist()})
        detail = [(b, t, int((~same[b, :, t]).nonzero()[0]), float(gaps[b, int((~same[b, :, t]).nonzero()[0]), t])) for b, t in bad]
        raise AssertionError(f"index flips on the near-tie fixture (clip, frame, stage, fp64 gap): {detail}")
    assert (q.cpu()[:, :, ::7] - T(g["near_q_probe"])).abs().max() < 1e-5


def test_pth_checkpoint_round_trip(tmp_path, real):
    """a checkpoint file as the reference's trainer writes it (`torch.save({'model': state_dict, ...}, 'NNNNN.pth')`,
    wrapper.py:428-444) -> `stream_driver.build_streaming_model(checkpoint=...)` -> the driver's encode / decode loop equals
    the model built directly from the state dict, bit for bit (indices, audio)."""
    from hilcodec_amd import stream_driver
    from hilcodec_amd.models.hilcodec.streaming import HILCodec as StreamingHILCodec
    g, mk, sd, _ = real
    path = tmp_path / "00010.pth"
    torch.save({"model": sd, "epoch": 10, "optim_g": {}, "optim_d": {}}, str(path))
    m_ckpt = stream_driver.build_streaming_model("hil_speech", str(path), DEV)
    smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
    m_direct = StreamingHILCodec(24000, **smk).eval()
    m_direct.load_offline_state_dict(sd)
    m_direct.remove_weight_reparameterizations()
    m_direct = m_direct.to(DEV)
    x = T(g["pcm"][24000:24000 + 320 * 20].astype(np.float32) / 32768.0).view(1, 1, -1).to(DEV)
    i1, _ = stream_driver.encode_stream(m_ckpt, x, 8)
    i2, _ = stream_driver.encode_stream(m_direct, x, 8)
    w1, _ = stream_driver.decode_stream(m_ckpt, i1, 8)
    w2, _ = stream_driver.decode_stream(m_direct, i2, 8)
    assert i1.dtype == torch.int16 and torch.equal(i1, i2) and torch.equal(w1, w2)
    # a bare state dict (no 'model' key) is accepted too
    bare = tmp_path / "bare.pth"
    torch.save(sd, str(bare))
    m_bare = stream_driver.build_streaming_model("hil_speech", str(bare), DEV)
    i3, _ = stream_driver.encode_stream(m_bare, x, 8)
    assert torch.equal(i3, i1)
