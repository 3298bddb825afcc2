"""CPU: the oracle (oracle/hilcodec_oracle.py) against the golden vectors that
oracle/make_golden.py produced from the REAL reference.  Bit-exact (same torch CPU kernels)."""
import numpy as np
import pytest
import torch

from hilcodec_amd import synth
from oracle import hilcodec_oracle as O

torch.set_num_threads(max(1, min(8, torch.get_num_threads())))


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def speech():
    return synth.model_kwargs("hil_speech"), synth.synth_state_dict("hil_speech", seed=7)


@pytest.mark.parametrize("name", ["hil_speech", "hil_music"])
def test_offline_end_to_end(golden, name):
    g = golden(f"offline_{name}")
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=int(g["weight_seed"]))
    x = synth.synth_clips(g["z"].shape[0], 24000, seed=int(g["clip_seed"]))
    with torch.no_grad():
        wav, num_replaces, loss, aux = O.codec_forward(sd, x, mk)
        assert torch.equal(aux["z"], T(g["z"]))
        assert torch.equal(aux["indices"], T(g["indices"]).long())
        assert torch.equal(wav, T(g["wav"]))
        assert torch.equal(aux["q"][:, :, ::15], T(g["q_probe"]))
        assert float(loss) == float(g["loss"])
        assert (num_replaces == g["num_replaces"]).all() and num_replaces.dtype == np.int64
        n = int(g["partial_n"])
        wav_n, _, loss_n, aux_n = O.codec_forward(sd, x, mk, n=n)
        assert torch.equal(aux_n["indices"], T(g["indices_n"]).long())
        assert torch.equal(wav_n[:, :, ::16], T(g["wav_n_probe"]))
        assert float(loss_n) == float(g["loss_n"])
        xr = x[:1, :, : int(g["ragged_len"])]
        wav_r, _, _, aux_r = O.codec_forward(sd, xr, mk)
        assert torch.equal(aux_r["z"], T(g["z_ragged"]))
        assert torch.equal(aux_r["indices"], T(g["indices_ragged"]).long())
        assert torch.equal(wav_r, T(g["wav_ragged"]))


def test_rvq_n_out_of_range(speech):
    mk, sd = speech
    z = torch.zeros(1, 128, 3)
    for bad in (0, 9):
        with pytest.raises(AssertionError):
            O.rvq_forward(sd, z, bad, 8)


def test_streaming(golden, speech):
    g = golden("stream_hil_speech")
    mk, sd = speech
    p = O.stream_prepare(sd, mk)
    hops = int(g["hops"])
    x = synth.synth_clips(1, 320 * hops, seed=int(g["clip_seed"]))
    ce, cd = O.stream_init_cache(mk, 1)
    zs, ids, ws = [], [], []
    with torch.no_grad():
        for h in range(hops):
            z, ce = O.stream_encoder(p, mk, x[:, :, 320 * h: 320 * (h + 1)], ce)
            idx = O.stream_quantize(p, z, 8)
            w, cd = O.stream_decoder(p, mk, O.stream_dequantize(p, idx, 8), cd)
            zs.append(z); ids.append(idx); ws.append(w)
            if h == 0:
                assert np.array_equal(np.array([c.double().sum().item() for c in ce]), g["e_first_sums"])
                assert np.array_equal(np.array([c.double().sum().item() for c in cd]), g["d_first_sums"])
        assert torch.equal(torch.cat(zs, 1), T(g["z"]))
        assert torch.equal(torch.cat(ids, 2), T(g["indices"]).long())
        assert torch.equal(torch.cat(ws, 2), T(g["wav"]))
        for i, c in enumerate(ce):
            assert torch.equal(c, T(g[f"e_out{i}"])), i
        for i, c in enumerate(cd):
            assert torch.equal(c, T(g[f"d_out{i}"])), i
        ce2, _ = O.stream_init_cache(mk, 1)
        z3, _ = O.stream_encoder(p, mk, x[:, :, :960], ce2)
        assert torch.equal(z3, T(g["z_chunk3"]))


def test_cache_template_shapes(speech):
    """Shapes documented for onnx/hil_speech_cache_{enc,dec}.npz."""
    mk, _ = speech
    enc, dec = O.stream_cache_shapes(mk)
    assert len(enc) == 22 and len(dec) == 30
    assert enc[0] == (1, 1023) and enc[1] == (64, 4) and enc[5] == (128, 2)
    assert enc[20] == (1024, 8) and enc[21] == (1024, 4)
    assert dec[0] == (1536, 4) and dec[1] == (1536, 1) and dec[2] == (768, 4) and dec[29] == (96, 4)
    assert sum(c * l for c, l in enc) == 32511 and sum(c * l for c, l in dec) == 43968


def test_rvq_kat(golden):
    g = golden("rvq")
    nq, K, D = 12, 1024, 128
    z = torch.from_numpy(synth.normalish(int(g["z_seed"]), 2 * D * 75)).view(2, D, 75)
    z = torch.nn.functional.normalize(z, dim=1) * D ** 0.5
    sd = {}
    for i in range(nq):
        sd[f"quantizer.layers.{i}.embed"] = torch.from_numpy(
            synth.normalish(synth.key_seed(int(g["codebook_seed"]), f"rvq{i}"), K * D) * np.float32(0.3 * 0.95 ** i)).view(K, D)
    q, nr, loss, idx = O.rvq_forward(sd, z, None, nq)
    assert torch.equal(idx, T(g["indices"]).long()) and torch.equal(q, T(g["q"]))
    assert float(loss) == float(g["loss"])
    q5, _, loss5, idx5 = O.rvq_forward(sd, z, 5, nq)
    assert torch.equal(idx5, T(g["indices_n5"]).long()) and float(loss5) == float(g["loss_n5"])
    ql, _, lossl, idxl = O.rvq_forward(sd, z, None, nq, variant="legacy")
    assert torch.equal(ql, T(g["q_legacy"]))
    # streaming layout: [B,T,C] -> [n,B,T]
    p = {f"vq.{i}.embed": sd[f"quantizer.layers.{i}.embed"] for i in range(nq)}
    ids = O.stream_quantize(p, z.transpose(1, 2), nq)
    assert torch.equal(ids.permute(1, 0, 2), T(g["indices"]).long())
    assert torch.allclose(O.stream_dequantize(p, ids, nq).transpose(1, 2), T(g["q"]), atol=1e-6)
    gaps = O.rvq_gaps_fp64(sd, z, idx)
    assert gaps.min() >= 0 and gaps.shape == idx.shape


def test_ops_kat(golden):
    g = golden("ops")
    F = torch.nn.functional

    def sub(tag):
        pre = tag + "."
        return {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}

    d = sub("pw")
    w, b = O.conv_weight(d, "conv.conv")
    assert torch.equal(F.conv1d(O.elu(d["x"]), w, b), d["y"])
    d = sub("dw5")
    w, b = O.conv_weight(d, "conv.conv")
    assert torch.equal(O.sconv1d(d["x"], w, b, groups=w.shape[0]), d["y"])
    for r in (2, 4, 5, 8):
        d = sub(f"dws{r}")
        w, b = O.conv_weight(d, "conv.conv")
        assert torch.equal(O.sconv1d(d["x"], w, b, stride=r, groups=w.shape[0]), d["y"])
        d = sub(f"dwt{r}")
        w, b = O.conv_weight(d, "convtr.convtr")
        assert torch.equal(O.sconvtr1d(d["x"], w, b, stride=r, groups=w.shape[0]), d["y"])
    for tag in ("pre", "post"):
        d = sub(tag)
        w, b = O.conv_weight(d, "conv.conv")
        assert torch.equal(O.sconv1d(d["x"], w, b), d["y"])
    for n_fft, hop in ((16, 1), (32, 4)):
        d = sub(f"spec{n_fft}")
        assert torch.equal(d["spec.weight"], synth.stft_basis(n_fft))
        assert torch.equal(O.causal_stft_mag(d["wav"], d["spec.weight"], hop, True, True), d["mag"])
        y = O.spec_block(d, "", d["x"], d["wav"], hop, -4.0, 2.8, 0.5773502691896258) if False else None
        sd = {"p." + k: v for k, v in d.items()}
        y = O.spec_block(sd, "p", d["x"], d["wav"], hop, -4.0, 2.8, 0.5773502691896258)
        assert torch.equal(y, d["y"])
    for idx in (0, 1, 2):
        d = sub(f"res{idx}")
        sd = {"p." + k: v for k, v in d.items()}
        assert torch.equal(O.resblock(sd, "p", d["x"], 0.5773502691896258, idx), d["y"])
    d = sub("l2")
    assert torch.equal(F.normalize(d["x"], p=2.0, dim=1, eps=1e-12) * 128 ** 0.5, d["y"])
    d = sub("ws")
    assert torch.equal(O.fold_weight_standardization(d["v"], d["g"], d["scale"]), d["w"])
    d = sub("cconv")
    y, c = O.causal_conv1d(d["x"], d["cache"], d["w"], d["b"], 5, 8)
    assert torch.equal(y, d["y"]) and torch.equal(c, d["cache_out"])
    d = sub("cconvtr")
    y, c = O.causal_convtr1d(d["x"], d["cache"], d["w"], None, 5, 8)
    assert torch.equal(y, d["y"]) and torch.equal(c, d["cache_out"])


def rvq_train_state(seed, nq, init, K=1024, D=128):
    st = {}
    for i in range(nq):
        e = torch.from_numpy(synth.normalish(synth.key_seed(seed, f"rvq{i}"), K * D) * np.float32(0.3 * 0.95 ** i)).view(K, D)
        st[f"layers.{i}.embed"] = e.clone()
        st[f"layers.{i}.ema_embed"] = e * init
        st[f"layers.{i}.ema_num"] = torch.ones(K) * init
    return st


def test_rvq_train_kat(golden):
    """training branch (EMA statistics + codebook update) against two steps of the reference in train mode"""
    g = golden("rvq_train")
    nq, D, B, Tn = 4, 128, 4, 75
    st = rvq_train_state(int(g["codebook_seed"]), nq, float(g["ema_num_initial"]))
    for step in range(2):
        z = torch.from_numpy(synth.normalish(int(g[f"z_seed{step}"]), B * D * Tn)).view(B, D, Tn)
        z = torch.nn.functional.normalize(z, dim=1) * D ** 0.5
        q, loss, idx, expired = O.rvq_train_step(st, z, None, nq, float(g["decay"]))
        assert torch.equal(idx, T(g[f"indices{step}"]).long())
        assert float(loss) == float(g[f"loss{step}"]) and torch.equal(q[:, :, ::5], T(g[f"q_probe{step}"]))
        assert not expired.any()
    for i in range(nq):
        assert torch.equal(st[f"layers.{i}.ema_num"], T(g["ema_num"][i]))
        assert torch.equal(st[f"layers.{i}.embed"][::16], T(g["embed_rows"][i]))
        assert torch.equal(st[f"layers.{i}.ema_embed"][::16], T(g["ema_embed_rows"][i]))
        assert abs(float(st[f"layers.{i}.embed"].double().sum()) - float(g["embed_sum"][i])) < 1e-9


def realistic_state_dict(g):
    """seeded synthetic conv weights + the reference's SHIPPED trained codebooks (tests/golden/realistic.npz)"""
    sd = synth.synth_state_dict("hil_speech", seed=int(g["weight_seed"]))
    for i in range(8):
        sd[f"quantizer.layers.{i}.embed"] = T(g["codebooks"][i]).clone()
    return sd


def test_realistic_and_adversarial_inputs(golden):
    """The oracle on the reference's own speech recording, its trained codebooks, and six adversarial clips (digital silence,
    silence -> signal, +-1 square wave, lone impulse, DC, full-scale sine) against the REAL reference's outputs
    (oracle/make_golden.py: realistic_golden): bit-exact, so the oracle is pinned on realistic data and on the clamp
    branches (conv.py:357, seanet.py:232), not only on noise-like inputs."""
    g = golden("realistic")
    mk = synth.model_kwargs("hil_speech")
    sd = realistic_state_dict(g)
    with torch.no_grad():
        x10 = T(g["pcm"][:240000].astype(np.float32) / 32768.0).view(1, 1, -1)
        wav, _, loss, aux = O.codec_forward(sd, x10, mk)
        assert torch.equal(aux["indices"], T(g["speech10_indices"]).long())
        assert torch.equal(aux["z"][:, :, ::25], T(g["speech10_z_probe"])) and torch.equal(wav[:, :, ::97], T(g["speech10_wav_probe"]))
        assert float(loss) == float(g["speech10_loss"])
        wav_a, _, loss_a, aux_a = O.codec_forward(sd, synth.adversarial_clips(), mk)
        assert torch.equal(aux_a["indices"], T(g["adv_indices"]).long())
        assert torch.equal(aux_a["z"][:, :, ::5], T(g["adv_z_probe"])) and torch.equal(wav_a[:, :, ::31], T(g["adv_wav_probe"]))
        assert torch.isfinite(wav_a).all() and float(loss_a) == float(g["adv_loss"])
        q, _, _, idx = O.rvq_forward(sd, T(g["near_z"]), None, 8)
        assert torch.equal(idx, T(g["near_indices"]).long()) and torch.equal(q[:, :, ::7], T(g["near_q_probe"]))


def test_shard_rank7_prefix(golden):
    """BASELINE configs[4]: the first 8 clips of rank 7's 256-clip shard (hil_music, clips 1792..1799), from the REAL reference."""
    g = golden("shard_rank7_hil_music")
    mk = synth.model_kwargs("hil_music")
    sd = synth.synth_state_dict("hil_music", seed=int(g["weight_seed"]))
    x = synth.synth_clips(2, 24000, seed=int(g["clip_seed"]), first=int(g["first"]) + 3)       # clips 1795, 1796: the oracle is pinned on two
    with torch.no_grad():
        wav, _, _, aux = O.codec_forward(sd, x, mk)
    assert torch.equal(aux["indices"], T(g["indices"]).long()[3:5])
    assert torch.equal(aux["z"][:, :, ::5], T(g["z_probe"])[3:5])
    assert torch.equal(wav[:, :, ::25], T(g["wav_probe"])[3:5])


@pytest.mark.parametrize("gname", ["ws_hil_speech", "ws125_hil_speech"])      # weight_scale 0.8 (O(1) activations) and 1.25 (|x| ~ 30)
def test_weight_standardization_whole_model(golden, gname):
    """`HILCodec(norm="weight_standardization")` (`conv.py:36-37`, `modules/weight_standardization.py:30-41`), offline and —
    through folded plain weights, the only way the reference's weight_norm-only streaming classes can carry such a
    checkpoint — streaming with every cache."""
    g = golden(gname)
    mk = synth.model_kwargs("hil_speech")
    sd = O.with_weight_standardization(synth.synth_state_dict("hil_speech", seed=int(g["weight_seed"])), float(g["ws_scale"]))
    x = synth.synth_clips(2, int(g["samples"]), seed=int(g["clip_seed"]))
    with torch.no_grad():
        wav, _, loss, aux = O.codec_forward(sd, x, mk)
        assert torch.equal(aux["z"], T(g["z"])) and torch.equal(aux["indices"], T(g["indices"]).long())
        assert torch.equal(wav, T(g["wav"])) and float(loss) == float(g["loss"])
        w, _ = O.conv_weight(sd, "decoder.model.4.convtr.convtr")
        assert torch.equal(w, T(g["fold_probe_convtr"]))
        w, _ = O.conv_weight(sd, "encoder.blocks.1.0.block.1.conv.conv")
        assert torch.equal(w, T(g["fold_probe_pw"]))
        p = O.stream_prepare(sd, mk)
        hops = int(g["s_hops"])
        xs = synth.synth_clips(1, 320 * hops, seed=int(g["s_clip_seed"]))
        ce, cd = O.stream_init_cache(mk, 1)
        zs, ids, ws = [], [], []
        for h in range(hops):
            z, ce = O.stream_encoder(p, mk, xs[:, :, 320 * h: 320 * (h + 1)], ce)
            idx = O.stream_quantize(p, z, 8)
            w, cd = O.stream_decoder(p, mk, O.stream_dequantize(p, idx, 8), cd)
            zs.append(z); ids.append(idx); ws.append(w)
        assert torch.equal(torch.cat(zs, 1), T(g["s_z"]))
        assert torch.equal(torch.cat(ids, 2), T(g["s_indices"]).long())
        assert torch.equal(torch.cat(ws, 2), T(g["s_wav"]))
        for i, c in enumerate(ce):
            assert torch.equal(c, T(g[f"e_out{i}"])), i
        for i, c in enumerate(cd):
            assert torch.equal(c, T(g[f"d_out{i}"])), i
