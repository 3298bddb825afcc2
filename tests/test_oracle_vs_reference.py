"""CPU, build container only: the oracle against the REAL reference modules imported from
/root/reference (skipped where the checkout is absent, e.g. on the GPU box).  Different seeds and
shapes than the committed goldens, so the restatement is pinned on more than one point."""
import numpy as np
import pytest
import torch

from hilcodec_amd import synth
from oracle import hilcodec_oracle as O
from oracle import refimport as R

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    return R.load_reference()


def test_state_dict_keys_and_shapes(ref):
    for name in ("hil_speech", "hil_music"):
        mk = synth.model_kwargs(name)
        import copy
        model = ref.OfflineHILCodec(sample_rate=24000, channels_audio=1, **copy.deepcopy(mk))
        want = {k: tuple(v.shape) for k, v in model.state_dict().items() if hasattr(v, "shape")}
        got = synth.offline_param_shapes(mk)
        assert list(got.keys()) == list(want.keys())
        assert got == want
        n_params = sum(p.numel() for p in model.parameters())
        assert n_params == 9577019          # scripts/PESQ STOI.ipynb:72-73


def test_model_kwargs_match_yaml():
    import os
    import yaml
    for name, f in (("hil_speech", "hilcodec_speech.yaml"), ("hil_music", "hilcodec_music.yaml")):
        y = yaml.safe_load(open(os.path.join(R.REFERENCE_ROOT, "configs", f)))["model_kwargs"]
        assert y == synth.model_kwargs(name)


@pytest.mark.parametrize("name,seed,T", [("hil_speech", 3, 6400), ("hil_music", 5, 3333)])
def test_offline_bit_exact(ref, name, seed, T):
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=seed)
    model = R.build_offline(ref, mk, sd)
    x = synth.synth_clips(2, T, seed=seed * 11)
    with torch.no_grad():
        z = model.encoder(x.clone())
        q, nr, loss, idx = model.quantizer(z, None, return_indices=True)
        wav = model.decoder(q)
        wav_o, nr_o, loss_o, aux = O.codec_forward(sd, x, mk)
    assert torch.equal(aux["z"], z) and torch.equal(aux["indices"], idx) and torch.equal(wav_o, wav)
    assert torch.equal(loss_o, loss) and (nr == nr_o).all()


def test_streaming_bit_exact_and_decoder_deviations(ref):
    mk = synth.model_kwargs("hil_speech")
    sd = synth.synth_state_dict("hil_speech", seed=13)
    model = R.build_offline(ref, mk, sd)
    sm = R.build_streaming(ref, mk, model)
    p = O.stream_prepare(sd, mk)
    x = synth.synth_clips(2, 1600, seed=77)
    ce, cd = sm.initialize_cache(x)
    oe, od = O.stream_init_cache(mk, 2)
    assert [tuple(c.shape) for c in ce] == [tuple(c.shape) for c in oe]
    assert [tuple(c.shape) for c in cd] == [tuple(c.shape) for c in od]
    with torch.no_grad():
        for h in range(0, 1600, 320):
            xin = x[:, :, h:h + 320]
            zr, ce = sm.encoder(xin, *ce)
            ir = sm.quantizer(zr, 8)
            wr, cd = sm.decoder(sm.dequantizer(ir, 8), *cd)
            zo, oe = O.stream_encoder(p, mk, xin, oe)
            io = O.stream_quantize(p, zo, 8)
            wo, od = O.stream_decoder(p, mk, O.stream_dequantize(p, io, 8), od)
            assert torch.equal(zr, zo) and torch.equal(ir, io) and torch.equal(wr, wo)
            assert all(torch.equal(a, b) for a, b in zip(ce, oe))
            assert all(torch.equal(a, b) for a, b in zip(cd, od))
        # the reference's streaming decoder is NOT its offline decoder
        z = O.encoder_forward(sd, x, mk)
        q, _, _, _ = O.rvq_forward(sd, z, 8, 8)
        _, od0 = O.stream_init_cache(mk, 2)
        w_stream, _ = O.stream_decoder(p, mk, q.transpose(1, 2), od0)
        assert (O.decoder_forward(sd, q, mk) - w_stream).abs().max() > 1e-2
        assert (O.decoder_forward(sd, q, mk, streaming_variant=True) - w_stream).abs().max() < 5e-6


def test_legacy_rvq_module(ref):
    D, K, nq = 128, 1024, 4
    old = ref.vq_old.ResidualVQ(num_quantizers=nq, dropout=False, dim=D, codebook_size=K, kmeans_init=False).eval()
    sd = {}
    for i in range(nq):
        e = torch.from_numpy(synth.normalish(500 + i, K * D) * np.float32(0.3)).view(K, D)
        old.layers[i]._codebook.embed.copy_(e)
        sd[f"quantizer.layers.{i}.embed"] = e
    z = torch.from_numpy(synth.normalish(9, 3 * D * 20)).view(3, D, 20)
    with torch.no_grad():
        q, nr, loss = old(z, 3)
        qo, nro, losso, _ = O.rvq_forward(sd, z, 3, nq, variant="legacy")
    assert torch.equal(q, qo) and torch.allclose(loss, losso) and (nr == nro).all()


def test_rvq_training_branch(ref):
    """oracle.rvq_train_step vs the reference ResidualVQ in train mode (n < Nq, expiry threshold on: the masks of
    expired codes must agree; the reference then replaces them with random batch vectors)."""
    D, K, nq, decay, init = 128, 1024, 3, 0.9, 0.5
    rvq = ref.vq_new.ResidualVQ(num_quantizers=nq, dropout=False, channel_last=False, dim=D, codebook_size=K,
                                kmeans_init=False, decay=decay, ema_num_threshold=0.0, ema_num_initial=init).train()
    st = {}
    for i in range(nq):
        e = torch.from_numpy(synth.normalish(800 + i, K * D) * np.float32(0.3)).view(K, D)
        rvq.layers[i].embed.copy_(e)
        rvq.layers[i].ema_embed.copy_(e * init)
        st[f"layers.{i}.embed"] = e.clone()
        st[f"layers.{i}.ema_embed"] = e * init
        st[f"layers.{i}.ema_num"] = torch.ones(K) * init
    for step in range(3):
        z = torch.from_numpy(synth.normalish(70 + step, 2 * D * 40)).view(2, D, 40)
        q, nr, loss, idx = rvq(z, 2, return_indices=True)
        qo, losso, idxo, expired = O.rvq_train_step(st, z, 2, nq, decay, ema_num_threshold=0.46)
        assert torch.equal(idx, idxo) and torch.equal(q, qo) and float(loss) == float(losso)
        for i in range(nq):
            assert torch.equal(rvq.layers[i].embed, st[f"layers.{i}.embed"])
            assert torch.equal(rvq.layers[i].ema_num, st[f"layers.{i}.ema_num"])
        assert expired.shape == (2, K)
        for i in range(2):
            assert torch.equal(expired[i], rvq.layers[i].ema_num < 0.46)
    assert expired.any()          # after three steps at decay 0.9 unused codes have dropped below 0.46


@pytest.mark.parametrize("threshold", [0.0, 0.46])
def test_legacy_rvq_training_branch(ref, threshold):
    """oracle.legacy_rvq_train_step vs the reference's older ResidualVQ in train mode: Laplace-smoothed division
    without an expiry threshold, plain division + expiry masks with one."""
    D, K, nq, decay, init = 128, 1024, 3, 0.9, 0.5
    rvq = ref.vq_old.ResidualVQ(num_quantizers=nq, dropout=False, dim=D, codebook_size=K, kmeans_init=False,
                                decay=decay, ema_num_threshold=0.0, ema_num_initial=init).train()
    st = {}
    for i in range(nq):
        e = torch.from_numpy(synth.normalish(820 + i, K * D) * np.float32(0.3)).view(K, D)
        cbk = rvq.layers[i]._codebook
        cbk.embed.copy_(e)
        cbk.ema_embed.copy_(e * init)
        st[f"layers.{i}.embed"] = e.clone()
        st[f"layers.{i}.ema_embed"] = e * init
        st[f"layers.{i}.ema_num"] = torch.ones(K) * init
    for step in range(3):
        z = torch.from_numpy(synth.normalish(75 + step, 2 * D * 40)).view(2, D, 40)
        q, nr, loss = rvq(z, 2)
        # the reference runs without a threshold (its replacement vectors are random); the oracle's division rule
        # follows `threshold`, so the smoothed form is compared at 0.0 and only the masks at 0.46
        qo, losso, expired = O.legacy_rvq_train_step(st if threshold == 0.0 else {k: v.clone() for k, v in st.items()},
                                                     z, 2, nq, decay, ema_num_threshold=threshold)
        if threshold == 0.0:
            assert torch.equal(q, qo) and float(loss) == float(losso)
            for i in range(nq):
                assert torch.equal(rvq.layers[i]._codebook.embed, st[f"layers.{i}.embed"])
                assert torch.equal(rvq.layers[i]._codebook.ema_num, st[f"layers.{i}.ema_num"])
        else:
            for i in range(nq):           # keep the oracle's state in lock-step with the reference's buffers
                for k in ("embed", "ema_embed", "ema_num"):
                    st[f"layers.{i}.{k}"] = getattr(rvq.layers[i]._codebook, k).clone()
            for i in range(2):
                assert torch.equal(expired[i], rvq.layers[i]._codebook.ema_num < 0.46)
