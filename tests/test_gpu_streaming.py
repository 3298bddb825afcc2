"""GPU parity of the streaming model (cache-in/cache-out protocol of models/hilcodec/streaming.py) against
golden vectors from the REAL reference streaming model and against the oracle."""
import numpy as np
import pytest
import torch

from hilcodec_amd import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def build_streaming(seed=7, name="hil_speech"):
    from hilcodec_amd.models.hilcodec.streaming import HILCodec
    mk = dict(synth.model_kwargs(name))
    sd = synth.synth_state_dict(name, seed=seed)
    for k in ("spec_learnable", "causal", "pad_mode"):
        mk.pop(k)
    model = HILCodec(24000, **mk).eval()
    model.load_offline_state_dict(sd)
    model.remove_weight_reparameterizations()
    return model, synth.model_kwargs(name), sd


def test_streaming_golden(golden):
    g = golden("stream_hil_speech")
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming(int(g["weight_seed"]))
    hops = int(g["hops"])
    x = synth.synth_clips(1, 320 * hops, seed=int(g["clip_seed"])).to(dev)
    ce, cd = model.initialize_cache(x)
    assert len(ce) == 22 and len(cd) == 30 and model.encoder.num_cache == 22
    zs, ids, ws = [], [], []
    for h in range(hops):
        z, ce = model.encoder(x[:, :, 320 * h: 320 * (h + 1)], *ce)
        idx = model.quantizer(z, 8)
        q = model.dequantizer(idx, 8)
        w, cd = model.decoder(q, *cd)
        zs.append(z); ids.append(idx); ws.append(w)
        if h == 0:
            assert np.allclose([c.double().sum().item() for c in ce], g["e_first_sums"], rtol=1e-4, atol=2e-3)
            assert np.allclose([c.double().sum().item() for c in cd], g["d_first_sums"], rtol=1e-4, atol=2e-3)
    z = torch.cat(zs, 1).cpu(); idx = torch.cat(ids, 2).cpu(); wav = torch.cat(ws, 2).cpu()
    assert z.shape == (1, hops, 128) and idx.shape == (8, 1, hops) and idx.dtype == torch.int64 and wav.shape == (1, 1, 320 * hops)
    assert (z - T(g["z"])).abs().max() < 2e-5
    assert torch.equal(idx, T(g["indices"]).long())
    assert (wav - T(g["wav"])).abs().max() < 1e-4
    for i, c in enumerate(ce):
        ref = T(g[f"e_out{i}"])
        assert c.shape == ref.shape and (c.cpu() - ref).abs().max() < 5e-5, f"e_out{i}"
    for i, c in enumerate(cd):
        ref = T(g[f"d_out{i}"])
        assert c.shape == ref.shape and (c.cpu() - ref).abs().max() < 5e-5, f"d_out{i}"
    # multi-frame chunk == frame-by-frame (caches are exact)
    ce2, _ = model.initialize_cache(x)
    z3, _ = model.encoder(x[:, :, :960], *ce2)
    assert (z3.cpu() - T(g["z_chunk3"])).abs().max() < 2e-5
    # whole-model forward: encoder -> quantizer -> dequantizer -> decoder with both cache lists
    ce3, cd3 = model.initialize_cache(x)
    w_all, ce4, cd4 = model(x[:, :, :640], 8, *ce3, *cd3)
    assert (w_all.cpu() - T(g["wav"])[:, :, :640]).abs().max() < 1e-4 and len(ce4) == 22 and len(cd4) == 30


def test_streaming_vs_oracle_batched_and_int16_indices():
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming(seed=11)
    p = O.stream_prepare(sd, mk)
    B, hops = 3, 4
    x = synth.synth_clips(B, 320 * hops, seed=5)
    ce, cd = model.initialize_cache(x.to(dev))
    oe, od = O.stream_init_cache(mk, B)
    for h in range(hops):
        xin = x[:, :, 320 * h: 320 * (h + 1)]
        z, ce = model.encoder(xin.to(dev), *ce)
        idx = model.quantizer(z, 4)
        zo, oe = O.stream_encoder(p, mk, xin, oe)
        io = O.stream_quantize(p, zo, 4)
        assert (z.cpu() - zo).abs().max() < 2e-5
        assert torch.equal(idx.cpu(), io)
        # the wire format of test_onnx.py is int16: the dequantizer must accept it
        q = model.dequantizer(idx.to(torch.int16), 4)
        qo = O.stream_dequantize(p, io, 4)
        assert torch.equal(q.cpu(), qo)
        w, cd = model.decoder(q, *cd)
        wo, od = O.stream_decoder(p, mk, qo, od)
        assert (w.cpu() - wo).abs().max() < 1e-4
        for a, b in zip(ce, oe):
            assert (a.cpu() - b).abs().max() < 5e-5
        for a, b in zip(cd, od):
            assert (a.cpu() - b).abs().max() < 5e-5


def test_streaming_unmerged_equals_merged():
    """merge_scaling is algebra only: the un-merged streaming model gives the same z up to rounding —
    except for the wav_std scalings, which (as in the reference) only exist in merged form."""
    from hilcodec_amd.models.hilcodec.streaming import HILCodec
    dev = torch.device("cuda:0")
    mk = dict(synth.model_kwargs("hil_speech"))
    sd = synth.synth_state_dict("hil_speech", seed=7)
    for k in ("spec_learnable", "causal", "pad_mode"):
        mk.pop(k)
    m1 = HILCodec(24000, **mk).eval(); m1.load_offline_state_dict(sd)
    m2 = HILCodec(24000, **mk).eval(); m2.load_offline_state_dict(sd); m2.remove_weight_reparameterizations()
    x = synth.synth_clips(1, 640, seed=2).to(dev)
    c1, _ = m1.initialize_cache(x); c2, _ = m2.initialize_cache(x)
    z1, _ = m1.encoder(x / 0.1122080159, *c1)      # un-merged conv_pre lacks the 1/wav_std
    z2, _ = m2.encoder(x, *c2)
    # the spectrogram branch of m1 sees the rescaled waveform too, so only compare shapes/finite here
    assert z1.shape == z2.shape and torch.isfinite(z1).all()
    assert "encoder.conv_pre.weight" in m2.state_dict() and "encoder.conv_pre.weight_g" in m1.state_dict()
    assert "encoder.spec_post.layer.bias" in m2.state_dict()


def test_streaming_layer_classes(golden):
    """CausalConv1d / CausalConvTranspose1d / CausalSTFT with the reference's forward(x, cache) protocol."""
    from hilcodec_amd.models.hilcodec import causal_layers as CL
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    g = golden("ops")
    c = CL.CausalConv1d(8, 8, 10, 5, groups=8, bias=True, norm="none")
    with torch.no_grad():
        c.weight.copy_(T(g["cconv.w"])); c.bias.copy_(T(g["cconv.b"]))
    y, nc = c(T(g["cconv.x"]).to(dev), T(g["cconv.cache"]).to(dev))
    assert (y.cpu() - T(g["cconv.y"])).abs().max() < 2e-6 and torch.equal(nc.cpu(), T(g["cconv.cache_out"]))
    assert c.initialize_cache(torch.zeros(3, 1)).shape == (3, 8, 5)
    ct = CL.CausalConvTranspose1d(8, 8, 10, 5, groups=8, bias=False, norm="none")
    with torch.no_grad():
        ct.weight.copy_(T(g["cconvtr.w"]))
    y, nc = ct(T(g["cconvtr.x"]).to(dev), T(g["cconvtr.cache"]).to(dev))
    assert (y.cpu() - T(g["cconvtr.y"])).abs().max() < 2e-6 and torch.equal(nc.cpu(), T(g["cconvtr.cache_out"]))
    st = CL.CausalSTFT(64, 2)
    wav = synth.synth_clips(2, 63 + 128, seed=1)
    ref = O.causal_stft_mag(wav, st.weight, 2, pad=False, clamp=False)
    assert (st(wav.to(dev)).cpu() - ref).abs().max() < 2e-5
    with pytest.raises(ValueError):
        CL.SConv1d(4, 4, 3, norm="spectral_norm")


def test_stream_driver_protocol(tmp_path):
    """test_onnx.py's protocol end to end: chunked encode -> int16 [n,B,T] .npy -> chunked decode -> wav;
    frame-by-frame and 5-frames-at-a-time give the same codes; the codes equal the oracle's."""
    import numpy as np
    from hilcodec_amd import stream_driver as SD, wire
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming()
    model = model.to(dev)
    x = torch.cat([synth.sweep_clip(4800), synth.synth_clips(1, 4800, seed=9)], dim=0)[:, :, :4800 + 0]
    x = torch.nn.functional.pad(x, (0, 100))                         # 4900 samples: the 100-sample tail is dropped
    idx1, ce1 = SD.encode_stream(model, x.to(dev), 8, num_frames=1)
    idx5, ce5 = SD.encode_stream(model, x.to(dev), 8, num_frames=5)
    assert idx1.dtype == torch.int16 and idx1.shape == (8, 2, 15) and torch.equal(idx1, idx5)
    p = O.stream_prepare(sd, mk)
    oe, od = O.stream_init_cache(mk, 2)
    zo, oe = O.stream_encoder(p, mk, x[:, :, :4800], oe)
    assert torch.equal(idx1.long().cpu(), O.stream_quantize(p, zo, 8))
    path = str(tmp_path / "hil_speech_quantized.npy")
    wire.save_indices_npy(path, idx1)
    assert np.load(path).dtype == np.int16
    timer = SD.StageClock(24000)
    w1, _ = SD.decode_stream(model, wire.load_indices_npy(path), 8, num_frames=1, timer=timer)
    w3, _ = SD.decode_stream(model, idx1, 8, num_frames=3)
    assert w1.shape == (2, 1, 4800) and (w1 - w3).abs().max() < 2e-6
    wo, _ = O.stream_decoder(p, mk, O.stream_dequantize(p, idx1.long().cpu(), 8), od)
    assert (w1.cpu() - wo).abs().max() < 1e-4
    rep = timer.report()
    assert rep["wav_seconds"] == 0.2 and rep["decoder_rtf"] > 0
    # fewer codebooks than trained (bitrate scalability): n = 2 uses the first two index planes
    w_n2, _ = SD.decode_stream(model, idx1, 2, num_frames=15)
    wo2, _ = O.stream_decoder(p, mk, O.stream_dequantize(p, idx1.long().cpu(), 2), O.stream_init_cache(mk, 2)[1])
    assert (w_n2.cpu() - wo2).abs().max() < 1e-4
    # WAV I/O of the driver
    wav_path = str(tmp_path / "o.wav")
    SD.write_wav(wav_path, w1[0, 0].cpu().numpy(), 24000)
    back = SD.read_wav(wav_path, 24000)
    assert back.shape == (4800,) and np.abs(back - w1[0, 0].cpu().numpy()).max() < 1.0 / 32767 + 1e-6


@pytest.mark.parametrize("groups", [1, 2, 3])
def test_graphed_hop_equals_eager(groups):
    """One HIP-graph replay per hop (hilcodec_amd/graph_step.py) against the eager loop: bit-identical indices, wav
    and caches over several hops, after a reset, and resumed from saved caches — also with the streams split into groups
    whose chains run side by side on separate HIP streams inside the graph (same arithmetic, no added latency)."""
    from hilcodec_amd.graph_step import GraphedHop
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming()
    B, hops = 5, 4
    x = synth.synth_clips(B, 320 * hops, seed=77).to(dev)
    ce, cd = model.initialize_cache(x)
    eager = []
    mid = None
    with torch.no_grad():
        for h in range(hops):
            z, ce = model.encoder(x[:, :, 320 * h: 320 * (h + 1)].contiguous(), *ce)
            idx = model.quantizer(z, 8)
            wav, cd = model.decoder(model.dequantizer(idx, 8), *cd)
            eager.append((idx.clone(), wav.clone()))
            if h == 1:
                mid = ([c.clone() for c in ce], [c.clone() for c in cd])
    g = GraphedHop(model, B, 320, 8, dev, groups=groups)
    for rep in range(2):                       # second pass after reset(): the graph carries no hidden state
        for h in range(hops):
            idx, wav = g.step(x[:, :, 320 * h: 320 * (h + 1)])
            assert torch.equal(idx, eager[h][0]) and torch.equal(wav, eager[h][1]), f"pass {rep} hop {h}"
        for (lo, hi), blocks in zip(g.bounds, g.gstate):
            cur = blocks[g.parity]
            for a, b in zip(cur.enc + cur.dec, list(ce) + list(cd)):
                assert torch.equal(a, b[lo:hi])
        g.reset()
    g.reset(*mid)                               # resume after hop 1
    for h in (2, 3):
        idx, wav = g.step(x[:, :, 320 * h: 320 * (h + 1)])
        assert torch.equal(idx, eager[h][0]) and torch.equal(wav, eager[h][1])


@pytest.mark.parametrize("groups", [1, 2])
def test_pipelined_hop_equals_eager(groups):
    """PipelinedHop (decoder of hop i-1 beside the encoder of hop i, two HIP streams inside one graph): indices of
    every hop and — one replay later — its wav bit-identical to the eager loop; flush() delivers the last hop; a
    second pass after reset() and a continuation after flush() give the same again.  groups = 2: the streams in two
    groups, four chains side by side."""
    from hilcodec_amd.graph_step import PipelinedHop
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming()
    B, hops = 5, 5
    x = synth.synth_clips(B, 320 * hops, seed=78).to(dev)
    ce, cd = model.initialize_cache(x)
    eager = []
    with torch.no_grad():
        for h in range(hops):
            z, ce = model.encoder(x[:, :, 320 * h: 320 * (h + 1)].contiguous(), *ce)
            idx = model.quantizer(z, 8)
            wav, cd = model.decoder(model.dequantizer(idx, 8), *cd)
            eager.append((idx.clone(), wav.clone()))
    g = PipelinedHop(model, B, 320, 8, dev, groups=groups)
    for rep in range(2):
        for h in range(hops):
            idx, wav = g.step(x[:, :, 320 * h: 320 * (h + 1)])
            assert torch.equal(idx, eager[h][0]), f"pass {rep} hop {h} indices"
            if h == 0:
                assert wav is None
            else:
                assert torch.equal(wav, eager[h - 1][1]), f"pass {rep} wav of hop {h - 1}"
            if h == 2 and rep == 1:            # drain in mid-stream, then carry on: the caches stay consistent
                assert torch.equal(g.flush(), eager[2][1])
                idx, wav = g.step(x[:, :, 320 * 3: 320 * 4])
                assert wav is None and torch.equal(idx, eager[3][0])
                idx, wav = g.step(x[:, :, 320 * 4: 320 * 5])
                assert torch.equal(idx, eager[4][0]) and torch.equal(wav, eager[3][1])
                break
        assert torch.equal(g.flush(), eager[hops - 1][1]) and g.flush() is None
        g.reset()


@pytest.mark.parametrize("name,n,hop_frames", [("hil_music", 12, 1), ("hil_speech", 8, 3), ("hil_music", 5, 2), ("hil_speech", 8, 4), ("hil_speech", 3, 5)])
def test_streaming_vs_oracle_other_configs(name, n, hop_frames):
    """hil_music (Nq = 12) and multi-frame hops (test_onnx.py `num_frames` > 1: 640 / 960 samples per call) against
    the oracle's streaming model; multi-frame hops take the flat-tiled fused block and the whole-clip wide kernels
    at other T than the single-frame goldens (round 6: the wide stage launches at 2 / 4 frames per stream — C = 512 whole-stream tiles of 16 / 32
    columns, C = 256 / 384 runs of 80 - 200 columns — and their fall-backs at 3 / 5 frames, where a stream does not tile 32 columns)."""
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming(seed=13, name=name)
    p = O.stream_prepare(sd, mk)
    B, hops, L = 2, 3, 320 * hop_frames
    x = synth.synth_clips(B, L * hops, seed=21)
    ce, cd = model.initialize_cache(x.to(dev))
    oe, od = O.stream_init_cache(mk, B)
    for h in range(hops):
        xin = x[:, :, L * h: L * (h + 1)]
        z, ce = model.encoder(xin.to(dev), *ce)
        zo, oe = O.stream_encoder(p, mk, xin, oe)
        assert z.shape == (B, hop_frames, 128) and (z.cpu() - zo).abs().max() < 2e-5
        idx = model.quantizer(z, n)
        io = O.stream_quantize(p, zo, n)
        assert torch.equal(idx.cpu(), io)
        w, cd = model.decoder(model.dequantizer(idx, n), *cd)
        wo, od = O.stream_decoder(p, mk, O.stream_dequantize(p, io, n), od)
        assert w.shape == (B, 1, L) and (w.cpu() - wo).abs().max() < 1e-4
        for a, b in zip(list(ce) + list(cd), list(oe) + list(od)):
            assert (a.cpu() - b).abs().max() < 5e-5


def test_streaming_1024_streams_full_size(golden):
    """BASELINE configs[3] at its REAL size (1024 concurrent streams, hop 320): the launch shapes of a full hop differ
    from the small-batch tests (row-tile heights, frames per RVQ workgroup, flat tiles that straddle many streams), so
    the same properties are checked there — stream 0 is the reference's golden stream, a stream's result does not
    depend on the other 1023 (bit-exact against a 4-stream run), two half batches equal the whole (the multi-GPU
    layout), and one HIP-graph replay per hop equals the eager loop."""
    from hilcodec_amd.graph_step import GraphedHop
    g = golden("stream_hil_speech")
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming(int(g["weight_seed"]))
    B, hops = 1024, int(g["hops"])
    x = synth.synth_clips(B, 320 * hops, seed=int(g["clip_seed"])).to(dev)       # stream 0 = the golden clip

    def run(xs, graphed=False):
        n = xs.shape[0]
        hopper = GraphedHop(model, n, 320, 8, dev) if graphed else None
        ce, cd = model.initialize_cache(xs)
        zs, ids, ws = [], [], []
        with torch.no_grad():
            for h in range(hops):
                xin = xs[:, :, 320 * h: 320 * (h + 1)].contiguous()
                if graphed:
                    idx, w = hopper.step(xin)
                    ids.append(idx.clone()); ws.append(w.clone())
                    continue
                z, ce = model.encoder(xin, *ce)
                idx = model.quantizer(z, 8)
                w, cd = model.decoder(model.dequantizer(idx, 8), *cd)
                zs.append(z); ids.append(idx); ws.append(w)
        caches = (hopper.cache_enc + hopper.cache_dec) if graphed else (list(ce) + list(cd))
        return (torch.cat(zs, 1) if zs else None), torch.cat(ids, 2), torch.cat(ws, 2), caches

    z, idx, wav, caches = run(x)
    assert z.shape == (B, hops, 128) and idx.shape == (8, B, hops) and wav.shape == (B, 1, 320 * hops)
    assert torch.isfinite(wav).all()
    # (1) the golden stream inside the batch of 1024
    assert (z[:1].cpu() - T(g["z"])).abs().max() < 2e-5
    assert torch.equal(idx[:, :1].cpu(), T(g["indices"]).long())
    assert (wav[:1].cpu() - T(g["wav"])).abs().max() < 1e-4
    for i in range(22):
        assert (caches[i][:1].cpu() - T(g[f"e_out{i}"])).abs().max() < 5e-5, f"e_out{i}"
    for i in range(30):
        assert (caches[22 + i][:1].cpu() - T(g[f"d_out{i}"])).abs().max() < 5e-5, f"d_out{i}"
    # (2) batch invariance, bit-exact
    pick = [0, 1, 511, 1023]
    z4, idx4, wav4, caches4 = run(x[pick].contiguous())
    assert torch.equal(z4, z[pick]) and torch.equal(idx4, idx[:, pick]) and torch.equal(wav4, wav[pick])
    for a, b in zip(caches4, caches):
        assert torch.equal(a, b[pick])
    # (3) shard invariance: streams are pinned to a GPU for life, half batches must equal the whole
    _, idx_a, wav_a, _ = run(x[:512].contiguous())
    _, idx_b, wav_b, _ = run(x[512:].contiguous())
    assert torch.equal(torch.cat([idx_a, idx_b], 1), idx) and torch.equal(torch.cat([wav_a, wav_b]), wav)
    # (4) graph replay == eager at full size, caches included
    _, idx_g, wav_g, caches_g = run(x, graphed=True)
    assert torch.equal(idx_g, idx) and torch.equal(wav_g, wav)
    for a, b in zip(caches_g, caches):
        assert torch.equal(a, b)
    # (5) every schedule bench.py reports, at full size, against the eager loop bit for bit: indices, wav (the pipelined
    # schedules deliver it one replay later), all 22 + 30 caches.  The reference's protocol is the plain loop
    # (`scripts/HILCodec Onnx.ipynb` cell 3, `test_onnx.py:75-135`).
    from hilcodec_amd.graph_step import PipelinedHop
    for make, pipelined in ((lambda: GraphedHop(model, B, 320, 8, dev, groups=2), False),
                            (lambda: PipelinedHop(model, B, 320, 8, dev, groups=1), True),
                            (lambda: PipelinedHop(model, B, 320, 8, dev, groups=2), True)):
        hopper = make()
        ids, ws = [], []
        with torch.no_grad():
            for h in range(hops):
                i_h, w_h = hopper.step(x[:, :, 320 * h: 320 * (h + 1)])
                ids.append(i_h.clone())
                if pipelined:
                    assert (w_h is None) == (h == 0)
                if w_h is not None:
                    ws.append(w_h.clone())
            if pipelined:
                ws.append(hopper.flush().clone())
        assert torch.equal(torch.cat(ids, 2), idx), type(hopper).__name__
        assert torch.equal(torch.cat(ws, 2), wav), type(hopper).__name__
        got = list(hopper.cache_enc) + list(hopper.cache_dec)
        assert len(got) == 52
        for i, (a, b) in enumerate(zip(got, caches)):
            assert torch.equal(a, b), f"{type(hopper).__name__} cache {i}"
        del hopper
        torch.cuda.empty_cache()


def test_grouped_schedules_export_and_resume():
    """A grouped schedule exports its caches in the reference's order (`cache_enc` / `cache_dec`, concatenated over the
    groups) and resumes from them: run 2 hops, export, build a NEW hopper, `reset(exported)`, the remaining hops equal the
    uninterrupted run bit for bit — GraphedHop(groups=2), and PipelinedHop(groups=2) after a flush()."""
    from hilcodec_amd.graph_step import GraphedHop, PipelinedHop
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming()
    B, hops = 6, 5
    x = synth.synth_clips(B, 320 * hops, seed=79).to(dev)
    ce, cd = model.initialize_cache(x)
    eager = []
    with torch.no_grad():
        for h in range(hops):
            z, ce = model.encoder(x[:, :, 320 * h: 320 * (h + 1)].contiguous(), *ce)
            idx = model.quantizer(z, 8)
            wav, cd = model.decoder(model.dequantizer(idx, 8), *cd)
            eager.append((idx.clone(), wav.clone()))
    g = GraphedHop(model, B, 320, 8, dev, groups=2)
    for h in range(2):
        g.step(x[:, :, 320 * h: 320 * (h + 1)])
    saved = ([c.clone() for c in g.cache_enc], [c.clone() for c in g.cache_dec])
    assert len(saved[0]) == 22 and len(saved[1]) == 30 and all(c.shape[0] == B for c in saved[0] + saved[1])
    g2 = GraphedHop(model, B, 320, 8, dev, groups=3)             # another grouping resumes from the same export
    g2.reset(*saved)
    for h in range(2, hops):
        idx, wav = g2.step(x[:, :, 320 * h: 320 * (h + 1)])
        assert torch.equal(idx, eager[h][0]) and torch.equal(wav, eager[h][1])
    p = PipelinedHop(model, B, 320, 8, dev, groups=2)
    for h in range(2):
        p.step(x[:, :, 320 * h: 320 * (h + 1)])
    assert torch.equal(p.flush(), eager[1][1])
    for a, b in zip(list(p.cache_enc) + list(p.cache_dec), saved[0] + saved[1]):
        assert torch.equal(a, b)
    p2 = PipelinedHop(model, B, 320, 8, dev, groups=1)
    p2.reset(p.cache_enc, p.cache_dec)
    for h in range(2, hops):
        idx, wav = p2.step(x[:, :, 320 * h: 320 * (h + 1)])
        assert torch.equal(idx, eager[h][0])
        assert wav is None if h == 2 else torch.equal(wav, eager[h - 1][1])
    assert torch.equal(p2.flush(), eager[hops - 1][1])


def test_stream_block_options_reach_both_halves():
    """`exec_options.stage_launches` (a stage's residual blocks — and its down- / up-sampling layer — as ONE launch, or one launch per
    block and layer as in round 3), `wide_blocks` (one launch per wide residual block of a hop, or two as in round 2),
    `decoder_stage_narrow` and `stream_defer_spec` are honoured by the ENCODER's and the DECODER's blocks and change no bit (model
    level; the op-level pins are in test_gpu_ops.py)."""
    from hilcodec_amd import ops
    dev = torch.device("cuda:0")
    model, mk, sd = build_streaming()
    B, hops = 8, 3
    x = synth.synth_clips(B, 320 * hops, seed=80).to(dev)

    def run(stages, wide, defer=True, narrow=True):
        model.encoder.exec_options.stream_defer_spec = defer
        for half in (model.encoder, model.decoder):
            half.exec_options.stage_launches = stages
            half.exec_options.wide_blocks = wide
            half.exec_options.decoder_stage_narrow = narrow
        ce, cd = model.initialize_cache(x)
        outs, kinds = [], []
        with torch.no_grad():
            for h in range(hops):
                with ops.timed_launches() as t:
                    z, ce = model.encoder(x[:, :, 320 * h: 320 * (h + 1)].contiguous(), *ce)
                    n_enc = len(t.records)
                    idx = model.quantizer(z, 8)
                    wav, cd = model.decoder(model.dequantizer(idx, 8), *cd)
                kinds.append((sum(r[0] == "resblock" for r in t.records[:n_enc]), sum(r[0] == "resblock" for r in t.records[n_enc:]), len(t.records)))
                outs.append((z.clone(), idx.clone(), wav.clone()))
        return outs, kinds[0], list(ce) + list(cd)

    try:
        chained, k_chain, c_chain = run(True, True)
        one, k_one, c_one = run(False, True)
        two, k_two, c_two = run(False, False)
        inline, _, c_inline = run(True, True, defer=False)          # SpecBlock branches added in-line (round 3) instead of by the down-sampling epilogues
        split, k_split, c_split = run(True, True, narrow=False)     # PipelinedHop's decoder: narrow / partial stages as up-sampling launch + chain
    finally:
        model.encoder.exec_options.stream_defer_spec = True
        for half in (model.encoder, model.decoder):
            half.exec_options.stage_launches = half.exec_options.wide_blocks = half.exec_options.decoder_stage_narrow = True
    # launches of the fused-block kernel per hop: encoder 4 stages x 2 blocks, decoder 4 x 3; without the wide forms the 4 + 6 wide
    # blocks are two GEMM launches each instead.
    assert k_one[:2] == (8, 12) and k_two[:2] == (4, 6), (k_one, k_two)
    # default (round 6): every encoder and every decoder stage is ONE launch (blocks + down- / up-sampling layer; the wide ones — C = 256 / 512,
    # C = 768 / 384 — in the narrow-tile shapes; the first with the first conv and stage 0's SpecBlock as its opening phase, the last with the closing
    # conv as its closing phase).  decoder_stage_narrow off: C = 192 / 96 get their up-sampling launch back (+ a chain each), and the closing conv its own.
    assert k_chain[:2] == (4, 4) and k_split[:2] == (4, 4) and k_split[2] - k_chain[2] == 3, (k_chain, k_split)
    for ref, other in ((chained, one), (chained, two), (chained, inline), (chained, split)):
        for (z1, i1, w1), (z2, i2, w2) in zip(ref, other):
            assert torch.equal(z1, z2) and torch.equal(i1, i2) and torch.equal(w1, w2)
    for a, b, c, d, e in zip(c_chain, c_one, c_two, c_inline, c_split):
        assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d) and torch.equal(a, e)


@pytest.mark.parametrize("gname", ["ws_hil_speech", "ws125_hil_speech"])
def test_streaming_weight_standardised_checkpoint(golden, gname):
    """A checkpoint of the offline `HILCodec(norm="weight_standardization")` through the streaming model, hop by hop, against
    the REAL reference's streaming model carrying the weights its own `WeightStandardization.compute_weight` produced
    (tests/golden/ws_hil_speech.npz; the reference's streaming classes know weight_norm only, so folded plain weights are
    the only way in — `oracle/make_golden.py: ws_golden`): z, indices, wav and all 52 caches after 5 hops — at `weight_scale` 0.8 (every cache O(1):
    the absolute 5e-5 of every other cache test) and at 1.25 (the decoder's caches reach |x| = 30: the one model-level case that drives large
    activations through the fused stage kernels and the caches; its bar is 5e-5 relative to the cache's largest value)."""
    from hilcodec_amd.models.hilcodec.streaming import HILCodec
    g = golden(gname)
    dev = torch.device("cuda:0")
    mk = {k: v for k, v in synth.model_kwargs("hil_speech").items() if k not in ("spec_learnable", "causal", "pad_mode")}
    sd = synth.synth_state_dict("hil_speech", seed=int(g["weight_seed"]))
    model = HILCodec(24000, **mk).eval()
    model.load_offline_state_dict(sd, norm="weight_standardization",
                                  norm_kwargs={"eps": float(g["ws_eps"]), "scale": float(g["ws_scale"])})
    model.remove_weight_reparameterizations()
    hops = int(g["s_hops"])
    x = synth.synth_clips(1, 320 * hops, seed=int(g["s_clip_seed"])).to(dev)
    ce, cd = model.initialize_cache(x)
    zs, ids, ws = [], [], []
    with torch.no_grad():
        for h in range(hops):
            z, ce = model.encoder(x[:, :, 320 * h: 320 * (h + 1)], *ce)
            idx = model.quantizer(z, 8)
            w, cd = model.decoder(model.dequantizer(idx, 8), *cd)
            zs.append(z); ids.append(idx); ws.append(w)
    assert (torch.cat(zs, 1).cpu() - T(g["s_z"])).abs().max() < 2e-5
    assert torch.equal(torch.cat(ids, 2).cpu(), T(g["s_indices"]).long())
    assert (torch.cat(ws, 2).cpu() - T(g["s_wav"])).abs().max() < 1e-4
    for i, c in enumerate(list(ce) + list(cd)):
        ref = T(g[f"e_out{i}"] if i < 22 else g[f"d_out{i - 22}"])
        bar = 5e-5 * max(1.0, float(ref.abs().max()))          # absolute at weight_scale 0.8 (|cache| <= 2.4), relative at 1.25 (<= 30)
        assert c.shape == ref.shape and (c.cpu() - ref).abs().max() < bar, (i, float((c.cpu() - ref).abs().max()), bar)
