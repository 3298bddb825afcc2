"""CPU: the dispatcher boundary (`torch.ops.hilcodec.*`) — every op is registered with a schema, a shape
function (fake / meta kernel) and a CPU kernel that refuses; the whole encoder -> RVQ -> decoder plan (offline and one
streaming hop with a persistent state block) walks through the ops on META tensors, i.e. exactly what a tracing
compiler sees.  No kernel is launched here; numerics of the same graph are the -m gpu tests."""
import pytest
import torch

import hilcodec_amd
from hilcodec_amd import engine, ops, synth

OPS = ["spec_block", "spec_block_conv_pre", "spec_block_pack", "pw_conv", "dws_conv", "dws_conv_stream", "up_conv_expand_taps", "up_conv", "resblock_pack", "resblock", "dw_conv",
       "dw_convtr", "conv_pre", "conv_post", "stft_logmag", "tail", "l2norm", "rvq_encode", "rvq_decode",
       "rvq_ema_stats", "rvq_ema_update", "resblock_chain", "encoder_stage", "decoder_stage", "decoder_stage_post", "encoder_stage0", "resblock_pack_rc"]


def test_every_op_is_registered_with_schema_and_refuses_cpu():
    for name in OPS:
        op = getattr(torch.ops.hilcodec, name).default
        assert str(op._schema).startswith(f"hilcodec::{name}(")
        assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "CUDA")
        assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "Meta")
    with pytest.raises(RuntimeError, match="GPU"):
        torch.ops.hilcodec.pw_conv(torch.zeros(1, 8, 8), torch.zeros(8, 8), None, None, 1.0, False, 1.0)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.l2norm(torch.zeros(1, 8, 8))
    # mutated arguments are declared (functionalisation / graph capture need to know): the streaming state outputs
    assert "Tensor(a!) hist_out" in str(torch.ops.hilcodec.dws_conv_stream.default._schema)
    assert "Tensor(a!)? hist1_out" in str(torch.ops.hilcodec.resblock.default._schema)


def _meta_model(name):
    mk = synth.model_kwargs(name)
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    model.load_state_dict(synth.synth_state_dict(name, seed=7), strict=False)
    es = engine.finalize_spec(engine.spec_to(model.encoder.build_spec("cpu"), "meta"))
    ds = engine.finalize_spec(engine.spec_to(model.decoder.build_spec("cpu"), "meta"))
    return model, mk, es, ds


@pytest.mark.parametrize("name", ["hil_speech", "hil_music"])
def test_offline_plan_on_meta_tensors(name):
    model, mk, es, ds = _meta_model(name)
    nq = mk["vq_kwargs"]["num_quantizers"]
    assert es.stages[0].blocks[0].pw1_packed is not None and es.stages[3].blocks[0].pw1_packed is not None   # C=64 / C=512 (round 4: the wide blocks too)
    assert ds.stages[1].taps is not None and ds.stages[0].taps is None                                    # stride 5 / 8
    assert es.stages[0].spec.fused is not None and es.stages[2].spec.fused is not None and es.stages[3].spec.fused is None
    assert es.stages[0].spec.fused[0].shape == (64 * 64,) and es.stages[1].spec.fused[2].numel() % (128 * 8) == 0
    x = torch.empty(3, 1, 24000, device="meta")
    z = engine.run_encoder(es, x)
    assert z.shape == (3, 128, 75) and z.device.type == "meta"
    cb = torch.empty(nq, 1024, 128, device="meta")
    idx, q, loss = ops.rvq_encode(z, cb, cb.transpose(1, 2).contiguous(), torch.empty(nq, 1024, device="meta"), nq,
                                  want_loss=True)
    assert idx.shape == (3, nq, 75) and idx.dtype == torch.int64 and q.shape == z.shape and loss.shape == ()
    idx4, q4, l4 = ops.rvq_encode(z, cb, cb.transpose(1, 2).contiguous(), torch.empty(nq, 1024, device="meta"), 4,
                                  stage_major=True, want_q=False)
    assert idx4.shape == (4, 3, 75) and q4 is None and l4 is None
    assert ops.rvq_decode(idx4, cb, 4).shape == (3, 75, 128)
    wav = engine.run_decoder(ds, q)
    assert wav.shape == (3, 1, 24000)
    # ragged length: ceil semantics of the strided layers (conv.py:61-68)
    assert engine.run_encoder(es, torch.empty(1, 1, 5000, device="meta")).shape == (1, 128, 16)


def test_streaming_hop_on_meta_tensors_with_state_block():
    from hilcodec_amd.models.hilcodec.streaming import HILCodec
    mk = dict(synth.model_kwargs("hil_speech"))
    for k in ("spec_learnable", "causal", "pad_mode"):
        mk.pop(k)
    m = HILCodec(24000, **mk).eval()
    m.load_offline_state_dict(synth.synth_state_dict("hil_speech", seed=7))
    m.remove_weight_reparameterizations()
    es = engine.finalize_spec(engine.spec_to(m.encoder.build_spec("cpu"), "meta"))
    ds = engine.finalize_spec(engine.spec_to(m.decoder.build_spec("cpu"), "meta"))
    B = 6
    ce = [torch.empty(B, c, l, device="meta") for c, l in engine.encoder_cache_shapes(es)]
    cd = [torch.empty(B, c, l, device="meta") for c, l in engine.decoder_cache_shapes(ds)]
    assert len(ce) == 22 and len(cd) == 30
    for hop in (320, 960):
        x = torch.empty(B, 1, hop, device="meta")
        # reference protocol: fresh cache tensors
        z, ne = engine.run_encoder(es, x, ce, channel_last_out=True)
        assert z.shape == (B, hop // 320, 128) and [t.shape for t in ne] == [t.shape for t in ce]
        assert all(a is not b for a, b in zip(ne, ce))
        w, nd = engine.run_decoder(ds, z.transpose(1, 2), cd)
        assert w.shape == (B, 1, hop) and [t.shape for t in nd] == [t.shape for t in cd]
        # persistent state block: the caller's buffers ARE the returned caches
        oe = [torch.empty_like(t) for t in ce]
        od = [torch.empty_like(t) for t in cd]
        _, ne2 = engine.run_encoder(es, x, ce, channel_last_out=True, caches_out=oe)
        _, nd2 = engine.run_decoder(ds, z.transpose(1, 2), cd, caches_out=od)
        assert all(a is b for a, b in zip(ne2, oe)) and all(a is b for a, b in zip(nd2, od))


def test_dynamo_fullgraph_traces_the_module_classes():
    """torch.compile(fullgraph=True) of the reference-API modules on META tensors: dynamo must get through the plan walk
    without a graph break (no data_ptr / ctypes / host fold in the traced region) and the fake kernels must propagate
    shapes; the same compile on real tensors with bit-exact comparison is tests/test_gpu_compile.py."""
    model, mk, es, ds = _meta_model("hil_speech")
    model.encoder._plan_cache, model.decoder._plan_cache = es, ds
    model.encoder._plan_cache_key = model.decoder._plan_cache_key = ("meta",)        # the device the plans were folded for
    enc = torch.compile(model.encoder, fullgraph=True, backend="aot_eager")
    dec = torch.compile(model.decoder, fullgraph=True, backend="aot_eager")
    with torch.no_grad():
        z = enc(torch.empty(2, 1, 4800, device="meta"))
        w = dec(z)
    assert z.shape == (2, 128, 15) and w.shape == (2, 1, 4800)
    # a plan folded for another device is refused instead of being baked into the graph as wrong-device constants
    model.encoder._plan_cache_key = ("cuda:1",)
    with pytest.raises(Exception, match="prepared for cuda:1"):
        torch.compile(model.encoder, fullgraph=True, backend="aot_eager")(torch.empty(2, 1, 4800, device="meta"))
    # without a prepared plan the traced region refuses instead of folding weights inside the graph
    fresh = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    with pytest.raises(Exception, match="prepare"):
        torch.compile(fresh.encoder, fullgraph=True, backend="aot_eager")(torch.empty(1, 1, 640, device="meta"))
