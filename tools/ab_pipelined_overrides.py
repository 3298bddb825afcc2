#!/usr/bin/env python
"""PipelinedHop's capture-time ExecOptions overrides, A/B (round 6: with every stage of a hop ONE launch the plain graph replay overtook the
pipelined schedule; which launch structure does the two-chain schedule want now?).
   python tools/ab_pipelined_overrides.py"""
import contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import engine, graph_step, synth
from hilcodec_amd.models.hilcodec.streaming import HILCodec as StreamingHILCodec

dev = torch.device("cuda:0")
mk = synth.model_kwargs("hil_speech")
smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
model = StreamingHILCodec(24000, **smk).eval()
model.load_offline_state_dict(synth.synth_state_dict("hil_speech", 7))
model.remove_weight_reparameterizations()
xs = [synth.synth_clips(1024, 320, seed=4321 + 7 * j).to(dev) for j in range(8)]
real = engine.exec_overrides
for name, over in (("decoder_stage_narrow=False (shipped)", dict(decoder_stage_narrow=False)), ("no override", {}),
                   ("wide_blocks=False, narrow=False", dict(decoder_stage_narrow=False, wide_blocks=False)),
                   ("stage_launches=False", dict(stage_launches=False))):
    # PipelinedHop asks for decoder_stage_narrow=False itself: replace what it asks for by `over`
    def patched(**fields):
        return real(**over) if over else contextlib.nullcontext()
    engine.exec_overrides = patched
    enc_over = real(**{k: v for k, v in over.items() if k != "decoder_stage_narrow"}) if over else contextlib.nullcontext()
    with enc_over:
        hop = graph_step.PipelinedHop(model, 1024, 320, 8, dev)
    for i in range(6):
        hop.step(xs[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60):
        hop.step(xs[i % 8])
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms per hop", flush=True)
    engine.exec_overrides = real
    del hop
    torch.cuda.empty_cache()
