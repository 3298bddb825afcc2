#!/usr/bin/env python
"""Per-launch table of one offline step (HIP events on the launch stream): which layer shapes are
far from their roofline.  GEMM rows: TFLOP/s (fp32-MFMA peak 157.3); HBM-bound rows: GB/s."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hilcodec_amd
from hilcodec_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--model", default="hil_speech")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--mode", default="offline", choices=["offline", "streaming"])
ap.add_argument("--no-stage", action="store_true", help="one launch per residual block and per down- / up-sampling layer (round 3) instead of one per stage")
args = ap.parse_args()
dev = torch.device("cuda:0")
mk = synth.model_kwargs(args.model)
model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
model.load_state_dict(synth.synth_state_dict(args.model, 7), strict=False)
for l in model.quantizer.layers:
    l.initted = True
model.encoder.exec_options.stage_launches = model.decoder.exec_options.stage_launches = not args.no_stage
x = synth.synth_clips(args.batch, 24000).to(dev)


def step():
    z = model.encoder(x)
    q, _, _, idx = model.quantizer(z, None, return_indices=True)
    return model.decoder(q)


if args.mode == "streaming":
    from hilcodec_amd.models.hilcodec.streaming import HILCodec as StreamingHILCodec
    smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
    smodel = StreamingHILCodec(24000, **smk).eval()
    smodel.load_offline_state_dict(synth.synth_state_dict(args.model, 7))
    smodel.remove_weight_reparameterizations()
    smodel.encoder.exec_options.stage_launches = smodel.decoder.exec_options.stage_launches = not args.no_stage
    nq = mk["vq_kwargs"]["num_quantizers"]
    xs = synth.synth_clips(args.batch, 320, seed=4321).to(dev)
    state = list(smodel.initialize_cache(xs))

    def step():
        z, state[0] = smodel.encoder(xs, *state[0])
        idx = smodel.quantizer(z, nq)
        q = smodel.dequantizer(idx, nq)
        wav, state[1] = smodel.decoder(q, *state[1])
        return wav


with torch.no_grad():
    step()
    torch.cuda.synchronize()
    with ops.timed_launches() as t:
        for _ in range(args.reps):
            step()
        torch.cuda.synchronize()
n = len(t.records) // args.reps
agg = {}
order = []
for i, (kind, work, e0, e1, tag) in enumerate(t.records):
    key = (i % n, kind, tag)
    if key not in agg:
        agg[key] = [work, 0.0]
        order.append(key)
    agg[key][1] += e0.elapsed_time(e1) * 1e-3 / args.reps
tot = sum(v[1] for v in agg.values())
print(f"{'#':>3} {'kind':10} {'shape':22} {'ms':>8} {'%':>6} {'rate':>10}")
for key in order:
    i, kind, tag = key
    work, sec = agg[key]
    rate = f"{work / sec / 1e12:7.1f} TF" if kind in ("pw_conv", "stft", "rvq_encode", "dws_conv", "resblock", "up_conv", "spec_block") else f"{work / sec / 1e9:7.0f} GB/s"
    print(f"{i:3d} {kind:10} {tag:22} {sec * 1e3:8.3f} {100 * sec / tot:6.2f} {rate:>10}")
print(f"total {tot * 1e3:.2f} ms/step")
