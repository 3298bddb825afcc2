#!/usr/bin/env python
"""Same-box, alternating A/B of the offline step (256 x 1 s, hil_speech) with and without a monkeypatch of the product's Python layer.
    python tools/ab_patch.py "ops.decoder_stage_post_supported=lambda *a: False" [--rounds 4] [--steps 10]
A = the tree as it is, B = with the patch applied (an attribute of hilcodec_amd.ops / hilcodec_amd.engine replaced).  Prints ms per step
of every round and whether the index checksum and the waveform agree."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hilcodec_amd
from hilcodec_amd import engine, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("patch")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--model", default="hil_speech")
args = ap.parse_args()
target, expr = args.patch.split("=", 1)
modname, attr = target.split(".", 1)
mod = {"ops": ops, "engine": engine}[modname]
orig = getattr(mod, attr)
new = eval(expr, {"ops": ops, "engine": engine, "torch": torch, "orig": orig})
dev = torch.device("cuda:0")
mk = synth.model_kwargs(args.model)
model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
model.load_state_dict(synth.synth_state_dict(args.model, 7), strict=False)
for l in model.quantizer.layers:
    l.initted = True
x = synth.synth_clips(256, 24000, seed=1234).to(dev)


def step():
    z = model.encoder(x)
    q, _, _, idx = model.quantizer(z, None, return_indices=True)
    return idx, model.decoder(q)


def timed(patched):
    setattr(mod, attr, new if patched else orig)
    with torch.no_grad():
        for _ in range(3):
            idx, wav = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            idx, wav = step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3, idx, wav


res = {False: [], True: []}
outs = {}
for r in range(args.rounds):
    for patched in (False, True):
        ms, idx, wav = timed(patched)
        res[patched].append(ms)
        outs[patched] = (idx.clone(), wav.clone())
setattr(mod, attr, orig)
print("A (tree)   :", " ".join(f"{m:.3f}" for m in res[False]), f" mean {sum(res[False]) / len(res[False]):.3f} ms")
print("B (patched):", " ".join(f"{m:.3f}" for m in res[True]), f" mean {sum(res[True]) / len(res[True]):.3f} ms   [{args.patch}]")
print("indices equal:", bool(torch.equal(outs[False][0], outs[True][0])), " wav equal:", bool(torch.equal(outs[False][1], outs[True][1])),
      " |dwav| max:", float((outs[False][1] - outs[True][1]).abs().max()))
