#!/usr/bin/env python
"""Is the offline step bound by the socket power limit rather than by instruction issue?  The same launches on operands that
toggle fewer or more bits: time per launch, sustained shader clock and socket power (rocm-smi, `bench.sustained_clock`).

  1. one wide depthwise-separable layer (K = M = 768, T = 600, 256 clips; the same launch every time), input = zeros /
     the layer's real activations' scale (N(0, 0.05)) / N(0, 1) / N(0, 1) with the weights N(0, 1) too;
  2. the whole encode -> RVQ -> decode step on silence / the synthetic clips of the bench / full-scale white noise.

If the launches were issue- or latency-bound the time per launch would not depend on the VALUES; under a power cap the
clock follows the switching activity of the operands.  usage: python tools/data_power_probe.py [seconds per variant]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from hilcodec_amd import ops

dev = torch.device("cuda:0")
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def timed(step, n):
    """ms per call over n back-to-back calls, after the clock has settled (the probe ran just before)"""
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        for i in range(n):
            step(i)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def probe(label, step, calls):
    s = bench.sustained_clock(step, 0, SECONDS) or {}
    ms = timed(step, calls)
    out = {"variant": label, "ms_per_call": round(ms, 4), "sclk_mhz": s.get("sclk_mhz"), "power_w": s.get("power_w")}
    print(json.dumps(out), flush=True)
    return out


def layer_variants():
    B, K, M, T = 256, 768, 768, 600
    g = torch.Generator(device=dev); g.manual_seed(3)
    dw = torch.randn(M, 5, device=dev, generator=g) * 0.4
    db = torch.randn(M, device=dev, generator=g) * 0.2
    w_small = torch.randn(K, M, device=dev, generator=g) / K ** 0.5
    w_big = torch.randn(K, M, device=dev, generator=g)
    flop = 2.0 * B * K * M * T
    res = []
    for label, xs, wt in [("x = 0", 0.0, w_small), ("x ~ N(0, 0.05)", 0.05, w_small), ("x ~ N(0, 1)", 1.0, w_small),
                          ("x ~ N(0, 1), w ~ N(0, 1)", 1.0, w_big)]:
        x = torch.randn(B, K, T, device=dev, generator=g) * xs
        step = lambda i, x=x, wt=wt: [ops.dws_conv(x, wt, dw, db, in_scale=0.9, in_elu=True, out_elu=True) for _ in range(20)]
        r = probe("dws_conv K768 M768 T600, " + label, step, 10)
        r["ms_per_call"] = round(r["ms_per_call"] / 20, 4)
        r["tflops"] = round(flop / (r["ms_per_call"] * 1e-3) / 1e12, 1)
        print("   -> per launch:", json.dumps(r), flush=True)
        res.append(r)
    return res


def step_variants():
    from hilcodec_amd import synth
    res = []
    step, _, ctx = bench.offline_workload("hil_speech", 256, 0, 24000, dev)
    model = ctx["model"]
    g = torch.Generator(device=dev); g.manual_seed(5)
    inputs = {"silence": torch.zeros(256, 1, 24000, device=dev),
              "bench clips (synth.synth_clips)": synth.synth_clips(256, 24000, seed=1234).to(dev),
              "white noise, uniform in [-1, 1)": torch.rand(256, 1, 24000, device=dev, generator=g) * 2 - 1}
    for label, x in inputs.items():
        def run(i, x=x):
            z = model.encoder(x)
            q, _, _, idx = model.quantizer(z, None, return_indices=True)
            return model.decoder(q)
        res.append(probe("offline step, input = " + label, run, 10))
    return res


if __name__ == "__main__":
    out = {"layer": layer_variants(), "step": step_variants(), "seconds_per_variant": SECONDS}
    print(json.dumps(out))
