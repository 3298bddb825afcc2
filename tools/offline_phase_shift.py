#!/usr/bin/env python
"""Experiment: does a flatter power profile buy clock?  The offline batch (256 clips) as two half-batches on two HIP streams, the
second chain started half a chain late, so that narrow encoder launches (light) run beside wide decoder launches (heavy)
instead of the whole chip moving through light and heavy phases together.  Compares ms per 256 clips, clock and power with the
plain step.  usage: python tools/offline_phase_shift.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from hilcodec_amd import ops

dev = torch.device("cuda:0")
step, _, ctx = bench.offline_workload("hil_speech", 256, 0, 24000, dev)
model = ctx["model"]
from hilcodec_amd import synth
x = synth.synth_clips(256, 24000, seed=1234).to(dev)
halves = [x[:128].contiguous(), x[128:].contiguous()]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ws = [ops.SchedWorkspace(dev), ops.SchedWorkspace(dev)]


def chain(xh):
    z = model.encoder(xh)
    q, _, _, idx = model.quantizer(z, None, return_indices=True)
    return idx, model.decoder(q)


def shifted(n, shift=True):
    """n passes over both halves; chain B starts when chain A's first encoder is done (shift) or at once"""
    main = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(main)
    ev = torch.cuda.Event()
    outs = [None, None]
    with torch.no_grad():
        for i in range(n):
            with torch.cuda.stream(streams[0]), ops.sched_workspace(ws[0]):
                if i == 0 and shift:
                    z = model.encoder(halves[0]); ev.record(streams[0])
                    q, _, _, idx = model.quantizer(z, None, return_indices=True)
                    outs[0] = (idx, model.decoder(q))
                else:
                    outs[0] = chain(halves[0])
            with torch.cuda.stream(streams[1]), ops.sched_workspace(ws[1]):
                if i == 0 and shift:
                    streams[1].wait_event(ev)
                outs[1] = chain(halves[1])
    for s in streams:
        main.wait_stream(s)
    return outs


def measure(label, fn, n):
    fn(2); torch.cuda.synchronize()
    s = bench.sustained_clock(lambda i: fn(2), 0, 3.0) or {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn(n); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print(json.dumps({"schedule": label, "ms_per_256_clips": round(ms, 3), "sclk_mhz": s.get("sclk_mhz"), "power_w": s.get("power_w")}), flush=True)
    return out


def plain(n):
    with torch.no_grad():
        for i in range(n):
            o = step(i)
    return o


o_plain = measure("one chain, 256 clips per launch (the bench's step)", plain, 10)
o_two = measure("two chains of 128 clips, started together", lambda n: shifted(n, False), 10)
o_shift = measure("two chains of 128 clips, the second one encoder late", lambda n: shifted(n, True), 10)
idx = torch.cat([o_shift[0][0], o_shift[1][0]], dim=0) if o_shift[0][0].shape[0] == 128 else None
print("indices equal to the plain step:", bool(idx is not None and torch.equal(idx, o_plain[0])), tuple(o_plain[0].shape), tuple(o_shift[0][0].shape))
