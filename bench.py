#!/usr/bin/env python
"""Throughput bench of the MI355X-native HILCodec encode -> RVQ -> decode path.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Default (BASELINE.json configs[1], the configuration the metric is quoted on): one "step" = one
offline pass (encoder -> RVQ(Nq) -> decoder) over a batch of 256 synthetic 1 s, 24 kHz clips per GPU,
inputs resident in HBM.  Metric = audio-seconds processed per wall second (xRT), whole job.  Clips
shard embarrassingly: rank r processes clips [r*256, (r+1)*256) (weak scaling); the only collective
is an all_gather of per-rank counters over RCCL after the timed region.

Other workloads: `--model hil_music` (configs[2]), `--mode streaming` (configs[3]: 1024 concurrent
streams per GPU, one 320-sample hop per step, the 52 cache tensors of every stream resident in HBM).
The DEFAULT invocation also times short runs of configs[2] and configs[3] after the headline region and
reports them under `other_configs` of the same JSON line (they are never part of `value`).

Multi-GPU on one GPU: `--force-dist` initialises the RCCL process group even at world size 1 (so that
`init_process_group("nccl", device_id=...)`, the barriers and the counters all_gather really execute);
`--emulate-rank r --emulate-world W` runs rank r's shard of the W-GPU job (e.g. configs[4]: hil_music,
2048 clips over 8 GPUs) in a single process — a shard check, not a scaling measurement.

Prints ONE JSON line (rank 0) with
  roofline     — dominant kernel = the fp32-MFMA GEMM core (pointwise convs and everything fused
                 onto them), every launch timed with HIP events on the launch stream INSIDE the timed
                 region; achieved = algorithmic FLOPs / summed launch time; peak 157.3 TFLOP/s.
  cpu_baseline — the CPU oracle (restatement of the reference in plain torch fp32 ops, bit-identical
                 to the reference, tests/test_oracle_vs_reference.py) timed on this box's host cores
                 over a bounded sample of the same workload (1 warm-up + 3 timed passes per thread
                 setting, median)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense; no xf32/TF32 on gfx950
HBM_PEAK_TBPS = 8.0                # MI355X_MICROARCH.md: HBM3E spec (6.29 measured with a float4 copy)
FLOP_PER_AUDIO_SECOND = {"hil_speech": 34.219e9, "hil_music": 34.298e9}
MFMA_KINDS = ("pw_conv", "dws_conv", "resblock", "up_conv", "spec_block")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--model", default=None, choices=["hil_speech", "hil_music"],
                    help="default: hil_speech (BASELINE configs[1]); with --gpus 8 / --emulate-world 8: hil_music (configs[4], 2048 clips over 8 GPUs)")
    ap.add_argument("--mode", default="offline", choices=["offline", "streaming"])
    ap.add_argument("--batch", type=int, default=None, help="clips (offline) / streams (streaming) per GPU")
    ap.add_argument("--samples", type=int, default=24000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the 2.5 s sustained clock / power sample")
    ap.add_argument("--no-launch-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="default run only: skip the short configs[2] / configs[3] lines")
    ap.add_argument("--other-configs", action="store_true", help="add the short configs[2] / configs[3] lines to a non-default offline "
                    "invocation too (sizes follow --batch: that many clips, four times as many streams)")
    ap.add_argument("--pipeline", action="store_true", help="with --graph: two-stage software pipeline over hops "
                    "(decoder of hop i-1 beside the encoder of hop i on a second HIP stream; +1 hop output latency)")
    ap.add_argument("--graph", action="store_true", help="streaming mode: replay each hop as one HIP graph "
                    "(hilcodec_amd/graph_step.py); per-launch timing is not available inside a graph")
    ap.add_argument("--groups", type=int, default=1, help="with --graph: split the streams into this many groups whose "
                    "chains run side by side on separate HIP streams inside the graph (same arithmetic, no added latency)")
    ap.add_argument("--exec-opt", action="append", default=[], metavar="NAME=0|1", help="A/B: set a boolean field of engine.ExecOptions on the "
                    "encoder and the decoder (stage_launches, wide_blocks, decoder_stage_narrow, stream_defer_spec); all of them change "
                    "launches, never results")
    ap.add_argument("--cpu-clips", type=int, default=8, help="clips of the bounded CPU-baseline sample (per timed pass)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even at world size 1")
    ap.add_argument("--emulate-rank", type=int, default=None, help="with --emulate-world W: run rank r's shard of the W-GPU job in this one process")
    ap.add_argument("--emulate-world", type=int, default=None)
    a = ap.parse_args()
    a.is_default = (a.model is None and a.mode == "offline" and a.batch is None and a.samples == 24000 and a.gpus == 1
                    and a.emulate_world is None and not a.exec_opt)
    if (a.emulate_rank is None) != (a.emulate_world is None):
        ap.error("--emulate-rank and --emulate-world go together")
    if a.emulate_world is not None and not 0 <= a.emulate_rank < a.emulate_world:
        ap.error("--emulate-rank must be in [0, --emulate-world)")
    if a.model is None:
        a.model = "hil_music" if ((a.gpus == 8 or a.emulate_world == 8) and a.mode == "offline") else "hil_speech"
    if a.mode == "offline":
        a.steps = 5 if a.steps is None else a.steps
        a.warmup = 2 if a.warmup is None else a.warmup
        a.batch = 256 if a.batch is None else a.batch
    else:
        a.steps = 75 if a.steps is None else a.steps
        a.warmup = 5 if a.warmup is None else a.warmup
        a.batch = 1024 if a.batch is None else a.batch
    return a


def smi_index_of(device_index: int, smi: str) -> int:
    """rocm-smi's index of the HIP device `device_index`: matched by PCI bus id — under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES remapping
    (common under launchers) a rank's HIP ordinal is NOT rocm-smi's physical index.  Falls back to the ordinal where either side does not say."""
    import re
    import subprocess
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        txt = subprocess.run([smi, "--showbus"], capture_output=True, text=True, timeout=10).stdout
        for m in re.finditer(r"GPU\[(\d+)\][^\n]*PCI Bus:\s*([0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2})", txt):
            if m.group(2).lower() == want:
                return int(m.group(1))
    except Exception:
        pass
    return device_index


def sustained_clock(step, first_index: int, seconds: float = 2.5, device_index: int = 0):
    """Shader clock and socket power while the benchmark's own steps run back to back: `rocm-smi` sampled from a
    thread (≈3 samples per second).  Returns medians, or None where rocm-smi is missing / prints something else."""
    import re
    import shutil
    import subprocess
    import threading
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    sclk, power, cap, stop = [], [], [], [False]
    smi_dev = smi_index_of(device_index, smi)

    def sampler():
        while not stop[0]:
            try:
                txt = subprocess.run([smi, "-d", str(smi_dev), "--showclocks", "--showpower", "--showmaxpower"], capture_output=True,
                                     text=True, timeout=10).stdout
            except Exception:
                return
            m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", txt)
            w = re.search(r"(?:Current|Average)[^\n]*Power \(W\):\s*([0-9.]+)", txt)
            c = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", txt)
            if m:
                sclk.append(float(m.group(1)))
            if w:
                power.append(float(w.group(1)))
            if c:
                cap[:] = [float(c.group(1))]

    th = threading.Thread(target=sampler, daemon=True)
    with torch.no_grad():
        step(first_index)
        torch.cuda.synchronize()
        th.start()
        t0 = time.perf_counter()
        i = first_index + 1
        while time.perf_counter() - t0 < seconds:
            step(i)
            i += 1
            torch.cuda.synchronize()
    stop[0] = True
    th.join(timeout=15)
    sclk, power = sclk[1:] or sclk, power[1:] or power      # the first sample straddles the ramp
    if not sclk:
        return None
    med = lambda v: sorted(v)[len(v) // 2]
    return {"sclk_mhz": med(sclk), "power_w": med(power) if power else None, "power_cap_w": cap[0] if cap else None,
            "samples": len(sclk),
            "how": "rocm-smi medians over %.1f s of the same steps after the timed region" % seconds}


def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_multiprocess(name, clips_per_worker: int, samples: int, workers: int, timeout: float = 240.0):
    """Whole-host CPU throughput the way a batch job would use a many-core box: `workers` independent single-threaded
    processes (clips are independent), each running the oracle on its own clips; value = all clips / (last end - first
    start) after a file rendezvous, so process start-up and `import torch` are outside the interval.  The workers see no
    GPU.  Returns None if anything goes wrong (the thread sweep stands alone then)."""
    import shutil
    import subprocess
    import tempfile
    rdv = tempfile.mkdtemp(prefix="hilc_cpu_")
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="",
               ROCR_VISIBLE_DEVICES="")
    procs = []
    try:
        for w in range(workers):
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "census.py"), "--model", name, "--cpu-worker",
                                           str(clips_per_worker), str(samples), str(w * clips_per_worker), rdv, str(w)],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT))
        t_end = time.time() + timeout
        while sum(os.path.exists(os.path.join(rdv, f"ready{w}")) for w in range(workers)) < workers:
            if time.time() > t_end or any(p.poll() not in (None, 0) for p in procs):
                raise RuntimeError("workers not ready")
            time.sleep(0.05)
        open(os.path.join(rdv, "go"), "w").close()
        outs = [json.loads(p.communicate(timeout=timeout)[0].strip().splitlines()[-1]) for p in procs]
        wall = max(o["t1"] for o in outs) - min(o["t0"] for o in outs)
        n = sum(o["clips"] for o in outs)
        return {"value": n * samples / 24000.0 / wall, "wall_s": wall, "workers": workers, "clips": n,
                "slowest_worker_s": max(o["t1"] - o["t0"] for o in outs), "fastest_worker_s": min(o["t1"] - o["t0"] for o in outs),
                "sample": f"{workers} single-threaded processes x {clips_per_worker} clips, one clip per call, interval from the "
                          f"first start to the last end after a rendezvous ({wall:.2f} s)"}
    except Exception:                                             # noqa: BLE001
        return None
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdv, ignore_errors=True)


def cpu_baseline(name, mk, sd, clips: int, samples: int, multiprocess: bool = True):
    """The CPU oracle (= the reference's arithmetic) timed on this box's host cores over a bounded sample of the same
    workload.  Thread sweep inside ONE process — 1, 8, 16, 32 and all physical cores; a pass = the same `clips`
    clips (1 thread, all cores) or the first 4 of them one clip per call (8 / 16 / 32 threads: BASELINE.md's survey measured
    the reference fastest that way), one untimed warm-up and THREE timed passes, the MEDIAN pass is the setting's figure —
    plus, where the host has the cores, one run of independent single-threaded PROCESSES (`cpu_multiprocess`), which is how a
    batch job fills a many-core box: torch's intra-op threading does not scale on this graph of small ops.  `value` = the
    best of all settings, `cores` = the cores that setting used.  The oracle's outputs for the sample are returned too: they
    double as the checker of the parity census below."""
    from hilcodec_amd import synth
    from tests import census                            # test infrastructure: the checker doubles as the timed CPU baseline
    x = synth.synth_clips(clips, samples, seed=1234)
    clip_s = samples / 24000.0
    phys = physical_cores()
    runs, keep = {}, None
    settings = [(1, clips, min(clips, 8))] + [(t, min(clips, 4), 1) for t in (8, 16, 32) if t < phys] + [(phys, clips, min(clips, 8))]
    for th, n, chunk in settings:
        if str(th) in runs:
            continue
        census.oracle_clips(name, sd, mk, x[:min(2, n)], chunk=min(2, chunk), threads=th)           # warm-up (untimed)
        times = []
        for _ in range(3):
            z_o, idx_o, wav_o, dt, _ = census.oracle_clips(name, sd, mk, x[:n], chunk=chunk, threads=th)
            times.append(dt)
        if n == clips and keep is None:
            keep = (z_o, idx_o, wav_o)
        med = sorted(times)[1]
        runs[str(th)] = {"value": n * clip_s / med, "cores": th, "passes_s": [round(t, 3) for t in times], "median_s": med,
                         "sample": f"{n} clips in chunks of {chunk}, 1 warm-up + 3 timed passes, median {med:.2f} s"}
    torch.set_num_threads(min(phys, 64))
    if multiprocess and phys >= 4:
        workers = min(phys, 64)
        mp = cpu_multiprocess(name, 2, samples, workers)
        if mp is not None:
            mp["cores"] = workers
            runs[f"{workers} processes x 1 thread"] = mp
    best = max(runs, key=lambda t: runs[t]["value"])
    base = {"value": runs[best]["value"], "unit": "audio-seconds/sec", "cores": runs[best]["cores"], "kind": "port",
            "sample": f"{runs[best]['sample']}; {clip_s:g} s clips, {name}, fp32, torch CPU ops (oracle = the reference's arithmetic); "
                      f"best of {len(runs)} settings ({', '.join(runs)}) on a host with {phys} physical cores / {os.cpu_count()} logical CPUs",
            "best_setting": best, "by_threads": runs}
    return base, keep


def parity_census(model, sd, nq, z, idx, wav, oracle_out):
    """Every RVQ index of the CPU-baseline sample (the first clips of the bench batch) against the oracle; the 64-clip
    census is tests/test_gpu_census.py, this is the same comparison riding on the cpu_baseline leg's outputs."""
    from tests import census
    z_o, idx_o, wav_o = oracle_out
    n = z_o.shape[0]
    pad = torch.zeros(idx.shape[0] - n, *idx_o.shape[1:], dtype=torch.int64)
    wav_ri = census.gpu_decode_indices(model, torch.cat([idx_o, pad]).to(idx.device))[:n]
    out = census.compare(sd, nq, z[:n].cpu(), idx[:n].cpu(), wav[:n].cpu(), wav_ri.cpu(), z_o, idx_o, wav_o)
    out.pop("flips", None)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# workloads: each returns (step, audio_seconds_per_step, context)
# ---------------------------------------------------------------------------------------------------------------------
def offline_workload(name: str, n_clips: int, first: int, T: int, dev):
    import hilcodec_amd
    from hilcodec_amd import synth
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=7)
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    model.load_state_dict(sd, strict=False)
    for l in model.quantizer.layers:
        l.initted = True
    x = synth.synth_clips(n_clips, T, seed=1234, first=first).to(dev)
    last = {}

    def step(i):
        z = model.encoder(x)
        q, _, _, idx = model.quantizer(z, None, return_indices=True)
        wav = model.decoder(q)
        last["z"] = z
        return idx, wav

    return step, n_clips * T / 24000.0, {"model": model, "sd": sd, "mk": mk, "last": last}


def streaming_workload(name: str, n_streams: int, first: int, dev, graph: bool, pipeline: bool, groups: int = 1):
    from hilcodec_amd import synth
    from hilcodec_amd.models.hilcodec.streaming import HILCodec as StreamingHILCodec
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=7)
    nq = mk["vq_kwargs"]["num_quantizers"]
    smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
    model = StreamingHILCodec(24000, **smk).eval()
    model.load_offline_state_dict(sd)
    model.remove_weight_reparameterizations()
    hop = 320
    nbuf = 8                                           # distinct input hops, cycled
    xs = [synth.synth_clips(n_streams, hop, seed=4321 + 7 * j, first=first).to(dev) for j in range(nbuf)]
    ctx = {"model": model, "sd": sd, "mk": mk, "xs": xs, "hop": hop, "nq": nq}
    if graph:
        from hilcodec_amd.graph_step import GraphedHop, PipelinedHop
        ctx["make_hopper"] = lambda: (PipelinedHop(model, n_streams, hop, nq, dev, groups=groups) if pipeline
                                      else GraphedHop(model, n_streams, hop, nq, dev, groups=groups))
        ctx["hopper"] = ctx["make_hopper"]()

        def step(i):
            return ctx["hopper"].step(xs[i % nbuf])
    else:
        from hilcodec_amd.graph_step import StateBlock
        blocks = (StateBlock(model, n_streams, dev), StateBlock(model, n_streams, dev))   # persistent ping-pong state in HBM

        def step(i):
            src, dst = blocks[i & 1], blocks[(i & 1) ^ 1]
            z, _ = model.encoder(xs[i % nbuf], *src.enc, cache_out=dst.enc)
            idx = model.quantizer(z, nq)
            q = model.dequantizer(idx, nq)
            wav, _ = model.decoder(q, *src.dec, cache_out=dst.dec)
            return idx, wav

    return step, n_streams * hop / 24000.0, ctx


def timed_region(step, steps: int, warmup: int, D, launch_timing: bool):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides."""
    import contextlib
    from hilcodec_amd import ops
    with torch.no_grad():
        for i in range(warmup):
            idx, wav = step(i)
        torch.cuda.synchronize()
        cm = ops.timed_launches() if launch_timing else contextlib.nullcontext()
        with cm as timer:
            D.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                idx, wav = step(warmup + i)
            torch.cuda.synchronize()
            D.barrier()
            dt = time.perf_counter() - t0
    return dt, idx, wav, timer


def other_config_lines(dev, D, clips: int = 256, streams: int = 1024, census_clips: int = 0):
    """Short, separately reported runs of BASELINE configs[2] and configs[3] (never part of `value`): the driver sees them in
    the one JSON line of the default invocation."""
    out = {}

    def line(workload, name, step, audio, steps, warmup, extra=None):
        dt, idx, _wav, _ = timed_region(step, steps, warmup, D, False)
        xrt = audio * steps / dt
        tf = xrt * FLOP_PER_AUDIO_SECOND[name] / 1e12
        d = {"workload": workload, "value": xrt, "unit": "audio-seconds/sec", "ms_per_step": dt / steps * 1e3, "steps": steps,
             "warmup": warmup, "whole_path_tflops": tf, "whole_path_frac": tf / FP32_MFMA_PEAK_TFLOPS,
             "index_checksum": int(idx.sum().item()), "dtype": "f32"}
        d.update(extra or {})
        return d

    step, audio, _ctx = offline_workload("hil_music", clips, 0, 24000, dev)
    out["configs[2]"] = line(f"hil_music, batch={clips}x1 s 24 kHz, Nq=12, offline encode+RVQ+decode", "hil_music", step, audio, 10, 2)
    if census_clips > 0 and clips >= census_clips:
        # the same comparison the headline model gets from the cpu_baseline leg, after the timed region: the oracle on the batch's first clips,
        # EVERY RVQ index of them, |dz|, |dwav| on the reference's own indices and end to end (tests/census.py)
        try:
            from hilcodec_amd import synth
            from tests import census
            idx, wav = step(0)
            torch.cuda.synchronize()
            x = synth.synth_clips(census_clips, 24000, seed=1234)
            z_o, idx_o, wav_o, _, _ = census.oracle_clips("hil_music", _ctx["sd"], _ctx["mk"], x, chunk=min(census_clips, 4))
            out["configs[2]"]["parity_census"] = parity_census(_ctx["model"], _ctx["sd"], _ctx["mk"]["vq_kwargs"]["num_quantizers"],
                                                               _ctx["last"]["z"], idx, wav, (z_o, idx_o, wav_o))
        except Exception as e:                                     # noqa: BLE001
            out["configs[2]"]["parity_census"] = {"error": f"{type(e).__name__}: {e}"}
    del step, _ctx
    torch.cuda.empty_cache()
    for key, pipeline, groups in (("configs[3] graph", False, 1), ("configs[3] graph, 2 stream groups", False, 2),
                                  ("configs[3] pipelined graph", True, 1), ("configs[3] pipelined graph, 2 stream groups", True, 2)):
        step, audio, _ctx = streaming_workload("hil_speech", streams, 0, dev, True, pipeline, groups)
        out[key] = line(f"hil_speech streaming, hop=320, {streams} concurrent streams, Nq=8, 22+30 caches per stream resident in HBM, "
                        "one HIP-graph replay per hop"
                        + (", streams in 2 groups on parallel HIP streams inside the graph (same arithmetic, no latency added by the grouping)" if groups > 1 else "")
                        + (", decoder of hop i-1 pipelined beside the encoder of hop i (+1 hop = 13.3 ms output latency)" if pipeline else ""),
                        "hil_speech", step, audio, 40, 4)
        del step, _ctx
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    from hilcodec_amd import distributed as D
    rank, world, local = D.env_rank_world()
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    D.init("nccl", dev, force=args.force_dist)

    from hilcodec_amd import _lib

    name = args.model
    B, T = args.batch, args.samples
    # this process's clips / streams of the global batch: its own rank's slice, or (emulation) another rank's
    shard_rank, shard_world = (args.emulate_rank, args.emulate_world) if args.emulate_world else (rank, world)
    lo, hi = D.shard_range(B * shard_world, shard_rank, shard_world)

    if args.mode == "offline":
        step, audio_per_step, ctx = offline_workload(name, hi - lo, lo, T, dev)
    else:
        step, audio_per_step, ctx = streaming_workload(name, hi - lo, lo, dev, args.graph, args.pipeline, args.groups)
        if args.graph:
            args.no_launch_timing = True
    model, sd, mk = ctx["model"], ctx["sd"], ctx["mk"]
    nq = mk["vq_kwargs"]["num_quantizers"]
    if args.exec_opt:
        for item in args.exec_opt:
            key, val = item.split("=")
            for half in (model.encoder, model.decoder):
                assert isinstance(getattr(half.exec_options, key), bool), key
                setattr(half.exec_options, key, bool(int(val)))
        if args.mode == "streaming" and args.graph:
            ctx["hopper"] = ctx["make_hopper"]()                 # captured with the defaults: capture again

    dt, idx, wav, timer = timed_region(step, args.steps, args.warmup, D, not args.no_launch_timing)

    per_rank = D.gather_counters({"clips": float((hi - lo) * args.steps), "audio_s": audio_per_step * args.steps,
                                  "wall_s": dt, "index_checksum": float(idx.sum().item())}, dev)   # the data path has NO collective: this gathers the counters (a second gather below carries each rank's clock / power)
    rank_sustained = None
    if world > 1 and not args.no_clock_probe:
        # all ranks keep stepping together for 2.5 s (the node's power budget is shared: a sub-linear curve can then be told apart
        # from eight chips throttling each other) and each samples ITS device; outside the timed region, one more 32-byte gather
        sus_r = sustained_clock(step, args.warmup + args.steps, device_index=local) or {}
        rank_sustained = D.gather_counters({"sclk_mhz": sus_r.get("sclk_mhz") or 0.0, "power_w": sus_r.get("power_w") or 0.0,
                                            "power_cap_w": sus_r.get("power_cap_w") or 0.0}, dev)
    if rank == 0:
        agg = D.aggregate(per_rank)
        value = agg["xrt"]
        cfg_ix = {("offline", "hil_speech"): 1, ("offline", "hil_music"): 2, ("streaming", "hil_speech"): 3}.get(
            (args.mode, name), None)
        if args.mode == "offline" and name == "hil_music" and shard_world == 8 and B == 256:
            cfg_ix = 4                                   # hil_music, 2048 clips sharded over 8 GPUs
        if args.mode == "offline":
            workload = (f"{name}, batch={B}x{T / 24000.0:g} s 24 kHz per GPU, Nq={nq}, offline encode+RVQ+decode "
                        f"(BASELINE configs[{cfg_ix}])")
        else:
            workload = (f"{name} streaming, hop=320, {B} concurrent streams per GPU, Nq={nq}, 22+30 caches per stream "
                        f"resident in HBM (BASELINE configs[{cfg_ix}])" + (", one HIP-graph replay per hop" if args.graph else "")
                        + (f", streams in {args.groups} groups on parallel HIP streams inside the graph" if args.graph and args.groups > 1 else "")
                        + (", decoder of hop i-1 pipelined beside the encoder of hop i (+1 hop output latency)"
                           if args.graph and args.pipeline else ""))
        if args.emulate_world:
            workload += (f" — EMULATED rank {shard_rank} of {shard_world}: this process ran clips [{lo}, {hi}) of the "
                         f"{B * shard_world}-clip job on one GPU (a shard check, not a scaling measurement)")
        out = {
            "metric": "audio-seconds/sec (xRT) encode+RVQ+decode, 24 kHz batch=256",
            "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": agg["wall_s"] / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "global_batch": B * shard_world, "samples_per_clip": T if args.mode == "offline" else 320,
                       "parallelism": f"clip-sharded x{shard_world}, replicated weights, counters all_gather only",
                       "shard": [lo, hi]},
            "frames_per_sec": value * 75.0,
            "index_checksum": int(agg["index_checksum"]),
            "ranks": {"backend": D.backend_name(), "rccl_ranks": world if D.is_initialized() else 0,
                      "process_group_initialized": D.is_initialized(), "collectives_in_timed_region": 0,
                      "wall_s_per_rank": [r["wall_s"] for r in per_rank],
                      "wall_skew_s": max(r["wall_s"] for r in per_rank) - min(r["wall_s"] for r in per_rank)},
            "build": {"csrc_sha16": _lib.source_hash(), "abi": _lib.ABI_VERSION},
        }
        whole_tflops = value * FLOP_PER_AUDIO_SECOND[name] / 1e12 / world
        roof = {"bound": "mfma", "achieved": None, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "traffic": None, "whole_path_tflops_per_gpu": whole_tflops,
                "whole_path_frac": whole_tflops / FP32_MFMA_PEAK_TFLOPS}
        if timer is not None:
            tot = timer.totals()
            kinds = [k for k in MFMA_KINDS if k in tot]
            launches = sum(tot[k][0] for k in kinds)
            flops = sum(tot[k][1] for k in kinds)
            secs = sum(tot[k][2] for k in kinds)
            roof.update({
                "kernel": "hilc::gemm_lin_kernel<MB,BOp,Epilogue> / gemm_lin_wr_kernel<BOp,Epilogue> / gemm_kernel<MB,Loader,Epilogue> + resblock_kernel<C,STREAM,..,NB,W8,DRU,POST,SPEC0> (residual blocks / whole stages: chain, + down- or up-sampling phase, + the closing conv behind the last decoder stage, + first conv and SpecBlock in front of the first encoder stage) + spec_block_kernel<N> ("
                          + "+".join(kinds) + "; fp32 v_mfma_f32_32x32x2_f32)",
                "achieved": flops / secs / 1e12, "frac": flops / secs / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                "launches_per_step": launches // args.steps, "avg_launch_us": secs / launches * 1e6,
                "flop_per_launch_avg": flops / launches,
                "share_of_gpu_time": secs / sum(v[2] for v in tot.values()),
            })
            out["kernel_time_breakdown_ms_per_step"] = {k: v[2] / args.steps * 1e3 for k, v in sorted(tot.items())}
            out["hbm_bound_ops_GBps"] = {k: v[1] / v[2] / 1e9 for k, v in tot.items()
                                         if k in ("dw_conv", "dw_convtr", "conv_pre", "conv_post")}
        # HBM traffic of the dominant kernel family comes from rocprofv3 PMC passes of this same command
        # (FETCH_SIZE*1024*2 + WRITE_SIZE*1024, separate passes; tools/summarize_profile.py) — it cannot be
        # sampled from inside the process, so the committed summary of the latest profiled build is quoted, together with
        # the build it was measured on: `stale` = the kernel sources have changed since.
        try:
            with open(os.path.join(ROOT, "profiles", "latest_mfma_family.json")) as f:
                prof = json.load(f)
            if args.mode == "offline" and name == "hil_speech" and B == 256 and T == 24000:
                roof["traffic"] = prof["hbm_bytes_per_launch"]
                roof["traffic_unit"] = "bytes per launch (avg over the family), rocprofv3 PMC, profiles/latest_mfma_family.json"
                roof["traffic_build"] = {"csrc_sha16": prof.get("csrc_sha16"), "git_sha": prof.get("git_sha"),
                                         "stale": prof.get("csrc_sha16") != _lib.source_hash()}
                roof["algorithmic_bytes_per_step"] = 196800 * B + 38150404 + nq * 524288 + 5602816
                # north_star asks for the fraction of the HBM roofline too: measured HBM bytes of the whole step (PMC, all kernels of
                # the family) over this run's step time against 8 TB/s — a few per cent: the path is matrix-bound, not HBM-bound
                step_bytes = (prof.get("hbm_read_GB_per_step", 0.0) + prof.get("hbm_write_GB_per_step", 0.0)) * 1e9
                if step_bytes:
                    roof["hbm_bytes_per_step"] = step_bytes
                    roof["hbm_peak_TBps"] = HBM_PEAK_TBPS
                    roof["hbm_frac"] = step_bytes / (agg["wall_s"] / args.steps) / (HBM_PEAK_TBPS * 1e12)
                    roof["hbm_frac_algorithmic"] = roof["algorithmic_bytes_per_step"] / (agg["wall_s"] / args.steps) / (HBM_PEAK_TBPS * 1e12)
        except (OSError, KeyError, ValueError):
            pass
        if rank_sustained is not None:
            roof["sustained_per_rank"] = rank_sustained
        if world == 1 and not args.no_clock_probe:
            # the fp32-MFMA peak assumes 2.4 GHz; this workload runs at the socket power limit and the clock follows
            # (profiles/r02_experiments.md).  Sampled AFTER the timed region, while the same steps keep running.
            sus = sustained_clock(step, args.warmup + args.steps, device_index=local)
            if sus is not None:
                sus["peak_at_sclk_tflops"] = FP32_MFMA_PEAK_TFLOPS * sus["sclk_mhz"] / 2400.0
                sus["whole_path_frac_at_sclk"] = whole_tflops / sus["peak_at_sclk_tflops"]
                roof["sustained"] = sus
        out["roofline"] = roof
        if world == 1 and not args.no_other_configs and (args.is_default or (args.other_configs and args.mode == "offline")):
            # auxiliary lines: a failure here (it would be a bug) must not cost the headline line its measurement — it is reported
            # in place of the lines, and tests/test_gpu_bench.py fails on it
            try:
                out["other_configs"] = other_config_lines(dev, D, B, 4 * B, 0 if args.no_cpu_baseline else min(4, args.cpu_clips))
            except Exception as e:                                     # noqa: BLE001
                out["other_configs"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], oracle_out = cpu_baseline(name, mk, sd, args.cpu_clips, T)
                if args.mode == "offline" and lo == 0 and hi - lo >= args.cpu_clips:
                    out["parity_census"] = parity_census(model, sd, nq, ctx["last"]["z"], idx, wav, oracle_out)
            except Exception as e:                                     # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
