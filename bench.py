#!/usr/bin/env python
"""Throughput bench of the MI355X-native HILCodec encode -> RVQ -> decode path.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one offline pass (encoder -> RVQ(Nq) -> decoder) over a batch of 256 synthetic 1 s,
24 kHz clips per GPU, inputs resident in HBM.  Metric = audio-seconds processed per wall second
(xRT), whole job.  Clips shard embarrassingly: every rank processes its own 256 clips (weak scaling);
the only collective is a gather of per-rank counters (RCCL), as in BASELINE.json's north star.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the fp32-MFMA pointwise-conv GEMM,
timed per launch with HIP events on the launch stream inside the timed region) and `cpu_baseline`
(the CPU oracle timed on the host cores over a bounded sample of the same workload)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, no xf32 on gfx950
FLOP_PER_AUDIO_SECOND = {"hil_speech": 34.219e9, "hil_music": 34.298e9}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="hil_speech", choices=["hil_speech", "hil_music"])
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--samples", type=int, default=24000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timing", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=16)
    return ap.parse_args()


def cpu_baseline(name, mk, sd, clips: int, samples: int):
    """The oracle (CPU restatement of the reference, plain torch fp32 ops == the reference's own
    arithmetic) on the host cores, bounded sample."""
    from hilcodec_amd import synth
    from oracle import hilcodec_oracle as O
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    x = synth.synth_clips(clips, samples, seed=1234)
    chunk = 8
    with torch.no_grad():
        O.codec_forward(sd, x[:1], mk)                        # warm-up
        t0 = time.perf_counter()
        for i in range(0, clips, chunk):
            O.codec_forward(sd, x[i:i + chunk], mk)
        dt = time.perf_counter() - t0
    audio_s = clips * samples / 24000.0
    return {"value": audio_s / dt, "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
            "sample": f"{clips} clips x {samples / 24000.0:g} s in chunks of {chunk}, {name}, fp32, torch CPU ops, "
                      f"{threads} threads, {dt:.2f} s wall"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import hilcodec_amd
    from hilcodec_amd import ops, synth

    name = args.model
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=7)
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    model.load_state_dict(sd, strict=False)
    for l in model.quantizer.layers:
        l.initted = True
    nq = mk["vq_kwargs"]["num_quantizers"]

    # this rank's shard of the global batch: clips [rank*B, (rank+1)*B)
    B, T = args.batch, args.samples
    x = synth.synth_clips(B, T, seed=1234, first=rank * B).to(dev)

    def step():
        z = model.encoder(x)
        q, _, _, idx = model.quantizer(z, None, return_indices=True)
        wav = model.decoder(q)
        return idx, wav

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    with torch.no_grad():
        for _ in range(args.warmup):
            idx, wav = step()
        torch.cuda.synchronize()
        timer = None
        if not args.no_launch_timing:
            timer = ops.LaunchTimer()
            ops.TIMER = timer
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            idx, wav = step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        ops.TIMER = None

    checksum = int(idx.sum().item())
    counters = torch.tensor([float(B * args.steps), B * args.steps * T / 24000.0, dt, float(checksum)],
                            dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        gathered = [torch.zeros_like(counters) for _ in range(world)]
        dist.all_gather(gathered, counters)
        gathered = torch.stack(gathered).cpu()
    else:
        gathered = counters.cpu().unsqueeze(0)

    if rank == 0:
        wall = float(gathered[:, 2].max())
        audio_s = float(gathered[:, 1].sum())
        value = audio_s / wall
        out = {
            "metric": "audio-seconds/sec (xRT) encode+RVQ+decode, 24 kHz batch=256",
            "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}, batch={B}x{T / 24000.0:g} s 24 kHz per GPU, Nq={nq}, offline "
                                   f"encode+RVQ+decode (BASELINE configs[{1 if name == 'hil_speech' else 2}])",
                       "global_batch": B * world, "samples_per_clip": T, "parallelism": f"clip-sharded x{world}"},
            "frames_per_sec": value * 75.0,
            "index_checksum": int(gathered[:, 3].sum()),
        }
        whole_tflops = value * FLOP_PER_AUDIO_SECOND[name] / 1e12 / world
        roof = {"bound": "mfma", "achieved": None, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "traffic": None, "whole_path_tflops_per_gpu": whole_tflops,
                "whole_path_frac": whole_tflops / FP32_MFMA_PEAK_TFLOPS}
        if timer is not None:
            tot = timer.totals()
            # dominant kernel = the fp32-MFMA GEMM core (every instantiation of hilc::gemm_kernel /
            # the fused residual-block kernel built on it): pointwise convs with their fused epilogues
            mfma_kinds = [k for k in ("pw_conv", "dws_conv", "resblock") if k in tot]
            launches = sum(tot[k][0] for k in mfma_kinds)
            flops = sum(tot[k][1] for k in mfma_kinds)
            secs = sum(tot[k][2] for k in mfma_kinds)
            roof.update({
                "kernel": "hilc::gemm_kernel<MB,Loader,Epilogue> family (" + "+".join(mfma_kinds) +
                          "; fp32 v_mfma_f32_32x32x2_f32)",
                "achieved": flops / secs / 1e12, "frac": flops / secs / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                "launches_per_step": launches // args.steps, "avg_launch_us": secs / launches * 1e6,
                "flop_per_launch_avg": flops / launches,
                "share_of_gpu_time": secs / sum(v[2] for v in tot.values()),
            })
            out["kernel_time_breakdown_ms_per_step"] = {k: v[2] / args.steps * 1e3 for k, v in sorted(tot.items())}
            hbm = {k: v[1] / v[2] / 1e9 for k, v in tot.items() if k in ("dw_conv", "dw_convtr", "conv_pre", "conv_post")}
            out["hbm_bound_ops_GBps"] = hbm
        out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, mk, sd, args.cpu_clips, T)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
