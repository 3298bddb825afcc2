"""Full-size parity census (test infrastructure; imported by tests/test_gpu_census.py and by bench.py's cpu_baseline leg).

The GPU path runs BASELINE's full batch (256 clips: the real launch shapes); the CPU oracle — bit-identical to the
reference, tests/test_oracle_vs_reference.py — runs the first `n_clips` of the same batch, and EVERY RVQ index of those
clips is compared.  A differing frame is classified by the oracle's own fp64 best-vs-second distance gap at the first
differing stage: below `tie_eps` it is a near-tie that the 1e-6-level difference in z may legitimately flip (the
reference's argmin on another BLAS would flip it too); at or above it is a genuine mismatch.

    python tests/census.py --model hil_speech --clips 64 [--lib path/to/other/libhilcodec_amd.so]

(`--lib` re-runs the census on another build of the kernels, e.g. the expm1f-ELU A/B build.)"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def compare(sd, nq, z_gpu, idx_gpu, wav_gpu, wav_refidx_gpu, z_o, idx_o, wav_o, tie_eps=1e-4) -> dict:
    """All tensors on the CPU, first dimension = the compared clips."""
    import torch
    from oracle import hilcodec_oracle as O
    n = z_o.shape[0]
    same = idx_gpu == idx_o
    flips = []
    if not bool(same.all()):
        gaps = O.rvq_gaps_fp64(sd, z_o, idx_o)
        for b, t in sorted({(b, t) for b, s, t in (~same).nonzero().tolist()}):
            s0 = int((~same[b, :, t]).nonzero()[0])
            flips.append({"clip": b, "frame": t, "stage": s0, "fp64_gap": float(gaps[b, s0, t])})
    clean = [b for b in range(n) if bool(same[b].all())]
    out = {
        "clips": n, "argmins": int(idx_o.numel()),
        "dz_max": float((z_gpu - z_o).abs().max()),
        "index_mismatch_frames": len(flips),
        "near_tie_flips": sum(f["fp64_gap"] < tie_eps for f in flips),
        "genuine_mismatches": sum(f["fp64_gap"] >= tie_eps for f in flips),
        "max_flip_gap": max([f["fp64_gap"] for f in flips], default=0.0),
        "flips": flips[:16],
        # decoder alone, on the reference's own indices (every clip), and end to end on the clips without a flip
        "dwav_max_on_reference_indices": float((wav_refidx_gpu - wav_o).abs().max()),
        "dwav_max_end_to_end": float((wav_gpu[clean] - wav_o[clean]).abs().max()) if clean else None,
        "tie_eps": tie_eps,
    }
    return out


def oracle_clips(name, sd, mk, x, chunk=8, threads=None):
    """The oracle on x `[n,1,T]` (CPU): returns (z, indices, wav, seconds, threads)."""
    import torch
    from oracle import hilcodec_oracle as O
    threads = threads or min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    zs, ids, ws = [], [], []
    with torch.no_grad():
        O.codec_forward(sd, x[:1], mk)                 # warm-up
        t0 = time.perf_counter()
        for i in range(0, x.shape[0], chunk):
            wav_o, _, _, aux = O.codec_forward(sd, x[i:i + chunk], mk)
            zs.append(aux["z"]); ids.append(aux["indices"]); ws.append(wav_o)
        dt = time.perf_counter() - t0
    return torch.cat(zs), torch.cat(ids), torch.cat(ws), dt, threads


def cpu_worker(name: str, clips: int, samples: int, first: int, rendezvous: str, wid: int) -> None:
    """One single-threaded CPU worker of the multi-process baseline (bench.py: cpu_baseline): builds the model's state
    dict, warms up on one clip, waits at a file rendezvous until every worker is ready, runs its `clips` clips one at a
    time through the oracle and prints its own wall-clock interval.  Never touches the GPU."""
    import torch
    from hilcodec_amd import synth
    from oracle import hilcodec_oracle as O
    torch.set_num_threads(1)
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=7)
    x = synth.synth_clips(clips, samples, seed=1234, first=first)
    with torch.no_grad():
        O.codec_forward(sd, x[:1, :, :min(samples, 4800)], mk)           # warm-up
        open(os.path.join(rendezvous, f"ready{wid}"), "w").close()
        go = os.path.join(rendezvous, "go")
        deadline = time.time() + 600
        while not os.path.exists(go):
            if time.time() > deadline:
                raise SystemExit("rendezvous timed out")
            time.sleep(0.01)
        t0 = time.time()
        chk = 0
        for i in range(clips):
            _, _, _, aux = O.codec_forward(sd, x[i:i + 1], mk)
            chk += int(aux["indices"].sum())
        t1 = time.time()
    print(json.dumps({"worker": wid, "t0": t0, "t1": t1, "clips": clips, "index_checksum": chk}), flush=True)


def gpu_forward(model, x_dev):
    import torch
    with torch.no_grad():
        z = model.encoder(x_dev)
        q, _, _, idx = model.quantizer(z, None, return_indices=True)
        wav = model.decoder(q)
    return z, idx, wav


def gpu_decode_indices(model, idx_dev):
    import torch
    from hilcodec_amd import ops
    cb = model.quantizer.spec(idx_dev.device).codebooks
    q = ops.rvq_decode(idx_dev, cb, idx_dev.shape[1], channel_last=False, stage_major=False)
    with torch.no_grad():
        return model.decoder(q)


def run_census(name="hil_speech", n_clips=64, batch=256, seed=1234, weight_seed=7, samples=24000) -> dict:
    import torch
    import hilcodec_amd
    from hilcodec_amd import synth
    dev = torch.device("cuda:0")
    mk = synth.model_kwargs(name)
    sd = synth.synth_state_dict(name, seed=weight_seed)
    nq = mk["vq_kwargs"]["num_quantizers"]
    model = hilcodec_amd.HILCodec(24000, 1, **mk).eval()
    model.load_state_dict(sd, strict=False)
    for l in model.quantizer.layers:
        l.initted = True
    x = synth.synth_clips(batch, samples, seed=seed)
    z, idx, wav = gpu_forward(model, x.to(dev))
    z_o, idx_o, wav_o, secs, threads = oracle_clips(name, sd, mk, x[:n_clips])
    pad = torch.zeros(batch - n_clips, *idx_o.shape[1:], dtype=torch.int64)
    wav_ri = gpu_decode_indices(model, torch.cat([idx_o, pad]).to(dev))[:n_clips]      # full-size decoder launch too
    torch.cuda.synchronize()
    out = compare(sd, nq, z[:n_clips].cpu(), idx[:n_clips].cpu(), wav[:n_clips].cpu(), wav_ri.cpu(), z_o, idx_o, wav_o)
    out.update({"model": name, "gpu_batch": batch, "oracle_seconds": secs, "oracle_threads": threads,
                "library": os.path.basename(hilcodec_amd._lib.LIB_PATH)})
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hil_speech")
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--cpu-worker", nargs=5, default=None, metavar=("CLIPS", "SAMPLES", "FIRST", "RENDEZVOUS", "ID"))
    a = ap.parse_args()
    if a.cpu_worker:
        c, smp, first, rdv, wid = a.cpu_worker
        cpu_worker(a.model, int(c), int(smp), int(first), rdv, int(wid))
        sys.exit(0)
    if a.lib:
        os.environ["HILC_LIB"] = os.path.abspath(a.lib)
    print(json.dumps(run_census(a.model, a.clips, a.batch)), flush=True)
