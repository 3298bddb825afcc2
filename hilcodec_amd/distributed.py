"""Multi-GPU layout of the path: one process per GPU, clips (or streams) sharded contiguously, weights
replicated, NO data-path collective — the only exchange is a gather of per-rank counters after the
timed region.  `torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by
the CPU tests."""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, device: torch.device = None, force: bool = False) -> Tuple[int, int]:
    """One process per GPU (`/root/reference/train.py:51-61` does the same with NCCL).  A single process needs no process
    group; `force=True` (or HILC_FORCE_DIST=1) creates it anyway, so that the RCCL initialisation, the barriers and the
    counters all_gather below execute for real at world size 1 (the only way to exercise them on a 1-GPU box)."""
    rank, world, _ = env_rank_world()
    force = force or os.environ.get("HILC_FORCE_DIST", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def is_initialized() -> bool:
    return dist.is_initialized()


def backend_name() -> str:
    """what the counters travel over: "rccl (torch.distributed nccl)" on GPUs, "gloo" in the CPU tests, or no group at all"""
    if not dist.is_initialized():
        return "none (single process, no process group)"
    b = dist.get_backend()
    return "rccl (torch.distributed nccl)" if b == "nccl" else str(b)


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of `total` independent clips/streams for `rank`
    (the first `total % world` ranks get one extra)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (RCCL on GPU tensors, gloo on CPU tensors); identity for a single process.  The one
    real exchange of the codec: the RVQ training statistics, all stages in ONE bucket (Nq * 516 KiB) instead of
    the reference's one all-reduce per stage (`models/hilcodec/vector_quantize.py:158-165`) — over xGMI a ring
    all-reduce is latency-bound at this size, so fewer, larger calls."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src)
    return t


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def gather_counters(counters: Dict[str, float], device: torch.device) -> List[Dict[str, float]]:
    """all_gather of a small dict of scalars (<= 64 B per rank); returns one dict per rank, on every rank."""
    keys = sorted(counters)
    t = torch.tensor([float(counters[k]) for k in keys], dtype=torch.float64, device=device)
    if dist.is_initialized():          # also at world size 1 (forced group): the collective really runs
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
    else:
        out = [t]
    return [dict(zip(keys, o.cpu().tolist())) for o in out]


def aggregate(per_rank: List[Dict[str, float]]) -> Dict[str, float]:
    """Whole-job figures: audio seconds add up, wall time is the slowest rank's."""
    wall = max(r["wall_s"] for r in per_rank)
    audio = sum(r["audio_s"] for r in per_rank)
    return {"wall_s": wall, "audio_s": audio, "clips": sum(r["clips"] for r in per_rank), "xrt": audio / wall,
            "index_checksum": sum(r.get("index_checksum", 0.0) for r in per_rank)}
