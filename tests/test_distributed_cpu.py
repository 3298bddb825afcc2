"""CPU, world_size 2 over gloo: the multi-GPU layout of the path — contiguous clip shards, replicated
weights, no data-path collective, one gather of counters."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_everything():
    from hilcodec_amd.distributed import shard_range
    for total in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(2048, 3, 8) == (768, 1024)        # BASELINE config 5: 256 clips per GPU


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from hilcodec_amd import distributed as D, synth
    from oracle import hilcodec_oracle as O
    torch.set_num_threads(2)
    r, w = D.init("gloo")
    assert (r, w) == (rank, world)
    total = 5                                   # uneven on purpose: shards of 3 and 2 clips
    lo, hi = D.shard_range(total, rank, world)
    mk = synth.model_kwargs("hil_speech")
    sd = synth.synth_state_dict("hil_speech", seed=7)       # replicated weights: every rank builds the same
    x = synth.synth_clips(hi - lo, 1600, seed=1234, first=lo)
    D.barrier()
    z = O.encoder_forward(sd, x, mk)            # CPU stand-in for the per-rank work; the GPU path is tested with -m gpu
    _, _, _, idx = O.rvq_forward(sd, z, None, 8)
    D.barrier()
    per_rank = D.gather_counters({"clips": hi - lo, "audio_s": (hi - lo) * 1600 / 24000.0, "wall_s": 1.0 + rank,
                                  "index_checksum": float(idx.sum())}, torch.device("cpu"))
    agg = D.aggregate(per_rank)
    q.put((rank, lo, hi, per_rank, agg, float(idx.sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, pr0, agg0, cs0), (r1, lo1, hi1, pr1, agg1, cs1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)
    assert pr0 == pr1 and agg0 == agg1                       # every rank sees the same gathered table
    assert agg0["clips"] == 5 and agg0["wall_s"] == 2.0      # max over ranks
    assert abs(agg0["audio_s"] - 5 * 1600 / 24000.0) < 1e-12 and abs(agg0["xrt"] - agg0["audio_s"] / 2.0) < 1e-12
    assert agg0["index_checksum"] == cs0 + cs1
    # sharding does not change results: the union of the shards equals the un-sharded batch
    sys.path.insert(0, ROOT)
    from hilcodec_amd import synth
    from oracle import hilcodec_oracle as O
    mk = synth.model_kwargs("hil_speech")
    sd = synth.synth_state_dict("hil_speech", seed=7)
    x = synth.synth_clips(5, 1600, seed=1234)
    _, _, _, idx = O.rvq_forward(sd, O.encoder_forward(sd, x, mk), None, 8)
    assert float(idx.sum()) == cs0 + cs1


def _ema_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    import torch.distributed as dist
    from hilcodec_amd import distributed as D, synth
    from oracle import hilcodec_oracle as O
    torch.set_num_threads(2)
    D.init("gloo")
    nq, K, Dm, total, Tn = 2, 1024, 128, 6, 30
    st = {}
    for i in range(nq):
        e = torch.from_numpy(synth.normalish(300 + i, K * Dm) * np.float32(0.3)).view(K, Dm)
        st[f"layers.{i}.embed"], st[f"layers.{i}.ema_embed"], st[f"layers.{i}.ema_num"] = e.clone(), e * 0.5, torch.ones(K) * 0.5
    z = torch.from_numpy(synth.normalish(55, total * Dm * Tn)).view(total, Dm, Tn)
    lo, hi = D.shard_range(total, rank, world)
    calls = []

    def hook(bucket):                      # the reference's dist.all_reduce(bucket) (vector_quantize.py:158-162)
        calls.append(bucket.numel())
        D.all_reduce_sum_(bucket)

    _, _, idx, _ = O.rvq_train_step(st, z[lo:hi], None, nq, 0.9, bucket_hook=hook)
    q.put((rank, calls, {k: v.numpy() for k, v in st.items()}, idx.numpy()))
    dist.destroy_process_group()


def test_two_rank_gloo_rvq_ema_all_reduce():
    """The one real exchange of the codec: per-rank cluster statistics summed over ranks.  Both
    ranks must end with the SAME codebooks, equal to a single process that saw the whole batch."""
    import numpy as np
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ema_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, calls0, st0, idx0), (_, calls1, st1, idx1) = res
    assert calls0 == calls1 == [1024 + 1024 * 128] * 2
    for k in st0:
        assert np.array_equal(st0[k], st1[k]), k
    sys.path.insert(0, ROOT)
    from hilcodec_amd import synth
    from oracle import hilcodec_oracle as O
    st = {}
    for i in range(2):
        e = torch.from_numpy(synth.normalish(300 + i, 1024 * 128) * np.float32(0.3)).view(1024, 128)
        st[f"layers.{i}.embed"], st[f"layers.{i}.ema_embed"], st[f"layers.{i}.ema_num"] = e.clone(), e * 0.5, torch.ones(1024) * 0.5
    z = torch.from_numpy(synth.normalish(55, 6 * 128 * 30)).view(6, 128, 30)
    _, _, idx, _ = O.rvq_train_step(st, z, None, 2, 0.9)
    assert np.array_equal(np.concatenate([idx0, idx1]), idx.numpy())
    for k in st:
        assert np.allclose(st0[k], st[k].numpy(), rtol=1e-6, atol=1e-7), k


def _cpu_rvq_kernels():
    """CPU stand-ins with the kernels' contracts (hilc_rvq_encode / _ema_stats / _ema_update), built on the oracle:
    they let the PRODUCT's multi-rank code (`ResidualVQ._train_update` -> `distributed.all_reduce_sum_` ->
    `replace_` -> `broadcast_`) run under gloo without a GPU; the kernels themselves are checked by -m gpu tests."""
    import torch.nn.functional as F
    from oracle import hilcodec_oracle as O

    def rvq_encode(z, cb, cbt, norms, n, channel_last=False, stage_major=False, want_q=True, want_loss=False, valu_only=False):
        sd = {f"quantizer.layers.{i}.embed": cb[i] for i in range(cb.shape[0])}
        zz = z.transpose(1, 2) if channel_last else z
        q, _, loss, idx = O.rvq_forward(sd, zz, int(n), cb.shape[0])
        if stage_major:
            idx = idx.transpose(0, 1).contiguous()
        return idx, (q.transpose(1, 2) if channel_last else q), (loss if want_loss else None)

    def rvq_ema_stats(z, cb, idx, n, channel_last=False, stage_major=False):
        res = (z if channel_last else z.transpose(1, 2)).reshape(-1, z.shape[-1 if channel_last else 1]).clone()
        K = cb.shape[1]
        rows = []
        for s in range(n):
            ind = (idx[s] if stage_major else idx[:, s]).reshape(-1)
            onehot = F.one_hot(ind, K).float()
            rows.append(torch.cat([onehot.sum(0), (onehot.t() @ res).reshape(-1)]))
            res = res - F.embedding(ind, cb[s])
        return torch.stack(rows)

    def rvq_ema_update(embed, ema_num, ema_embed, bucket, decay):
        n, K, C = embed.shape
        ema_num.mul_(decay).add_(bucket[:, :K], alpha=1 - decay)
        ema_embed.mul_(decay).add_(bucket[:, K:].view(n, K, C), alpha=1 - decay)
        embed.copy_(ema_embed / ema_num.unsqueeze(2))

    return rvq_encode, rvq_ema_stats, rvq_ema_update


def _make_product_rvq(threshold):
    import numpy as np
    from hilcodec_amd import synth
    from hilcodec_amd.models.hilcodec.vector_quantize import ResidualVQ
    nq, K, Dm = 2, 1024, 128
    rvq = ResidualVQ(num_quantizers=nq, dropout=False, channel_last=False, dim=Dm, codebook_size=K, kmeans_init=False,
                     decay=0.9, ema_num_threshold=threshold, ema_num_initial=0.5).train()
    for i, l in enumerate(rvq.layers):
        e = torch.from_numpy(synth.normalish(300 + i, K * Dm) * np.float32(0.3)).view(K, Dm)
        l.embed.copy_(e)
        l.ema_embed.copy_(e * 0.5)
    return rvq


def _product_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    import torch.distributed as dist
    from hilcodec_amd import distributed as D, ops, synth
    torch.set_num_threads(2)
    D.init("gloo")
    ops.rvq_encode, ops.rvq_ema_stats, ops.rvq_ema_update = _cpu_rvq_kernels()
    total, Tn = 6, 30
    z = torch.from_numpy(synth.normalish(55, total * 128 * Tn)).view(total, 128, Tn)
    lo, hi = D.shard_range(total, rank, world)
    out = {}
    for name, thr in (("plain", 0.0), ("expiry", 0.46)):
        rvq = _make_product_rvq(thr)
        torch.manual_seed(100 + rank)                  # ranks draw DIFFERENT replacement candidates: rank 0's must win
        qz, num_replaces, loss = rvq(z[lo:hi])
        out[name] = ({k: v.numpy().copy() for k, v in rvq.state_dict().items() if not k.endswith("_extra_state")},
                     num_replaces.copy(), float(loss))
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_rank_gloo_product_rvq_training_path():
    """The PRODUCT's data-parallel RVQ training step (not the oracle's): one bucketed all-reduce for all stages, then
    EMA update; with an expiry threshold, dead codes are replaced by rank 0's broadcast choice.  Both ranks must end
    with identical codebooks; without expiry they must equal a single process that saw the whole batch."""
    import numpy as np
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_product_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name in ("plain", "expiry"):
        st0, nr0, _ = res[0][name]
        st1, nr1, _ = res[1][name]
        assert (nr0 == nr1).all()
        for k in st0:
            assert np.array_equal(st0[k], st1[k]), (name, k)
    assert res[0]["expiry"][1].sum() > 0 and not res[0]["plain"][1].any()
    # single process, whole batch, same stand-in kernels
    sys.path.insert(0, ROOT)
    from hilcodec_amd import ops, synth
    saved = (ops.rvq_encode, ops.rvq_ema_stats, ops.rvq_ema_update)
    try:
        ops.rvq_encode, ops.rvq_ema_stats, ops.rvq_ema_update = _cpu_rvq_kernels()
        rvq = _make_product_rvq(0.0)
        z = torch.from_numpy(synth.normalish(55, 6 * 128 * 30)).view(6, 128, 30)
        rvq(z)
        single = {k: v.numpy() for k, v in rvq.state_dict().items() if not k.endswith("_extra_state")}
    finally:
        ops.rvq_encode, ops.rvq_ema_stats, ops.rvq_ema_update = saved
    for k, v in single.items():
        assert np.allclose(res[0]["plain"][0][k], v, rtol=1e-6, atol=1e-7), k
