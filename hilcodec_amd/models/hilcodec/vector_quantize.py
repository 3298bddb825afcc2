"""Residual VQ with the reference's API (`models/hilcodec/vector_quantize.py`) on the gfx950 RVQ kernels
(`csrc/rvq.hip`).

Eval branch = the forward hot path.  Training branch of `ResidualVQ.forward`: code search with the
same kernel, EMA cluster statistics + codebook update on the GPU (`hilc_rvq_ema_stats` / `hilc_rvq_ema_update`),
ONE RCCL all-reduce for all stages, dead-code expiry (`vector_quantize.py:101-130,155-172`).  k-means
initialisation (`:32-58,91-99`) is not implemented: un-initialised codebooks raise."""
from __future__ import annotations

import random
import typing as tp

import numpy as np
import torch
from torch import Tensor, nn

from ... import distributed, engine, fold, ops


class EuclideanCodebook(nn.Module):
    """`EuclideanCodebook` (`vector_quantize.py:61-176`): buffers `embed`, `ema_embed`, `ema_num` and
    the `initted` extra state, so reference checkpoints load unchanged."""

    def __init__(self, dim: int, codebook_size: int, kmeans_init: bool = False, kmeans_iters: int = 20,
                 decay: float = 0.8, eps: float = 1e-7, ema_num_threshold: float = 0.0,
                 ema_num_initial: float = 1.0):
        super().__init__()
        self.decay = decay
        embed = (torch.randn if not kmeans_init else torch.zeros)(codebook_size, dim)
        self.codebook_size = codebook_size
        self.kmeans_iters = kmeans_iters
        self.eps = eps
        self.ema_num_threshold = ema_num_threshold
        self.ema_num_initial = ema_num_initial
        self.initted = not kmeans_init
        self.register_buffer("embed", embed)
        self.register_buffer("ema_embed", embed.clone() * ema_num_initial)
        self.register_buffer("ema_num", torch.ones(codebook_size) * ema_num_initial)

    def get_extra_state(self) -> tp.Dict[str, bool]:
        return {"initted": self.initted}

    def set_extra_state(self, state: tp.Dict[str, tp.Any]) -> None:
        self.initted = state["initted"]

    @torch.no_grad()
    def replace_(self, samples: Tensor, mask: Tensor) -> int:
        """`replace` (`vector_quantize.py:101-112`): expired codes <- random vectors of the batch (rank 0's choice is
        broadcast).  Like the reference, `ema_num` keeps its value: there `self.ema_num[idx].fill_(...)` fills a
        temporary produced by advanced indexing, so the buffer is never written."""
        idx = torch.nonzero(mask).squeeze(1)
        num = idx.size(0)
        ns = samples.shape[0]
        if ns >= num:
            pick = torch.randperm(ns, device=samples.device)[:num]
        else:
            pick = torch.randint(0, ns, (num,), device=samples.device)
        new_embed = distributed.broadcast_(samples[pick].float().contiguous(), 0)
        tgt = idx.to(self.embed.device)
        self.embed[tgt, :] = new_embed.to(self.embed.device)
        self.ema_embed[tgt, :] = (new_embed * self.ema_num_initial).to(self.ema_embed.device)
        return num

    @torch.no_grad()
    def forward(self, x: Tensor) -> tp.Tuple[Tensor, int, Tensor]:
        """x `[..., dim]` -> (quantize `[..., dim]`, num_replace 0, embed_ind `[...]`)."""
        if self.training:
            raise NotImplementedError("EMA codebook training is outside the MI355X forward hot path")
        if not self.initted:
            raise RuntimeError("codebook not initialised (the reference would run k-means on the input here, "
                               "vector_quantize.py:139-140); load trained codebooks first")
        shape = x.shape
        flat = x.reshape(1, -1, shape[-1]).contiguous().float()        # [1, N, C] channel-last
        cb, cbt, norms = fold.codebook_tables([self.embed])
        dev = x.device
        idx, q, _ = ops.rvq_encode(flat, cb.to(dev), cbt.to(dev), norms.to(dev), 1, channel_last=True,
                                   stage_major=True, want_q=True)
        return q.view(shape), 0, idx.view(shape[:-1])


class ResidualVQ(nn.Module):
    """`ResidualVQ` (`vector_quantize.py:179-243`).
    forward(x `[B,C,T]`, n=None, return_indices=False) ->
        (quantized `[B,C,T]`, num_replaces np.int64[Nq], mse loss 0-d[, indices `[B,n,T]` int64]).
    Extension: `n` may be a sequence / tensor of B ints (clip b is quantised with its own n_b stages; rows
    >= n_b of `indices` hold -1); each clip's result equals a uniform call with n = n_b."""

    def __init__(self, num_quantizers: int, dropout: bool = False,
                 dropout_index: tp.Optional[tp.List[int]] = None, channel_last: bool = False, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([EuclideanCodebook(**kwargs) for _ in range(num_quantizers)])
        self.dropout = dropout
        if dropout_index is None:
            dropout_index = list(range(1, num_quantizers + 1))
        self.dropout_index = dropout_index
        self.channel_last = channel_last
        # per-quantiser launch option (not part of the reference's API): keep batches of 8 192 frames and more on the VALU form of
        # hilc_rvq_encode (HILC_RVQ_VALU_ONLY) instead of the matrix pipe — same fmaf chains, same indices
        self.rvq_valu_only = False
        self._key = None
        self._spec = None

    def spec(self, dev) -> engine.RvqSpec:
        if torch.compiler.is_compiling():      # the device tables are constants of a compiled graph (see _PlanModule.plan)
            if getattr(self, "_spec", None) is None:
                raise RuntimeError("run the quantizer once eagerly before torch.compile")
            return self._spec
        key = (str(dev),) + tuple((l.embed.data_ptr(), l.embed._version) for l in self.layers)
        if key != self._key:
            cb, cbt, norms = fold.codebook_tables([l.embed for l in self.layers])
            self._spec = engine.RvqSpec(cb.to(dev), cbt.to(dev), norms.to(dev))
            self._key = key
        return self._spec

    @torch.no_grad()
    def _train_update(self, x: Tensor, high: int):
        """Code search + EMA update of the first `high` codebooks (`vector_quantize.py:132-176` training branch).
        Every stage searches with its PRE-update table, exactly as in the reference (stage i's update only touches
        table i, which later stages never read), so search -> statistics -> one all-reduce -> update is equivalent
        to the reference's per-stage interleaving."""
        dev = x.device
        layers = list(self.layers[:high])
        sp = self.spec(dev)
        xin = x.detach().contiguous().float()
        idx, q, _ = ops.rvq_encode(xin, sp.codebooks, sp.codebooks_t, sp.norms, high, channel_last=self.channel_last,
                                   stage_major=False, want_q=True, valu_only=self.rvq_valu_only)
        bucket = ops.rvq_ema_stats(xin, sp.codebooks, idx, high, channel_last=self.channel_last, stage_major=False)
        distributed.all_reduce_sum_(bucket)
        decay = layers[0].decay
        if any(l.decay != decay for l in layers):
            raise NotImplementedError("per-layer EMA decay")
        embed = torch.stack([l.embed.detach().float() for l in layers]).to(dev).contiguous()
        ema_num = torch.stack([l.ema_num.detach().float() for l in layers]).to(dev).contiguous()
        ema_embed = torch.stack([l.ema_embed.detach().float() for l in layers]).to(dev).contiguous()
        ops.rvq_ema_update(embed, ema_num, ema_embed, bucket, decay)
        num_replaces = np.zeros(len(self.layers), dtype=np.int64)
        for i, l in enumerate(layers):
            l.embed.copy_(embed[i])
            l.ema_num.copy_(ema_num[i])
            l.ema_embed.copy_(ema_embed[i])
            # dead-code expiry (`:101-130`): rare, host-side torch ops; samples = this stage's input residual
            if l.ema_num_threshold != 0.0:
                expired = ema_num[i] < l.ema_num_threshold
                if bool(expired.any()):
                    res = xin if self.channel_last else xin.transpose(1, 2)
                    for j in range(i):
                        res = res - torch.nn.functional.embedding(idx[:, j], sp.codebooks[j])
                    num_replaces[i] = l.replace_(res.reshape(-1, res.shape[-1]), expired)
        self._key = None          # the cached device tables are stale now
        return idx, q, num_replaces

    def forward(self, x: Tensor, n: tp.Optional[int] = None, return_indices: bool = False):
        if self.training:
            for l in self.layers:
                if not l.initted:
                    raise RuntimeError("k-means codebook initialisation (vector_quantize.py:91-99) is not implemented: "
                                       "load or set the codebooks first")
            if n is not None:
                assert 1 <= n <= len(self.layers), f"'n' must be in range of 1 <= n <= {len(self.layers)}"
                high = n
            elif self.dropout:
                high = random.sample(self.dropout_index, 1)[0]
            else:
                high = len(self.layers)
            idx, q, num_replaces = self._train_update(x, high)
            # Values of the reference's training outputs (`:233-235`).  This branch UPDATES CODEBOOKS ONLY: x reaches it
            # from raw kernels without an autograd graph (and HILCodec.forward runs under no_grad), so neither the
            # commitment loss nor the straight-through term carries a gradient here — encoder training is out of scope.
            loss = torch.nn.functional.mse_loss(x, q)
            q = q + x - x.detach()
            if return_indices:
                return q, num_replaces, loss, idx
            return q, num_replaces, loss
        for l in self.layers:
            if not l.initted:
                raise RuntimeError("codebook not initialised (kmeans_init=True and no checkpoint loaded); the "
                                   "reference would silently run k-means on this input (vector_quantize.py:139-140)")
        num_replaces = np.zeros(len(self.layers), dtype=np.int64)
        if n is not None and not ops.is_scalar_n(n):
            high = n       # one n per clip (mixed-bitrate batch); ops.per_clip_n applies the assert per entry
        elif n is not None:
            n = int(n)
            assert 1 <= n <= len(self.layers), f"'n' must be in range of 1 <= n <= {len(self.layers)}"
            high = n
        else:
            high = len(self.layers)
        sp = self.spec(x.device)
        idx, q, loss = ops.rvq_encode(x.contiguous().float(), sp.codebooks, sp.codebooks_t, sp.norms, high,
                                      channel_last=self.channel_last, stage_major=False, want_q=True,
                                      want_loss=True, valu_only=self.rvq_valu_only)
        if return_indices:
            return q, num_replaces, loss, idx
        return q, num_replaces, loss
