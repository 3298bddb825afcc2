"""The older residual VQ named by the north star (`modules/vector_quantize.py` of the reference) on the gfx950 RVQ
kernels (`csrc/rvq.hip`): `EuclideanCodebook` / `VectorQuantize` / `ResidualVQ` with the reference's constructor
signatures, buffer names (`initted`, `embed`, `ema_embed`, `ema_num` — a reference `state_dict` loads with
`load_state_dict(strict=True)`) and return contracts:

  EuclideanCodebook.forward(x [..., dim])                          -> (quantize [..., dim], num_replace)
  VectorQuantize.forward(x, calculate_commitment_loss=False)       -> (quantize, num_replace, commit_loss | None)
  ResidualVQ.forward(x, n=None)                                    -> (quantized_out, num_replaces np.int64[Nq], mse loss)

Eval = the forward hot path (one fused launch for all stages in `ResidualVQ`).  The training branch
(`modules/vector_quantize.py:167-195`: EMA statistics, Laplace smoothing, dead-code replacement) runs the code search
and the cluster statistics on the kernels and the (tiny, once-per-step) table update as torch ops on the same device;
it updates codebooks only — nothing here builds an autograd graph.  Options this path does not implement raise:
`use_shape_gain` (`ShapeGainCodebook` / `ResidualShapeGainVQ` are unused by HILCodec), k-means initialisation
(an un-initialised codebook raises instead of silently quantising against zeros)."""
from __future__ import annotations

import random
import typing as tp

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import distributed, engine, fold, ops


class EuclideanCodebook(nn.Module):
    """`modules/vector_quantize.py:76-195`."""

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
        self.register_buffer("initted", Tensor([not kmeans_init]))
        self.register_buffer("embed", embed)
        self.register_buffer("ema_embed", embed.clone() * ema_num_initial)
        self.register_buffer("ema_num", torch.ones(codebook_size) * ema_num_initial)

    def _require_initted(self) -> None:
        if not bool(self.initted):
            raise RuntimeError("codebook not initialised (kmeans_init=True and no checkpoint loaded): the reference "
                               "would run k-means on this input (modules/vector_quantize.py:149-150), which this "
                               "path does not implement; load or set the codebook first")

    @torch.no_grad()
    def replace(self, samples: Tensor, mask: Tensor) -> int:
        """`:119-128`: expired codes <- random vectors of the batch; rank 0's choice is broadcast."""
        idx = torch.nonzero(mask).squeeze(1)
        num, ns = idx.size(0), samples.shape[0]
        if ns >= num:
            pick = torch.randperm(ns, device=samples.device)[:num]
        else:
            pick = torch.randint(0, ns, (num,), device=samples.device)
        new_embed = distributed.broadcast_(samples[pick].detach().float().contiguous(), 0).to(self.embed.device)
        tgt = idx.to(self.embed.device)
        self.embed[tgt, :] = new_embed
        self.ema_embed[tgt, :] = new_embed * self.ema_num_initial
        self.ema_num[tgt] = self.ema_num_initial
        return num

    @torch.no_grad()
    def expire_codes_(self, batch_samples: Tensor) -> int:
        """`:130-138`."""
        if self.ema_num_threshold == 0.0:
            return 0
        expired = self.ema_num < self.ema_num_threshold
        if not torch.any(expired):
            return 0
        return self.replace(batch_samples.reshape(-1, batch_samples.shape[-1]), expired)

    @torch.no_grad()
    def forward(self, x: Tensor) -> tp.Tuple[Tensor, int]:
        """x `[..., dim]` (channel-last, as `VectorQuantize` hands it over)."""
        self._require_initted()
        shape = x.shape
        dev = x.device
        flat = x.reshape(1, -1, shape[-1]).contiguous().float()        # [1, N, dim]
        cb, cbt, norms = (t.to(dev) for t in fold.codebook_tables([self.embed]))
        idx, q, _ = ops.rvq_encode(flat, cb, cbt, norms, 1, channel_last=True, stage_major=True, want_q=True)
        quantize = q.view(shape)
        num_replace = 0
        if self.training:
            # `:167-193`: counts and per-code sums of this batch (deterministic kernel, fixed frame order), summed
            # over ranks in one bucket, EMA, (Laplace-smoothed) normalisation, dead-code replacement
            K, C = self.embed.shape
            bucket = ops.rvq_ema_stats(flat, cb, idx, 1, channel_last=True, stage_major=True)
            distributed.all_reduce_sum_(bucket)
            ema_num_new = bucket[0, :K].to(self.ema_num.device)
            ema_embed_new = bucket[0, K:].view(K, C).to(self.ema_embed.device)
            self.ema_num.mul_(self.decay).add_(ema_num_new, alpha=1 - self.decay)
            self.ema_embed.mul_(self.decay).add_(ema_embed_new, alpha=1 - self.decay)
            if self.ema_num_threshold <= 0.0:
                ema_num = (self.ema_num + self.eps) / (self.ema_num.sum() + self.codebook_size * self.eps) \
                    * self.ema_num.sum()
            else:
                ema_num = self.ema_num
            self.embed.copy_(self.ema_embed / ema_num.unsqueeze(1))
            num_replace = self.expire_codes_(x)
        return quantize, num_replace


class VectorQuantize(nn.Module):
    """`modules/vector_quantize.py:376-419`."""

    def __init__(self, commitment: float = 1., use_shape_gain: bool = False, channel_last: bool = False,
                 gradient_flow: bool = True, **kwargs):
        super().__init__()
        if use_shape_gain:
            raise NotImplementedError("ShapeGainCodebook (modules/vector_quantize.py:198-372) is not part of the "
                                      "HILCodec path")
        self.commitment = commitment
        self.use_shape_gain = use_shape_gain
        self.channel_last = channel_last
        self._codebook = EuclideanCodebook(**kwargs)      # unknown kwargs raise TypeError, as in the reference
        self.gradient_flow = gradient_flow

    def forward(self, x: Tensor, calculate_commitment_loss: bool = False):
        if not self.channel_last:
            x = x.transpose(1, 2)                          # [B,C,T] -> [B,T,C]
        quantize, num_replace = self._codebook(x)
        commit_loss = F.mse_loss(quantize.detach(), x) * self.commitment if calculate_commitment_loss else None
        if self.gradient_flow and self.training:
            quantize = x + quantize - x.detach()
        if not self.channel_last:
            quantize = quantize.transpose(1, 2)
        return quantize, num_replace, commit_loss


class ResidualVQ(nn.Module):
    """`modules/vector_quantize.py:471-516`."""

    def __init__(self, num_quantizers: int, dropout: bool = False,
                 dropout_index: tp.Optional[tp.List[int]] = None, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantize(gradient_flow=False, **kwargs) for _ in range(num_quantizers)])
        self.dropout = dropout
        if dropout_index is None:
            dropout_index = list(range(1, num_quantizers + 1))
        self.dropout_index = dropout_index
        self.use_shape_gain = self.layers[0].use_shape_gain
        self.rvq_valu_only = False      # launch option, see models/hilcodec/vector_quantize.py
        self._key = None
        self._spec = None

    def spec(self, dev) -> engine.RvqSpec:
        if torch.compiler.is_compiling():      # the device tables are constants of a compiled graph (see _PlanModule.plan)
            if getattr(self, "_spec", None) is None:
                raise RuntimeError("run the quantizer once eagerly before torch.compile")
            return self._spec
        embeds = [l._codebook.embed for l in self.layers]
        key = (str(dev),) + tuple((e.data_ptr(), e._version) for e in embeds)
        if key != self._key:
            cb, cbt, norms = fold.codebook_tables(embeds)
            self._spec = engine.RvqSpec(cb.to(dev), cbt.to(dev), norms.to(dev))
            self._key = key
        return self._spec

    def forward(self, x: Tensor, n: tp.Optional[int] = None):
        num_replaces = np.zeros(len(self.layers), dtype=np.int64)
        if n is not None:
            assert 1 <= n <= len(self.layers), f"'n' must be in range of 1 <= n <= {len(self.layers)}"
            high = int(n)
        elif self.training and self.dropout:
            high = random.sample(self.dropout_index, 1)[0]
        else:
            high = len(self.layers)
        if self.training:
            # the reference's stage loop: every stage updates its own table after its search (`:501-505`)
            quantized_out = 0.
            residual = x.detach()
            for i, layer in enumerate(self.layers[:high]):
                quantized, num_replace, _ = layer(residual, calculate_commitment_loss=False)
                num_replaces[i] = num_replace
                residual = residual - quantized
                quantized_out = quantized_out + quantized
            loss = F.mse_loss(x, quantized_out)
            return quantized_out + x - x.detach(), num_replaces, loss
        for l in self.layers[:high]:
            l._codebook._require_initted()
        sp = self.spec(x.device)
        _, q, loss = ops.rvq_encode(x.contiguous().float(), sp.codebooks, sp.codebooks_t, sp.norms, high,
                                    channel_last=self.layers[0].channel_last, stage_major=False, want_q=True,
                                    want_loss=True, valu_only=self.rvq_valu_only)
        return q, num_replaces, loss
