"""Residual VQ with the reference's API (`models/hilcodec/vector_quantize.py`), eval branch only,
on the gfx950 RVQ kernel (`csrc/rvq.hip`).

Training-side behaviour (EMA cluster statistics, k-means init, dead-code expiry and their
collectives, `vector_quantize.py:32-130,155-172`) is out of scope of the forward hot path: calling
these modules in training mode raises."""
from __future__ import annotations

import typing as tp

import numpy as np
import torch
from torch import Tensor, nn

from ... import engine, fold, ops


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
        self._key = None
        self._spec = None

    def spec(self, dev) -> engine.RvqSpec:
        key = (str(dev),) + tuple((l.embed.data_ptr(), l.embed._version) for l in self.layers)
        if key != self._key:
            cb, cbt, norms = fold.codebook_tables([l.embed for l in self.layers])
            self._spec = engine.RvqSpec(cb.to(dev), cbt.to(dev), norms.to(dev))
            self._key = key
        return self._spec

    def forward(self, x: Tensor, n: tp.Optional[int] = None, return_indices: bool = False):
        if self.training:
            raise NotImplementedError("training-mode RVQ (dropout / EMA updates) is outside the forward hot path")
        for l in self.layers:
            if not l.initted:
                raise RuntimeError("codebook not initialised (kmeans_init=True and no checkpoint loaded); the "
                                   "reference would silently run k-means on this input (vector_quantize.py:139-140)")
        num_replaces = np.zeros(len(self.layers), dtype=np.int64)
        if n is not None and not isinstance(n, int):
            high = n       # one n per clip (mixed-bitrate batch); ops.per_clip_n applies the assert per entry
        elif n is not None:
            assert 1 <= n <= len(self.layers), f"'n' must be in range of 1 <= n <= {len(self.layers)}"
            high = n
        else:
            high = len(self.layers)
        sp = self.spec(x.device)
        idx, q, loss = ops.rvq_encode(x.contiguous().float(), sp.codebooks, sp.codebooks_t, sp.norms, high,
                                      channel_last=self.channel_last, stage_major=False, want_q=True,
                                      want_loss=True)
        if return_indices:
            return q, num_replaces, loss, idx
        return q, num_replaces, loss
