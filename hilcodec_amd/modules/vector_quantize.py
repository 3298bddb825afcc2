"""The older residual VQ named by the north star (`modules/vector_quantize.py` of the reference):
`EuclideanCodebook` / `VectorQuantize` / `ResidualVQ`, eval branch, same kernel.  Codebooks live at
`.layers[i]._codebook.embed`; `forward(x [B,C,T], n=None) -> (quantized, num_replaces, loss)`.
`ShapeGainCodebook` / `ResidualShapeGainVQ` are unused by HILCodec and out of scope."""
from __future__ import annotations

import typing as tp

import numpy as np
import torch
from torch import Tensor, nn

from .. import engine, fold, ops


class EuclideanCodebook(nn.Module):
    """`modules/vector_quantize.py:76-195` (buffers `embed`, `embed_avg`, `cluster_size`, `initted`)."""

    def __init__(self, dim: int, codebook_size: int, kmeans_init: bool = False, kmeans_iters: int = 10,
                 decay: float = 0.8, eps: float = 1e-5, threshold_ema_dead_code: float = 2.0, **_ignored):
        super().__init__()
        self.decay = decay
        embed = (torch.randn if not kmeans_init else torch.zeros)(codebook_size, dim)
        self.codebook_size = codebook_size
        self.kmeans_iters = kmeans_iters
        self.eps = eps
        self.threshold_ema_dead_code = threshold_ema_dead_code
        self.register_buffer("initted", torch.Tensor([not kmeans_init]))
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed", embed)
        self.register_buffer("embed_avg", embed.clone())

    @torch.no_grad()
    def forward(self, x: Tensor) -> tp.Tuple[Tensor, int]:
        if self.training:
            raise NotImplementedError("EMA codebook training is outside the MI355X forward hot path")
        shape = x.shape
        flat = x.reshape(1, -1, shape[-1]).contiguous().float()
        cb, cbt, norms = fold.codebook_tables([self.embed])
        dev = x.device
        _, q, _ = ops.rvq_encode(flat, cb.to(dev), cbt.to(dev), norms.to(dev), 1, channel_last=True,
                                 stage_major=True, want_q=True)
        return q.view(shape), 0


class VectorQuantize(nn.Module):
    """`modules/vector_quantize.py:376-419`: x `[B,C,T]` -> (quantize `[B,C,T]`, num_replace)."""

    def __init__(self, dim: int, codebook_size: int, **kwargs):
        super().__init__()
        self._codebook = EuclideanCodebook(dim=dim, codebook_size=codebook_size, **kwargs)

    @property
    def codebook(self):
        return self._codebook.embed

    def forward(self, x: Tensor) -> tp.Tuple[Tensor, int]:
        q, nr = self._codebook(x.transpose(1, 2).contiguous())
        return q.transpose(1, 2).contiguous(), nr


class ResidualVQ(nn.Module):
    """`modules/vector_quantize.py:471-516`."""

    def __init__(self, *, num_quantizers: int, dropout: bool = False,
                 dropout_index: tp.Optional[tp.List[int]] = None, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantize(**kwargs) for _ in range(num_quantizers)])
        self.dropout = dropout
        self.dropout_index = dropout_index
        self._key = None
        self._spec = None

    def spec(self, dev) -> engine.RvqSpec:
        embeds = [l._codebook.embed for l in self.layers]
        key = (str(dev),) + tuple((e.data_ptr(), e._version) for e in embeds)
        if key != self._key:
            cb, cbt, norms = fold.codebook_tables(embeds)
            self._spec = engine.RvqSpec(cb.to(dev), cbt.to(dev), norms.to(dev))
            self._key = key
        return self._spec

    def forward(self, x: Tensor, n: tp.Optional[int] = None):
        if self.training:
            raise NotImplementedError("training-mode RVQ is outside the forward hot path")
        num_replaces = np.zeros(len(self.layers), dtype=np.int64)
        high = len(self.layers) if n is None else n
        if n is not None:
            assert 1 <= n <= len(self.layers), f"'n' must be in range of 1 <= n <= {len(self.layers)}"
        sp = self.spec(x.device)
        _, q, loss = ops.rvq_encode(x.contiguous().float(), sp.codebooks, sp.codebooks_t, sp.norms, high,
                                    channel_last=False, stage_major=False, want_q=True, want_loss=True)
        return q, num_replaces, loss
