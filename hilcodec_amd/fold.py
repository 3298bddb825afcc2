"""Host-side, one-off weight preparation ("fold") for the gfx950 kernels.

The reference recomputes every weight re-parameterisation on every forward (107 hooks per call,
SURVEY.md §3.1).  Here each is folded once on the host with the *same torch function the
reference's hook calls*, so the folded fp32 weights are bit-identical to what the reference
convolves with; the kernels then only see plain weights in their preferred layouts.

  weight_norm            torch._weight_norm(v, g, 0)   (torch.nn.utils.weight_norm, conv.py:26-41)
  weight_standardization g*scale*(v-mean)*rsqrt(max(var*fan_in, eps))   (modules/weight_standardization.py:30-41)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor


def weight_norm_fold(v: Tensor, g: Tensor) -> Tensor:
    return torch._weight_norm(v.detach().float().cpu(), g.detach().float().cpu(), 0)


def weight_standardization_fold(v: Tensor, g: Optional[Tensor], scale: Optional[Tensor] = None,
                                eps: float = 1e-7) -> Tensor:
    v = v.detach().float().cpu()
    axes = list(range(1, v.dim()))
    fan_in = 1.0
    for a in axes:
        fan_in *= v.size(a)
    var, mean = torch.var_mean(v, dim=axes, unbiased=False, keepdim=True)
    w = (v - mean) * torch.rsqrt(torch.clamp(var * fan_in, min=eps))
    if g is not None:
        g = g.detach().float().cpu()
        if scale is not None:
            g = g * scale.detach().float().cpu()
        w = g * w
    return w


def pointwise_layout(w: Tensor) -> Tensor:
    """`[M, K, 1]` conv weight -> `[K, M]` (k-major: the GEMM's A slices are contiguous rows)."""
    assert w.dim() == 3 and w.shape[2] == 1
    return w[:, :, 0].t().contiguous()


def depthwise_layout(w: Tensor) -> Tensor:
    """`[C, 1, k]` (conv or conv-transpose, groups=C) -> `[C, k]`."""
    assert w.dim() == 3 and w.shape[1] == 1
    return w[:, 0, :].contiguous()


def stft_basis_layout(basis: Tensor) -> Tensor:
    """Reference basis `[n_fft+2, 1, n_fft]` = [cos_0..cos_{N/2}; sin_0..sin_{N/2}] * hann
    (conv.py:329-345) -> `[n_fft][m_pad]` with (cos_k, sin_k) interleaved along the GEMM's M axis
    (so one MFMA lane holds both halves of a bin) and zero-padded to a multiple of 32 rows."""
    m, one, n_fft = basis.shape
    assert one == 1 and m == n_fft + 2
    nb = n_fft // 2 + 1
    b = basis[:, 0, :].detach().float().cpu()
    inter = torch.stack([b[:nb], b[nb:]], dim=1).reshape(2 * nb, n_fft)      # row 2k = cos_k, 2k+1 = sin_k
    m_pad = (m + 31) // 32 * 32
    out = torch.zeros(m_pad, n_fft, dtype=torch.float32)
    out[:m] = inter
    return out.t().contiguous()


def codebook_tables(embeds) -> Tuple[Tensor, Tensor, Tensor]:
    """list of `[K, C]` codebooks -> (`[Nq,K,C]`, `[Nq,C,K]`, norms `[Nq,K]`).
    norms use the reference's own expression `embed.t().pow(2).sum(0)` (vector_quantize.py:150)."""
    cb = torch.stack([e.detach().float().cpu() for e in embeds], dim=0).contiguous()
    cbt = cb.transpose(1, 2).contiguous()
    norms = torch.stack([e.detach().float().cpu().t().pow(2).sum(0) for e in embeds], dim=0).contiguous()
    return cb, cbt, norms
