"""Cache-carrying causal layers with the reference's names and call protocol
(`models/hilcodec/causal_layers.py`): `forward(x, cache) -> (y, new_cache)` where the cache is the
layer-input history that the offline model replaces by zero padding.  Executed by the gfx950
kernels through their `hist` / `hist_out` arguments (`include/hilcodec_amd.h`)."""
from __future__ import annotations

import typing as tp

import torch
from torch import Tensor, nn

from ... import fold, ops
from .modules.conv import ConvParams, _FoldCache
from ...synth import stft_basis


class CausalSTFT(nn.Module):
    """`causal_layers.py:72-144`: conv-as-DFT magnitude WITHOUT padding — the caller prepends the
    `n_fft-1` history samples (`cache_len`)."""

    def __init__(self, n_fft: int, hop_size: int, win_size: tp.Optional[int] = None,
                 win_type: tp.Optional[str] = "hann", window: tp.Optional[Tensor] = None,
                 norm: tp.Optional[str] = "backward", magnitude: bool = True, device=None, dtype=None):
        assert magnitude, "STFTOnnx only supports magnitude=True"
        super().__init__()
        if win_size not in (None, n_fft) or win_type != "hann" or window is not None or norm != "backward":
            raise NotImplementedError("only the configuration HILCodec uses (hann, n_fft window, backward norm)")
        self.n_fft, self.hop_size, self.cache_len, self.norm, self.magnitude = n_fft, hop_size, n_fft - 1, norm, True
        self.register_buffer("weight", stft_basis(n_fft))
        self._cache = _FoldCache()

    def initialize_cache(self, x: Tensor) -> Tensor:
        return torch.zeros(x.size(0), self.cache_len, dtype=x.dtype, device=x.device)

    def basis_t(self, device) -> Tensor:
        return self._cache.get((str(device), self.weight.data_ptr(), self.weight._version),
                               lambda: fold.stft_basis_layout(self.weight).to(device))

    def forward(self, x: Tensor) -> Tensor:
        """x `[B,1,H*L + n_fft - H]` (history already prepended) -> `[B, n_fft/2+1, L]`."""
        if x.dim() == 2:
            x = x.unsqueeze(1)
        x = x.contiguous().float()
        hist, cur = x[:, :, : self.cache_len].contiguous(), x[:, :, self.cache_len:].contiguous()
        return ops.stft_logmag(cur, self.basis_t(x.device), self.n_fft, self.hop_size, normalize=2, hist=hist)


class CausalConv1d(ConvParams):
    """`causal_layers.py:147-165` (depthwise or pointwise-free k>1 conv with an explicit cache)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1,
                 groups: int = 1, bias: bool = True, norm: str = "none"):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups, bias=bias,
                         norm=norm)
        self.causal_padding = dilation * (kernel_size - 1) - (stride - 1)
        self._cache = _FoldCache()

    def initialize_cache(self, x: Tensor) -> Tensor:
        return torch.zeros(x.size(0), self.in_channels, self.causal_padding, device=x.device)

    def folded(self, device):
        def build():
            w, b = self.effective_weight(), self.effective_bias()
            if self.groups == 1 and self.in_channels == 1:
                w = w[:, 0, :].contiguous()
            elif self.groups == 1 and self.out_channels == 1:
                w = w[0].contiguous()
            else:
                w = fold.depthwise_layout(w)
            return w.to(device), None if b is None else b.to(device)
        return self._cache.get((str(device), self.version_key()), build)

    def forward(self, x: Tensor, cache: Tensor) -> tp.Tuple[Tensor, Tensor]:
        w, b = self.folded(x.device)
        x = x.contiguous().float()
        cache = cache.contiguous().float()
        if self.groups == 1 and self.out_channels == 1:
            return ops.conv_post(x, w, b, in_elu=False, do_tanh=False, hist=cache, want_hist=True)
        if self.groups != self.in_channels:
            raise NotImplementedError("only depthwise / Cout=1 cache-carrying convs are part of HILCodec")
        return ops.dw_conv(x, w, b, stride=self.stride, hist=cache, want_hist=True)


class CausalConvTranspose1d(ConvParams):
    """`causal_layers.py:168-188` (depthwise, k = 2*stride: one frame of cache)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1,
                 groups: int = 1, bias: bool = True, norm: str = "none"):
        if kernel_size != 2 * stride or groups != in_channels or in_channels != out_channels or bias:
            raise NotImplementedError("only bias-free depthwise transposed convs with kernel = 2*stride are on the hot path")
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups, bias=bias,
                         norm=norm, transposed=True)
        self.causal_padding = (dilation * (kernel_size - 1)) // stride
        self._cache = _FoldCache()

    def initialize_cache(self, x: Tensor) -> Tensor:
        return torch.zeros(x.size(0), self.in_channels, self.causal_padding, device=x.device)

    def folded(self, device):
        return self._cache.get((str(device), self.version_key()),
                               lambda: fold.depthwise_layout(self.effective_weight()).to(device))

    def forward(self, x: Tensor, cache: Tensor) -> tp.Tuple[Tensor, Tensor]:
        return ops.dw_convtr(x.contiguous().float(), self.folded(x.device), self.stride,
                             hist=cache.contiguous().float(), want_hist=True)


class PointwiseConv1d(ConvParams):
    """What the reference's `SConv1d` factory returns for kernel_size == 1: a plain (weight-normed) Conv1d."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, norm: str = "none"):
        super().__init__(in_channels, out_channels, 1, bias=bias, norm=norm)
        self._cache = _FoldCache()

    def folded(self, device):
        def build():
            b = self.effective_bias()
            return fold.pointwise_layout(self.effective_weight()).to(device), None if b is None else b.to(device)
        return self._cache.get((str(device), self.version_key()), build)

    def forward(self, x: Tensor) -> Tensor:
        w, b = self.folded(x.device)
        return ops.pw_conv(x.contiguous().float(), w, b)


def SConv1d(in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1,
            groups: int = 1, bias: bool = True, norm: str = "weight_norm") -> nn.Module:
    """`causal_layers.py:191-204`."""
    if norm != "weight_norm":
        raise ValueError(f"Unknown norm: {norm}")
    if kernel_size == 1:
        return PointwiseConv1d(in_channels, out_channels, bias=bias, norm=norm)
    return CausalConv1d(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups, bias=bias,
                        norm=norm)


def SConvTranspose1d(in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, dilation: int = 1,
                     groups: int = 1, bias: bool = True, norm: str = "weight_norm") -> nn.Module:
    """`causal_layers.py:207-220`."""
    if norm != "weight_norm":
        raise ValueError(f"Unknown norm: {norm}")
    return CausalConvTranspose1d(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups,
                                 bias=bias, norm=norm)
