"""Convolution wrappers with the reference's names, arguments and state-dict keys
(`models/hilcodec/modules/conv.py`), executing on the hand-written gfx950 kernels.

Only what the HILCodec hot path uses is supported — causal, zero ('constant') padding, dilation 1,
groups in {1, C}, norms in {'weight_norm', 'weight_standardization', 'none'}; anything else raises
`NotImplementedError` instead of silently computing something different.
"""
from __future__ import annotations

import math
import typing as tp

import torch
from torch import Tensor, nn

from .... import fold, ops
from ....synth import stft_basis

CONV_NORMALIZATIONS = frozenset(["none", "weight_norm", "weight_standardization"])


class ConvParams(nn.Module):
    """Stands where the reference has a (re-parameterised) `nn.Conv1d` / `nn.ConvTranspose1d`:
    owns `weight_g`/`weight_v` (+`weight_scale`) or `weight`, and `bias`, under the same names."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1,
                 dilation: int = 1, groups: int = 1, bias: bool = True, norm: str = "weight_norm",
                 transposed: bool = False, nonlinearity: str = "linear",
                 norm_kwargs: tp.Optional[dict] = None):
        super().__init__()
        if norm not in CONV_NORMALIZATIONS:
            raise NotImplementedError(f"norm '{norm}' is not on the MI355X hot path (supported: {sorted(CONV_NORMALIZATIONS)})")
        if dilation != 1:
            raise NotImplementedError("dilation != 1 (both shipped configs use dilation_base 1)")
        if groups not in (1, in_channels) or (groups != 1 and in_channels != out_channels):
            raise NotImplementedError("groups must be 1 or in_channels == out_channels (depthwise)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.groups = kernel_size, stride, groups
        self.transposed, self.norm_type = transposed, norm
        norm_kwargs = dict(norm_kwargs or {})
        shape = ((in_channels, out_channels // groups, kernel_size) if transposed
                 else (out_channels, in_channels // groups, kernel_size))
        w = torch.empty(shape)
        nn.init.kaiming_normal_(w, nonlinearity=nonlinearity)          # conv.py:124,168
        if norm == "weight_norm":
            self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(-1, 1, 1))
            self.weight_v = nn.Parameter(w)
        elif norm == "weight_standardization":
            self.ws_eps = float(norm_kwargs.get("eps", 1e-7))
            zero_init = bool(norm_kwargs.get("zero_init", False))
            self.weight_v = nn.Parameter(w)
            if norm_kwargs.get("learnable_gain", True):
                g = torch.zeros(shape[0], 1, 1) if zero_init else torch.ones(shape[0], 1, 1)
                self.weight_g = nn.Parameter(g)
            else:
                self.register_buffer("weight_g", None)
            scale = norm_kwargs.get("scale", None)
            self.register_buffer("weight_scale", None if scale is None else torch.ones(1) * scale)
        else:
            self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))       # conv.py:125-126
        else:
            self.register_parameter("bias", None)

    # -- folding -------------------------------------------------------------------------
    def effective_weight(self) -> Tensor:
        """fp32 CPU weight exactly as the reference's forward-pre-hook would produce it."""
        if hasattr(self, "weight") and isinstance(getattr(self, "weight", None), nn.Parameter):
            return self.weight.detach().float().cpu()
        if self.norm_type == "weight_norm":
            return fold.weight_norm_fold(self.weight_v, self.weight_g)
        return fold.weight_standardization_fold(self.weight_v, self.weight_g, self.weight_scale, self.ws_eps)

    def effective_bias(self) -> tp.Optional[Tensor]:
        return None if self.bias is None else self.bias.detach().float().cpu()

    def remove_reparameterization(self) -> None:
        """`torch.nn.utils.remove_weight_norm` equivalent: keep a plain `weight`."""
        if isinstance(getattr(self, "weight", None), nn.Parameter):
            return
        w = self.effective_weight().to(self.weight_v.device)
        for name in ("weight_g", "weight_v", "weight_scale"):
            if name in self._parameters:
                del self._parameters[name]
            elif name in self._buffers:
                del self._buffers[name]
        self.weight = nn.Parameter(w)
        self.norm_type = "none"

    def version_key(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + [b for b in self.buffers() if b is not None])


class _FoldCache:
    """device copy of the folded weights, rebuilt when a parameter changes."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, key, build):
        if key != self.key:
            self.val = build()
            self.key = key
        return self.val


class NormConv1d(nn.Module):
    """`conv.py:115-134` — holds `.conv`; normalisation layers after the conv are not on the path."""

    def __init__(self, *args, causal: bool = False, norm: str = "none", nonlinearity: str = "linear",
                 norm_kwargs: tp.Dict[str, tp.Any] = {}, **kwargs):
        super().__init__()
        self.conv = ConvParams(*args, norm=norm, nonlinearity=nonlinearity, norm_kwargs=norm_kwargs, **kwargs)
        self.norm_type = norm


class NormConvTranspose1d(nn.Module):
    """`conv.py:159-179` — holds `.convtr`."""

    def __init__(self, *args, causal: bool = False, norm: str = "none", nonlinearity: str = "linear",
                 norm_kwargs: tp.Dict[str, tp.Any] = {}, **kwargs):
        super().__init__()
        self.convtr = ConvParams(*args, norm=norm, transposed=True, nonlinearity=nonlinearity,
                                 norm_kwargs=norm_kwargs, **kwargs)
        self.norm_type = norm


class SConv1d(nn.Module):
    """`SConv1d` (`conv.py:202-236`): causal left zero-pad `(k-1)-(s-1)` (+ right "extra" so the last
    window is full), then Conv1d.  Pointwise (k=1) -> MFMA GEMM; depthwise -> `hilc_dw_conv`;
    Cin=1 -> `hilc_conv_pre`; Cout=1 -> `hilc_conv_post`."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1,
                 dilation: int = 1, groups: int = 1, bias: bool = True, causal: bool = False,
                 norm: str = "none", norm_kwargs: tp.Dict[str, tp.Any] = {},
                 pad_mode: str = "constant", nonlinearity: str = "linear"):
        super().__init__()
        if pad_mode not in ("constant", "zero"):
            raise NotImplementedError("only zero ('constant') padding is on the MI355X hot path")
        if kernel_size > 1 and not causal:
            raise NotImplementedError("non-causal k>1 convolutions are not on the hot path (causal: True in both configs)")
        self.conv = NormConv1d(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups,
                               bias=bias, causal=causal, norm=norm, norm_kwargs=norm_kwargs,
                               nonlinearity=nonlinearity)
        self.causal = causal
        self.pad_mode = pad_mode
        self._cache = _FoldCache()

    def _folded(self, device):
        p = self.conv.conv

        def build():
            w, b = p.effective_weight(), p.effective_bias()
            if p.kernel_size == 1 and p.groups == 1:
                w = fold.pointwise_layout(w)
            elif p.groups == 1 and p.in_channels == 1:
                w = w[:, 0, :].contiguous()
            elif p.groups == 1 and p.out_channels == 1:
                w = w[0].contiguous()
            else:
                w = fold.depthwise_layout(w)
            return w.to(device), None if b is None else b.to(device)
        return self._cache.get((str(device), p.version_key()), build)

    def forward(self, x: Tensor) -> Tensor:
        p = self.conv.conv
        w, b = self._folded(x.device)
        x = x.contiguous().float()
        if p.kernel_size == 1 and p.groups == 1:
            return ops.pw_conv(x, w, b)
        if p.groups == 1 and p.in_channels == 1:
            if p.stride != 1:
                raise NotImplementedError
            return ops.conv_pre(x, w, b)
        if p.groups == 1 and p.out_channels == 1:
            if p.stride != 1:
                raise NotImplementedError
            return ops.conv_post(x, w, b, in_elu=False, do_tanh=False)
        if p.groups == 1:
            raise NotImplementedError("dense k>1 convolutions with Cin,Cout>1 are not part of HILCodec")
        return ops.dw_conv(x, w, b, stride=p.stride)


class SConvTranspose1d(nn.Module):
    """`SConvTranspose1d` (`conv.py:239-282`), causal, trim_right_ratio 1, depthwise k = 2*stride."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1,
                 dilation: int = 1, groups: int = 1, causal: bool = False, norm: str = "none",
                 trim_right_ratio: float = 1.0, norm_kwargs: tp.Dict[str, tp.Any] = {},
                 pad_mode: str = "constant", bias: bool = True, nonlinearity: str = "linear"):
        super().__init__()
        if not causal or trim_right_ratio != 1.0:
            raise NotImplementedError("only causal transposed convs with trim_right_ratio=1 are on the hot path")
        if groups != in_channels or in_channels != out_channels or kernel_size != 2 * stride:
            raise NotImplementedError("only depthwise transposed convs with kernel = 2*stride are on the hot path")
        if bias:
            raise NotImplementedError("HILCodec's up-sampling transposed convs carry no bias (seanet.py:433-438)")
        self.convtr = NormConvTranspose1d(in_channels, out_channels, kernel_size, stride, dilation=dilation,
                                          groups=groups, bias=bias, nonlinearity=nonlinearity, causal=causal,
                                          norm=norm, norm_kwargs=norm_kwargs)
        self.causal = causal
        self.trim_right_ratio = trim_right_ratio
        self._cache = _FoldCache()

    def _folded(self, device):
        p = self.convtr.convtr
        return self._cache.get((str(device), p.version_key()),
                               lambda: fold.depthwise_layout(p.effective_weight()).to(device))

    def forward(self, x: Tensor) -> Tensor:
        p = self.convtr.convtr
        return ops.dw_convtr(x.contiguous().float(), self._folded(x.device), p.stride)


class CausalSTFT(nn.Module):
    """`CausalSTFT` (`conv.py:285-358`): conv-as-DFT magnitude.  Holds the reference's fixed basis
    as buffer `weight` `[n_fft+2, 1, n_fft]`; forward returns the magnitude `[B, n_fft/2+1, L]`."""

    def __init__(self, n_fft: int, hop_size: int, win_size: tp.Optional[int] = None,
                 win_type: tp.Optional[str] = "hann", window: tp.Optional[Tensor] = None,
                 norm: tp.Optional[str] = "backward", pad_mode: str = "constant", learnable: bool = False,
                 eps: float = 1e-12, device=None, dtype=None):
        super().__init__()
        if win_size not in (None, n_fft) or win_type != "hann" or window is not None or norm != "backward" \
                or learnable or pad_mode != "constant":
            raise NotImplementedError("only the configuration HILCodec uses (hann, n_fft window, backward norm, fixed)")
        self.n_fft, self.hop_size, self.cache_len, self.eps = n_fft, hop_size, n_fft - 1, eps
        self.register_buffer("weight", stft_basis(n_fft))
        self._cache = _FoldCache()

    def basis_t(self, device) -> Tensor:
        return self._cache.get((str(device), self.weight.data_ptr(), self.weight._version),
                               lambda: fold.stft_basis_layout(self.weight).to(device))

    def forward(self, x: Tensor) -> Tensor:
        if x.dim() == 2:
            x = x.unsqueeze(1)
        return ops.stft_logmag(x.contiguous().float(), self.basis_t(x.device), self.n_fft, self.hop_size,
                               normalize=2)


def get_extra_padding_for_conv1d(x: Tensor, kernel_size: int, stride: int, padding_total: int = 0) -> int:
    """`conv.py:61-68` (the kernels implement this implicitly: samples past the end read as zero)."""
    length = x.shape[-1]
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length
