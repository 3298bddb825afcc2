"""Offline codec `HILCodec` with the reference's constructor, attributes, state-dict keys and
`forward` contract (`models/hilcodec/models.py:24-124`), running encoder -> RVQ -> decoder on the
hand-written gfx950 kernels."""
from __future__ import annotations

import typing as tp

import numpy as np
import torch
from torch import Tensor, nn

from . import modules as m
from .modules.conv import ConvParams
from .vector_quantize import ResidualVQ

Array = tp.Union[np.ndarray, list]


class HILCodec(nn.Module):
    def __init__(self, sample_rate: int, channels_audio: int = 1, channels_enc: int = 32, channels_dec: int = 32,
                 n_fft_base: int = 64, n_residual_enc: int = 1, n_residual_dec: int = 1,
                 res_scale_enc: tp.Optional[float] = None, res_scale_dec: tp.Optional[float] = None,
                 strides: tp.List[int] = [8, 5, 4, 2], activation: str = "ELU",
                 activation_kwargs: dict = {"alpha": 1.0}, norm: str = "weight_norm",
                 norm_kwargs: tp.Dict[str, tp.Any] = {}, kernel_size: int = 5, last_kernel_size: int = 5,
                 residual_kernel_size: int = 5, dilation_base: int = 1, skip: str = "identity",
                 final_activation: tp.Optional[str] = "Tanh", vq: str = "ResidualVQ",
                 vq_kwargs: tp.Dict[str, tp.Any] = {}, act_all: bool = False, expansion: int = 1,
                 groups: int = -1, encoder_l2norm: bool = True, bias: bool = True, spec: str = "stft",
                 spec_compression: str = "", spec_learnable: bool = False, pad_mode: str = "constant",
                 causal: bool = True, zero_init: bool = True, inout_norm: bool = True):
        assert spec in ["stft", ""]
        assert skip in ["1x1", "scale", "channelwise_scale", "identity"]
        if expansion != 1 and groups != -1:
            raise RuntimeError(f"Both expansion({expansion}) and groups({groups}) are set. "
                               f"Either set expansion=1 or set groups=-1")
        super().__init__()
        self.norm = norm
        channels_vq = vq_kwargs["dim"]
        self.encoder = m.SEANetEncoder(
            channels_audio, channels_vq, channels_enc, n_fft_base, n_residual_enc, strides, activation,
            activation_kwargs, norm, norm_kwargs, kernel_size, last_kernel_size, residual_kernel_size,
            dilation_base, skip, causal=causal, act_all=act_all, expansion=expansion, groups=groups,
            l2norm=encoder_l2norm, bias=bias, spec=spec, spec_compression=spec_compression,
            res_scale=res_scale_enc, pad_mode=pad_mode, spec_learnable=spec_learnable, zero_init=zero_init,
            inout_norm=inout_norm)
        self.decoder = m.SEANetDecoder(
            channels_audio, channels_vq, channels_dec, n_residual_dec, strides, activation, activation_kwargs,
            norm, norm_kwargs, kernel_size, last_kernel_size, residual_kernel_size, dilation_base, skip,
            causal=causal, final_activation=final_activation, act_all=act_all, expansion=expansion,
            groups=groups, bias=bias, res_scale=res_scale_dec, pad_mode=pad_mode, zero_init=zero_init,
            inout_norm=inout_norm)
        if vq == "ResidualVQ":
            self.quantizer = ResidualVQ(channel_last=False, **vq_kwargs)
        elif vq == "":
            self.quantizer = None
        else:
            raise ValueError(f"Unknown vq: {vq}")
        self.sample_rate = sample_rate
        self.channels = channels_audio

    @torch.no_grad()
    def forward(self, x: Tensor, n: tp.Optional[int] = None) -> tp.Tuple[Tensor, Array, Tensor]:
        """x `[B,1,T]` -> (wav `[B,1,ceil(T/320)*320]` fp32, num_replaces int64[Nq], loss_vq 0-d fp32)
        — `models.py:111-118`."""
        x = self.encoder(x)
        if self.quantizer is not None:
            x, num_replaces, loss_vq = self.quantizer(x, n)
        else:
            num_replaces, loss_vq = [], torch.zeros(1, dtype=torch.float32, device=x.device)
        x = self.decoder(x)
        return x.float(), num_replaces, loss_vq

    def remove_weight_reparameterizations(self):
        """`models.py:120-124`: only the `NormConv1d` convs are touched (a transposed conv keeps its
        parametrisation, exactly as in the reference)."""
        if self.norm == "weight_norm":
            for module in self.modules():
                if isinstance(module, m.NormConv1d):
                    module.conv.remove_reparameterization()
