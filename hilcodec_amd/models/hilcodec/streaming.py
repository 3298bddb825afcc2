"""Streaming (cache-in / cache-out) HILCodec with the reference's class names, constructor arguments,
state-dict keys and call protocol (`models/hilcodec/streaming.py`), on the gfx950 kernels.

    x, cache_enc = model.encoder(wav_in [B,1,320*m], *cache_enc)     -> x [B,m,128]
    indices      = model.quantizer(x, n)                             -> [n,B,m] int64
    q            = model.dequantizer(indices, n)                     -> [B,m,128]
    wav, cache_dec = model.decoder(q, *cache_dec)                    -> wav [B,1,320*m]

(the loop of `scripts/HILCodec Onnx.ipynb` cell 3 / `test_onnx.py:75-93,123-135`).  Caches are the
reference's: 22 encoder tensors (`Encoder.initialize_cache`, :458-470) and 30 decoder tensors
(:599-607), same order, shapes and contents, so `onnx/*_cache_{enc,dec}.npz` templates apply.
As in the reference the model is meant to be used after `remove_weight_reparameterizations()`
(which also folds the constant scales into the weights, `merge_scaling`); the un-merged forward is
supported too and, like the reference's, then lacks the wav_std scaling of conv_pre / conv_post.
The reference's two decoder deviations from its offline model (pre_scale = 1 in decoder
residual blocks; only the final conv's weight is scaled by wav_std) are reproduced because this
module tree is constructed the same way."""
from __future__ import annotations

import typing as tp
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor, nn

from ... import engine, fold, ops
from .causal_layers import CausalSTFT, SConv1d, SConvTranspose1d
from .modules.conv import ConvParams
from .modules.seanet import _Placeholder, _PlanModule


class EuclideanCodebook(nn.Module):
    """`streaming.py:25-72`: forward(x [B,T,C]) -> (quantized [B,T,C], embed_ind [B,T])."""

    def __init__(self, dim: int = 128, codebook_size: int = 1024, kmeans_init: bool = False, kmeans_iters: int = 20,
                 decay: float = 0.8, eps: float = 1e-7, ema_num_threshold: float = 0.0, ema_num_initial: float = 1.0):
        super().__init__()
        self.decay, self.codebook_size, self.eps, self.ema_num_initial = decay, codebook_size, eps, ema_num_initial
        self.register_buffer("embed", torch.randn(codebook_size, dim))
        self.register_buffer("ema_num", torch.ones(codebook_size) * ema_num_initial)

    def forward(self, x: Tensor) -> tp.Tuple[Tensor, Tensor]:
        cb, cbt, norms = [t.to(x.device) for t in fold.codebook_tables([self.embed])]
        idx, q, _ = ops.rvq_encode(x.contiguous().float(), cb, cbt, norms, 1, channel_last=True, stage_major=True)
        return q, idx[0]

    def decode(self, embed_ind: Tensor) -> Tensor:
        cb = self.embed.detach().float().unsqueeze(0).contiguous().to(embed_ind.device)
        return ops.rvq_decode(embed_ind.unsqueeze(0).contiguous(), cb, 1, channel_last=True, stage_major=True)


class _CodebookStack(nn.Module):
    def _tables(self, dev) -> engine.RvqSpec:
        if torch.compiler.is_compiling():      # the device tables are constants of a compiled graph (see _PlanModule.plan)
            if getattr(self, "_spec", None) is None:
                raise RuntimeError("run the quantizer once eagerly before torch.compile")
            return self._spec
        key = (str(dev),) + tuple((l.embed.data_ptr(), l.embed._version) for l in self.layers)
        if getattr(self, "_key", None) != key:
            cb, cbt, norms = fold.codebook_tables([l.embed for l in self.layers])
            self._spec = engine.RvqSpec(cb.to(dev), cbt.to(dev), norms.to(dev))
            self._key = key
        return self._spec


class ResidualVQ(_CodebookStack):
    """`streaming.py:75-100`: forward(x [B,T,C], n) -> indices [n,B,T] int64."""

    def __init__(self, num_quantizers: int = 16, dropout: bool = False, dropout_index: Optional[List[int]] = None,
                 **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([EuclideanCodebook(**kwargs) for _ in range(num_quantizers)])
        self.rvq_valu_only = False      # launch option, see models/hilcodec/vector_quantize.py

    def forward(self, x: Tensor, n: int) -> Tensor:
        sp = self._tables(x.device)
        if ops.is_scalar_n(n):
            n = min(int(n), len(self.layers))           # reference: `self.layers[:n]`
            if n < 1:
                raise RuntimeError("stack expects a non-empty TensorList")   # torch.stack([]) in the reference
        # else: one n per stream (mixed-bitrate batch); rows >= n_b of the result hold -1
        idx, _, _ = ops.rvq_encode(x.contiguous().float(), sp.codebooks, sp.codebooks_t, sp.norms, n,
                                   channel_last=True, stage_major=True, want_q=False, valu_only=self.rvq_valu_only)
        return idx


class EuclideanCodebookDeq(nn.Module):
    """`streaming.py:103-131`."""

    def __init__(self, dim: int = 128, codebook_size: int = 1024, kmeans_init: bool = False, kmeans_iters: int = 20,
                 decay: float = 0.8, eps: float = 1e-7, ema_num_threshold: float = 0.0, ema_num_initial: float = 1.0):
        super().__init__()
        self.decay, self.codebook_size, self.eps, self.ema_num_initial = decay, codebook_size, eps, ema_num_initial
        self.register_buffer("embed", torch.randn(codebook_size, dim))
        self.register_buffer("ema_num", torch.ones(codebook_size) * ema_num_initial)

    def forward(self, embed_ind: Tensor) -> Tensor:
        cb = self.embed.detach().float().unsqueeze(0).contiguous().to(embed_ind.device)
        return ops.rvq_decode(embed_ind.unsqueeze(0).contiguous(), cb, 1, channel_last=True, stage_major=True)


class Dequantizer(_CodebookStack):
    """`streaming.py:134-157`: forward(indices [n,B,T], n) -> [B,T,C]."""

    def __init__(self, num_quantizers: int = 16, dropout: bool = False, dropout_index: Optional[List[int]] = None,
                 **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([EuclideanCodebookDeq(**kwargs) for _ in range(num_quantizers)])

    def forward(self, indices: Tensor, n: int) -> Tensor:
        sp = self._tables(indices.device)
        if indices.dtype != torch.int64:
            indices = indices.long()                   # test_onnx.py stores int16 (:96-100)
        if ops.is_scalar_n(n):
            n = int(n)
        return ops.rvq_decode(indices.contiguous(), sp.codebooks, n, channel_last=True, stage_major=True)


class DWSBlock(nn.Module):
    """`streaming.py:160-192`: [ELU, pointwise 1x1 (no bias)] -> depthwise causal conv with cache."""

    def __init__(self, act, activation_params: dict, in_chs: int, out_chs: int, kernel_size: int, stride: int = 1,
                 dilation: int = 1, norm: str = "weight_norm", act_all: bool = False, transposed: bool = False,
                 expansion: int = 1, groups: int = -1, bias: bool = True):
        super().__init__()
        if act_all or expansion != 1 or groups != -1 or transposed:
            raise NotImplementedError("act_all / expansion / groups / transposed DWS blocks are not on the hot path")
        self.pointwise = nn.Sequential(_Placeholder("ELU"), SConv1d(in_chs, out_chs, kernel_size=1, norm=norm, bias=False))
        self.depthwise = SConv1d(out_chs, out_chs, kernel_size=kernel_size, stride=stride, dilation=dilation,
                                 groups=out_chs, norm=norm, bias=bias)

    def initialize_cache(self, x: Tensor) -> Tensor:
        return self.depthwise.initialize_cache(x)

    def forward(self, x: Tensor, cache: Tensor) -> tp.Tuple[Tensor, Tensor]:
        w, _ = self.pointwise[1].folded(x.device)
        h = ops.pw_conv(x.contiguous().float(), w, in_elu=True)
        return self.depthwise(h, cache)


class ResBlock(nn.Module):
    """`streaming.py:195-276`."""

    def __init__(self, dim: int, kernel_size: int = 3, dilations: tp.List[int] = [1, 1], activation: str = "ELU",
                 activation_params: dict = {"alpha": 1.0}, norm: str = "weight_norm", compress: int = 2,
                 act_all: bool = False, expansion: int = 1, groups: int = -1, bias: bool = True,
                 res_scale: float = 1.0, idx: int = 0):
        super().__init__()
        if compress != 1 or list(dilations) != [1, 1] or activation != "ELU":
            raise NotImplementedError("compress != 1 / dilations != [1,1] / non-ELU are not on the hot path")
        self.pre_scale = (1 + idx * res_scale ** 2) ** -0.5
        self.block = nn.ModuleList([
            DWSBlock(None, activation_params, dim, dim, kernel_size, dilation=1, norm=norm, act_all=act_all,
                     expansion=expansion, groups=groups, bias=bias) for _ in dilations])
        self.res_scale = res_scale
        self.res_scale_param = nn.Parameter(torch.zeros(1))
        self.merged = False

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        return [b.initialize_cache(x) for b in self.block]

    def merge_scaling(self) -> None:
        """`streaming.py:240-250`: fold res_scale*res_scale_param into the last depthwise conv."""
        scale = self.res_scale * self.res_scale_param.data
        conv = self.block[-1].depthwise
        conv.weight.data.mul_(scale)
        if conv.bias is not None:
            conv.bias.data.mul_(scale)
        self.merged = True

    def spec(self, dev) -> engine.ResBlockSpec:
        pw1, _ = self.block[0].pointwise[1].folded(dev)
        dw1, b1 = self.block[0].depthwise.folded(dev)
        pw2, _ = self.block[1].pointwise[1].folded(dev)
        dw2, b2 = self.block[1].depthwise.folded(dev)
        out_scale = 1.0 if self.merged else float((self.res_scale * self.res_scale_param.detach().float().cpu())[0])
        return engine.ResBlockSpec(pw1, dw1, b1, pw2, dw2, b2, self.pre_scale, out_scale)

    def forward(self, x: Tensor, cache: tp.List[Tensor]) -> tp.Tuple[Tensor, tp.List[Tensor]]:
        new_cache: tp.List[Tensor] = []
        y = engine._resblock(engine.finalize_block(self.spec(x.device), streaming=True), x.contiguous().float(), cache, new_cache)
        return y, new_cache


class L2Norm(nn.Module):
    """`streaming.py:279-286`."""

    def __init__(self, channels: int, eps: float = 1e-12):
        super().__init__()
        self.eps = eps
        self.scale = channels ** 0.5

    def forward(self, x: Tensor) -> Tensor:
        return ops.l2norm(x.contiguous().float(), self.eps, self.scale)


class Scale(nn.Module):
    """`streaming.py:289-301` (non-learnable)."""

    def __init__(self, dim: int, value: float = 1.0, learnable: bool = True, inplace: bool = False):
        super().__init__()
        if learnable:
            raise NotImplementedError("learnable Scale is not used by HILCodec")
        self.scale = value
        self.inplace = inplace


class SpecBlock(nn.Module):
    """`streaming.py:304-365`."""

    def __init__(self, n_fft: int, channels: int, stride: int, norm: str, bias: bool, mean: float = 0.0,
                 std: float = 1.0, res_scale: float = 1.0) -> None:
        super().__init__()
        self.mean, self.std = mean, std
        self.scale = res_scale
        self.spec = CausalSTFT(n_fft=n_fft, hop_size=stride, magnitude=True)
        self.layer = SConv1d(n_fft // 2 + 1, channels, 1, norm=norm, bias=bias)
        self.scale_param = nn.Parameter(torch.zeros(1))
        self.merged = False

    def merge_scaling(self) -> None:
        """`streaming.py:321-344`: y = W@(x-mean)/std * s  ->  (W*s/std) @ x + (-mean/std * sum(W)) * s."""
        if self.merged:
            return
        bias2 = self.layer.weight.data.sum((1, 2)).mul(-self.mean / self.std)
        self.layer.weight.data.div_(self.std)
        if self.layer.bias is not None:
            self.layer.bias.data.add_(bias2)
        else:
            del self.layer._parameters["bias"]
            self.layer.register_buffer("bias", bias2)
        scale = self.scale * self.scale_param.data
        self.layer.weight.data.mul_(scale)
        self.layer.bias.data.mul_(scale)
        self.merged = True

    def spec_spec(self, dev) -> engine.SpecBlockSpec:
        wt, b = self.layer.folded(dev)
        out_scale = 1.0 if self.merged else float((self.scale * self.scale_param.detach().float().cpu())[0])
        return engine.SpecBlockSpec(self.spec.basis_t(dev), self.spec.n_fft, self.spec.hop_size, float(self.mean),
                                    float(self.std), not self.merged, wt, b, out_scale)

    def forward(self, x: Tensor, wav: Tensor) -> Tensor:
        """x `[B,C,L]`, wav `[B,1,hop*L + n_fft - hop]` (history prepended) -> `[B,C,L]`."""
        wav = wav.contiguous().float()
        cl = self.spec.cache_len
        hist, cur = wav[:, :, :cl].contiguous(), wav[:, :, cl:].contiguous()
        return engine._spec_block(self.spec_spec(x.device), x.contiguous().float().clone(), cur, hist)


def _check_stream_options(activation, activation_params, dilation_base, compress, act_all, expansion, groups, norm, bias):
    if activation != "ELU" or float(activation_params.get("alpha", 1.0)) != 1.0:
        raise NotImplementedError("activation must be ELU(alpha=1)")
    if dilation_base != 1 or compress != 1 or act_all or expansion != 1 or groups != -1 or not bias:
        raise NotImplementedError("only the shipped HILCodec hyper-parameters are on the hot path")
    if norm != "weight_norm":
        raise ValueError(f"Unknown norm: {norm}")


class Encoder(_PlanModule):
    """`streaming.py:368-517`: forward(x [B,1,320m], *cache_in) -> (z [B,m,dimension], cache_out list[22])."""

    _stream_plan = True

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_fft_base: int = 64,
                 n_residual_layers: int = 2, ratios: tp.List[int] = [8, 5, 4, 2], activation: str = "ELU",
                 activation_params: dict = {"alpha": 1.0}, norm: str = "weight_norm", kernel_size: int = 5,
                 last_kernel_size: int = 5, residual_kernel_size: int = 5, dilation_base: int = 1, skip: str = "1x1",
                 compress: int = 1, act_all: bool = False, expansion: int = 1, groups: int = -1, l2norm: bool = True,
                 bias: bool = True, res_scale: float = 0.5, wav_std: float = 0.1122080159,
                 spec_means: tp.List[float] = [-4.554, -4.315, -4.021, -3.726, -3.477],
                 spec_stds: tp.List[float] = [2.830, 2.837, 2.817, 2.796, 2.871]):
        super().__init__()
        _check_stream_options(activation, activation_params, dilation_base, compress, act_all, expansion, groups, norm, bias)
        if channels != 1:
            raise NotImplementedError("mono audio only")
        self.dimension = dimension
        self.n_filters = n_filters
        self.ratios = list(reversed(ratios))
        self.n_residual_layers = n_residual_layers
        self.hop_length = np.prod(self.ratios)
        self.res_scale = res_scale
        mult = 1
        self.wav_std = wav_std
        self.conv_pre = ConvParams(channels, mult * n_filters, kernel_size, bias=bias, norm=norm)
        self.conv_pre_cache_len = kernel_size - 1
        self.blocks = nn.ModuleList()
        self.spec_blocks = nn.ModuleList()
        self.downsample_pointwise = nn.ModuleList()
        self.downsample_depthwise = nn.ModuleList()
        stride = 1
        self.scale_layer = Scale(1, value=(1 + n_residual_layers * res_scale ** 2) ** -0.5, learnable=False, inplace=True)
        for spec_mean, spec_std, ratio in zip(spec_means, spec_stds, self.ratios):
            self.blocks.append(nn.ModuleList([
                ResBlock(mult * n_filters, kernel_size=residual_kernel_size, dilations=[dilation_base ** j, 1], norm=norm,
                         activation=activation, activation_params=activation_params, compress=compress,
                         act_all=act_all, expansion=expansion, groups=groups, bias=bias, res_scale=res_scale, idx=j)
                for j in range(1, n_residual_layers + 1)]))
            self.spec_blocks.append(SpecBlock(mult * n_fft_base, mult * n_filters, stride, norm, bias=False,
                                              mean=spec_mean, std=spec_std, res_scale=res_scale))
            stride *= ratio
            self.downsample_pointwise.append(nn.Sequential(
                _Placeholder("ELU"), SConv1d(mult * n_filters, mult * n_filters * 2, 1, norm=norm, bias=False)))
            self.downsample_depthwise.append(SConv1d(mult * n_filters * 2, mult * n_filters * 2, kernel_size=ratio * 2,
                                                     stride=ratio, groups=mult * n_filters * 2, norm=norm, bias=bias))
            mult *= 2
        self.spec_post = SpecBlock(mult * n_fft_base, mult * n_filters, stride, norm, bias=False, mean=spec_means[-1],
                                   std=spec_stds[-1], res_scale=res_scale)
        self.conv_post_act = _Placeholder("ELU")
        self.conv_post_depthwise = SConv1d(mult * n_filters, mult * n_filters, last_kernel_size,
                                           groups=mult * n_filters, norm=norm, bias=False)
        self.conv_post_pointwise = SConv1d(mult * n_filters, dimension, 1, norm=norm, bias=bias)
        self.l2norm = L2Norm(dimension) if l2norm else nn.Identity()
        self.num_cache = len(self.initialize_cache(torch.zeros(1)))
        self.merged = False

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        out: tp.List[Tensor] = [torch.zeros(x.size(0), 1, self.spec_post.spec.cache_len, device=x.device)]
        for blocks, down in zip(self.blocks, self.downsample_depthwise):
            for block in blocks:
                out.extend(block.initialize_cache(x))
            out.append(down.initialize_cache(x))
        out.append(self.conv_post_depthwise.initialize_cache(x))
        return out

    def merge_scaling(self) -> None:
        """`streaming.py:472-480`: conv_pre.weight /= wav_std."""
        if self.merged:
            return
        self.conv_pre.weight.data.div_(self.wav_std)
        self.merged = True

    def build_spec(self, dev) -> engine.EncoderSpec:
        stage_scale = float(self.scale_layer.scale)
        stages = []
        for s, ratio in enumerate(self.ratios):
            pw_wt, _ = self.downsample_pointwise[s][1].folded(dev)
            dw_w, dw_b = self.downsample_depthwise[s].folded(dev)
            stages.append(engine.EncStageSpec(self.spec_blocks[s].spec_spec(dev), [rb.spec(dev) for rb in self.blocks[s]],
                                              stage_scale, pw_wt, dw_w, dw_b, ratio))
        post_dw, _ = self.conv_post_depthwise.folded(dev)
        post_pw, post_b = self.conv_post_pointwise.folded(dev)
        pb = self.conv_pre.effective_bias()
        return engine.EncoderSpec(
            self.conv_pre.effective_weight()[:, 0, :].contiguous().to(dev), None if pb is None else pb.to(dev), 1.0,
            stages, self.spec_post.spec_spec(dev), post_dw, post_pw, post_b, isinstance(self.l2norm, L2Norm),
            self.dimension, self.spec_post.spec.cache_len)

    def _plan_key(self, dev):
        flags = (self.merged,) + tuple(m.merged for m in self.modules() if isinstance(m, (ResBlock, SpecBlock)))
        return super()._plan_key(dev) + flags

    def forward(self, x: Tensor, *args, cache_out: tp.Optional[tp.Sequence[Tensor]] = None
                ) -> tp.Tuple[Tensor, tp.List[Tensor]]:
        """`cache_out` (extension): persistent buffers that receive the new caches (ping-pong state block in HBM);
        without it the new caches are fresh tensors, as in the reference."""
        if len(args) != self.num_cache:
            raise RuntimeError(f"expected {self.num_cache} cache tensors, got {len(args)}")
        return engine.run_encoder(self.plan(x.device), x, list(args), channel_last_out=True, caches_out=cache_out,
                                  opts=self.exec_options)


class Decoder(_PlanModule):
    """`streaming.py:520-648`: forward(q [B,m,dimension], *cache_in) -> (wav [B,1,320m], cache_out list[30])."""

    _stream_plan = True

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_residual_layers: int = 1,
                 ratios: tp.List[int] = [8, 5, 4, 2], activation: str = "ELU", activation_params: dict = {"alpha": 1.0},
                 norm: str = "weight_norm", kernel_size: int = 7, last_kernel_size: int = 7,
                 residual_kernel_size: int = 3, dilation_base: int = 2, skip: str = "1x1", compress: int = 2,
                 final_activation: tp.Optional[str] = None, final_activation_params: tp.Optional[dict] = None,
                 act_all: bool = False, expansion: int = 1, groups: int = -1, bias: bool = True,
                 res_scale: tp.Optional[float] = None, wav_std: float = 0.1122080159):
        super().__init__()
        _check_stream_options(activation, activation_params, dilation_base, compress, act_all, expansion, groups, norm, bias)
        if channels != 1 or res_scale is None or final_activation not in (None, "Tanh"):
            raise NotImplementedError("mono audio, res_scale set, final activation Tanh/None only")
        self.dimension = dimension
        self.channels = channels
        self.n_filters = n_filters
        self.ratios = list(ratios)
        self.n_residual_layers = n_residual_layers
        self.hop_length = np.prod(self.ratios)
        self.wav_std = wav_std
        self.final_activation = final_activation
        mult = int(2 ** len(self.ratios))
        self.conv_pre_pointwise = SConv1d(dimension, mult * n_filters, 1, norm=norm, bias=False)
        self.conv_pre_depthwise = SConv1d(mult * n_filters, mult * n_filters, kernel_size, groups=mult * n_filters,
                                          norm=norm, bias=bias)
        self.blocks = nn.ModuleList()
        self.upsample_act = nn.ModuleList()
        self.upsample_depthwise = nn.ModuleList()
        self.upsample_pointwise = nn.ModuleList()
        self.scale_layer = Scale(1, value=(1 + n_residual_layers * res_scale ** 2) ** -0.5, learnable=False, inplace=True)
        for ratio in self.ratios:
            self.upsample_act.append(_Placeholder("ELU"))
            self.upsample_depthwise.append(SConvTranspose1d(mult * n_filters, mult * n_filters, kernel_size=ratio * 2,
                                                            stride=ratio, groups=mult * n_filters, norm=norm, bias=False))
            self.upsample_pointwise.append(SConv1d(mult * n_filters, mult * n_filters // 2, 1, norm=norm, bias=bias))
            self.blocks.append(nn.ModuleList([
                ResBlock(mult * n_filters // 2, kernel_size=residual_kernel_size, dilations=[dilation_base ** j, 1],
                         activation=activation, activation_params=activation_params, norm=norm, compress=compress,
                         act_all=act_all, expansion=expansion, groups=groups, bias=bias, res_scale=res_scale)
                for j in range(n_residual_layers)]))          # no idx -> pre_scale = 1 (streaming.py:576-583)
            mult //= 2
        self.conv_post_act = _Placeholder("ELU")
        self.conv_post = SConv1d(n_filters, channels, last_kernel_size, norm=norm, bias=bias)
        self.final_act = _Placeholder(final_activation or "Identity")
        self.merged = False

    def initialize_cache(self, x: Tensor) -> tp.List[Tensor]:
        out: tp.List[Tensor] = [self.conv_pre_depthwise.initialize_cache(x)]
        for blocks, up in zip(self.blocks, self.upsample_depthwise):
            out.append(up.initialize_cache(x))
            for block in blocks:
                out.extend(block.initialize_cache(x))
        out.append(self.conv_post.initialize_cache(x))
        return out

    def merge_scaling(self) -> None:
        """`streaming.py:609-617`: conv_post.weight *= wav_std (the bias is NOT scaled)."""
        if self.merged:
            return
        self.conv_post.weight.data.mul_(self.wav_std)
        self.merged = True

    def build_spec(self, dev) -> engine.DecoderSpec:
        stage_scale = float(self.scale_layer.scale)
        pre_pw, _ = self.conv_pre_pointwise.folded(dev)
        pre_dw, pre_b = self.conv_pre_depthwise.folded(dev)
        stages = []
        for i, ratio in enumerate(self.ratios):
            tr_w = self.upsample_depthwise[i].folded(dev)
            pw_wt, pw_b = self.upsample_pointwise[i].folded(dev)
            stages.append(engine.DecStageSpec(stage_scale if i > 0 else 1.0, tr_w, ratio, pw_wt, pw_b,
                                              [rb.spec(dev) for rb in self.blocks[i]]))
        w, b = self.conv_post.folded(dev)
        return engine.DecoderSpec(pre_pw, pre_dw, pre_b, stages, stage_scale, w, b, 1.0, self.final_activation == "Tanh")

    def _plan_key(self, dev):
        flags = (self.merged,) + tuple(m.merged for m in self.modules() if isinstance(m, ResBlock))
        return super()._plan_key(dev) + flags

    def forward(self, x: Tensor, *args, cache_out: tp.Optional[tp.Sequence[Tensor]] = None
                ) -> tp.Tuple[Tensor, tp.List[Tensor]]:
        q = x.float().transpose(1, 2).contiguous()       # [B,T',C] -> [B,C,T'] (layout copy only)
        return engine.run_decoder(self.plan(x.device), q, list(args), caches_out=cache_out, opts=self.exec_options)


class HILCodec(nn.Module):
    """`streaming.py:651-747`."""

    def __init__(self, sample_rate: int = 16_000, channels_audio: int = 1, channels_enc: int = 64,
                 channels_dec: int = 96, n_fft_base: int = 64, n_residual_enc: int = 2, n_residual_dec: int = 3,
                 res_scale_enc: tp.Optional[float] = 0.5773502691896258,
                 res_scale_dec: tp.Optional[float] = 0.5773502691896258, strides: tp.List[int] = [8, 5, 4, 2],
                 activation: str = "ELU", activation_kwargs: dict = {"alpha": 1.0}, norm: str = "weight_norm",
                 kernel_size: int = 5, last_kernel_size: int = 5, residual_kernel_size: int = 5,
                 dilation_base: int = 1, skip: str = "identity", compress: int = 1,
                 final_activation: tp.Optional[str] = "Tanh", use_vq: bool = True, vq: str = "ResidualVQ",
                 vq_kwargs: tp.Dict[str, tp.Any] = dict(dim=128, ), act_all: bool = False, expansion: int = 1,
                 groups: int = -1, encoder_l2norm: bool = True, bias: bool = True, spec: str = "stft",
                 spec_compression: str = "log", zero_init: bool = True, inout_norm: bool = True):
        assert spec == "stft"
        assert spec_compression == "log"
        assert skip == "identity", skip
        assert zero_init == True   # noqa: E712
        assert inout_norm == True  # noqa: E712
        if expansion != 1 and groups != -1:
            raise RuntimeError(f"Both expansion({expansion}) and groups({groups}) are set. "
                               f"Either set expansion=1 or set groups=-1")
        super().__init__()
        self.norm = norm
        channels_vq = vq_kwargs["dim"]
        self.encoder = Encoder(channels_audio, channels_vq, channels_enc, n_fft_base, n_residual_enc, strides,
                               activation, activation_kwargs, norm, kernel_size, last_kernel_size,
                               residual_kernel_size, dilation_base, skip, compress, act_all=act_all,
                               expansion=expansion, groups=groups, l2norm=encoder_l2norm, bias=bias,
                               res_scale=res_scale_enc)
        self.decoder = Decoder(channels_audio, channels_vq, channels_dec, n_residual_dec, strides, activation,
                               activation_kwargs, norm, kernel_size, last_kernel_size, residual_kernel_size,
                               dilation_base, skip, compress, final_activation=final_activation, act_all=act_all,
                               expansion=expansion, groups=groups, bias=bias, res_scale=res_scale_dec)
        self.quantizer = ResidualVQ(**vq_kwargs)
        self.dequantizer = Dequantizer(**vq_kwargs)
        self.sample_rate = sample_rate
        self.channels = channels_audio

    def initialize_cache(self, x: Tensor) -> tp.Tuple[tp.List[Tensor], tp.List[Tensor]]:
        return self.encoder.initialize_cache(x), self.decoder.initialize_cache(x)

    @torch.no_grad()
    def forward(self, x: Tensor, n: int, *args):
        """`streaming.py:726-738`.  The reference feeds the quantizer's INDICES straight into its decoder,
        which cannot work; here the indices are de-quantised first, which is what every
        caller of the reference does by hand (notebook cell 3, test_onnx.py)."""
        cache = [*args]
        cache_enc = cache[:self.encoder.num_cache]
        cache_dec = cache[self.encoder.num_cache:]
        x, cache_enc = self.encoder(x, *cache_enc)
        idx = self.quantizer(x, n)
        x = self.dequantizer(idx, n)
        x, cache_dec = self.decoder(x, *cache_dec)
        return x, cache_enc, cache_dec

    def remove_weight_reparameterizations(self):
        """`streaming.py:740-747`: remove weight_norm everywhere, then every `merge_scaling`."""
        if self.norm == "weight_norm":
            for module in self.modules():
                if isinstance(module, ConvParams):
                    module.remove_reparameterization()
        for module in self.modules():
            if hasattr(module, "merge_scaling"):
                module.merge_scaling()

    def load_offline_state_dict(self, sd: tp.Dict[str, Tensor], norm: str = "weight_norm",
                                norm_kwargs: tp.Optional[dict] = None) -> None:
        """Fill this streaming model from an OFFLINE checkpoint (`checkpoint['model']` of the reference)
        with the correspondence of `scripts/HILCodec Onnx.ipynb` cell 1.  Call
        `remove_weight_reparameterizations()` afterwards, as the notebook does.

        `norm` = the re-parameterisation the OFFLINE model was built with.  The reference's streaming classes know
        weight_norm only (`causal_layers.py:200-204,216-220` raise ValueError otherwise), so a checkpoint of an offline
        `HILCodec(norm="weight_standardization", norm_kwargs=...)` (`conv.py:36-37`) has one way into the streaming
        model: every conv folded to a plain weight with `modules/weight_standardization.py:30-41`'s expression
        (`fold.weight_standardization_fold`; `norm_kwargs`: `eps`, `scale`, as given to the offline constructor) — this
        model's convs then hold plain `weight`s, as after `remove_weight_norm`."""
        if norm not in ("weight_norm", "weight_standardization"):
            raise ValueError(f"Unknown norm: {norm}")
        ws = norm == "weight_standardization"
        if ws:
            kw = dict(norm_kwargs or {})
            # the reference passes **norm_kwargs to `weight_standardization` (`modules/weight_standardization.py:44-53`: name, dim, eps, scale,
            # learnable_gain, zero_init).  The fold standardises over every axis but 0, which is `dim = 0`; `learnable_gain` / `zero_init` only
            # decide how weight_g was created (it is in the checkpoint or it is not).  Anything else would be folded over the wrong axes:
            unknown = set(kw) - {"eps", "scale", "dim", "learnable_gain", "zero_init", "name"}
            if unknown:
                raise ValueError(f"load_offline_state_dict: unsupported weight-standardisation arguments {sorted(unknown)}")
            if kw.get("dim", 0) not in (0, (0,), [0]) or kw.get("name", "weight") != "weight":
                raise ValueError("load_offline_state_dict: only weight standardisation of `weight` over dim = 0 (the reference's default and "
                                 f"its configs') can be folded, got dim = {kw.get('dim')!r}, name = {kw.get('name')!r}")
            ws_eps = float(kw.get("eps", 1e-7))
            ws_scale = None if kw.get("scale") is None else torch.ones(1) * float(kw["scale"])
            for module in self.modules():
                if isinstance(module, ConvParams):
                    module.remove_reparameterization()
        own = self.state_dict()
        new: tp.Dict[str, Tensor] = {}

        def conv(dst: str, src: str):
            if ws and f"{src}.weight_v" in sd:
                scale = sd.get(f"{src}.weight_scale", ws_scale)
                new[f"{dst}.weight"] = fold.weight_standardization_fold(sd[f"{src}.weight_v"], sd.get(f"{src}.weight_g"),
                                                                         scale, ws_eps)
                if f"{src}.bias" in sd:
                    new[f"{dst}.bias"] = sd[f"{src}.bias"]
                return
            for suffix in ("weight_g", "weight_v", "bias", "weight"):
                if f"{src}.{suffix}" in sd:
                    new[f"{dst}.{suffix}"] = sd[f"{src}.{suffix}"]

        def res(dst: str, src: str):
            conv(f"{dst}.block.0.pointwise.1", f"{src}.block.1.conv.conv")
            conv(f"{dst}.block.0.depthwise", f"{src}.block.2.conv.conv")
            conv(f"{dst}.block.1.pointwise.1", f"{src}.block.4.conv.conv")
            conv(f"{dst}.block.1.depthwise", f"{src}.block.5.conv.conv")
            new[f"{dst}.res_scale_param"] = sd[f"{src}.res_scale_param"]

        e = self.encoder
        conv("encoder.conv_pre", "encoder.conv_pre.1.conv.conv")
        for s in range(len(e.ratios)):
            for j in range(e.n_residual_layers):
                res(f"encoder.blocks.{s}.{j}", f"encoder.blocks.{s}.{j}")
            conv(f"encoder.spec_blocks.{s}.layer", f"encoder.spec_blocks.{s}.layer.conv.conv")
            new[f"encoder.spec_blocks.{s}.scale_param"] = sd[f"encoder.spec_blocks.{s}.scale_param"]
            conv(f"encoder.downsample_pointwise.{s}.1", f"encoder.downsample.{s}.2.conv.conv")
            conv(f"encoder.downsample_depthwise.{s}", f"encoder.downsample.{s}.3.conv.conv")
        conv("encoder.spec_post.layer", "encoder.spec_post.layer.conv.conv")
        new["encoder.spec_post.scale_param"] = sd["encoder.spec_post.scale_param"]
        conv("encoder.conv_post_depthwise", "encoder.conv_post.1.conv.conv")
        conv("encoder.conv_post_pointwise", "encoder.conv_post.2.conv.conv")
        d = self.decoder
        conv("decoder.conv_pre_pointwise", "decoder.model.0.conv.conv")
        conv("decoder.conv_pre_depthwise", "decoder.model.1.conv.conv")
        pos = 2
        for i in range(len(d.ratios)):
            pos += 2
            conv(f"decoder.upsample_depthwise.{i}", f"decoder.model.{pos}.convtr.convtr")
            conv(f"decoder.upsample_pointwise.{i}", f"decoder.model.{pos + 1}.conv.conv")
            pos += 2
            for j in range(d.n_residual_layers):
                res(f"decoder.blocks.{i}.{j}", f"decoder.model.{pos}")
                pos += 1
        conv("decoder.conv_post", f"decoder.model.{pos + 2}.conv.conv")
        for i in range(len(self.quantizer.layers)):
            for tgt in ("quantizer", "dequantizer"):
                new[f"{tgt}.layers.{i}.embed"] = sd[f"quantizer.layers.{i}.embed"]
                new[f"{tgt}.layers.{i}.ema_num"] = sd[f"quantizer.layers.{i}.ema_num"]
        missing = [k for k in own if k not in new and not k.endswith("spec.weight")]
        if missing:
            raise KeyError(f"offline checkpoint lacks tensors for: {missing[:5]}...")
        self.load_state_dict(new, strict=False)
