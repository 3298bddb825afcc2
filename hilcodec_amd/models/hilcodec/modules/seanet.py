"""Offline SEANet encoder / decoder with the reference's class names, constructor arguments,
attributes and state-dict keys (`models/hilcodec/modules/seanet.py`), executing on the hand-written
gfx950 kernels through a folded execution plan (`hilcodec_amd/engine.py`).

The sub-modules exist to own parameters under the reference's key layout (so a reference checkpoint
`checkpoint['model']` loads with `load_state_dict`) and to be individually callable; the top-level
`SEANetEncoder.forward` / `SEANetDecoder.forward` do not call them one by one but run the plan.
"""
from __future__ import annotations

import typing as tp

import numpy as np
import torch
from torch import Tensor, nn

from .... import engine, fold, ops
from .conv import CausalSTFT, ConvParams, SConv1d, SConvTranspose1d, _FoldCache


class _Placeholder(nn.Module):
    """Parameter-free slot (ELU / Scale / Tanh / Identity of the reference's nn.Sequential) kept so
    that Sequential indices — and therefore state-dict keys — match the reference."""

    def __init__(self, kind: str, value: tp.Optional[float] = None):
        super().__init__()
        self.kind, self.value = kind, value

    def extra_repr(self) -> str:
        return self.kind if self.value is None else f"{self.kind}({self.value})"


def _pw_dev(conv: SConv1d, dev):
    p = conv.conv.conv
    b = p.effective_bias()
    return fold.pointwise_layout(p.effective_weight()).to(dev), None if b is None else b.to(dev)


def _dw_dev(conv, dev):
    p = conv.conv.conv if hasattr(conv, "conv") else conv.convtr.convtr
    b = p.effective_bias()
    return fold.depthwise_layout(p.effective_weight()).to(dev), None if b is None else b.to(dev)


def _check_block_options(skip, act_all, expansion, groups, activation, activation_params, causal, pad_mode, dilation_base=1):
    if skip != "identity":
        raise NotImplementedError("skip must be 'identity' (both shipped configs)")
    if act_all or expansion != 1 or groups != -1:
        raise NotImplementedError("act_all/expansion/groups variants are not on the hot path")
    if activation != "ELU" or float(activation_params.get("alpha", 1.0)) != 1.0:
        raise NotImplementedError("activation must be ELU(alpha=1)")
    if not causal or pad_mode != "constant":
        raise NotImplementedError("only the causal, zero-padded model is on the hot path")
    if dilation_base != 1:
        raise NotImplementedError("dilation_base must be 1")


class SEANetResnetBlock(nn.Module):
    """`SEANetResnetBlock` (`seanet.py:55-148`), skip='identity': two [ELU, pw 1x1, dw k] pairs,
    `out = block(x*pre_scale) * (res_scale*res_scale_param) + x`."""

    def __init__(self, dim: int, kernel_size: int = 3, dilations: tp.List[int] = [1, 1],
                 activation: str = "ELU", activation_params: dict = {"alpha": 1.0}, norm: str = "weight_norm",
                 norm_params: tp.Dict[str, tp.Any] = {}, causal: bool = False, pad_mode: str = "constant",
                 skip: str = "1x1", act_all: bool = False, expansion: int = 1, groups: int = -1,
                 bias: bool = True, res_scale: tp.Optional[float] = None, idx: int = 0, zero_init: bool = True):
        super().__init__()
        _check_block_options(skip, act_all, expansion, groups, activation, activation_params, causal, pad_mode)
        if list(dilations) != [1, 1]:
            raise NotImplementedError("dilations must be [1, 1]")
        if res_scale is None:
            raise NotImplementedError("res_scale=None is unreachable from the shipped configs (SURVEY §3.4)")
        self.pre_scale = (1 + idx * res_scale ** 2) ** -0.5
        block: tp.List[nn.Module] = []
        for _ in dilations:
            block += [
                _Placeholder("ELU"),
                SConv1d(dim, dim, kernel_size=1, norm=norm, norm_kwargs=norm_params, bias=False, nonlinearity="relu"),
                SConv1d(dim, dim, kernel_size=kernel_size, groups=dim, norm=norm, norm_kwargs=norm_params,
                        causal=causal, pad_mode=pad_mode, bias=bias, nonlinearity="linear"),
            ]
        self.block = nn.Sequential(*block)
        self.shortcut = nn.Identity()
        self.res_scale = res_scale
        if zero_init:
            self.res_scale_param = nn.Parameter(torch.zeros(1))
        else:
            self.res_scale_param = None
        self._cache = _FoldCache()

    def out_scale(self) -> float:
        """`scale = res_scale * res_scale_param` evaluated exactly like the reference (fp32 tensor op)."""
        if self.res_scale_param is None:
            return float(torch.tensor(self.res_scale, dtype=torch.float32))
        return float((self.res_scale * self.res_scale_param.detach().float().cpu())[0])

    def spec(self, dev, pre_scale: tp.Optional[float] = None) -> engine.ResBlockSpec:
        pw1, _ = _pw_dev(self.block[1], dev)
        dw1, b1 = _dw_dev(self.block[2], dev)
        pw2, _ = _pw_dev(self.block[4], dev)
        dw2, b2 = _dw_dev(self.block[5], dev)
        return engine.ResBlockSpec(pw1, dw1, b1, pw2, dw2, b2,
                                   self.pre_scale if pre_scale is None else pre_scale, self.out_scale())

    def _key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def forward(self, x: Tensor) -> Tensor:
        sp = self._cache.get((str(x.device), self._key()), lambda: engine.finalize_block(self.spec(x.device)))
        return engine._resblock(sp, x.contiguous().float(), None, None)


class L2Norm(nn.Module):
    """`L2Norm` (`seanet.py:151-162`)."""

    def __init__(self, channels: int, eps: float = 1e-12, inout_norm: bool = True):
        super().__init__()
        self.scale = channels ** 0.5
        self.eps = eps
        self.inout_norm = inout_norm

    def forward(self, x: Tensor) -> Tensor:
        return ops.l2norm(x.contiguous().float(), self.eps, self.scale if self.inout_norm else 1.0)


class Scale(nn.Module):
    """`Scale` (`seanet.py:165-178`), non-learnable form only (the only one HILCodec instantiates)."""

    def __init__(self, dim: int, value: float = 1.0, learnable: bool = True, inplace: bool = False):
        super().__init__()
        if learnable:
            raise NotImplementedError("learnable Scale is not used by HILCodec")
        self.scale = value
        self.inplace = inplace


class SpecBlock(nn.Module):
    """`SpecBlock` (`seanet.py:181-246`): x += conv1x1((log(max(|STFT(wav)|,1e-5)) - mean)/std) * scale."""

    def __init__(self, spec: str, spec_compression: str, n_fft: int, channels: int, stride: int, norm: str,
                 norm_params: tp.Dict[str, tp.Any], bias: bool, pad_mode: str, learnable: bool,
                 causal: bool = True, mean: float = 0.0, std: float = 1.0,
                 res_scale: tp.Optional[float] = 1.0, zero_init: bool = True, inout_norm: bool = True):
        super().__init__()
        if spec != "stft" or spec_compression != "log" or learnable or not causal or not inout_norm:
            raise NotImplementedError("SpecBlock: only spec='stft', compression='log', fixed causal basis, inout_norm")
        self.learnable = learnable
        self.spec = CausalSTFT(n_fft=n_fft, hop_size=stride, pad_mode=pad_mode, learnable=learnable)
        self.compression = "log"
        self.inout_norm = inout_norm
        self.mean, self.std = mean, std
        self.scale = res_scale
        self.scale_param = None
        self.layer = SConv1d(n_fft // 2 + 1, channels, 1, norm=norm, norm_kwargs=norm_params, bias=bias,
                             pad_mode=pad_mode)
        if zero_init:
            self.scale_param = nn.Parameter(torch.zeros(1))
        self._cache = _FoldCache()

    def out_scale(self) -> float:
        scale = 1.0 if self.scale is None else self.scale
        if self.scale_param is None:
            return float(torch.tensor(scale, dtype=torch.float32))
        return float((self.scale_param.detach().float().cpu() * scale)[0])      # seanet.py:241-244

    def spec_spec(self, dev) -> engine.SpecBlockSpec:
        wt, b = _pw_dev(self.layer, dev)
        return engine.SpecBlockSpec(self.spec.basis_t(dev), self.spec.n_fft, self.spec.hop_size,
                                    float(self.mean), float(self.std), True, wt, b, self.out_scale())

    def _key(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def forward(self, x: Tensor, wav: Tensor) -> Tensor:
        sp = self._cache.get((str(x.device), self._key()), lambda: self.spec_spec(x.device))
        return engine._spec_block(sp, x.contiguous().float().clone(), wav.contiguous().float(), None)


class _PlanModule(nn.Module):
    """Mixin: folded plan cached per (device, parameter versions); per-model execution options."""
    _stream_plan = False      # streaming Encoder / Decoder: the plan also holds packed weights for the wide fused blocks

    @property
    def exec_options(self) -> engine.ExecOptions:
        """this module's own execution options (arithmetic mode of the decoder's GEMMs, capture side stream): never
        shared between models, never process-global"""
        opts = self.__dict__.get("_exec_options")
        if opts is None:
            opts = self.__dict__["_exec_options"] = engine.ExecOptions()
        return opts

    def _plan_key(self, dev):
        return (str(dev),) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    @staticmethod
    def _norm_dev(dev):
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None and torch.cuda.is_available():
            dev = torch.device("cuda", torch.cuda.current_device())      # "cuda" and "cuda:0" are one plan
        return dev

    def plan(self, dev):
        dev = self._norm_dev(dev)
        if torch.compiler.is_compiling():
            # inside torch.compile the folded plan is a constant of the graph: fold (host work, data_ptr keys) cannot be
            # traced, so it must exist already — `prepare(device)` or any earlier eager call builds it.  The cheap part of
            # the key (the device) is still compared: a plan folded for another device would bake wrong-device pointers
            # into the graph.  After in-place weight updates / merge_scaling / load_state_dict: prepare() again, recompile.
            cached = getattr(self, "_plan_cache", None)
            if cached is None:
                raise RuntimeError("call .prepare(device) (or run the module once eagerly) before torch.compile")
            if self._plan_cache_key[0] != str(dev):
                raise RuntimeError(f"the folded plan was prepared for {self._plan_cache_key[0]}, the input is on {dev}: "
                                   "call .prepare(device) for this device before compiling")
            return cached
        key = self._plan_key(dev)
        if getattr(self, "_plan_cache_key", None) != key:
            self._plan_cache = engine.finalize_spec(self.build_spec(dev), streaming=self._stream_plan)
            self._plan_cache_key = key
        return self._plan_cache

    def prepare(self, dev):
        """Fold the weights for `dev` now (otherwise done lazily by the first forward)."""
        self.plan(torch.device(dev))
        return self


class SEANetEncoder(_PlanModule):
    """`SEANetEncoder` (`seanet.py:249-378`): x `[B,1,T]` -> `[B,dimension,ceil(T/hop_length)]`."""

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_fft_base: int = 64,
                 n_residual_layers: int = 1, ratios: tp.List[int] = [8, 5, 4, 2], activation: str = "ELU",
                 activation_params: dict = {"alpha": 1.0}, norm: str = "weight_norm",
                 norm_params: tp.Dict[str, tp.Any] = {}, kernel_size: int = 7, last_kernel_size: int = 7,
                 residual_kernel_size: int = 3, dilation_base: int = 2, skip: str = "1x1",
                 causal: bool = False, pad_mode: str = "constant", act_all: bool = False, expansion: int = 1,
                 groups: int = -1, l2norm: bool = False, bias: bool = True, spec: str = "stft",
                 spec_compression: str = "", spec_learnable: bool = False,
                 res_scale: tp.Optional[float] = None, wav_std: float = 0.1122080159,
                 spec_means: tp.List[float] = [-4.554, -4.315, -4.021, -3.726, -3.477],
                 spec_stds: tp.List[float] = [2.830, 2.837, 2.817, 2.796, 2.871],
                 zero_init: bool = True, inout_norm: bool = True):
        super().__init__()
        _check_block_options(skip, act_all, expansion, groups, activation, activation_params, causal, pad_mode,
                             dilation_base)
        if channels != 1:
            raise NotImplementedError("mono audio only (channels_audio=1)")
        if res_scale is None or not inout_norm or not bias:
            raise NotImplementedError("res_scale=None / inout_norm=False / bias=False are not on the hot path")
        self.dimension = dimension
        self.n_filters = n_filters
        self.ratios = list(reversed(ratios))
        self.n_residual_layers = n_residual_layers
        self.hop_length = np.prod(self.ratios)
        self.wav_std = wav_std
        self.res_scale = res_scale
        self.l2norm = l2norm

        mult = 1
        self.conv_pre = nn.Sequential(
            _Placeholder("Scale", 1 / wav_std),
            SConv1d(channels, mult * n_filters, kernel_size, norm=norm, norm_kwargs=norm_params, causal=causal,
                    pad_mode=pad_mode, bias=bias))
        self.blocks = nn.ModuleList()
        self.spec_blocks = nn.ModuleList()
        self.downsample = nn.ModuleList()
        stride = 1
        for block_idx, ratio in enumerate(self.ratios):
            block = []
            for j in range(1, n_residual_layers + 1):
                block.append(SEANetResnetBlock(
                    mult * n_filters, kernel_size=residual_kernel_size, dilations=[dilation_base ** j, 1],
                    norm=norm, norm_params=norm_params, activation=activation, activation_params=activation_params,
                    causal=causal, pad_mode=pad_mode, skip=skip, act_all=act_all, expansion=expansion,
                    groups=groups, bias=bias, res_scale=res_scale, idx=j, zero_init=zero_init))
            self.blocks.append(nn.Sequential(*block))
            self.spec_blocks.append(SpecBlock(
                spec, spec_compression, mult * n_fft_base, mult * n_filters, stride, norm, norm_params, bias=False,
                pad_mode=pad_mode, learnable=spec_learnable, causal=causal, mean=spec_means[block_idx],
                std=spec_stds[block_idx], res_scale=res_scale, zero_init=zero_init, inout_norm=inout_norm))
            stride *= ratio
            self.downsample.append(nn.Sequential(
                _Placeholder("Scale", (1 + n_residual_layers * res_scale ** 2) ** -0.5),
                _Placeholder("ELU"),
                SConv1d(mult * n_filters, mult * n_filters * 2, 1, norm=norm, norm_kwargs=norm_params, bias=False,
                        nonlinearity="relu"),
                SConv1d(mult * n_filters * 2, mult * n_filters * 2, kernel_size=ratio * 2, stride=ratio,
                        groups=mult * n_filters * 2, norm=norm, norm_kwargs=norm_params, causal=causal,
                        pad_mode=pad_mode, bias=bias)))
            mult *= 2
        self.spec_post = SpecBlock(
            spec, spec_compression, mult * n_fft_base, mult * n_filters, stride, norm, norm_params, bias=False,
            pad_mode=pad_mode, learnable=spec_learnable, causal=causal, mean=spec_means[-1], std=spec_stds[-1],
            res_scale=res_scale, zero_init=zero_init, inout_norm=inout_norm)
        self.conv_post = nn.Sequential(
            _Placeholder("ELU"),
            SConv1d(mult * n_filters, mult * n_filters, last_kernel_size, groups=mult * n_filters, norm=norm,
                    norm_kwargs=norm_params, causal=causal, pad_mode=pad_mode, bias=False, nonlinearity="relu"),
            SConv1d(mult * n_filters, dimension, 1, norm=norm, norm_kwargs=norm_params, bias=bias),
            L2Norm(dimension, inout_norm=inout_norm) if l2norm else nn.Identity())
        if l2norm:
            with torch.no_grad():
                self.conv_post[-2].conv.conv.bias.normal_()            # seanet.py:359-366

    def build_spec(self, dev) -> engine.EncoderSpec:
        pre = self.conv_pre[1].conv.conv
        stage_scale = (1 + self.n_residual_layers * self.res_scale ** 2) ** -0.5
        stages = []
        for s, ratio in enumerate(self.ratios):
            pw_wt, _ = _pw_dev(self.downsample[s][2], dev)
            dw_w, dw_b = _dw_dev(self.downsample[s][3], dev)
            stages.append(engine.EncStageSpec(
                self.spec_blocks[s].spec_spec(dev), [rb.spec(dev) for rb in self.blocks[s]],
                stage_scale, pw_wt, dw_w, dw_b, ratio))
        post_dw, _ = _dw_dev(self.conv_post[1], dev)
        post_pw, post_b = _pw_dev(self.conv_post[2], dev)
        pb = pre.effective_bias()
        return engine.EncoderSpec(
            pre.effective_weight()[:, 0, :].contiguous().to(dev), None if pb is None else pb.to(dev),
            1 / self.wav_std, stages, self.spec_post.spec_spec(dev), post_dw, post_pw, post_b,
            bool(self.l2norm), self.dimension, self.spec_post.spec.n_fft - 1)

    def forward(self, x: Tensor) -> Tensor:
        return engine.run_encoder(self.plan(x.device), x, opts=self.exec_options)


class SEANetDecoder(_PlanModule):
    """`SEANetDecoder` (`seanet.py:381-479`): z `[B,dimension,F]` -> `[B,1,F*hop_length]`."""

    def __init__(self, channels: int = 1, dimension: int = 128, n_filters: int = 32, n_residual_layers: int = 1,
                 ratios: tp.List[int] = [8, 5, 4, 2], activation: str = "ELU",
                 activation_params: dict = {"alpha": 1.0}, norm: str = "weight_norm",
                 norm_params: tp.Dict[str, tp.Any] = {}, kernel_size: int = 7, last_kernel_size: int = 7,
                 residual_kernel_size: int = 3, dilation_base: int = 2, skip: str = "1x1", causal: bool = False,
                 pad_mode: str = "constant", trim_right_ratio: float = 1.0,
                 final_activation: tp.Optional[str] = None, final_activation_params: tp.Optional[dict] = None,
                 act_all: bool = False, expansion: int = 1, groups: int = -1, bias: bool = True,
                 res_scale: tp.Optional[float] = None, wav_std: float = 0.1122080159, zero_init: bool = True,
                 inout_norm: bool = True):
        super().__init__()
        _check_block_options(skip, act_all, expansion, groups, activation, activation_params, causal, pad_mode,
                             dilation_base)
        if channels != 1:
            raise NotImplementedError("mono audio only (channels_audio=1)")
        if res_scale is None or not inout_norm or not bias:
            raise NotImplementedError("res_scale=None / inout_norm=False / bias=False are not on the hot path")
        if final_activation not in (None, "Tanh"):
            raise NotImplementedError("final_activation must be Tanh or None")
        self.dimension = dimension
        self.channels = channels
        self.n_filters = n_filters
        self.ratios = list(ratios)
        self.n_residual_layers = n_residual_layers
        self.hop_length = np.prod(self.ratios)
        self.wav_std = wav_std
        self.res_scale = res_scale
        self.final_activation = final_activation

        mult = int(2 ** len(self.ratios))
        model: tp.List[nn.Module] = [
            SConv1d(dimension, mult * n_filters, 1, norm=norm, norm_kwargs=norm_params, bias=False),
            SConv1d(mult * n_filters, mult * n_filters, kernel_size, groups=mult * n_filters, norm=norm,
                    norm_kwargs=norm_params, causal=causal, pad_mode=pad_mode, bias=bias)]
        stage_scale = (1 + n_residual_layers * res_scale ** 2) ** -0.5
        self._stage_slots = []
        for i, ratio in enumerate(self.ratios):
            first = len(model)
            model += [
                _Placeholder("Scale", stage_scale) if i > 0 else _Placeholder("Identity"),
                _Placeholder("ELU"),
                SConvTranspose1d(mult * n_filters, mult * n_filters, kernel_size=ratio * 2, stride=ratio,
                                 groups=mult * n_filters, norm=norm, norm_kwargs=norm_params, causal=causal,
                                 trim_right_ratio=trim_right_ratio, bias=False, nonlinearity="relu"),
                SConv1d(mult * n_filters, mult * n_filters // 2, 1, norm=norm, norm_kwargs=norm_params, bias=bias)]
            for j in range(n_residual_layers):
                model.append(SEANetResnetBlock(
                    mult * n_filters // 2, kernel_size=residual_kernel_size, dilations=[dilation_base ** j, 1],
                    activation=activation, activation_params=activation_params, norm=norm, norm_params=norm_params,
                    causal=causal, pad_mode=pad_mode, skip=skip, act_all=act_all, expansion=expansion,
                    groups=groups, bias=bias, res_scale=res_scale, idx=j, zero_init=zero_init))
            self._stage_slots.append((first, ratio))
            mult //= 2
        self._post_slot = len(model) + 2
        model += [
            _Placeholder("Scale", stage_scale),
            _Placeholder("ELU"),
            SConv1d(n_filters, channels, last_kernel_size, norm=norm, norm_kwargs=norm_params, causal=causal,
                    pad_mode=pad_mode, bias=bias, nonlinearity="relu"),
            _Placeholder("Scale", wav_std)]
        if final_activation is not None:
            model.append(_Placeholder(final_activation))
        self.model = nn.Sequential(*model)

    def build_spec(self, dev, streaming_variant: bool = False) -> engine.DecoderSpec:
        m = self.model
        stage_scale = (1 + self.n_residual_layers * self.res_scale ** 2) ** -0.5
        pre_pw, _ = _pw_dev(m[0], dev)
        pre_dw, pre_b = _dw_dev(m[1], dev)
        stages = []
        for i, (first, ratio) in enumerate(self._stage_slots):
            tr_w, _ = _dw_dev(m[first + 2], dev)
            pw_wt, pw_b = _pw_dev(m[first + 3], dev)
            blocks = [m[first + 4 + j].spec(dev, 1.0 if streaming_variant else None)
                      for j in range(self.n_residual_layers)]
            stages.append(engine.DecStageSpec(stage_scale if i > 0 else 1.0, tr_w, ratio, pw_wt, pw_b, blocks))
        post = m[self._post_slot].conv.conv
        w = post.effective_weight()[0].contiguous()
        b = post.effective_bias()
        if streaming_variant:                      # streaming.py:609-617: weight only
            w = w * self.wav_std
            out_scale = 1.0
        else:
            out_scale = self.wav_std               # seanet.py:464-466: conv AND bias
        return engine.DecoderSpec(pre_pw, pre_dw, pre_b, stages, stage_scale, w.to(dev),
                                  None if b is None else b.to(dev), out_scale, self.final_activation == "Tanh")

    def forward(self, z: Tensor) -> Tensor:
        return engine.run_decoder(self.plan(z.device), z, opts=self.exec_options)
