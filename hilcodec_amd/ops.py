"""PyTorch-ROCm custom ops (`torch.ops.hilcodec.*`) over the C ABI of `include/hilcodec_amd.h`.

Every entry point of the hot path is registered with the dispatcher through `torch.library`:
  * schema + a CUDA(HIP) implementation = torch tensors -> device pointers -> the hand-written gfx950 kernels
    (ctypes launch on the current HIP stream; PyTorch supplies memory and the stream, no arithmetic);
  * a CPU implementation that raises (there is deliberately no fallback path);
  * a fake (meta) implementation = the op's shape function,
so the module classes are ordinary traceable `nn.Module`s like the reference's (`torch.compile(fullgraph=True)`,
`torch.export`, HIP-graph capture all see plain ops).  All ops are functional (outputs are fresh tensors) except where
a caller hands in persistent streaming state to be overwritten (`hist_out` arguments, declared as mutated).

The public functions below are what `engine.py` and the module classes call; they only normalise arguments and
dispatch through `torch.ops.hilcodec`.  All tensors must be fp32 (indices int64), contiguous, on a GPU."""
from __future__ import annotations

import contextlib
import contextvars
import numbers
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from ._lib import check, lib

_LIB = torch.library.Library("hilcodec", "DEF")


def _ptr(t: Optional[Tensor], dtype=torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("hilcodec_amd ops need GPU tensors (no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("expected a contiguous tensor")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _new(ref: Tensor, *shape, dtype=torch.float32) -> Tensor:
    return torch.empty(shape, device=ref.device, dtype=dtype)


class LaunchTimer:
    """Optional per-launch HIP-event timing on the stream the kernels are launched on (bench.py).
    `work` is the algorithmic work of the launch (flops for the GEMMs, bytes for the HBM-bound ops)."""

    def __init__(self):
        self.records = []          # (kind, work, start_event, end_event, tag)

    def totals(self):
        out = {}
        for kind, work, e0, e1, _tag in self.records:
            t = out.setdefault(kind, [0, 0.0, 0.0])
            t[0] += 1
            t[1] += float(work)
            t[2] += e0.elapsed_time(e1) * 1e-3
        return out                 # kind -> [launches, work, seconds]


_TIMER: contextvars.ContextVar = contextvars.ContextVar("hilcodec_launch_timer", default=None)


@contextlib.contextmanager
def timed_launches(timer: Optional[LaunchTimer] = None):
    """`with ops.timed_launches() as t:` — every launch issued by THIS context (thread / task) inside the block is
    bracketed by HIP events on its launch stream and recorded in `t`.  A context variable, not a module global: another
    thread's launches are neither timed nor slowed down."""
    t = timer if timer is not None else LaunchTimer()
    token = _TIMER.set(t)
    try:
        yield t
    finally:
        _TIMER.reset(token)


class _timed:
    def __init__(self, kind: str, work: float, tag: str = ""):
        self.kind, self.work, self.tag = kind, work, tag
        self.timer = _TIMER.get()

    def __enter__(self):
        if self.timer is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.e1.record()
            self.timer.records.append((self.kind, self.work, self.e0, self.e1, self.tag))
        return False


def _no_cpu(*_a, **_k):
    raise RuntimeError("hilcodec_amd ops need GPU tensors (no CPU fallback)")


def _register(name: str, schema: str, impl, fake) -> None:
    _LIB.define(f"{name}{schema}")
    _LIB.impl(name, impl, "CUDA")
    _LIB.impl(name, _no_cpu, "CPU")
    torch.library.register_fake(f"hilcodec::{name}", fake, lib=_LIB)


# ======================================================================================================
# GEMM family
# ======================================================================================================
def _pw_conv(x, wt, bias, res, in_scale, in_elu, out_scale):
    B, K, T = x.shape
    M = wt.shape[1]
    if wt.shape[0] != K:
        raise RuntimeError(f"pw_conv: weight rows {wt.shape[0]} != input channels {K}")
    y = _new(x, B, M, T)
    with _timed("pw_conv", 2.0 * B * T * K * M, f"K{K} M{M} T{T}"):
        check(lib.hilc_pw_conv(_ptr(x), _ptr(wt), _ptr(bias), _ptr(res), _ptr(y), B, K, M, T,
                               in_scale, int(in_elu), out_scale, _stream()), "hilc_pw_conv")
    return y


_register("pw_conv", "(Tensor x, Tensor wt, Tensor? bias, Tensor? res, float in_scale, bool in_elu, float out_scale)"
          " -> Tensor", _pw_conv,
          lambda x, wt, bias, res, in_scale, in_elu, out_scale: x.new_empty(x.shape[0], wt.shape[1], x.shape[2]))


def _dws_conv(x, wt, dw_w, dw_b, res, stride, in_scale, in_elu, out_scale, out_elu):
    B, K, T = x.shape
    M, k = wt.shape[1], dw_w.shape[1]
    To = (T + stride - 1) // stride
    y = _new(x, B, M, To)
    with _timed("dws_conv", 2.0 * B * T * K * M, f"K{K} M{M} T{T} k{k} s{stride}"):
        check(lib.hilc_dws_conv(_ptr(x), _ptr(wt), _ptr(dw_w), _ptr(dw_b), _ptr(res), _ptr(y), B, K, M, T, k,
                                stride, in_scale, int(in_elu), out_scale, int(out_elu), _stream()), "hilc_dws_conv")
    return y


_register("dws_conv", "(Tensor x, Tensor wt, Tensor dw_w, Tensor? dw_b, Tensor? res, int stride, float in_scale, "
          "bool in_elu, float out_scale, bool out_elu) -> Tensor", _dws_conv,
          lambda x, wt, dw_w, dw_b, res, stride, in_scale, in_elu, out_scale, out_elu:
          x.new_empty(x.shape[0], wt.shape[1], (x.shape[2] + stride - 1) // stride))


def _dws_conv_stream(x, wt, dw_w, dw_b, hist, hist_out, res, stride, in_scale, in_elu, out_scale, out_elu):
    B, K, T = x.shape
    M, k = wt.shape[1], dw_w.shape[1]
    pad = k - stride
    for h in (hist, hist_out):
        if h is not None and tuple(h.shape) != (B, M, pad):
            raise RuntimeError(f"cache must be [{B},{M},{pad}], got {tuple(h.shape)}")
    y = _new(x, B, M, T // stride)
    with _timed("dws_conv", 2.0 * B * T * K * M, f"K{K} M{M} T{T} k{k} s{stride} stream"):
        check(lib.hilc_dws_conv_stream(_ptr(x), _ptr(wt), _ptr(dw_w), _ptr(dw_b), _ptr(hist), _ptr(hist_out),
                                       _ptr(res), _ptr(y), B, K, M, T, k, stride, in_scale, int(in_elu), out_scale,
                                       int(out_elu), _stream()), "hilc_dws_conv_stream")
    return y


_register("dws_conv_stream", "(Tensor x, Tensor wt, Tensor dw_w, Tensor? dw_b, Tensor? hist, Tensor(a!) hist_out, "
          "Tensor? res, int stride, float in_scale, bool in_elu, float out_scale, bool out_elu) -> Tensor",
          _dws_conv_stream,
          lambda x, wt, dw_w, dw_b, hist, hist_out, res, stride, in_scale, in_elu, out_scale, out_elu:
          x.new_empty(x.shape[0], wt.shape[1], x.shape[2] // stride))


def _up_conv_expand_taps(tr_w, stride):
    K = tr_w.shape[0]
    out = _new(tr_w, K * stride * 8)
    check(lib.hilc_up_conv_expand_taps(_ptr(tr_w), _ptr(out), K, stride, _stream()), "hilc_up_conv_expand_taps")
    return out


_register("up_conv_expand_taps", "(Tensor tr_w, int stride) -> Tensor", _up_conv_expand_taps,
          lambda tr_w, stride: tr_w.new_empty(tr_w.shape[0] * stride * 8))


def _up_conv(x, hist, hist_out, tr_w, taps, wt, bias, stride, in_scale, in_elu):
    B, K, Tin = x.shape
    M = wt.shape[1]
    for h in (hist, hist_out):
        if h is not None and h.numel() != B * K:
            raise RuntimeError(f"cache must be [{B},{K},1], got {tuple(h.shape)}")
    y = _new(x, B, M, Tin * stride)
    with _timed("up_conv", 2.0 * B * Tin * stride * K * M,
                f"K{K} M{M} Tin{Tin} r{stride}" + (" stream" if hist is not None or hist_out is not None else "")):
        check(lib.hilc_up_conv_expanded(_ptr(x), _ptr(hist), _ptr(hist_out), _ptr(tr_w), _ptr(taps), _ptr(wt),
                                        _ptr(bias), _ptr(y), B, K, M, Tin, stride, in_scale, int(in_elu), _stream()),
              "hilc_up_conv")
    return y


_register("up_conv", "(Tensor x, Tensor? hist, Tensor(a!)? hist_out, Tensor tr_w, Tensor? taps, Tensor wt, Tensor? bias, "
          "int stride, float in_scale, bool in_elu) -> Tensor", _up_conv,
          lambda x, hist, hist_out, tr_w, taps, wt, bias, stride, in_scale, in_elu:
          x.new_empty(x.shape[0], wt.shape[1], x.shape[2] * stride))


def _resblock_pack(wt):
    Cc = wt.shape[0]
    out = _new(wt, Cc * Cc)
    check(lib.hilc_resblock_pack_weights(_ptr(wt), _ptr(out), Cc, _stream()), "hilc_resblock_pack_weights")
    return out


_register("resblock_pack", "(Tensor wt) -> Tensor", _resblock_pack, lambda wt: wt.new_empty(wt.shape[0] * wt.shape[0]))

_SCHED = {}


class SchedWorkspace:
    """Ticket words of the residual-block kernel's tile scheduler (two ints per launch, zero at launch, re-armed by the
    kernel itself) owned by ONE schedule object (graph_step.GraphedHop / PipelinedHop): allocated outside graph capture,
    one slot per launch in call order, so that (i) no allocation or memset node lands inside a capture, (ii) launches that
    run concurrently — the two branches of a pipelined hop, two graphs replayed on different streams — never share a
    ticket (shared words would hand out duplicate or skipped tiles)."""

    def __init__(self, device, slots: int = 64):
        self.words = torch.zeros(2 * slots, dtype=torch.int32, device=device)
        self.slots, self.next = slots, 0

    def take(self) -> Tensor:
        if self.next >= self.slots:
            raise RuntimeError("SchedWorkspace: more residual-block launches than slots")
        w = self.words[2 * self.next:2 * self.next + 2]
        self.next += 1
        return w


_SCHED_WS: contextvars.ContextVar = contextvars.ContextVar("hilcodec_sched_workspace", default=None)


@contextlib.contextmanager
def sched_workspace(ws: Optional[SchedWorkspace]):
    """the residual-block launches of this block take their ticket words from `ws`, slot 0 first"""
    if ws is not None:
        ws.next = 0
    token = _SCHED_WS.set(ws)
    try:
        yield ws
    finally:
        _SCHED_WS.reset(token)


def _sched_buffer(device) -> Tensor:
    """Two zeroed ints for the residual-block kernel's ticket scheduler; the kernel re-arms them itself, so a buffer is
    written by the host only once.  Inside `sched_workspace(...)`: the owner's next slot.  Otherwise one buffer per
    (device, stream): launches on one stream are ordered, so they may share; eager launches only — a captured graph must
    bring its own workspace (allocating here during capture would put a memset node and the buffer into that graph)."""
    ws = _SCHED_WS.get()
    if ws is not None:
        return ws.take()
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("residual-block launch inside a graph capture without ops.sched_workspace(...): "
                           "the ticket words must be allocated by the owner of the graph, outside the capture")
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _SCHED.get(key)
    if buf is None:
        buf = torch.zeros(2, dtype=torch.int32, device=device)
        _SCHED[key] = buf
    return buf


def _resblock(x, w1p, dw1_w, dw1_b, w2p, dw2_w, dw2_b, hist1, hist2, hist1_out, hist2_out, pre_scale, out_scale):
    B, Cc, T = x.shape
    streaming = hist1_out is not None or hist1 is not None
    for h in (hist1, hist2, hist1_out, hist2_out):
        if h is not None and tuple(h.shape) != (B, Cc, 4):
            raise RuntimeError(f"resblock caches must be [{B},{Cc},4], got {tuple(h.shape)}")
    y = torch.empty_like(x)
    with _timed("resblock", 4.0 * B * T * Cc * Cc, f"C{Cc} T{T}" + (" stream" if streaming else "")):
        check(lib.hilc_resblock_balanced(_ptr(x), _ptr(w1p), _ptr(dw1_w), _ptr(dw1_b), _ptr(w2p), _ptr(dw2_w),
                                         _ptr(dw2_b), _ptr(hist1), _ptr(hist2), _ptr(hist1_out), _ptr(hist2_out),
                                         _ptr(y), _ptr(_sched_buffer(x.device), torch.int32), int(streaming), B, Cc, T,
                                         pre_scale, out_scale, _stream()), "hilc_resblock")
    return y


_register("resblock", "(Tensor x, Tensor w1p, Tensor dw1_w, Tensor dw1_b, Tensor w2p, Tensor dw2_w, Tensor dw2_b, "
          "Tensor? hist1, Tensor? hist2, Tensor(a!)? hist1_out, Tensor(b!)? hist2_out, float pre_scale, float out_scale)"
          " -> Tensor", _resblock,
          lambda x, w1p, dw1_w, dw1_b, w2p, dw2_w, dw2_b, hist1, hist2, hist1_out, hist2_out, pre_scale, out_scale:
          torch.empty_like(x))


def _resblock_chain(x, params, hist_in, hist_out, pre_scales, out_scales):
    from ._lib import ResblockParams
    B, Cc, T = x.shape
    n = len(pre_scales)
    streaming = len(hist_in) > 0                   # no caches: the offline causal model (zero padding in front of every clip)
    if len(params) != 6 * n or len(out_scales) != n or (streaming and (len(hist_in) != 2 * n or len(hist_out) != 2 * n)) or \
            (not streaming and len(hist_out) != 0):
        raise RuntimeError("resblock_chain: 6 parameter tensors per block, and (streaming) 2 caches in and 2 caches out per block")
    for h in list(hist_in) + list(hist_out):
        if tuple(h.shape) != (B, Cc, 4):
            raise RuntimeError(f"resblock caches must be [{B},{Cc},4], got {tuple(h.shape)}")
    blocks = (ResblockParams * n)()
    for i in range(n):
        w1p, d1w, d1b, w2p, d2w, d2b = params[6 * i:6 * i + 6]
        h = [_ptr(t) for t in (hist_in[2 * i], hist_in[2 * i + 1], hist_out[2 * i], hist_out[2 * i + 1])] if streaming else [None] * 4
        blocks[i] = ResblockParams(_ptr(w1p), _ptr(d1w), _ptr(d1b), _ptr(w2p), _ptr(d2w), _ptr(d2b), h[0], h[1], h[2], h[3],
                                   float(pre_scales[i]), float(out_scales[i]))
    y = torch.empty_like(x)
    import ctypes
    with _timed("resblock", 4.0 * n * B * T * Cc * Cc, f"C{Cc} T{T}" + (" stream" if streaming else "") + f" chain x{n}"):
        check(lib.hilc_resblock_chain(_ptr(x), _ptr(y), ctypes.cast(blocks, ctypes.c_void_p), n, int(streaming), B, Cc, T, _stream()),
              "hilc_resblock_chain")
    return y


_register("resblock_chain", "(Tensor x, Tensor[] params, Tensor[] hist_in, Tensor(a!)[] hist_out, float[] pre_scales, "
          "float[] out_scales) -> Tensor", _resblock_chain,
          lambda x, params, hist_in, hist_out, pre_scales, out_scales: torch.empty_like(x))


def _encoder_stage(x, params, hist_in, hist_out, pre_scales, out_scales, w_lo, w_hi, ddw_w, ddw_b, dhist, dhist_out, dres, in_scale,
                   stride):
    import ctypes
    from ._lib import DownParams, ResblockParams
    B, Cc, T = x.shape
    n = len(pre_scales)
    streaming = dhist_out is not None
    if len(params) != 6 * n or len(out_scales) != n or len(hist_in) != (2 * n if streaming else 0) or len(hist_out) != len(hist_in):
        raise RuntimeError("encoder_stage: 6 parameter tensors per block, and (streaming) 2 caches in and 2 caches out per block")
    if T % stride != 0:
        raise RuntimeError("encoder_stage: T must be a multiple of the stride")
    blocks = (ResblockParams * n)()
    for i in range(n):
        w1p, d1w, d1b, w2p, d2w, d2b = params[6 * i:6 * i + 6]
        h = [_ptr(t) for t in (hist_in[2 * i], hist_in[2 * i + 1], hist_out[2 * i], hist_out[2 * i + 1])] if streaming else [None] * 4
        blocks[i] = ResblockParams(_ptr(w1p), _ptr(d1w), _ptr(d1b), _ptr(w2p), _ptr(d2w), _ptr(d2b), h[0], h[1], h[2], h[3],
                                   float(pre_scales[i]), float(out_scales[i]))
    To = T // stride
    for t, shape in ((dhist, (B, 2 * Cc, stride)), (dhist_out, (B, 2 * Cc, stride)), (dres, (B, 2 * Cc, To))):
        if t is not None and tuple(t.shape) != shape:
            raise RuntimeError(f"encoder_stage: expected {shape}, got {tuple(t.shape)}")
    y = torch.empty(B, 2 * Cc, To, device=x.device, dtype=torch.float32)
    down = DownParams(_ptr(w_lo), _ptr(w_hi), _ptr(ddw_w), _ptr(ddw_b), _ptr(dhist), _ptr(dhist_out), _ptr(dres), _ptr(y),
                      float(in_scale), int(stride))
    tag = f"C{Cc} T{T}" + (" stream" if streaming else "")
    with _timed("resblock", 4.0 * n * B * T * Cc * Cc + 4.0 * B * T * Cc * Cc, tag + f" stage x{n} + down s{stride}"):
        check(lib.hilc_encoder_stage(_ptr(x), ctypes.cast(blocks, ctypes.c_void_p), n, ctypes.cast(ctypes.pointer(down), ctypes.c_void_p),
                                     int(streaming), B, Cc, T, _stream()), "hilc_encoder_stage")
    return y


_register("encoder_stage", "(Tensor x, Tensor[] params, Tensor[] hist_in, Tensor(a!)[] hist_out, float[] pre_scales, float[] out_scales, "
          "Tensor w_lo, Tensor w_hi, Tensor ddw_w, Tensor ddw_b, Tensor? dhist, Tensor(b!)? dhist_out, Tensor? dres, float in_scale, "
          "int stride) -> Tensor", _encoder_stage,
          lambda x, params, hist_in, hist_out, pre_scales, out_scales, w_lo, w_hi, ddw_w, ddw_b, dhist, dhist_out, dres, in_scale, stride:
          x.new_empty(x.shape[0], 2 * x.shape[1], x.shape[2] // stride))


def _decoder_stage(xin, tr_w, w_lo, w_hi, bias, uhist, uhist_out, in_scale, stride, params, hist_in, hist_out, pre_scales, out_scales):
    import ctypes
    from ._lib import ResblockParams, UpParams
    B, K2, Tin = xin.shape
    Cc, T = K2 // 2, Tin * stride
    n = len(pre_scales)
    streaming = len(hist_in) > 0 or uhist_out is not None        # no caches at all: the offline causal model
    if len(params) != 6 * n or len(out_scales) != n or len(hist_in) != (2 * n if streaming else 0) or len(hist_out) != len(hist_in):
        raise RuntimeError("decoder_stage: 6 parameter tensors per block, and (streaming) 2 caches in and 2 caches out per block")
    for h in (uhist, uhist_out):
        if h is not None and h.numel() != B * K2:
            raise RuntimeError(f"decoder_stage: the up-sampling cache must be [{B},{K2},1], got {tuple(h.shape)}")
    blocks = (ResblockParams * n)()
    for i in range(n):
        w1p, d1w, d1b, w2p, d2w, d2b = params[6 * i:6 * i + 6]
        h = [_ptr(t) for t in (hist_in[2 * i], hist_in[2 * i + 1], hist_out[2 * i], hist_out[2 * i + 1])] if streaming else [None] * 4
        blocks[i] = ResblockParams(_ptr(w1p), _ptr(d1w), _ptr(d1b), _ptr(w2p), _ptr(d2w), _ptr(d2b), h[0], h[1], h[2], h[3],
                                   float(pre_scales[i]), float(out_scales[i]))
    up = UpParams(_ptr(xin), _ptr(tr_w), _ptr(w_lo), _ptr(w_hi), _ptr(bias), _ptr(uhist), _ptr(uhist_out), float(in_scale), int(stride))
    y = torch.empty(B, Cc, T, device=xin.device, dtype=torch.float32)
    with _timed("resblock", 4.0 * n * B * T * Cc * Cc + 4.0 * B * T * Cc * Cc, f"C{Cc} T{T}" + (" stream" if streaming else "") + f" up r{stride} + stage x{n}"):
        check(lib.hilc_decoder_stage(ctypes.cast(ctypes.pointer(up), ctypes.c_void_p), ctypes.cast(blocks, ctypes.c_void_p), n, _ptr(y),
                                     int(streaming), B, Cc, T, _stream()), "hilc_decoder_stage")
    return y


_register("decoder_stage", "(Tensor xin, Tensor tr_w, Tensor w_lo, Tensor w_hi, Tensor? bias, Tensor? uhist, Tensor(a!)? uhist_out, "
          "float in_scale, int stride, Tensor[] params, Tensor[] hist_in, Tensor(b!)[] hist_out, float[] pre_scales, float[] out_scales) "
          "-> Tensor", _decoder_stage,
          lambda xin, tr_w, w_lo, w_hi, bias, uhist, uhist_out, in_scale, stride, params, hist_in, hist_out, pre_scales, out_scales:
          xin.new_empty(xin.shape[0], xin.shape[1] // 2, xin.shape[2] * stride))


def _encoder_stage0(wav, wav_hist, dft_packed, nyq_sin, pw_packed, bias, pre_w, pre_b, pre_in_scale, mean, std, normalize, out_scale, params, hist_in,
                    hist_out, pre_scales, out_scales, w_lo, w_hi, ddw_w, ddw_b, dhist, dhist_out, dres, in_scale, stride):
    import ctypes
    from ._lib import DownParams, ResblockParams, Spec0Params
    B, one, T = wav.shape
    Cc, k = pre_w.shape
    n = len(pre_scales)
    streaming = dhist_out is not None
    if one != 1 or Cc != 64 or len(params) != 6 * n or len(out_scales) != n or T % stride != 0:
        raise RuntimeError("encoder_stage0: wav [B,1,T], a 64-channel first conv, 6 parameter tensors per block, T a multiple of the stride")
    if len(hist_in) != (2 * n if streaming else 0) or len(hist_out) != len(hist_in):
        raise RuntimeError("encoder_stage0: (streaming) 2 caches in and 2 caches out per block")
    if wav_hist is not None and (wav_hist.shape[0] != B or wav_hist.numel() // B < 63):
        raise RuntimeError(f"encoder_stage0: the waveform history must be [{B},1,>=63], got {tuple(wav_hist.shape)}")
    blocks = (ResblockParams * n)()
    for i in range(n):
        w1p, d1w, d1b, w2p, d2w, d2b = params[6 * i:6 * i + 6]
        h = [_ptr(t) for t in (hist_in[2 * i], hist_in[2 * i + 1], hist_out[2 * i], hist_out[2 * i + 1])] if streaming else [None] * 4
        blocks[i] = ResblockParams(_ptr(w1p), _ptr(d1w), _ptr(d1b), _ptr(w2p), _ptr(d2w), _ptr(d2b), h[0], h[1], h[2], h[3],
                                   float(pre_scales[i]), float(out_scales[i]))
    To = T // stride
    for t, shape in ((dhist, (B, 2 * Cc, stride)), (dhist_out, (B, 2 * Cc, stride)), (dres, (B, 2 * Cc, To))):
        if t is not None and tuple(t.shape) != shape:
            raise RuntimeError(f"encoder_stage0: expected {shape}, got {tuple(t.shape)}")
    y = torch.empty(B, 2 * Cc, To, device=wav.device, dtype=torch.float32)
    wh = wav_hist.reshape(B, -1) if wav_hist is not None else None
    spec = Spec0Params(_ptr(wav), _ptr(dft_packed), _ptr(nyq_sin), _ptr(pw_packed), _ptr(bias), _ptr(pre_w), _ptr(pre_b),
                       _ptr(wh), int(wh.shape[1]) if wh is not None else 0, float(pre_in_scale),
                       float(mean), float(std), float(out_scale), int(normalize), 64, 1, int(pre_w.shape[1]))
    down = DownParams(_ptr(w_lo), _ptr(w_hi), _ptr(ddw_w), _ptr(ddw_b), _ptr(dhist), _ptr(dhist_out), _ptr(dres), _ptr(y), float(in_scale), int(stride))
    work = 2.0 * B * T * 64 * (64 + 1 + 32 + 1) + 2.0 * B * T * Cc * k + 4.0 * n * B * T * Cc * Cc + 4.0 * B * T * Cc * Cc
    with _timed("resblock", work, f"C{Cc} T{T}" + (" stream" if streaming else "") + f" conv_pre + spec N64 + stage x{n} + down s{stride}"):
        check(lib.hilc_encoder_stage0(ctypes.cast(ctypes.pointer(spec), ctypes.c_void_p), ctypes.cast(blocks, ctypes.c_void_p), n,
                                      ctypes.cast(ctypes.pointer(down), ctypes.c_void_p), int(streaming), B, T, _stream()), "hilc_encoder_stage0")
    return y


_register("encoder_stage0", "(Tensor wav, Tensor? wav_hist, Tensor dft_packed, Tensor nyq_sin, Tensor pw_packed, Tensor? bias, Tensor pre_w, Tensor? pre_b, "
          "float pre_in_scale, float mean, float std, int normalize, float out_scale, Tensor[] params, Tensor[] hist_in, Tensor(a!)[] hist_out, "
          "float[] pre_scales, float[] out_scales, Tensor w_lo, Tensor w_hi, Tensor ddw_w, Tensor ddw_b, Tensor? dhist, Tensor(b!)? dhist_out, "
          "Tensor? dres, float in_scale, int stride) -> Tensor", _encoder_stage0,
          lambda wav, wav_hist, dft_packed, nyq_sin, pw_packed, bias, pre_w, pre_b, pre_in_scale, mean, std, normalize, out_scale, params, hist_in,
          hist_out, pre_scales, out_scales, w_lo, w_hi, ddw_w, ddw_b, dhist, dhist_out, dres, in_scale, stride:
          wav.new_empty(wav.shape[0], 2 * pre_w.shape[0], wav.shape[2] // stride))


def _decoder_stage_post(xin, tr_w, w_lo, w_hi, bias, uhist, uhist_out, in_scale, stride, params, hist_in, hist_out, pre_scales, out_scales,
                        post_w, post_b, post_hist, post_hist_out, post_in_scale, post_out_scale, do_tanh):
    import ctypes
    from ._lib import PostParams, ResblockParams, UpParams
    B, K2, Tin = xin.shape
    Cc, T = K2 // 2, Tin * stride
    n = len(pre_scales)
    streaming = len(hist_in) > 0 or uhist_out is not None or post_hist_out is not None      # no caches at all: the offline causal model
    if len(params) != 6 * n or len(out_scales) != n or len(hist_in) != (2 * n if streaming else 0) or len(hist_out) != len(hist_in):
        raise RuntimeError("decoder_stage_post: 6 parameter tensors per block, and (streaming) 2 caches in and 2 caches out per block")
    if post_w.dim() != 2 or post_w.shape[0] != Cc:
        raise RuntimeError(f"decoder_stage_post: the closing conv's taps must be [{Cc}, k], got {tuple(post_w.shape)}")
    for h in (uhist, uhist_out):
        if h is not None and h.numel() != B * K2:
            raise RuntimeError(f"decoder_stage_post: the up-sampling cache must be [{B},{K2},1], got {tuple(h.shape)}")
    for h in (post_hist, post_hist_out):
        if h is not None and tuple(h.shape) != (B, Cc, post_w.shape[1] - 1):
            raise RuntimeError(f"decoder_stage_post: the closing conv's cache must be [{B},{Cc},{post_w.shape[1] - 1}], got {tuple(h.shape)}")
    blocks = (ResblockParams * n)()
    for i in range(n):
        w1p, d1w, d1b, w2p, d2w, d2b = params[6 * i:6 * i + 6]
        h = [_ptr(t) for t in (hist_in[2 * i], hist_in[2 * i + 1], hist_out[2 * i], hist_out[2 * i + 1])] if streaming else [None] * 4
        blocks[i] = ResblockParams(_ptr(w1p), _ptr(d1w), _ptr(d1b), _ptr(w2p), _ptr(d2w), _ptr(d2b), h[0], h[1], h[2], h[3],
                                   float(pre_scales[i]), float(out_scales[i]))
    up = UpParams(_ptr(xin), _ptr(tr_w), _ptr(w_lo), _ptr(w_hi), _ptr(bias), _ptr(uhist), _ptr(uhist_out), float(in_scale), int(stride))
    wav = torch.empty(B, 1, T, device=xin.device, dtype=torch.float32)
    post = PostParams(_ptr(post_w), _ptr(post_b), _ptr(wav), _ptr(post_hist), _ptr(post_hist_out), float(post_in_scale), float(post_out_scale),
                      int(do_tanh), int(post_w.shape[1]))
    with _timed("resblock", 4.0 * n * B * T * Cc * Cc + 4.0 * B * T * Cc * Cc, f"C{Cc} T{T}" + (" stream" if streaming else "") + f" up r{stride} + stage x{n} + conv_post"):
        check(lib.hilc_decoder_stage_post(ctypes.cast(ctypes.pointer(up), ctypes.c_void_p), ctypes.cast(blocks, ctypes.c_void_p), n,
                                          ctypes.cast(ctypes.pointer(post), ctypes.c_void_p), int(streaming), B, Cc, T, _stream()), "hilc_decoder_stage_post")
    return wav


_register("decoder_stage_post", "(Tensor xin, Tensor tr_w, Tensor w_lo, Tensor w_hi, Tensor? bias, Tensor? uhist, Tensor(a!)? uhist_out, float in_scale, "
          "int stride, Tensor[] params, Tensor[] hist_in, Tensor(b!)[] hist_out, float[] pre_scales, float[] out_scales, Tensor post_w, Tensor? post_b, "
          "Tensor? post_hist, Tensor(c!)? post_hist_out, float post_in_scale, float post_out_scale, bool do_tanh) -> Tensor",
          _decoder_stage_post,
          lambda xin, tr_w, w_lo, w_hi, bias, uhist, uhist_out, in_scale, stride, params, hist_in, hist_out, pre_scales, out_scales, post_w, post_b,
          post_hist, post_hist_out, post_in_scale, post_out_scale, do_tanh: xin.new_empty(xin.shape[0], 1, xin.shape[2] * stride))


def _resblock_pack_rc(wt, row_classes):
    Cc = wt.shape[0]
    out = torch.empty(Cc * Cc, device=wt.device, dtype=torch.float32)
    check(lib.hilc_resblock_pack_weights_rc(_ptr(wt), _ptr(out), Cc, int(row_classes), _stream()), "hilc_resblock_pack_weights_rc")
    return out


_register("resblock_pack_rc", "(Tensor wt, int row_classes) -> Tensor", _resblock_pack_rc,
          lambda wt, row_classes: wt.new_empty(wt.shape[0] * wt.shape[0]))


# ======================================================================================================
# HBM-bound element-wise ops
# ======================================================================================================
def _dw_conv(x, hist, w, bias, res, hist_out, stride, in_scale, in_elu, out_scale, out_elu):
    B, Cc, T = x.shape
    k = w.shape[1]
    To = (T + stride - 1) // stride
    y = _new(x, B, Cc, To)
    with _timed("dw_conv", 4.0 * B * Cc * (T + To * (2 if res is not None else 1)), f"C{Cc} T{T} k{k} s{stride}"):
        check(lib.hilc_dw_conv(_ptr(x), _ptr(hist), _ptr(w), _ptr(bias), _ptr(res), _ptr(y), _ptr(hist_out),
                               B, Cc, T, k, stride, in_scale, int(in_elu), out_scale, int(out_elu), _stream()),
              "hilc_dw_conv")
    return y


_register("dw_conv", "(Tensor x, Tensor? hist, Tensor w, Tensor? bias, Tensor? res, Tensor(a!)? hist_out, int stride, "
          "float in_scale, bool in_elu, float out_scale, bool out_elu) -> Tensor", _dw_conv,
          lambda x, hist, w, bias, res, hist_out, stride, in_scale, in_elu, out_scale, out_elu:
          x.new_empty(x.shape[0], x.shape[1], (x.shape[2] + stride - 1) // stride))


def _dw_convtr(x, hist, w, hist_out, stride, in_scale, in_elu):
    B, Cc, T = x.shape
    if w.shape[1] != 2 * stride:
        raise RuntimeError("dw_convtr: kernel size must be 2 * stride")
    y = _new(x, B, Cc, T * stride)
    with _timed("dw_convtr", 4.0 * B * Cc * T * (1 + stride), f"C{Cc} T{T} r{stride}"):
        check(lib.hilc_dw_convtr(_ptr(x), _ptr(hist), _ptr(w), _ptr(y), _ptr(hist_out), B, Cc, T, stride,
                                 in_scale, int(in_elu), _stream()), "hilc_dw_convtr")
    return y


_register("dw_convtr", "(Tensor x, Tensor? hist, Tensor w, Tensor(a!)? hist_out, int stride, float in_scale, bool in_elu)"
          " -> Tensor", _dw_convtr,
          lambda x, hist, w, hist_out, stride, in_scale, in_elu: x.new_empty(x.shape[0], x.shape[1], x.shape[2] * stride))


def _conv_pre(wav, hist, w, bias, in_scale):
    B, one, T = wav.shape
    if one != 1:
        raise RuntimeError("conv_pre expects a [B,1,T] waveform")
    Cc, k = w.shape
    y = _new(wav, B, Cc, T)
    hl = hist.shape[-1] if hist is not None else 0
    with _timed("conv_pre", 4.0 * B * T * (1 + Cc)):
        check(lib.hilc_conv_pre(_ptr(wav), _ptr(hist), hl, _ptr(w), _ptr(bias), _ptr(y), B, Cc, T, k,
                                in_scale, _stream()), "hilc_conv_pre")
    return y


_register("conv_pre", "(Tensor wav, Tensor? hist, Tensor w, Tensor? bias, float in_scale) -> Tensor", _conv_pre,
          lambda wav, hist, w, bias, in_scale: wav.new_empty(wav.shape[0], w.shape[0], wav.shape[2]))


def _conv_post(x, hist, w, bias, hist_out, in_scale, in_elu, out_scale, do_tanh):
    B, Cc, T = x.shape
    k = w.shape[1]
    y = _new(x, B, 1, T)
    with _timed("conv_post", 4.0 * B * T * (1 + Cc)):
        check(lib.hilc_conv_post(_ptr(x), _ptr(hist), _ptr(w), _ptr(bias), _ptr(y), _ptr(hist_out), B, Cc, T, k,
                                 in_scale, int(in_elu), out_scale, int(do_tanh), _stream()), "hilc_conv_post")
    return y


_register("conv_post", "(Tensor x, Tensor? hist, Tensor w, Tensor? bias, Tensor(a!)? hist_out, float in_scale, "
          "bool in_elu, float out_scale, bool do_tanh) -> Tensor", _conv_post,
          lambda x, hist, w, bias, hist_out, in_scale, in_elu, out_scale, do_tanh: x.new_empty(x.shape[0], 1, x.shape[2]))


def _stft_logmag(wav, hist, basis_t, n_fft, hop, mean, std, normalize):
    B, one, T = wav.shape
    Tf = (T - 1) // hop + 1
    spec = _new(wav, B, n_fft // 2 + 1, Tf)
    hl = hist.shape[-1] if hist is not None else 0
    with _timed("stft", 2.0 * B * Tf * n_fft * (n_fft + 2), f"N{n_fft} hop{hop}"):
        check(lib.hilc_stft_logmag(_ptr(wav), _ptr(hist), hl, _ptr(basis_t), _ptr(spec), B, T, n_fft, hop,
                                   mean, std, normalize, _stream()), "hilc_stft_logmag")
    return spec


_register("stft_logmag", "(Tensor wav, Tensor? hist, Tensor basis_t, int n_fft, int hop, float mean, float std, "
          "int normalize) -> Tensor", _stft_logmag,
          lambda wav, hist, basis_t, n_fft, hop, mean, std, normalize:
          wav.new_empty(wav.shape[0], n_fft // 2 + 1, (wav.shape[2] - 1) // hop + 1))


def _spec_block_pack(w, n_fft, which):
    out = _new(w, lib.hilc_spec_block_packed_floats(n_fft, which))
    check(lib.hilc_spec_block_pack(_ptr(w), _ptr(out), w.shape[0], n_fft, which, _stream()), "hilc_spec_block_pack")
    return out


_register("spec_block_pack", "(Tensor w, int n_fft, int which) -> Tensor", _spec_block_pack,
          lambda w, n_fft, which: w.new_empty(lib.hilc_spec_block_packed_floats(n_fft, which)))


def _spec_block(wav, hist, dft_packed, nyq_sin, pw_packed, bias, x, n_fft, hop, mean, std, normalize, out_scale):
    B, one, T = wav.shape
    hl = hist.shape[-1] if hist is not None else 0
    Tf = (T - 1) // hop + 1
    if x is not None and tuple(x.shape) != (B, n_fft, Tf):
        raise RuntimeError(f"spec_block: x must be [{B},{n_fft},{Tf}], got {tuple(x.shape)}")
    y = torch.empty(B, n_fft, Tf, device=wav.device, dtype=torch.float32)
    with _timed("spec_block", 2.0 * B * Tf * n_fft * (n_fft + 1 + n_fft // 2 + 1), f"N{n_fft} hop{hop}"):
        check(lib.hilc_spec_block(_ptr(wav), _ptr(hist), hl, _ptr(dft_packed), _ptr(nyq_sin), _ptr(pw_packed), _ptr(bias), _ptr(x), _ptr(y),
                                  B, T, n_fft, hop, mean, std, normalize, out_scale, _stream()), "hilc_spec_block")
    return y


_register("spec_block", "(Tensor wav, Tensor? hist, Tensor dft_packed, Tensor nyq_sin, Tensor pw_packed, Tensor? bias, Tensor? x, "
          "int n_fft, int hop, float mean, float std, int normalize, float out_scale) -> Tensor", _spec_block,
          lambda wav, hist, dft_packed, nyq_sin, pw_packed, bias, x, n_fft, hop, mean, std, normalize, out_scale:
          wav.new_empty(wav.shape[0], n_fft, (wav.shape[2] - 1) // hop + 1))


def _spec_block_conv_pre(wav, hist, dft_packed, nyq_sin, pw_packed, bias, pre_w, pre_b, pre_in_scale, n_fft, hop, mean, std,
                         normalize, out_scale):
    B, one, T = wav.shape
    hl = hist.shape[-1] if hist is not None else 0
    Cc, k = pre_w.shape
    y = _new(wav, B, Cc, T)
    with _timed("spec_block", 2.0 * B * T * n_fft * (n_fft + 1 + n_fft // 2 + 1) + 2.0 * B * T * Cc * k,
                f"N{n_fft} hop{hop} +conv_pre"):
        check(lib.hilc_spec_block_conv_pre(_ptr(wav), _ptr(hist), hl, _ptr(dft_packed), _ptr(nyq_sin), _ptr(pw_packed), _ptr(bias),
                                           _ptr(pre_w), _ptr(pre_b), pre_in_scale, _ptr(y), B, T, n_fft, hop, k, mean, std,
                                           normalize, out_scale, _stream()), "hilc_spec_block_conv_pre")
    return y


_register("spec_block_conv_pre", "(Tensor wav, Tensor? hist, Tensor dft_packed, Tensor nyq_sin, Tensor pw_packed, Tensor? bias, "
          "Tensor pre_w, Tensor? pre_b, float pre_in_scale, int n_fft, int hop, float mean, float std, int normalize, "
          "float out_scale) -> Tensor", _spec_block_conv_pre,
          lambda wav, hist, dft_packed, nyq_sin, pw_packed, bias, pre_w, pre_b, pre_in_scale, n_fft, hop, mean, std, normalize,
          out_scale: wav.new_empty(wav.shape[0], pre_w.shape[0], wav.shape[2]))


def _tail(x, hist, out):
    B, Cc, T = x.shape
    pad = out.shape[-1]
    hl = hist.shape[-1] if hist is not None else 0
    check(lib.hilc_tail(_ptr(x), _ptr(hist), _ptr(out), B * Cc, T, pad, hl, _stream()), "hilc_tail")


_register("tail", "(Tensor x, Tensor? hist, Tensor(a!) out) -> ()", _tail, lambda x, hist, out: None)


def _l2norm(x, eps, scale, channel_last_out):
    B, Cc, T = x.shape
    y = _new(x, *((B, T, Cc) if channel_last_out else (B, Cc, T)))
    check(lib.hilc_l2norm(_ptr(x), _ptr(y), B, Cc, T, eps, scale, int(channel_last_out), _stream()), "hilc_l2norm")
    return y


_register("l2norm", "(Tensor x, float eps, float scale, bool channel_last_out) -> Tensor", _l2norm,
          lambda x, eps, scale, channel_last_out:
          x.new_empty(*((x.shape[0], x.shape[2], x.shape[1]) if channel_last_out else x.shape)))


# ======================================================================================================
# residual VQ
# ======================================================================================================
def _zct(z, channel_last):
    if channel_last:
        B, T, Cc = z.shape
    else:
        B, Cc, T = z.shape
    return B, Cc, T


RVQ_VALU_ONLY = 1      # include/hilcodec_amd.h: HILC_RVQ_VALU_ONLY


def _rvq_encode(z, codebooks, codebooks_t, norms, n_clip, n, channel_last, stage_major, want_q, want_loss, flags):
    B, Cc, T = _zct(z, channel_last)
    Nq, K, _ = codebooks.shape
    rows = max(1, min(n, Nq))
    idx = _new(z, *((rows, B, T) if stage_major else (B, rows, T)), dtype=torch.int64)
    q = torch.empty_like(z) if want_q else None
    ferr = _new(z, B * T) if want_loss else None
    with _timed("rvq_encode", 2.0 * B * T * K * Cc * rows):
        check(lib.hilc_rvq_encode_mixed(_ptr(z), _ptr(codebooks), _ptr(codebooks_t), _ptr(norms),
                                        _ptr(n_clip, torch.int32), _ptr(idx, torch.int64), _ptr(q), _ptr(ferr),
                                        B, Cc, T, K, Nq, n, int(channel_last), int(stage_major), int(flags), _stream()),
              "hilc_rvq_encode")
    loss = None
    if want_loss:
        loss = _new(z)
        check(lib.hilc_mse_finalize(_ptr(ferr), _ptr(loss), B * T, float(B) * T * Cc, _stream()), "hilc_mse_finalize")
    # the dispatcher wants real tensors for every declared output
    return idx, (q if q is not None else _new(z, 0)), (loss if loss is not None else _new(z, 0))


def _rvq_encode_fake(z, codebooks, codebooks_t, norms, n_clip, n, channel_last, stage_major, want_q, want_loss, flags):
    B, Cc, T = _zct(z, channel_last)
    rows = max(1, min(n, codebooks.shape[0]))
    idx = z.new_empty(*((rows, B, T) if stage_major else (B, rows, T)), dtype=torch.int64)
    return idx, (torch.empty_like(z) if want_q else z.new_empty(0)), (z.new_empty(()) if want_loss else z.new_empty(0))


_register("rvq_encode", "(Tensor z, Tensor codebooks, Tensor codebooks_t, Tensor norms, Tensor? n_clip, int n, "
          "bool channel_last, bool stage_major, bool want_q, bool want_loss, int flags) -> (Tensor, Tensor, Tensor)",
          _rvq_encode, _rvq_encode_fake)


def _rvq_decode(indices, codebooks, n_clip, n, channel_last, stage_major):
    if stage_major:
        _, B, T = indices.shape
    else:
        B, _, T = indices.shape
    Nq, K, Cc = codebooks.shape
    q = _new(codebooks, *((B, T, Cc) if channel_last else (B, Cc, T)))
    check(lib.hilc_rvq_decode_mixed(_ptr(indices, torch.int64), _ptr(codebooks), _ptr(n_clip, torch.int32), _ptr(q),
                                    B, Cc, T, K, Nq, n, int(channel_last), int(stage_major), _stream()),
          "hilc_rvq_decode")
    return q


def _rvq_decode_fake(indices, codebooks, n_clip, n, channel_last, stage_major):
    B, T = (indices.shape[1], indices.shape[2]) if stage_major else (indices.shape[0], indices.shape[2])
    Cc = codebooks.shape[2]
    return codebooks.new_empty(*((B, T, Cc) if channel_last else (B, Cc, T)))


_register("rvq_decode", "(Tensor indices, Tensor codebooks, Tensor? n_clip, int n, bool channel_last, bool stage_major)"
          " -> Tensor", _rvq_decode, _rvq_decode_fake)


def _rvq_ema_stats(z, codebooks, indices, n, channel_last, stage_major):
    B, Cc, T = _zct(z, channel_last)
    Nq, K, _ = codebooks.shape
    rows = indices.shape[0] if stage_major else indices.shape[1]
    bucket = _new(z, n, K + K * Cc)
    with _timed("rvq_ema_stats", 4.0 * B * T * Cc * n):
        check(lib.hilc_rvq_ema_stats(_ptr(z), _ptr(codebooks), _ptr(indices, torch.int64), _ptr(bucket), B, Cc, T, K,
                                     n, rows, int(channel_last), int(stage_major), _stream()), "hilc_rvq_ema_stats")
    return bucket


_register("rvq_ema_stats", "(Tensor z, Tensor codebooks, Tensor indices, int n, bool channel_last, bool stage_major)"
          " -> Tensor", _rvq_ema_stats,
          lambda z, codebooks, indices, n, channel_last, stage_major:
          z.new_empty(n, codebooks.shape[1] * (1 + codebooks.shape[2])))


def _rvq_ema_update(embed, ema_num, ema_embed, bucket, decay):
    n, K, Cc = embed.shape
    if tuple(ema_num.shape) != (n, K) or tuple(ema_embed.shape) != (n, K, Cc) or tuple(bucket.shape) != (n, K + K * Cc):
        raise RuntimeError("rvq_ema_update: inconsistent shapes")
    check(lib.hilc_rvq_ema_update(_ptr(embed), _ptr(ema_num), _ptr(ema_embed), _ptr(bucket), float(decay), K, Cc, n,
                                  _stream()), "hilc_rvq_ema_update")


_register("rvq_ema_update", "(Tensor(a!) embed, Tensor(b!) ema_num, Tensor(c!) ema_embed, Tensor bucket, float decay)"
          " -> ()", _rvq_ema_update, lambda embed, ema_num, ema_embed, bucket, decay: None)

_OPS = torch.ops.hilcodec


# ======================================================================================================
# public wrappers (traceable: arguments in, torch.ops.hilcodec.* out)
# ======================================================================================================
def pw_conv(x: Tensor, wt: Tensor, bias: Optional[Tensor] = None, res: Optional[Tensor] = None,
            in_scale: float = 1.0, in_elu: bool = False, out_scale: float = 1.0) -> Tensor:
    """x `[B,K,T]`, wt `[K,M]` -> `[B,M,T]`; see hilc_pw_conv."""
    return _OPS.pw_conv(x, wt, bias, res, float(in_scale), bool(in_elu), float(out_scale))


def dws_conv(x: Tensor, wt: Tensor, dw_w: Tensor, dw_b: Optional[Tensor] = None, res: Optional[Tensor] = None,
             stride: int = 1, in_scale: float = 1.0, in_elu: bool = False, out_scale: float = 1.0,
             out_elu: bool = False) -> Tensor:
    """Fused pointwise -> depthwise causal conv (offline): x `[B,K,T]`, wt `[K,M]`, dw_w `[M,k]` ->
    `[B,M,ceil(T/stride)]`; see hilc_dws_conv."""
    return _OPS.dws_conv(x, wt, dw_w, dw_b, res, int(stride), float(in_scale), bool(in_elu), float(out_scale),
                         bool(out_elu))


def dws_conv_stream_supported(T: int, k: int, stride: int, has_res: bool = False, B: int = 1, M: int = 1) -> bool:
    """whole-clip tiles: a hop of at most 128 samples per stream, a multiple of the stride; longer hops for the
    down-sampling form (k = 2 * stride): per-clip tiles with a recomputed halo — which take a shortcut `res` only in their
    flat-tile form (csrc/gemm.hip: stride <= 8, fewer than 2^31 outputs, a tile holds at most two stream starts)"""
    if T > 128:
        if not (k == 2 * stride and stride <= 16 and T % 4 == 0 and T % stride == 0):
            return False
        if not has_res:
            return True
        h = (stride + 3) // 4 * 4
        n_out = (128 - h - stride) // stride + 1
        while (n_out * stride) % 4 != 0:
            n_out -= 1
        to = T // stride
        return stride <= 8 and B * to + T < (1 << 31) and B * M * to < (1 << 31) and to * 2 >= n_out + 1
    return T % stride == 0 and stride <= k <= 32


def dws_conv_stream_profitable(T: int, k: int, stride: int, has_res: bool = False, B: int = 1, M: int = 1) -> bool:
    """where the fused hop beats pointwise GEMM + cached depthwise conv (measured, tools/layer_profile.py
    --mode streaming): not for 2- or 3-sample hops of the tiled core, whose depthwise taps are nearly all cache reads."""
    if T == 1 and stride == 1:
        return k <= 32  # single-frame layers: the latency-bound 32 x 32-tile kernel (csrc/frame1.hip), taps in its epilogue
    return dws_conv_stream_supported(T, k, stride, has_res, B, M) and T // stride >= 1 and T >= 4


def _state_out(given: Optional[Tensor], like: Tensor, *shape) -> Tensor:
    """the next hop's cache: the caller's persistent buffer, or (reference protocol) a fresh tensor"""
    return given if given is not None else torch.empty(shape, device=like.device, dtype=torch.float32)


def dws_conv_stream(x: Tensor, wt: Tensor, dw_w: Tensor, dw_b: Optional[Tensor], hist: Optional[Tensor],
                    res: Optional[Tensor] = None, stride: int = 1, in_scale: float = 1.0, in_elu: bool = False,
                    out_scale: float = 1.0, out_elu: bool = False, hist_out: Optional[Tensor] = None):
    """Streaming hop of a depthwise-separable block (hilc_dws_conv_stream): x `[B,K,T]` (T <= 128, or any T % 4 == 0 for k = 2 * stride), cache
    `[B,M,k-stride]` (last pointwise outputs of the previous hop) -> (y `[B,M,T/stride]`, new cache)."""
    hout = _state_out(hist_out, x, x.shape[0], wt.shape[1], dw_w.shape[1] - stride)
    y = _OPS.dws_conv_stream(x, wt, dw_w, dw_b, hist, hout, res, int(stride), float(in_scale), bool(in_elu),
                             float(out_scale), bool(out_elu))
    return y, hout


def up_conv_taps(tr_w: Tensor, stride: int) -> Optional[Tensor]:
    """Expanded tap table for strides without a vector tap path (hilc_up_conv_expand_taps); built once per spec."""
    if stride in (2, 4, 8):
        return None
    return _OPS.up_conv_expand_taps(tr_w, int(stride))


def up_conv(x: Tensor, tr_w: Tensor, wt: Tensor, bias: Optional[Tensor], stride: int, in_scale: float = 1.0,
            in_elu: bool = True, hist: Optional[Tensor] = None, want_hist: bool = False,
            taps: Optional[Tensor] = None, hist_out: Optional[Tensor] = None):
    """Fused [Scale, ELU, depthwise transposed conv (k=2*stride), pointwise conv + bias]:
    x `[B,K,Tin]`, tr_w `[K,2*stride]`, wt `[K,M]` -> `[B,M,Tin*stride]`; see hilc_up_conv.
    Streaming: hist `[B,K,1]` = the activated last input frame of the previous hop -> (y, new cache)."""
    if taps is None and stride not in (2, 4, 8):
        taps = up_conv_taps(tr_w, stride)
    hout = _state_out(hist_out, x, x.shape[0], x.shape[1], 1) if want_hist else None
    y = _OPS.up_conv(x, hist, hout, tr_w, taps, wt, bias, int(stride), float(in_scale), bool(in_elu))
    return (y, hout) if want_hist else y


def resblock_supported(C: int, T: int, B: int = 1, streaming: bool = False) -> bool:
    """mirror of hilc_resblock_supported / hilc_resblock_stream_supported (plain Python so that a tracing compiler can
    evaluate it; the C entry points are checked against this in tests/test_api_cpu.py); the streaming form walks a flat
    32-bit column space and also takes the wide blocks of a hop (C = 256 / 384; C = 512 / 768 where whole streams tile 32
    columns)"""
    if T <= 0 or T % 4 != 0:
        return False
    if not streaming:
        return C in (64, 96, 128, 192, 256, 384, 512, 768)
    if B * C * T * 4 >= (1 << 32):
        return False
    return C in (64, 96, 128, 192, 256, 384) or (C in (512, 768) and 32 % T == 0)


def resblock_chain_supported(C: int, T: int, nblk: int, B: int = 1, streaming: bool = True) -> bool:
    """mirror of hilc_resblock_chain_supported: the blocks of one stage in one launch"""
    if nblk < 2 or nblk > (3 if C in (96, 192) or C == (768 if streaming else 384) else 2) or T <= 0 or T % 4 != 0:
        return False
    if not streaming:
        return C in (64, 96, 128, 192, 256, 384, 512)
    if B * C * T * 4 >= (1 << 32):
        return False
    return C in (64, 96, 128, 192) or (C in (512, 768) and 32 % T == 0)


def resblock_chain_row_classes(C: int, streaming: bool = True) -> int:
    """mirror of hilc_resblock_chain_row_classes(_offline): the row split of the packed weights a chain launch reads"""
    if not streaming:
        return 8 if C in (512, 768) else (4 if C in (256, 384) else (2 if C in (128, 192) else 1))
    return 8 if C >= 512 or C == 256 else (12 if C == 384 else (1 if C in (64, 96) else 2))


def resblock_chain_pack(wt: Tensor, streaming: bool = True) -> Tensor:
    """k-major `[C,C]` pointwise weights -> the packed layout of hilc_resblock_chain for this width"""
    return _OPS.resblock_pack_rc(wt, resblock_chain_row_classes(wt.shape[0], streaming))


def resblock_chain(x: Tensor, blocks: Sequence[Sequence], hist: Optional[Sequence[Sequence[Tensor]]] = None,
                   hist_out: Optional[Sequence[Optional[Sequence[Tensor]]]] = None):
    """The residual blocks of ONE stage of a streaming hop in one launch (hilc_resblock_chain): `blocks[i]` =
    (w1p, dw1_w, dw1_b, w2p, dw2_w, dw2_b, pre_scale, out_scale) with w1p / w2p packed by `resblock_chain_pack`, `hist[i]` =
    that block's two caches, `hist_out[i]` = where its new caches go (None: fresh tensors).  Returns (y, [new caches of block 0,
    of block 1, ...]) — equal, bit for bit, to the blocks launched one by one."""
    B, Cc, _ = x.shape
    params, hin, hout, pre, post = [], [], [], [], []
    if hist is None:            # offline: (w1p, ...) packed with resblock_chain_pack(w, streaming=False); returns y only
        for blk in blocks:
            params.extend(blk[:6])
            pre.append(float(blk[6]))
            post.append(float(blk[7]))
        return _OPS.resblock_chain(x, params, [], [], pre, post)
    for i, blk in enumerate(blocks):
        params.extend(blk[:6])
        pre.append(float(blk[6]))
        post.append(float(blk[7]))
        hin.extend(hist[i])
        given = hist_out[i] if hist_out is not None and hist_out[i] is not None else (None, None)
        hout.extend([_state_out(given[0], x, B, Cc, 4), _state_out(given[1], x, B, Cc, 4)])
    y = _OPS.resblock_chain(x, params, hin, hout, pre, post)
    return y, hout


def decoder_stage_supported(C: int, T: int, nblk: int, stride: int, B: int = 1, streaming: bool = True) -> bool:
    """mirror of hilc_decoder_stage_supported (C = output channels of the stage, T = its samples per stream / clip)"""
    if not 1 <= nblk <= 3 or T <= 0 or T % 4 != 0 or stride <= 0 or T % stride != 0 or (streaming and B * C * T * 4 >= (1 << 32)):
        return False
    if C == 768:
        return stride == 8 and (32 % T == 0 if streaming else nblk == 1)
    if C == 384:
        return stride == 5
    return (C == 192 and stride == 4) or (C == 96 and stride == 2)


def decoder_stage(xin: Tensor, up: Sequence, blocks: Sequence[Sequence], hist: Optional[Sequence[Sequence[Tensor]]] = None,
                  up_hist: Optional[Tensor] = None, hist_out: Optional[Sequence[Optional[Sequence[Tensor]]]] = None,
                  up_hist_out: Optional[Tensor] = None):
    """A decoder stage in ONE launch (hilc_decoder_stage): its up-sampling layer `up` = (tr_w `[2C,2r]` — stride 5: the EXPANDED table of
    `up_conv_taps` —, w_lo, w_hi,
    bias `[C]`, in_scale, stride) — w_lo / w_hi = the two ROW halves of the k-major `[2C,C]` pointwise weight packed with
    `resblock_chain_pack` — and its residual blocks (`blocks[i]`, `hist[i]`, `hist_out[i]` as in `resblock_chain`).
    xin `[B,2C,T/r]`, up_hist `[B,2C,1]` -> (y `[B,C,T]`, [block caches...], up-sampling cache); hist None: the offline model -> y."""
    B, K2, _ = xin.shape
    Cc = K2 // 2
    tr_w, w_lo, w_hi, bias, in_scale, stride = up
    params, hin, hout, pre, post = [], [], [], [], []
    if hist is None:                 # offline: y only
        for blk in blocks:
            params.extend(blk[:6])
            pre.append(float(blk[6]))
            post.append(float(blk[7]))
        return _OPS.decoder_stage(xin, tr_w, w_lo, w_hi, bias, None, None, float(in_scale), int(stride), params, [], [], pre, post)
    for i, blk in enumerate(blocks):
        params.extend(blk[:6])
        pre.append(float(blk[6]))
        post.append(float(blk[7]))
        hin.extend(hist[i])
        given = hist_out[i] if hist_out is not None and hist_out[i] is not None else (None, None)
        hout.extend([_state_out(given[0], xin, B, Cc, 4), _state_out(given[1], xin, B, Cc, 4)])
    uout = _state_out(up_hist_out, xin, B, K2, 1)
    y = _OPS.decoder_stage(xin, tr_w, w_lo, w_hi, bias, up_hist, uout, float(in_scale), int(stride), params, hin, hout, pre, post)
    return y, hout, uout


def encoder_stage0_supported(T: int, nblk: int, stride: int, n_fft: int, hop: int, pre_ksize: int, B: int = 1, streaming: bool = False) -> bool:
    """mirror of hilc_encoder_stage0_supported: the encoder's FIRST stage with the first conv and its SpecBlock in the same launch (a hop: T >= 128
    so that a tile holds at most one stream start, and the 32-bit offsets of the streaming form)"""
    if not (1 <= nblk <= 2 and stride == 2 and n_fft == 64 and hop == 1 and pre_ksize == 5 and T > 0 and T % 4 == 0):
        return False
    return not streaming or (T >= 128 and B * 128 * T * 4 < (1 << 32))


def encoder_stage0(wav: Tensor, spec: Sequence, blocks: Sequence[Sequence], down: Sequence, res: Optional[Tensor] = None,
                   hist: Optional[Sequence[Sequence[Tensor]]] = None, hist_out: Optional[Sequence[Optional[Sequence[Tensor]]]] = None,
                   down_hist: Optional[Tensor] = None, down_hist_out: Optional[Tensor] = None, wav_hist: Optional[Tensor] = None):
    """The encoder's first conv, stage-0 SpecBlock, residual blocks and down-sampling layer in ONE launch (hilc_encoder_stage0):
    `spec` = (dft_packed, nyq_sin, pw_packed, bias, pre_w `[64,5]`, pre_b, pre_in_scale, mean, std, normalize, out_scale) as given to
    `spec_block_conv_pre`; `blocks`, `down`, `res` as in `encoder_stage`.  Offline (hist None): wav `[B,1,T]` -> `[B,128,T/2]`, equal bit for
    bit to `encoder_stage(spec_block_conv_pre(wav, ...), blocks, down)`.  A streaming hop (round 6): `wav_hist` `[B,1,H >= 63]` = the waveform
    cache, `hist` / `hist_out` / `down_hist` / `down_hist_out` as in `encoder_stage` -> (y, [block caches...], down cache)."""
    dft, nyq, pw, bias, pre_w, pre_b, pre_in, mean, std, normalize, out_scale = spec
    w_lo, w_hi, ddw_w, ddw_b, in_scale, stride = down
    B, Cc = wav.shape[0], pre_w.shape[0]
    params, hin, hout, pre, post = [], [], [], [], []
    for i, blk in enumerate(blocks):
        params.extend(blk[:6])
        pre.append(float(blk[6]))
        post.append(float(blk[7]))
        if hist is not None:
            hin.extend(hist[i])
            given = hist_out[i] if hist_out is not None and hist_out[i] is not None else (None, None)
            hout.extend([_state_out(given[0], wav, B, Cc, 4), _state_out(given[1], wav, B, Cc, 4)])
    if hist is None:
        return _OPS.encoder_stage0(wav, None, dft, nyq, pw, bias, pre_w, pre_b, float(pre_in), float(mean), float(std), int(normalize), float(out_scale),
                                   params, [], [], pre, post, w_lo, w_hi, ddw_w, ddw_b, None, None, res, float(in_scale), int(stride))
    dout = _state_out(down_hist_out, wav, B, 2 * Cc, int(stride))
    y = _OPS.encoder_stage0(wav, wav_hist, dft, nyq, pw, bias, pre_w, pre_b, float(pre_in), float(mean), float(std), int(normalize), float(out_scale),
                            params, hin, hout, pre, post, w_lo, w_hi, ddw_w, ddw_b, down_hist, dout, res, float(in_scale), int(stride))
    return y, hout, dout


def decoder_stage_post_supported(C: int, T: int, nblk: int, stride: int, ksize: int) -> bool:
    """mirror of hilc_decoder_stage_post_supported: the offline decoder's LAST stage with the closing conv in the same launch"""
    return C == 96 and stride == 2 and nblk == 3 and ksize == 5 and T > 0 and T % 4 == 0


def decoder_stage_post(xin: Tensor, up: Sequence, blocks: Sequence[Sequence], post: Sequence, hist: Optional[Sequence[Sequence[Tensor]]] = None,
                       up_hist: Optional[Tensor] = None, post_hist: Optional[Tensor] = None,
                       hist_out: Optional[Sequence[Optional[Sequence[Tensor]]]] = None, up_hist_out: Optional[Tensor] = None,
                       post_hist_out: Optional[Tensor] = None):
    """The decoder's last stage AND its closing layer in ONE launch (hilc_decoder_stage_post): `up`, `blocks` as in `decoder_stage`,
    `post` = (w `[C,5]`, bias `[1]` or None, in_scale, out_scale, do_tanh) as given to `conv_post`.  Offline (hist None):
    xin `[B,2C,T/r]` -> wav `[B,1,T]`, equal bit for bit to `conv_post(decoder_stage(...))`.  Streaming hop (round 6): `hist` / `up_hist` /
    `hist_out` / `up_hist_out` as in `decoder_stage`, `post_hist` `[B,C,4]` = the closing conv's cache ->
    (wav, [block caches...], up-sampling cache, closing conv's cache), equal bit for bit to `decoder_stage` + `conv_post` with the same caches."""
    tr_w, w_lo, w_hi, bias, in_scale, stride = up
    B, K2, _ = xin.shape
    Cc = K2 // 2
    params, hin, hout, pre, post_s = [], [], [], [], []
    for i, blk in enumerate(blocks):
        params.extend(blk[:6])
        pre.append(float(blk[6]))
        post_s.append(float(blk[7]))
        if hist is not None:
            hin.extend(hist[i])
            given = hist_out[i] if hist_out is not None and hist_out[i] is not None else (None, None)
            hout.extend([_state_out(given[0], xin, B, Cc, 4), _state_out(given[1], xin, B, Cc, 4)])
    pw, pb, p_in, p_out, p_tanh = post
    if hist is None:
        return _OPS.decoder_stage_post(xin, tr_w, w_lo, w_hi, bias, None, None, float(in_scale), int(stride), params, [], [], pre, post_s, pw, pb,
                                       None, None, float(p_in), float(p_out), bool(p_tanh))
    uout = _state_out(up_hist_out, xin, B, K2, 1)
    pout = _state_out(post_hist_out, xin, B, Cc, pw.shape[1] - 1)
    wav = _OPS.decoder_stage_post(xin, tr_w, w_lo, w_hi, bias, up_hist, uout, float(in_scale), int(stride), params, hin, hout, pre, post_s, pw, pb,
                                  post_hist, pout, float(p_in), float(p_out), bool(p_tanh))
    return wav, hout, uout, pout


def encoder_stage_supported(C: int, T: int, nblk: int, stride: int, B: int = 1, streaming: bool = True) -> bool:
    """mirror of hilc_encoder_stage_supported (+ the 32-bit offsets of the streaming form)"""
    if nblk < 1 or nblk > 2 or T <= 0 or T % 4 != 0 or stride <= 0 or T % stride != 0 or (streaming and B * 2 * C * T * 4 >= (1 << 32)):
        return False
    if (C == 64 and stride == 2) or (C == 128 and stride == 4):
        return True
    if not streaming:
        return (C == 256 and stride == 5) or (C == 512 and stride == 8)                      # the wide stages of the offline model
    return (C == 256 and stride == 5 and T % 40 == 0) or (C == 512 and stride == 8 and T % 8 == 0 and 32 % T == 0)   # ... of a hop


def encoder_stage(x: Tensor, blocks: Sequence[Sequence], down: Sequence, hist: Optional[Sequence[Sequence[Tensor]]] = None,
                  hist_out: Optional[Sequence[Optional[Sequence[Tensor]]]] = None, down_hist: Optional[Tensor] = None,
                  down_hist_out: Optional[Tensor] = None, res: Optional[Tensor] = None):
    """An encoder stage in ONE launch (hilc_encoder_stage): its residual blocks (`blocks[i]` as in `resblock_chain`) and its
    down-sampling layer `down` = (w_lo, w_hi, dw_w `[2C,2r]`, dw_b `[2C]`, in_scale, stride) — w_lo / w_hi = the two column halves
    of the k-major `[C,2C]` pointwise weight, packed with `resblock_chain_pack`.  Offline (hist None): -> y `[B,2C,T/r]`.
    Streaming: hist / hist_out per block as in `resblock_chain`, `down_hist` `[B,2C,r]` -> (y, [block caches...], down cache).
    `res` `[B,2C,T/r]` is added to the output (the next stage's SpecBlock branch)."""
    B, Cc, _ = x.shape
    w_lo, w_hi, ddw_w, ddw_b, in_scale, stride = down
    params, hin, hout, pre, post = [], [], [], [], []
    for i, blk in enumerate(blocks):
        params.extend(blk[:6])
        pre.append(float(blk[6]))
        post.append(float(blk[7]))
        if hist is not None:
            hin.extend(hist[i])
            given = hist_out[i] if hist_out is not None and hist_out[i] is not None else (None, None)
            hout.extend([_state_out(given[0], x, B, Cc, 4), _state_out(given[1], x, B, Cc, 4)])
    if hist is None:
        return _OPS.encoder_stage(x, params, [], [], pre, post, w_lo, w_hi, ddw_w, ddw_b, None, None, res, float(in_scale), int(stride))
    dout = _state_out(down_hist_out, x, B, 2 * Cc, int(stride))
    y = _OPS.encoder_stage(x, params, hin, hout, pre, post, w_lo, w_hi, ddw_w, ddw_b, down_hist, dout, res, float(in_scale), int(stride))
    return y, hout, dout


def resblock_pack(wt: Tensor) -> Tensor:
    """k-major `[C,C]` pointwise weights -> the fused block's packed layout (hilc_resblock_pack_weights)."""
    return _OPS.resblock_pack(wt)


def resblock(x: Tensor, w1p: Tensor, dw1_w: Tensor, dw1_b: Tensor, w2p: Tensor, dw2_w: Tensor, dw2_b: Tensor,
             pre_scale: float, out_scale: float, hist: Optional[Sequence[Tensor]] = None,
             hist_out: Optional[Sequence[Tensor]] = None):
    """Fully fused residual block (hilc_resblock): x `[B,C,T]` -> new tensor `[B,C,T]`; w1p / w2p = PACKED
    pointwise weights (`resblock_pack`).  Streaming: hist = (cache of depthwise 1, cache of depthwise 2), each
    `[B,C,4]` -> (y, [new caches]).  k-major `[C,C]` matrices are accepted too and packed on the fly (tests, tools);
    the plans hold the packed form."""
    if w1p.dim() == 2:
        w1p, w2p = resblock_pack(w1p), resblock_pack(w2p)
    if hist is None:
        return _OPS.resblock(x, w1p, dw1_w, dw1_b, w2p, dw2_w, dw2_b, None, None, None, None, float(pre_scale),
                             float(out_scale))
    B, Cc, _ = x.shape
    o1 = _state_out(hist_out[0] if hist_out is not None else None, x, B, Cc, 4)
    o2 = _state_out(hist_out[1] if hist_out is not None else None, x, B, Cc, 4)
    y = _OPS.resblock(x, w1p, dw1_w, dw1_b, w2p, dw2_w, dw2_b, hist[0], hist[1], o1, o2, float(pre_scale),
                      float(out_scale))
    return y, [o1, o2]


def dw_conv(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, res: Optional[Tensor] = None,
            stride: int = 1, hist: Optional[Tensor] = None, want_hist: bool = False,
            in_scale: float = 1.0, in_elu: bool = False, out_scale: float = 1.0, out_elu: bool = False,
            hist_out: Optional[Tensor] = None):
    """x `[B,C,T]`, w `[C,k]` -> `[B,C,ceil(T/stride)]` (+ new history `[B,C,k-stride]` if want_hist)."""
    hout = _state_out(hist_out, x, x.shape[0], x.shape[1], w.shape[1] - stride) if want_hist else None
    y = _OPS.dw_conv(x, hist, w, bias, res, hout, int(stride), float(in_scale), bool(in_elu), float(out_scale),
                     bool(out_elu))
    return (y, hout) if want_hist else y


def dw_convtr(x: Tensor, w: Tensor, stride: int, hist: Optional[Tensor] = None, want_hist: bool = False,
              in_scale: float = 1.0, in_elu: bool = False, hist_out: Optional[Tensor] = None):
    """x `[B,C,T]`, w `[C,2*stride]` -> `[B,C,T*stride]`."""
    hout = _state_out(hist_out, x, x.shape[0], x.shape[1], 1) if want_hist else None
    y = _OPS.dw_convtr(x, hist, w, hout, int(stride), float(in_scale), bool(in_elu))
    return (y, hout) if want_hist else y


def conv_pre(wav: Tensor, w: Tensor, bias: Optional[Tensor], in_scale: float = 1.0,
             hist: Optional[Tensor] = None) -> Tensor:
    """wav `[B,1,T]`, w `[C,k]` -> `[B,C,T]`; hist `[B,1,L]` (L >= k-1) = waveform history."""
    return _OPS.conv_pre(wav, hist, w, bias, float(in_scale))


def conv_post(x: Tensor, w: Tensor, bias: Optional[Tensor], in_scale: float = 1.0, in_elu: bool = True,
              out_scale: float = 1.0, do_tanh: bool = True, hist: Optional[Tensor] = None,
              want_hist: bool = False, hist_out: Optional[Tensor] = None):
    """x `[B,C,T]`, w `[C,k]` -> `[B,1,T]`."""
    hout = _state_out(hist_out, x, x.shape[0], x.shape[1], w.shape[1] - 1) if want_hist else None
    y = _OPS.conv_post(x, hist, w, bias, hout, float(in_scale), bool(in_elu), float(out_scale), bool(do_tanh))
    return (y, hout) if want_hist else y


def stft_logmag(wav: Tensor, basis_t: Tensor, n_fft: int, hop: int, mean: float = 0.0, std: float = 1.0,
                normalize=True, hist: Optional[Tensor] = None) -> Tensor:
    """wav `[B,1,T]` -> `[B, n_fft/2+1, (T-1)//hop+1]`; normalize: False/0 log-mag, True/1 (log-mag - mean)/std,
    2 plain magnitude."""
    return _OPS.stft_logmag(wav, hist, basis_t, int(n_fft), int(hop), float(mean), float(std), int(normalize))


def spec_block_supported(n_fft: int, hop: int, C: int, T: int) -> bool:
    """mirror of hilc_spec_block_supported (plain Python: traceable)"""
    if n_fft not in (64, 128, 256) or hop != {64: 1, 128: 2, 256: 8}[n_fft] or C != n_fft or T <= 0:
        return False
    return ((T - 1) // hop + 1) % 4 == 0


def spec_block_profitable(n_fft: int, hop: int, C: int, T: int) -> bool:
    """supported, and the launch fills its 128-frame tiles: clips of 32 ... 511 frames (a streaming hop: 40 frames per stream at n_fft = 256)
    walk the flat frame space, longer ones per-clip tiles that must be >= 60 % full"""
    if not spec_block_supported(n_fft, hop, C, T):
        return False
    tf = (T - 1) // hop + 1
    if 32 <= tf < 512:          # round 6: short clips (a streaming hop) tile the FLAT frame space — every tile full (csrc/spec.hip: FLAT)
        return True
    return tf * 10 >= ((tf + 127) // 128) * 128 * 6


def spec_block_tables(basis_t: Tensor, wt: Tensor, n_fft: int):
    """The fused SpecBlock's operands from the un-fused ones: basis_t `[n_fft][m_pad]` (interleaved cos_k, sin_k columns,
    k = 0..n_fft/2: hilc_stft_logmag's layout) -> packed DFT matrix with exactly n_fft rows + the sin_{n_fft/2} row;
    wt `[n_fft/2+1][C]` -> packed conv weight.  Pure column selection: no arithmetic touches the basis."""
    cols = [0, n_fft] + list(range(2, n_fft))              # cos_0, cos_{N/2}, then (cos_k, sin_k), k = 1..N/2-1
    dft = basis_t[:, cols].contiguous()
    nyq = basis_t[:, n_fft + 1].contiguous()
    return (_OPS.spec_block_pack(dft, int(n_fft), 0), nyq, _OPS.spec_block_pack(wt.contiguous(), int(n_fft), 1))


def spec_block(wav: Tensor, dft_packed: Tensor, nyq_sin: Tensor, pw_packed: Tensor, bias: Optional[Tensor], x: Optional[Tensor],
               n_fft: int, hop: int, mean: float = 0.0, std: float = 1.0, normalize=True, out_scale: float = 1.0,
               hist: Optional[Tensor] = None) -> Tensor:
    """One-launch SpecBlock (hilc_spec_block): wav `[B,1,T]`, x `[B,n_fft,T/hop]` -> x + out_scale * (W spec + bias);
    hist `[B,1,L]` (L >= n_fft-1) = the waveform before t = 0 (streaming hop).  x None: the branch alone."""
    return _OPS.spec_block(wav, hist, dft_packed, nyq_sin, pw_packed, bias, x, int(n_fft), int(hop), float(mean), float(std),
                           int(normalize), float(out_scale))


def spec_block_conv_pre(wav: Tensor, dft_packed: Tensor, nyq_sin: Tensor, pw_packed: Tensor, bias: Optional[Tensor],
                        pre_w: Tensor, pre_b: Optional[Tensor], pre_in_scale: float, n_fft: int, hop: int, mean: float = 0.0,
                        std: float = 1.0, normalize=True, out_scale: float = 1.0, hist: Optional[Tensor] = None) -> Tensor:
    """First encoder stage in one launch: conv_pre(wav) + SpecBlock branch (hilc_spec_block_conv_pre)."""
    return _OPS.spec_block_conv_pre(wav, hist, dft_packed, nyq_sin, pw_packed, bias, pre_w, pre_b, float(pre_in_scale), int(n_fft),
                                    int(hop), float(mean), float(std), int(normalize), float(out_scale))


def tail(x: Tensor, hist: Optional[Tensor], pad: int, out: Optional[Tensor] = None) -> Tensor:
    """Last `pad` samples of cat([hist, x], -1) along time; x `[B,C,T]`, hist `[B,C,L]`."""
    out = _state_out(out, x, x.shape[0], x.shape[1], pad)
    _OPS.tail(x, hist, out)
    return out


def l2norm(x: Tensor, eps: float = 1e-12, scale: float = 1.0, channel_last_out: bool = False) -> Tensor:
    return _OPS.l2norm(x, float(eps), float(scale), bool(channel_last_out))


def is_scalar_n(n) -> bool:
    """anything integer-like (python / numpy ints, 0-dim arrays and tensors) is ONE n for the whole batch"""
    return isinstance(n, numbers.Integral) or (isinstance(n, (Tensor, np.ndarray)) and n.ndim == 0)


def per_clip_n(n, B: int, Nq: int, device):
    """`n` as the reference takes it (one int for the whole batch) or one int per clip (mixed-bitrate batch).
    Returns (rows, int32 [B] device tensor or None).  Every entry obeys the reference's assert
    (`models/hilcodec/vector_quantize.py:213-214`)."""
    if is_scalar_n(n):
        return int(n), None
    host = torch.as_tensor(n).detach().to("cpu", torch.int64).reshape(-1)
    if host.numel() != B:
        raise RuntimeError(f"per-clip n needs {B} entries, got {host.numel()}")
    lo, hi = int(host.min()), int(host.max())
    assert 1 <= lo and hi <= Nq, f"'n' must be in range of 1 <= n <= {Nq}"
    return hi, host.to(torch.int32).to(device)


def rvq_encode(z: Tensor, codebooks: Tensor, codebooks_t: Tensor, norms: Tensor, n,
               channel_last: bool = False, stage_major: bool = False, want_q: bool = True,
               want_loss: bool = False, valu_only: bool = False):
    """Returns (indices int64, q or None, loss 0-d or None).  `n`: int, or one int per clip (rows of `indices`
    beyond a clip's own n hold -1).  `valu_only` (HILC_RVQ_VALU_ONLY of the C ABI): keep batches of 8 192 frames and more on the
    VALU form instead of the matrix pipe — same fmaf chains, same bits; the quantiser modules pass their `rvq_valu_only` attribute."""
    B = z.shape[0]
    n, n_clip = per_clip_n(n, B, codebooks.shape[0], z.device)
    idx, q, loss = _OPS.rvq_encode(z, codebooks, codebooks_t, norms, n_clip, n, bool(channel_last), bool(stage_major),
                                   bool(want_q), bool(want_loss), RVQ_VALU_ONLY if valu_only else 0)
    return idx, (q if want_q else None), (loss if want_loss else None)


def rvq_decode(indices: Tensor, codebooks: Tensor, n, channel_last: bool = True,
               stage_major: bool = True) -> Tensor:
    B = indices.shape[1] if stage_major else indices.shape[0]
    n, n_clip = per_clip_n(n, B, codebooks.shape[0], indices.device)
    return _OPS.rvq_decode(indices, codebooks, n_clip, n, bool(channel_last), bool(stage_major))


def rvq_ema_stats(z: Tensor, codebooks: Tensor, indices: Tensor, n: int, channel_last: bool = False,
                  stage_major: bool = False) -> Tensor:
    """Training-side cluster statistics of the first n stages: bucket `[n, K + K*C]` (counts | residual sums),
    the reference's per-stage all-reduce payload (`vector_quantize.py:155-162`) for all stages at once."""
    return _OPS.rvq_ema_stats(z, codebooks, indices, int(n), bool(channel_last), bool(stage_major))


def rvq_ema_update(embed: Tensor, ema_num: Tensor, ema_embed: Tensor, bucket: Tensor, decay: float) -> None:
    """In place on stacked `[n,K,C]` / `[n,K]` tensors: EMA of counts and sums, embed = ema_embed / ema_num."""
    _OPS.rvq_ema_update(embed, ema_num, ema_embed, bucket, float(decay))
