"""Thin Python wrappers: torch CUDA(HIP) tensors -> device pointers -> C ABI (`include/hilcodec_amd.h`).

PyTorch is used for device memory and the current stream only; every arithmetic step runs in the
hand-written gfx950 kernels of `csrc/`.  All tensors must be fp32 (indices int64), contiguous, on a
GPU; anything else raises — there is deliberately no fallback path."""
from __future__ import annotations

import numbers
from typing import Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from ._lib import check, lib


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


TIMER: Optional[LaunchTimer] = None


class _timed:
    def __init__(self, kind: str, work: float, tag: str = ""):
        self.kind, self.work, self.tag = kind, work, tag

    def __enter__(self):
        if TIMER is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if TIMER is not None:
            self.e1.record()
            TIMER.records.append((self.kind, self.work, self.e0, self.e1, self.tag))
        return False


def pw_conv(x: Tensor, wt: Tensor, bias: Optional[Tensor] = None, res: Optional[Tensor] = None,
            out: Optional[Tensor] = None, in_scale: float = 1.0, in_elu: bool = False,
            out_scale: float = 1.0) -> Tensor:
    """x `[B,K,T]`, wt `[K,M]` -> `[B,M,T]`; see hilc_pw_conv."""
    B, K, T = x.shape
    M = wt.shape[1]
    assert wt.shape[0] == K
    y = out if out is not None else torch.empty(B, M, T, device=x.device, dtype=torch.float32)
    with _timed("pw_conv", 2.0 * B * T * K * M, f"K{K} M{M} T{T}"):
        check(lib.hilc_pw_conv(_ptr(x), _ptr(wt), _ptr(bias), _ptr(res), _ptr(y), B, K, M, T,
                               in_scale, int(in_elu), out_scale, _stream()), "hilc_pw_conv")
    return y


def dws_conv(x: Tensor, wt: Tensor, dw_w: Tensor, dw_b: Optional[Tensor] = None, res: Optional[Tensor] = None,
             stride: int = 1, in_scale: float = 1.0, in_elu: bool = False, out_scale: float = 1.0,
             out_elu: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """Fused pointwise -> depthwise causal conv (offline): x `[B,K,T]`, wt `[K,M]`, dw_w `[M,k]` ->
    `[B,M,ceil(T/stride)]`; see hilc_dws_conv."""
    B, K, T = x.shape
    M = wt.shape[1]
    k = dw_w.shape[1]
    To = (T + stride - 1) // stride
    y = out if out is not None else torch.empty(B, M, To, device=x.device, dtype=torch.float32)
    with _timed("dws_conv", 2.0 * B * T * K * M, f"K{K} M{M} T{T} k{k} s{stride}"):
        check(lib.hilc_dws_conv(_ptr(x), _ptr(wt), _ptr(dw_w), _ptr(dw_b), _ptr(res), _ptr(y), B, K, M, T, k,
                                stride, in_scale, int(in_elu), out_scale, int(out_elu), _stream()), "hilc_dws_conv")
    return y


def dws_conv_stream_supported(T: int, k: int, stride: int) -> bool:
    """whole-clip tiles: a hop of at most 128 samples per stream, a multiple of the stride"""
    return T <= 128 and T % stride == 0 and stride <= k <= 32


def dws_conv_stream_profitable(T: int, k: int, stride: int) -> bool:
    """where the fused hop beats pointwise GEMM + cached depthwise conv (measured, tools/layer_profile.py
    --mode streaming): not for single-sample hops, whose depthwise taps are nearly all cache reads."""
    return dws_conv_stream_supported(T, k, stride) and T // stride >= 1 and T >= 4


def dws_conv_stream(x: Tensor, wt: Tensor, dw_w: Tensor, dw_b: Optional[Tensor], hist: Optional[Tensor],
                    res: Optional[Tensor] = None, stride: int = 1, in_scale: float = 1.0, in_elu: bool = False,
                    out_scale: float = 1.0, out_elu: bool = False, out: Optional[Tensor] = None):
    """Streaming hop of a depthwise-separable block (hilc_dws_conv_stream): x `[B,K,T]`, T <= 128, cache
    `[B,M,k-stride]` (last pointwise outputs of the previous hop) -> (y `[B,M,T/stride]`, new cache)."""
    B, K, T = x.shape
    M = wt.shape[1]
    k = dw_w.shape[1]
    pad = k - stride
    if hist is not None and tuple(hist.shape) != (B, M, pad):
        raise RuntimeError(f"cache must be [{B},{M},{pad}], got {tuple(hist.shape)}")
    y = out if out is not None else torch.empty(B, M, T // stride, device=x.device, dtype=torch.float32)
    hout = torch.empty(B, M, pad, device=x.device, dtype=torch.float32)
    with _timed("dws_conv", 2.0 * B * T * K * M, f"K{K} M{M} T{T} k{k} s{stride} stream"):
        check(lib.hilc_dws_conv_stream(_ptr(x), _ptr(wt), _ptr(dw_w), _ptr(dw_b), _ptr(hist), _ptr(hout), _ptr(res),
                                       _ptr(y), B, K, M, T, k, stride, in_scale, int(in_elu), out_scale,
                                       int(out_elu), _stream()), "hilc_dws_conv_stream")
    return y, hout


_TAPS = {}


def up_conv_taps(tr_w: Tensor, stride: int) -> Optional[Tensor]:
    """Expanded tap table for strides without a vector tap path (hilc_up_conv_expand_taps), cached per weight tensor."""
    if stride in (2, 4, 8):
        return None
    key = (tr_w.data_ptr(), tr_w._version, tuple(tr_w.shape), tr_w.device.index)
    hit = _TAPS.get(key)
    if hit is not None and hit[0]() is tr_w:
        return hit[1]
    K = tr_w.shape[0]
    out = torch.empty(K * stride * 8, device=tr_w.device, dtype=torch.float32)
    check(lib.hilc_up_conv_expand_taps(_ptr(tr_w), _ptr(out), K, stride, _stream()), "hilc_up_conv_expand_taps")
    if len(_TAPS) > 64:
        _TAPS.clear()
    import weakref
    _TAPS[key] = (weakref.ref(tr_w), out)
    return out


def up_conv(x: Tensor, tr_w: Tensor, wt: Tensor, bias: Optional[Tensor], stride: int, in_scale: float = 1.0,
            in_elu: bool = True, hist: Optional[Tensor] = None, want_hist: bool = False):
    """Fused [Scale, ELU, depthwise transposed conv (k=2*stride), pointwise conv + bias]:
    x `[B,K,Tin]`, tr_w `[K,2*stride]`, wt `[K,M]` -> `[B,M,Tin*stride]`; see hilc_up_conv.
    Streaming: hist `[B,K,1]` = the activated last input frame of the previous hop -> (y, new cache)."""
    B, K, Tin = x.shape
    M = wt.shape[1]
    y = torch.empty(B, M, Tin * stride, device=x.device, dtype=torch.float32)
    taps = up_conv_taps(tr_w, stride)
    if hist is None and not want_hist:
        with _timed("up_conv", 2.0 * B * Tin * stride * K * M, f"K{K} M{M} Tin{Tin} r{stride}"):
            check(lib.hilc_up_conv_expanded(_ptr(x), None, None, _ptr(tr_w), _ptr(taps), _ptr(wt), _ptr(bias), _ptr(y),
                                            B, K, M, Tin, stride, in_scale, int(in_elu), _stream()), "hilc_up_conv")
        return y
    if hist is not None and hist.numel() != B * K:
        raise RuntimeError(f"cache must be [{B},{K},1], got {tuple(hist.shape)}")
    hout = torch.empty(B, K, 1, device=x.device, dtype=torch.float32) if want_hist else None
    with _timed("up_conv", 2.0 * B * Tin * stride * K * M, f"K{K} M{M} Tin{Tin} r{stride} stream"):
        check(lib.hilc_up_conv_expanded(_ptr(x), _ptr(hist), _ptr(hout), _ptr(tr_w), _ptr(taps), _ptr(wt), _ptr(bias),
                                        _ptr(y), B, K, M, Tin, stride, in_scale, int(in_elu), _stream()),
              "hilc_up_conv_stream")
    return (y, hout) if want_hist else y


def resblock_supported(C: int, T: int) -> bool:
    return bool(lib.hilc_resblock_supported(C, T))


_SCHED = {}
_PACKED = {}


def resblock_pack(wt: Tensor) -> Tensor:
    """k-major `[C,C]` pointwise weights -> the fused block's packed layout (hilc_resblock_pack_weights), cached per
    weight tensor (same storage and version => same packed copy)."""
    key = (wt.data_ptr(), wt._version, tuple(wt.shape), wt.device.index)
    hit = _PACKED.get(key)
    if hit is not None and hit[0]() is wt:
        return hit[1]
    Cc = wt.shape[0]
    out = torch.empty(Cc * Cc, device=wt.device, dtype=torch.float32)
    check(lib.hilc_resblock_pack_weights(_ptr(wt), _ptr(out), Cc, _stream()), "hilc_resblock_pack_weights")
    if len(_PACKED) > 256:
        _PACKED.clear()
    import weakref
    _PACKED[key] = (weakref.ref(wt), out)
    return out



def _sched_buffer(device) -> Tensor:
    """Two zeroed ints per (device, stream) for the residual-block kernel's ticket scheduler; the kernel re-arms
    them itself, so the buffer is written by the host only once."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _SCHED.get(key)
    if buf is None:
        buf = torch.zeros(2, dtype=torch.int32, device=device)
        _SCHED[key] = buf
    return buf


def resblock(x: Tensor, w1t: Tensor, dw1_w: Tensor, dw1_b: Tensor, w2t: Tensor, dw2_w: Tensor, dw2_b: Tensor,
             pre_scale: float, out_scale: float, hist: Optional[Sequence[Tensor]] = None):
    """Fully fused residual block (hilc_resblock): x `[B,C,T]` -> new tensor `[B,C,T]`.
    Streaming: hist = (cache of depthwise 1, cache of depthwise 2), each `[B,C,4]` -> (y, [new caches])."""
    B, Cc, T = x.shape
    y = torch.empty_like(x)
    w1t, w2t = resblock_pack(w1t), resblock_pack(w2t)
    if hist is None:
        with _timed("resblock", 4.0 * B * T * Cc * Cc, f"C{Cc} T{T}"):
            check(lib.hilc_resblock_balanced(_ptr(x), _ptr(w1t), _ptr(dw1_w), _ptr(dw1_b), _ptr(w2t), _ptr(dw2_w),
                                             _ptr(dw2_b), None, None, None, None, _ptr(y),
                                             _ptr(_sched_buffer(x.device), torch.int32), 0, B, Cc, T, pre_scale,
                                             out_scale, _stream()), "hilc_resblock")
        return y
    h1, h2 = hist
    if tuple(h1.shape) != (B, Cc, 4) or tuple(h2.shape) != (B, Cc, 4):
        raise RuntimeError(f"resblock caches must be [{B},{Cc},4], got {tuple(h1.shape)} / {tuple(h2.shape)}")
    o1, o2 = torch.empty_like(h1), torch.empty_like(h2)
    with _timed("resblock", 4.0 * B * T * Cc * Cc, f"C{Cc} T{T} stream"):
        check(lib.hilc_resblock_balanced(_ptr(x), _ptr(w1t), _ptr(dw1_w), _ptr(dw1_b), _ptr(w2t), _ptr(dw2_w),
                                         _ptr(dw2_b), _ptr(h1), _ptr(h2), _ptr(o1), _ptr(o2), _ptr(y),
                                         _ptr(_sched_buffer(x.device), torch.int32), 1, B, Cc, T, pre_scale,
                                         out_scale, _stream()), "hilc_resblock_stream")
    return y, [o1, o2]


def dw_conv(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, res: Optional[Tensor] = None,
            stride: int = 1, hist: Optional[Tensor] = None, want_hist: bool = False,
            in_scale: float = 1.0, in_elu: bool = False, out_scale: float = 1.0, out_elu: bool = False,
            out: Optional[Tensor] = None):
    """x `[B,C,T]`, w `[C,k]` -> `[B,C,ceil(T/stride)]` (+ new history `[B,C,k-stride]` if want_hist)."""
    B, Cc, T = x.shape
    k = w.shape[1]
    To = (T + stride - 1) // stride
    y = out if out is not None else torch.empty(B, Cc, To, device=x.device, dtype=torch.float32)
    hout = torch.empty(B, Cc, k - stride, device=x.device, dtype=torch.float32) if want_hist else None
    with _timed("dw_conv", 4.0 * B * Cc * (T + To * (2 if res is not None else 1)), f"C{Cc} T{T} k{k} s{stride}"):
        check(lib.hilc_dw_conv(_ptr(x), _ptr(hist), _ptr(w), _ptr(bias), _ptr(res), _ptr(y), _ptr(hout),
                               B, Cc, T, k, stride, in_scale, int(in_elu), out_scale, int(out_elu), _stream()),
              "hilc_dw_conv")
    return (y, hout) if want_hist else y


def dw_convtr(x: Tensor, w: Tensor, stride: int, hist: Optional[Tensor] = None, want_hist: bool = False,
              in_scale: float = 1.0, in_elu: bool = False):
    """x `[B,C,T]`, w `[C,2*stride]` -> `[B,C,T*stride]`."""
    B, Cc, T = x.shape
    assert w.shape[1] == 2 * stride
    y = torch.empty(B, Cc, T * stride, device=x.device, dtype=torch.float32)
    hout = torch.empty(B, Cc, 1, device=x.device, dtype=torch.float32) if want_hist else None
    with _timed("dw_convtr", 4.0 * B * Cc * T * (1 + stride), f"C{Cc} T{T} r{stride}"):
        check(lib.hilc_dw_convtr(_ptr(x), _ptr(hist), _ptr(w), _ptr(y), _ptr(hout), B, Cc, T, stride,
                                 in_scale, int(in_elu), _stream()), "hilc_dw_convtr")
    return (y, hout) if want_hist else y


def conv_pre(wav: Tensor, w: Tensor, bias: Optional[Tensor], in_scale: float = 1.0,
             hist: Optional[Tensor] = None) -> Tensor:
    """wav `[B,1,T]`, w `[C,k]` -> `[B,C,T]`; hist `[B,1,L]` (L >= k-1) = waveform history."""
    B, one, T = wav.shape
    assert one == 1
    Cc, k = w.shape
    y = torch.empty(B, Cc, T, device=wav.device, dtype=torch.float32)
    hl = hist.shape[-1] if hist is not None else 0
    with _timed("conv_pre", 4.0 * B * T * (1 + Cc)):
        check(lib.hilc_conv_pre(_ptr(wav), _ptr(hist), hl, _ptr(w), _ptr(bias), _ptr(y), B, Cc, T, k,
                                in_scale, _stream()), "hilc_conv_pre")
    return y


def conv_post(x: Tensor, w: Tensor, bias: Optional[Tensor], in_scale: float = 1.0, in_elu: bool = True,
              out_scale: float = 1.0, do_tanh: bool = True, hist: Optional[Tensor] = None,
              want_hist: bool = False):
    """x `[B,C,T]`, w `[C,k]` -> `[B,1,T]`."""
    B, Cc, T = x.shape
    k = w.shape[1]
    y = torch.empty(B, 1, T, device=x.device, dtype=torch.float32)
    hout = torch.empty(B, Cc, k - 1, device=x.device, dtype=torch.float32) if want_hist else None
    with _timed("conv_post", 4.0 * B * T * (1 + Cc)):
        check(lib.hilc_conv_post(_ptr(x), _ptr(hist), _ptr(w), _ptr(bias), _ptr(y), _ptr(hout), B, Cc, T, k,
                                 in_scale, int(in_elu), out_scale, int(do_tanh), _stream()), "hilc_conv_post")
    return (y, hout) if want_hist else y


def stft_logmag(wav: Tensor, basis_t: Tensor, n_fft: int, hop: int, mean: float = 0.0, std: float = 1.0,
                normalize=True, hist: Optional[Tensor] = None) -> Tensor:
    """wav `[B,1,T]` -> `[B, n_fft/2+1, (T-1)//hop+1]`; normalize: False/0 log-mag, True/1 (log-mag - mean)/std,
    2 plain magnitude."""
    B, one, T = wav.shape
    Tf = (T - 1) // hop + 1
    spec = torch.empty(B, n_fft // 2 + 1, Tf, device=wav.device, dtype=torch.float32)
    hl = hist.shape[-1] if hist is not None else 0
    with _timed("stft", 2.0 * B * Tf * n_fft * (n_fft + 2), f"N{n_fft} hop{hop}"):
        check(lib.hilc_stft_logmag(_ptr(wav), _ptr(hist), hl, _ptr(basis_t), _ptr(spec), B, T, n_fft, hop,
                                   mean, std, int(normalize), _stream()), "hilc_stft_logmag")
    return spec


def tail(x: Tensor, hist: Optional[Tensor], pad: int) -> Tensor:
    """Last `pad` samples of cat([hist, x], -1) along time; x `[B,C,T]`, hist `[B,C,L]`."""
    B, Cc, T = x.shape
    out = torch.empty(B, Cc, pad, device=x.device, dtype=torch.float32)
    hl = hist.shape[-1] if hist is not None else 0
    check(lib.hilc_tail(_ptr(x), _ptr(hist), _ptr(out), B * Cc, T, pad, hl, _stream()), "hilc_tail")
    return out


def l2norm(x: Tensor, eps: float = 1e-12, scale: float = 1.0, channel_last_out: bool = False) -> Tensor:
    B, Cc, T = x.shape
    y = torch.empty((B, T, Cc) if channel_last_out else (B, Cc, T), device=x.device, dtype=torch.float32)
    check(lib.hilc_l2norm(_ptr(x), _ptr(y), B, Cc, T, eps, scale, int(channel_last_out), _stream()), "hilc_l2norm")
    return y


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
               want_loss: bool = False):
    """Returns (indices int64, q or None, loss 0-d or None).  `n`: int, or one int per clip (rows of `indices`
    beyond a clip's own n hold -1)."""
    if channel_last:
        B, T, Cc = z.shape
    else:
        B, Cc, T = z.shape
    Nq, K, _ = codebooks.shape
    n, n_clip = per_clip_n(n, B, Nq, z.device)
    nn = max(1, min(n, Nq))
    idx = torch.empty((nn, B, T) if stage_major else (B, nn, T), device=z.device, dtype=torch.int64)
    q = torch.empty_like(z) if want_q else None
    ferr = torch.empty(B * T, device=z.device, dtype=torch.float32) if want_loss else None
    with _timed("rvq_encode", 2.0 * B * T * K * Cc * nn):
        check(lib.hilc_rvq_encode_mixed(_ptr(z), _ptr(codebooks), _ptr(codebooks_t), _ptr(norms),
                                        _ptr(n_clip, torch.int32), _ptr(idx, torch.int64), _ptr(q), _ptr(ferr),
                                        B, Cc, T, K, Nq, n, int(channel_last), int(stage_major), _stream()),
              "hilc_rvq_encode")
    loss = None
    if want_loss:
        loss = torch.empty((), device=z.device, dtype=torch.float32)
        check(lib.hilc_mse_finalize(_ptr(ferr), _ptr(loss), B * T, float(B) * T * Cc, _stream()), "hilc_mse_finalize")
    return idx, q, loss


def rvq_decode(indices: Tensor, codebooks: Tensor, n, channel_last: bool = True,
               stage_major: bool = True) -> Tensor:
    if stage_major:
        _, B, T = indices.shape
    else:
        B, _, T = indices.shape
    Nq, K, Cc = codebooks.shape
    n, n_clip = per_clip_n(n, B, Nq, indices.device)
    q = torch.empty((B, T, Cc) if channel_last else (B, Cc, T), device=indices.device, dtype=torch.float32)
    check(lib.hilc_rvq_decode_mixed(_ptr(indices, torch.int64), _ptr(codebooks), _ptr(n_clip, torch.int32), _ptr(q),
                                    B, Cc, T, K, Nq, n, int(channel_last), int(stage_major), _stream()),
          "hilc_rvq_decode")
    return q


def rvq_ema_stats(z: Tensor, codebooks: Tensor, indices: Tensor, n: int, channel_last: bool = False,
                  stage_major: bool = False) -> Tensor:
    """Training-side cluster statistics of the first n stages: bucket `[n, K + K*C]` (counts | residual sums),
    the reference's per-stage all-reduce payload (`vector_quantize.py:155-162`) for all stages at once."""
    if channel_last:
        B, T, Cc = z.shape
    else:
        B, Cc, T = z.shape
    Nq, K, _ = codebooks.shape
    rows = indices.shape[0] if stage_major else indices.shape[1]
    bucket = torch.empty(n, K + K * Cc, device=z.device, dtype=torch.float32)
    with _timed("rvq_ema_stats", 4.0 * B * T * Cc * n):
        check(lib.hilc_rvq_ema_stats(_ptr(z), _ptr(codebooks), _ptr(indices, torch.int64), _ptr(bucket), B, Cc, T, K,
                                     n, rows, int(channel_last), int(stage_major), _stream()), "hilc_rvq_ema_stats")
    return bucket


def rvq_ema_update(embed: Tensor, ema_num: Tensor, ema_embed: Tensor, bucket: Tensor, decay: float) -> None:
    """In place on stacked `[n,K,C]` / `[n,K]` tensors: EMA of counts and sums, embed = ema_embed / ema_num."""
    n, K, Cc = embed.shape
    if tuple(ema_num.shape) != (n, K) or tuple(ema_embed.shape) != (n, K, Cc) or tuple(bucket.shape) != (n, K + K * Cc):
        raise RuntimeError("rvq_ema_update: inconsistent shapes")
    check(lib.hilc_rvq_ema_update(_ptr(embed), _ptr(ema_num), _ptr(ema_embed), _ptr(bucket), float(decay), K, Cc, n,
                                  _stream()), "hilc_rvq_ema_update")
