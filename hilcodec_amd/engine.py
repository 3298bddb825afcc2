"""Execution plans for the HILCodec hot path on MI355X.

A *spec* is the folded, device-resident description of one sub-network (encoder / decoder / RVQ):
plain fp32 weights in kernel layouts plus the handful of scalars the reference applies around them.
`run_encoder` / `run_decoder` walk a spec and enqueue the hand-written gfx950 kernels (`ops.py` ->
C ABI) on the current HIP stream; with `caches=None` they compute the offline causal model
(`models/hilcodec/modules/seanet.py:368-378,477-479`), with a cache list they compute one streaming
step with the reference's cache protocol (`models/hilcodec/streaming.py:482-517,619-648`).  Both
behaviours of the reference's two decoders are data in the spec (per-block
`pre_scale`, where wav_std is applied), not code forks.
"""
from __future__ import annotations

import contextlib
import contextvars
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops

# offline path: fuse every pointwise->depthwise pair into one launch (hilc_dws_conv).  The unfused
# kernels remain the streaming path (per-hop tiles are a few samples wide) and a debugging aid.
FUSE_DWS = True
# narrow layers (C <= 192) are HBM-bound unless the whole residual block is one launch (hilc_resblock)
FUSE_UPSAMPLE = True      # decoder up-sampling: transposed conv computed inside the pointwise GEMM's loader
FUSE_RESBLOCK = True
FUSE_STREAM = True        # streaming hops: cache-aware fused kernels instead of pointwise GEMM + depthwise launches
FUSE_RESBLOCK_MAX_C = 192


@dataclass
class ExecOptions:
    """Launch structure of ONE model's encoder or decoder (held by the module, passed to run_encoder / run_decoder): two models
    in one process may run different schedules side by side — nothing here is process-global.  Every setting computes the SAME
    bits (tests pin each against the others); the defaults are the fastest measured.  The arithmetic is the reference's fp32
    throughout: the split-bf16 decoder mode of rounds 2-4 and the launch variants that were measured slower (batched cache
    updates, four-wave C = 256 chains, the 256-column lockstep shape) are gone from the library (round 5; `profiles/r05_experiments.md`).

    stage_launches: a whole stage per launch — its residual blocks back to back per tile and its down-sampling (encoder) /
      up-sampling (decoder) layer as the last / first phase (`hilc_encoder_stage`, `hilc_decoder_stage`, `hilc_resblock_chain`).
      False: one launch per residual block and per layer (round 3's structure).
    wide_blocks: the residual blocks with C >= 256 in the fused kernel (one launch per block or stage); False: two
      depthwise-separable GEMM launches per block with the mid tensor in HBM (round 3's structure).
    decoder_stage_narrow: (with stage_launches) the decoder stages that run ONE long workgroup per CU — C = 192 / 96, and the
      form that holds one block behind the up-sampling phase (offline C = 768) — take their up-sampling layer into the launch
      (C = 96: and the closing conv).  False: up-sampling launch + chain (round 4's structure for those stages; `PipelinedHop`
      captured with it until round 6, when one launch per stage made it the slower choice there too).
    stream_defer_spec: streaming hop: the SpecBlock branches of stages >= 1 computed alone and added by the down-sampling
      epilogue in front (False: in-line, as in round 3)."""
    stage_launches: bool = True
    wide_blocks: bool = True
    decoder_stage_narrow: bool = True
    stream_defer_spec: bool = True


_DEFAULT_OPTIONS = ExecOptions()

# The HIP stream that takes the STFT front halves of the un-fused SpecBlocks while a hop is warmed up / captured
# (graph_step).  A context variable like ops._TIMER / ops._SCHED_WS: it belongs to the schedule being built on THIS thread,
# not to the model — two schedules built concurrently on one model from different threads do not see each other's stream.
_SIDE_STREAM: contextvars.ContextVar = contextvars.ContextVar("hilcodec_side_stream", default=None)


# Per-schedule overrides of a model's ExecOptions (a schedule object that wants another launch structure for ITS capture —
# PipelinedHop — says so here instead of writing into the module's options: context-local, like the side stream)
_OPT_OVERRIDES: contextvars.ContextVar = contextvars.ContextVar("hilcodec_exec_overrides", default=None)


@contextlib.contextmanager
def exec_overrides(**fields):
    token = _OPT_OVERRIDES.set(dict(_OPT_OVERRIDES.get() or {}, **fields))
    try:
        yield
    finally:
        _OPT_OVERRIDES.reset(token)


def _effective(opts: "ExecOptions") -> "ExecOptions":
    over = None if torch.compiler.is_compiling() else _OPT_OVERRIDES.get()
    if not over:
        return opts
    import dataclasses
    return dataclasses.replace(opts, **over)


@contextlib.contextmanager
def spectra_side_stream(stream):
    token = _SIDE_STREAM.set(stream)
    try:
        yield
    finally:
        _SIDE_STREAM.reset(token)


FUSE_SPECBLOCK = True     # long encoder stages (n_fft <= 256): STFT -> log-mag -> 1x1 conv -> += in one launch
STREAM_STAGE0 = True      # streaming hop: first conv + stage-0 SpecBlock as the opening phase of the first stage launch (hilc_encoder_stage0)


@dataclass
class ResBlockSpec:
    pw1_wt: Tensor
    dw1_w: Tensor
    dw1_b: Optional[Tensor]
    pw2_wt: Tensor
    dw2_w: Tensor
    dw2_b: Optional[Tensor]
    pre_scale: float        # (1 + idx*res_scale^2)^-1/2  (seanet.py:84), 1.0 in the streaming decoder
    out_scale: float        # res_scale * res_scale_param (seanet.py:144-148); 1.0 when merged into dw2
    pw1_packed: Optional[Tensor] = None   # MFMA-lane-order copies for the fused block (finalize_spec): C = 64 ... 768
    pw2_packed: Optional[Tensor] = None
    pw1_chain: Optional[Tensor] = None    # the same for a chain launch (streaming plans; another row split below C = 192, else the tensors above)
    pw2_chain: Optional[Tensor] = None


@dataclass
class SpecBlockSpec:
    basis_t: Tensor
    n_fft: int
    hop: int
    mean: float
    std: float
    normalize: bool         # False when (y-mean)/std is merged into the layer (streaming.py:321-344)
    wt: Tensor              # [n_fft/2+1, C]
    bias: Optional[Tensor]
    out_scale: float        # res_scale * scale_param; 1.0 when merged
    fused: Optional[tuple] = None   # (dft_packed, nyq_sin, pw_packed) of the one-launch kernel (finalize_spec)


@dataclass
class EncStageSpec:
    spec: SpecBlockSpec
    blocks: List[ResBlockSpec]
    down_in_scale: float    # (1 + n_res*res_scale^2)^-1/2
    down_pw_wt: Tensor
    down_dw_w: Tensor
    down_dw_b: Optional[Tensor]
    ratio: int
    down_lo: Optional[Tensor] = None      # the two column halves of down_pw_wt packed for the one-launch stage (finalize_spec; C = 64 / 128)
    down_hi: Optional[Tensor] = None


@dataclass
class EncoderSpec:
    pre_w: Tensor
    pre_b: Optional[Tensor]
    pre_in_scale: float     # 1/wav_std, or 1.0 when merged into pre_w (streaming.py:472-480)
    stages: List[EncStageSpec]
    spec_post: SpecBlockSpec
    post_dw_w: Tensor
    post_pw_wt: Tensor
    post_pw_b: Optional[Tensor]
    l2norm: bool
    dim: int
    wav_cache_len: int      # n_fft of spec_post - 1


@dataclass
class DecStageSpec:
    in_scale: float         # 1.0 for the first stage, (1 + n_res*res_scale^2)^-1/2 after
    tr_w: Tensor
    ratio: int
    pw_wt: Tensor
    pw_b: Optional[Tensor]
    blocks: List[ResBlockSpec]
    taps: Optional[Tensor] = None         # expanded tap table for strides without a vector tap path (finalize_spec)
    up_lo: Optional[Tensor] = None        # the two row halves of pw_wt packed for the one-launch stage (C = 192 / 96; streaming plans also C = 768)
    up_hi: Optional[Tensor] = None


@dataclass
class DecoderSpec:
    pre_pw_wt: Tensor
    pre_dw_w: Tensor
    pre_dw_b: Optional[Tensor]
    stages: List[DecStageSpec]
    post_in_scale: float
    post_w: Tensor
    post_b: Optional[Tensor]
    post_out_scale: float   # wav_std (offline: scales conv AND bias, seanet.py:464-466); 1.0 if merged into post_w
    tanh: bool


@dataclass
class RvqSpec:
    codebooks: Tensor       # [Nq, K, C]
    codebooks_t: Tensor     # [Nq, C, K]
    norms: Tensor           # [Nq, K]

    @property
    def num_quantizers(self) -> int:
        return self.codebooks.shape[0]


STREAM_WIDE_C = (256, 384, 512, 768)     # the wide blocks (csrc/resblock_kernel.h: NARROW shapes)


def finalize_block(rb: "ResBlockSpec", streaming: bool = False) -> "ResBlockSpec":
    c = rb.pw1_wt.shape[0]
    narrow = c <= FUSE_RESBLOCK_MAX_C and ops.resblock_supported(c, 4)
    if (rb.pw1_packed is None and rb.pw1_wt.device.type in ("cuda", "meta") and rb.pw1_wt.shape[1] == c
            and (narrow or c in STREAM_WIDE_C)):
        rb.pw1_packed = ops.resblock_pack(rb.pw1_wt)
        rb.pw2_packed = ops.resblock_pack(rb.pw2_wt)
    if rb.pw1_packed is not None and rb.pw1_chain is None and (c <= FUSE_RESBLOCK_MAX_C or c in STREAM_WIDE_C):
        same = ops.resblock_chain_row_classes(c, streaming) == (8 if c >= 512 else (4 if c >= 256 else (2 if c == 192 else 1)))     # hilc_resblock_pack_weights' own split
        rb.pw1_chain = rb.pw1_packed if same else ops.resblock_chain_pack(rb.pw1_wt, streaming)
        rb.pw2_chain = rb.pw2_packed if same else ops.resblock_chain_pack(rb.pw2_wt, streaming)
    return rb


def finalize_spec(spec, streaming: bool = False):
    """Device-side, one-off derived tables of an Encoder/DecoderSpec whose tensors already live on the GPU: packed
    pointwise weights of the fused residual blocks, expanded up-sampling taps.  (They used to be hidden caches keyed
    by data_ptr inside the op wrappers; as part of the spec they are plain graph inputs for torch.compile.)"""
    for st in spec.stages:
        sb = getattr(st, "spec", None)
        if (isinstance(sb, SpecBlockSpec) and sb.fused is None and sb.basis_t.device.type in ("cuda", "meta")
                and sb.n_fft in (64, 128, 256) and sb.wt.shape[1] == sb.n_fft):
            sb.fused = ops.spec_block_tables(sb.basis_t, sb.wt, sb.n_fft)
        for rb in st.blocks:
            finalize_block(rb, streaming)
        if (isinstance(st, EncStageSpec) and st.down_lo is None and st.down_pw_wt.device.type in ("cuda", "meta")
                and st.down_pw_wt.shape[0] in (64, 128, 256, 512) and st.down_pw_wt.shape[1] == 2 * st.down_pw_wt.shape[0]
                and all(rb.pw1_chain is not None for rb in st.blocks)):
            c = st.down_pw_wt.shape[0]
            st.down_lo = ops.resblock_chain_pack(st.down_pw_wt[:, :c].contiguous(), streaming)
            st.down_hi = ops.resblock_chain_pack(st.down_pw_wt[:, c:].contiguous(), streaming)
        if isinstance(st, DecStageSpec) and st.tr_w.device.type in ("cuda", "meta"):
            st.taps = ops.up_conv_taps(st.tr_w, st.ratio)
            c = st.pw_wt.shape[1]
            if (st.up_lo is None and st.pw_wt.shape[0] == 2 * c and {768: 8, 384: 5, 192: 4, 96: 2}.get(c, 0) == st.ratio
                    and all(rb.pw1_chain is not None for rb in st.blocks)):
                st.up_lo = ops.resblock_chain_pack(st.pw_wt[:c].contiguous(), streaming)
                st.up_hi = ops.resblock_chain_pack(st.pw_wt[c:].contiguous(), streaming)
    return spec


def spec_to(spec, device):
    """Copy of a spec (any of the dataclasses above, nested) with every tensor moved to `device`."""
    import dataclasses
    if isinstance(spec, Tensor):
        return spec.to(device)
    if isinstance(spec, (list, tuple)):
        return type(spec)(spec_to(v, device) for v in spec)
    if dataclasses.is_dataclass(spec):
        return type(spec)(**{f.name: spec_to(getattr(spec, f.name), device) for f in dataclasses.fields(spec)})
    return spec


def _to(dev, *ts):
    return [None if t is None else t.to(device=dev, dtype=torch.float32).contiguous() for t in ts]


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def _fusable(rb: ResBlockSpec, x: Tensor, streaming: bool = False, wide: Optional[bool] = None) -> bool:
    if wide is None:
        wide = streaming
    return (FUSE_RESBLOCK and rb.pw1_packed is not None and rb.dw1_w.shape[1] == 5 and rb.dw2_w.shape[1] == 5
            and rb.dw1_b is not None and rb.dw2_b is not None
            and (x.shape[1] <= FUSE_RESBLOCK_MAX_C or wide)
            and ops.resblock_supported(x.shape[1], x.shape[2], x.shape[0], streaming))


def _resblock(rb: ResBlockSpec, x: Tensor, caches: Optional[Sequence[Tensor]], new_caches: Optional[list],
              outs: Optional[Sequence[Tensor]] = None, opts: ExecOptions = _DEFAULT_OPTIONS) -> Tensor:
    """One residual block; streaming: `caches` = its two depthwise caches, `outs` = where the next hop's caches go
    (persistent state block) or None (fresh tensors, the reference's protocol)."""
    o0, o1 = (outs[0], outs[1]) if outs is not None else (None, None)
    if caches is None and _fusable(rb, x, False, opts.wide_blocks):
        # one launch per block: x is read once, y written once, everything else stays in LDS
        return ops.resblock(x, rb.pw1_packed, rb.dw1_w, rb.dw1_b, rb.pw2_packed, rb.dw2_w, rb.dw2_b,
                            rb.pre_scale, rb.out_scale)
    if caches is not None and x.shape[2] >= 4 and _fusable(rb, x, True, opts.wide_blocks):
        # streaming hop: same kernel, the two depthwise caches patch the first tile's halo columns; the wide blocks of a hop
        # (C = 256 ... 768) take its narrow-tile shapes
        y, cs = ops.resblock(x, rb.pw1_packed, rb.dw1_w, rb.dw1_b, rb.pw2_packed, rb.dw2_w, rb.dw2_b,
                             rb.pre_scale, rb.out_scale, hist=(caches[0], caches[1]), hist_out=outs)
        new_caches.extend(cs)
        return y
    if (caches is not None and FUSE_STREAM and ops.dws_conv_stream_profitable(x.shape[2], rb.dw1_w.shape[1], 1)
            and ops.dws_conv_stream_profitable(x.shape[2], rb.dw2_w.shape[1], 1)):
        # wide layers of a streaming hop (T <= 128): two launches, whole-clip tiles, caches read/written in the epilogue
        g, c0 = ops.dws_conv_stream(x, rb.pw1_wt, rb.dw1_w, rb.dw1_b, caches[0], in_scale=rb.pre_scale,
                                    in_elu=True, out_elu=True, hist_out=o0)
        y, c1 = ops.dws_conv_stream(g, rb.pw2_wt, rb.dw2_w, rb.dw2_b, caches[1], res=x,
                                    out_scale=rb.out_scale, hist_out=o1)
        new_caches.extend([c0, c1])
        return y
    if caches is None and FUSE_DWS and rb.dw1_w.shape[1] == 5 and rb.dw2_w.shape[1] == 5:
        # two launches per block: [ELU, pw, dw, ELU] and [pw, dw, *scale + shortcut]; the pointwise
        # outputs never leave LDS
        g = ops.dws_conv(x, rb.pw1_wt, rb.dw1_w, rb.dw1_b, in_scale=rb.pre_scale, in_elu=True, out_elu=True)
        return ops.dws_conv(g, rb.pw2_wt, rb.dw2_w, rb.dw2_b, res=x, out_scale=rb.out_scale)
    h = ops.pw_conv(x, rb.pw1_wt, in_scale=rb.pre_scale, in_elu=True)
    if caches is None:
        g = ops.dw_conv(h, rb.dw1_w, rb.dw1_b)
        h2 = ops.pw_conv(g, rb.pw2_wt, in_scale=1.0, in_elu=True)
        return ops.dw_conv(h2, rb.dw2_w, rb.dw2_b, res=x, out_scale=rb.out_scale)
    g, c0 = ops.dw_conv(h, rb.dw1_w, rb.dw1_b, hist=caches[0], want_hist=True, hist_out=o0)
    h2 = ops.pw_conv(g, rb.pw2_wt, in_scale=1.0, in_elu=True)
    y, c1 = ops.dw_conv(h2, rb.dw2_w, rb.dw2_b, res=x, out_scale=rb.out_scale, hist=caches[1], want_hist=True,
                        hist_out=o1)
    new_caches.extend([c0, c1])
    return y


def _stage_blocks(blocks: Sequence[ResBlockSpec], x: Tensor, caches: Optional[Sequence[Tensor]], ci: int, new_caches: Optional[list],
                  caches_out: Optional[Sequence[Tensor]], opts: ExecOptions) -> Tensor:
    """The residual blocks of one stage: ONE chain launch where the kernel exists (the blocks run back to back per tile, the
    activations between them stay in registers: `ops.resblock_chain`); otherwise block by block."""
    n = len(blocks)
    streaming = caches is not None
    if (streaming and FUSE_RESBLOCK and FUSE_STREAM and opts.stage_launches and n >= 2 and x.shape[2] >= 4
            and all(rb.pw1_chain is not None and rb.dw1_w.shape[1] == 5 and rb.dw2_w.shape[1] == 5 and rb.dw1_b is not None
                    and rb.dw2_b is not None for rb in blocks)
            and (x.shape[1] <= FUSE_RESBLOCK_MAX_C or opts.wide_blocks)
            and ops.resblock_chain_supported(x.shape[1], x.shape[2], n, x.shape[0])):
        y, cs = ops.resblock_chain(
            x, [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in blocks],
            [caches[ci + 2 * i: ci + 2 * i + 2] for i in range(n)],
            [caches_out[ci + 2 * i: ci + 2 * i + 2] for i in range(n)] if caches_out is not None else None)
        new_caches.extend(cs)
        return y
    if (not streaming and FUSE_RESBLOCK and opts.stage_launches and n >= 2
            and all(rb.pw1_chain is not None and _fusable(rb, x, False, opts.wide_blocks) for rb in blocks)
            and ops.resblock_chain_supported(x.shape[1], x.shape[2], n, x.shape[0], streaming=False)):
        return ops.resblock_chain(
            x, [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in blocks])
    for i, rb in enumerate(blocks):
        x = _resblock(rb, x, caches[ci + 2 * i: ci + 2 * i + 2] if streaming else None, new_caches,
                      caches_out[ci + 2 * i: ci + 2 * i + 2] if caches_out is not None else None, opts=opts)
    return x


def _stage_fusable_blocks(st: "DecStageSpec", x: Tensor, streaming: bool, partial: bool = True) -> int:
    """How many of a decoder stage's residual blocks run in ONE launch with its up-sampling layer (`ops.decoder_stage`): all of them, the
    first one (where LDS holds the carry slots / the halo form of one block only), or 0 = no stage launch for this shape."""
    if st.up_lo is None or st.pw_b is None or not st.blocks:
        return 0
    if not all(rb.pw1_chain is not None and rb.dw1_w.shape[1] == 5 and rb.dw2_w.shape[1] == 5 and rb.dw1_b is not None
               and rb.dw2_b is not None for rb in st.blocks):
        return 0
    c, t = st.pw_wt.shape[1], x.shape[2] * st.ratio
    for nb in ((len(st.blocks), 1) if partial else (len(st.blocks),)):
        if ops.decoder_stage_supported(c, t, nb, st.ratio, x.shape[0], streaming=streaming):
            return nb
    return 0


def _spec_fused(sb: SpecBlockSpec, wav: Tensor, wav_hist: Optional[Tensor]) -> bool:
    return bool(FUSE_SPECBLOCK and sb.fused is not None and (wav_hist is None or wav_hist.shape[-1] >= sb.n_fft - 1)
                and ops.spec_block_profitable(sb.n_fft, sb.hop, sb.wt.shape[1], wav.shape[2]))


def _spec_block(sb: SpecBlockSpec, x: Tensor, wav: Tensor, wav_hist: Optional[Tensor]) -> Tensor:
    if _spec_fused(sb, wav, wav_hist):
        return ops.spec_block(wav, sb.fused[0], sb.fused[1], sb.fused[2], sb.bias, x, sb.n_fft, sb.hop, sb.mean, sb.std,
                              sb.normalize, sb.out_scale, hist=wav_hist)
    s = ops.stft_logmag(wav, sb.basis_t, sb.n_fft, sb.hop, sb.mean, sb.std, sb.normalize, hist=wav_hist)
    return ops.pw_conv(s, sb.wt, sb.bias, res=x, out_scale=sb.out_scale)


def _spec_branch(sb: SpecBlockSpec, wav: Tensor, wav_hist: Optional[Tensor]) -> Tensor:
    """The SpecBlock's branch ALONE: out_scale * (W logspec(wav) + bias) `[B, C, T_f]` — what `x.add_()` adds
    (`seanet.py:220-246`; streaming, merged: `streaming.py:346-366`).  It depends on the waveform only."""
    if _spec_fused(sb, wav, wav_hist):
        return ops.spec_block(wav, sb.fused[0], sb.fused[1], sb.fused[2], sb.bias, None, sb.n_fft, sb.hop, sb.mean, sb.std,
                              sb.normalize, sb.out_scale, hist=wav_hist)
    s = ops.stft_logmag(wav, sb.basis_t, sb.n_fft, sb.hop, sb.mean, sb.std, sb.normalize, hist=wav_hist)
    return ops.pw_conv(s, sb.wt, sb.bias, out_scale=sb.out_scale)


def _early_branches(todo: Sequence[SpecBlockSpec], wav: Tensor, wav_hist: Optional[Tensor], side) -> Optional[dict]:
    """Streaming hop inside a captured graph: the SpecBlock branches of the later stages depend on the waveform only, so they
    are computed on `side` (`spectra_side_stream`) beside the first encoder stages (their launches fill idle CUs instead of
    standing in the chain); the consumer waits for each one's event.  Same launches, same results."""
    if side is None or torch.compiler.is_compiling() or not wav.is_cuda or not todo:
        return None
    main = torch.cuda.current_stream(wav.device)
    capturing = torch.cuda.is_current_stream_capturing()
    early = {}
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for sb in todo:
            r = _spec_branch(sb, wav, wav_hist)
            if not capturing:
                r.record_stream(main)
            done = torch.cuda.Event()
            done.record(side)
            early[id(sb)] = (r, done)
    return early


def _contig(caches: Optional[Sequence[Tensor]]):
    return None if caches is None else [c.contiguous() for c in caches]


# The linear-addressing GEMM cores and the fused residual block address an activation tensor with 32-bit byte offsets
# (csrc/gemm_epilogues.h: lin_ok, csrc/resblock.hip): a tensor of 4 GiB or more silently took the generic core / two
# launches (10-20 % slower).  Clips are independent and results batch-invariant bit for bit (tests/test_gpu_fullsize.py),
# so an offline batch whose LARGEST activation would reach the limit is run as equal clip chunks that each stay below it.
OFFSET_LIMIT_BYTES = 1 << 32


def _clip_chunks(batch: int, per_clip_elems: int) -> List[Tuple[int, int]]:
    """[(lo, hi)] — one range if the whole batch fits 32-bit byte offsets, else the fewest equal chunks that do"""
    most = max(1, (OFFSET_LIMIT_BYTES - 1) // (4 * max(1, per_clip_elems)))
    if batch <= most:
        return [(0, batch)]
    n = -(-batch // most)
    size = -(-batch // n)
    return [(lo, min(batch, lo + size)) for lo in range(0, batch, size)]


def _encoder_clip_elems(es: "EncoderSpec", T: int) -> int:
    """largest [C, T_s] activation of one clip that a launch of the offline encoder reads or writes"""
    best, t = es.pre_w.shape[0] * T, T
    for st in es.stages:
        best = max(best, st.down_pw_wt.shape[0] * t)
        t = -(-t // st.ratio)
        best = max(best, st.down_dw_w.shape[0] * t)
    return best


def _decoder_clip_elems(ds: "DecoderSpec", F: int) -> int:
    best, t = ds.pre_dw_w.shape[0] * F, F
    for st in ds.stages:
        best = max(best, st.tr_w.shape[0] * t)
        t *= st.ratio
        best = max(best, st.pw_wt.shape[1] * t)
    return best


def run_encoder(es: EncoderSpec, wav: Tensor, caches: Optional[Sequence[Tensor]] = None,
                channel_last_out: bool = False, caches_out: Optional[Sequence[Tensor]] = None,
                opts: ExecOptions = _DEFAULT_OPTIONS):
    """wav `[B,1,T]` -> z `[B,dim,ceil(T/hop)]` (or `[B,T',dim]`), and the new cache list if streaming.
    `caches_out` (streaming, optional): persistent buffers, same shapes as `caches` and distinct from them, that
    receive the next hop's caches (ping-pong state block); without it every cache is a fresh tensor, which is the
    reference's protocol (`streaming.py:482-517`: cache_out never aliases cache_in)."""
    if wav.dim() != 3 or wav.shape[1] != 1:
        raise RuntimeError(f"expected [B,1,T] waveform, got {tuple(wav.shape)}")
    opts = _effective(opts)
    wav = wav.contiguous().float()
    streaming = caches is not None
    if not streaming and not torch.compiler.is_compiling():
        chunks = _clip_chunks(wav.shape[0], _encoder_clip_elems(es, wav.shape[2]))
        if len(chunks) > 1:
            return torch.cat([run_encoder(es, wav[lo:hi], None, channel_last_out, None, opts) for lo, hi in chunks], dim=0)
    caches = _contig(caches)
    new_caches: Optional[list] = [] if streaming else None

    def out(i):
        return caches_out[i] if caches_out is not None else None

    wav_hist = None
    ci = 0
    if streaming:
        wav_hist = caches[0]
        new_caches.append(ops.tail(wav, wav_hist, es.wav_cache_len, out=out(0)))
        ci = 1
    sb0 = es.stages[0].spec
    fuse_pre = (FUSE_SPECBLOCK and sb0.fused is not None and sb0.n_fft == 64 and sb0.hop == 1
                and es.pre_w.shape == (64, 5) and ops.spec_block_supported(64, 1, 64, wav.shape[2])
                and (wav_hist is None or wav_hist.shape[-1] >= 63))
    st0 = es.stages[0]
    fuse_stage0 = (fuse_pre and (not streaming or (FUSE_STREAM and STREAM_STAGE0)) and FUSE_RESBLOCK and opts.stage_launches and st0.down_lo is not None and st0.down_dw_b is not None
                   and st0.down_dw_w.shape[1] == 2 * st0.ratio and wav.shape[2] % st0.ratio == 0
                   and all(rb.pw1_chain is not None and rb.dw1_w.shape[1] == 5 and rb.dw2_w.shape[1] == 5 and rb.dw1_b is not None
                           and rb.dw2_b is not None for rb in st0.blocks)
                   and ops.encoder_stage0_supported(wav.shape[2], len(st0.blocks), st0.ratio, 64, 1, es.pre_w.shape[1], wav.shape[0], streaming))
    blocks0 = [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in st0.blocks] if fuse_stage0 else None
    if fuse_stage0 and not streaming:
        # first conv + first SpecBlock + the whole first stage in one launch: neither the [64 x T] tensor in front of the stage nor the one
        # behind its blocks ever exists
        x = ops.encoder_stage0(
            wav, (sb0.fused[0], sb0.fused[1], sb0.fused[2], sb0.bias, es.pre_w, es.pre_b, es.pre_in_scale, sb0.mean, sb0.std, sb0.normalize,
                  sb0.out_scale),
            blocks0, (st0.down_lo, st0.down_hi, st0.down_dw_w, st0.down_dw_b, st0.down_in_scale, st0.ratio))
    elif fuse_stage0:
        x = None      # launched below, once the side branch that computes stage 1's SpecBlock has been forked
    elif fuse_pre:
        # first conv + first SpecBlock in one launch: the [64 x T] tensor between them never exists
        x = ops.spec_block_conv_pre(wav, sb0.fused[0], sb0.fused[1], sb0.fused[2], sb0.bias, es.pre_w, es.pre_b,
                                    es.pre_in_scale, 64, 1, sb0.mean, sb0.std, sb0.normalize, sb0.out_scale, hist=wav_hist)
    else:
        x = ops.conv_pre(wav, es.pre_w, es.pre_b, in_scale=es.pre_in_scale, hist=wav_hist)
    # Streaming hop: the SpecBlock branch of stage s + 1 (and spec_post's) is `x.add_(branch)` right after stage s's
    # down-sampling layer (`streaming.py:497-511`), and the branch depends on the waveform only.  It is computed on its own —
    # inside a captured graph beside the earlier stages, on the side stream — and ADDED BY THE DOWN-SAMPLING LAYER'S EPILOGUE
    # (`res`): one launch and one read + write of x less per stage on the critical path, the same two roundings in the same
    # order (`fadd(down, branch)`), bit-identical to the in-line form.
    later = [st.spec for st in es.stages[1:]] + [es.spec_post]
    # (offline the same deferral was measured in round 5: 73.8 -> 75.2 ms per step, same box — the branch-only launches write what the
    #  in-line ones read-modify-write, and the stage launches gain a `res` read: kept in-line)
    defer = streaming and FUSE_STREAM and opts.stream_defer_spec
    side = None if torch.compiler.is_compiling() else _SIDE_STREAM.get()          # (a tracing compiler cannot read a ContextVar; it never forks streams)
    early = _early_branches(later, wav, wav_hist, side) if defer else None

    def branch_of(sb):
        if early is not None and id(sb) in early:
            r, done = early[id(sb)]
            torch.cuda.current_stream(wav.device).wait_event(done)
            return r
        return _spec_branch(sb, wav, wav_hist)

    for si, st in enumerate(es.stages):
        if fuse_stage0 and si == 0:
            if streaming:
                # a hop's first conv + first SpecBlock + first stage (streaming.py:490-511) in one launch, with the waveform cache in front
                # of every stream's t = 0, the blocks' caches and the down-sampling layer's
                nb0 = len(st.blocks)
                x, cs_, c = ops.encoder_stage0(
                    wav, (sb0.fused[0], sb0.fused[1], sb0.fused[2], sb0.bias, es.pre_w, es.pre_b, es.pre_in_scale, sb0.mean, sb0.std,
                          sb0.normalize, sb0.out_scale),
                    blocks0, (st.down_lo, st.down_hi, st.down_dw_w, st.down_dw_b, st.down_in_scale, st.ratio),
                    res=branch_of(later[0]) if defer else None,
                    hist=[caches[ci + 2 * i: ci + 2 * i + 2] for i in range(nb0)],
                    hist_out=[caches_out[ci + 2 * i: ci + 2 * i + 2] for i in range(nb0)] if caches_out is not None else None,
                    down_hist=caches[ci + 2 * nb0], down_hist_out=out(ci + 2 * nb0), wav_hist=wav_hist)
                new_caches.extend(cs_)
                new_caches.append(c)
            ci += 2 * len(st.blocks) + 1
            continue
        if not (fuse_pre and si == 0) and not (defer and si > 0):
            x = _spec_block(st.spec, x, wav, wav_hist)
        nxt = later[si] if defer else None
        nb = len(st.blocks)
        if (FUSE_RESBLOCK and opts.stage_launches and st.down_lo is not None and st.down_dw_b is not None
                and st.down_dw_w.shape[1] == 2 * st.ratio and x.shape[2] % st.ratio == 0 and (not streaming or FUSE_STREAM)
                and all(rb.pw1_chain is not None and rb.dw1_w.shape[1] == 5 and rb.dw2_w.shape[1] == 5 and rb.dw1_b is not None
                        and rb.dw2_b is not None for rb in st.blocks)
                and (x.shape[1] <= FUSE_RESBLOCK_MAX_C or opts.wide_blocks)
                and ops.encoder_stage_supported(x.shape[1], x.shape[2], nb, st.ratio, x.shape[0], streaming)):
            # the whole stage — its residual blocks and its down-sampling layer — is one launch; the stage's output never reaches HBM
            blocks = [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in st.blocks]
            down = (st.down_lo, st.down_hi, st.down_dw_w, st.down_dw_b, st.down_in_scale, st.ratio)
            if streaming:
                x, cs_, c = ops.encoder_stage(
                    x, blocks, down, hist=[caches[ci + 2 * i: ci + 2 * i + 2] for i in range(nb)],
                    hist_out=[caches_out[ci + 2 * i: ci + 2 * i + 2] for i in range(nb)] if caches_out is not None else None,
                    down_hist=caches[ci + 2 * nb], down_hist_out=out(ci + 2 * nb), res=branch_of(nxt) if defer else None)
                new_caches.extend(cs_)
                new_caches.append(c)
            else:
                x = ops.encoder_stage(x, blocks, down)
            ci += 2 * nb + 1
            continue
        x = _stage_blocks(st.blocks, x, caches, ci, new_caches, caches_out, opts)
        ci += 2 * len(st.blocks)
        # (a shortcut the fused layer cannot take — a long hop at a stride above 8 — falls through to pointwise GEMM + depthwise conv, which can)
        if streaming and FUSE_STREAM and ops.dws_conv_stream_profitable(x.shape[2], st.down_dw_w.shape[1], st.ratio, defer, x.shape[0],
                                                                         st.down_dw_w.shape[0]):
            x, c = ops.dws_conv_stream(x, st.down_pw_wt, st.down_dw_w, st.down_dw_b, caches[ci], res=branch_of(nxt) if defer else None,
                                       stride=st.ratio, in_scale=st.down_in_scale, in_elu=True, hist_out=out(ci))
            new_caches.append(c)
        elif streaming:
            h = ops.pw_conv(x, st.down_pw_wt, in_scale=st.down_in_scale, in_elu=True)
            x, c = ops.dw_conv(h, st.down_dw_w, st.down_dw_b, res=branch_of(nxt) if defer else None, stride=st.ratio, hist=caches[ci],
                               want_hist=True, hist_out=out(ci))
            new_caches.append(c)
        elif FUSE_DWS and st.down_dw_w.shape[1] == 2 * st.ratio:
            x = ops.dws_conv(x, st.down_pw_wt, st.down_dw_w, st.down_dw_b, stride=st.ratio,
                             in_scale=st.down_in_scale, in_elu=True)
        else:
            h = ops.pw_conv(x, st.down_pw_wt, in_scale=st.down_in_scale, in_elu=True)
            x = ops.dw_conv(h, st.down_dw_w, st.down_dw_b, stride=st.ratio)
        ci += 1
    if not defer:
        x = _spec_block(es.spec_post, x, wav, wav_hist)
    if streaming:
        h, c = ops.dw_conv(x, es.post_dw_w, None, in_elu=True, hist=caches[ci], want_hist=True, hist_out=out(ci))
        new_caches.append(c)
    else:
        h = ops.dw_conv(x, es.post_dw_w, None, in_elu=True)
    h = ops.pw_conv(h, es.post_pw_wt, es.post_pw_b)
    if es.l2norm:
        z = ops.l2norm(h, eps=1e-12, scale=float(es.dim) ** 0.5, channel_last_out=channel_last_out)
    else:
        z = h.transpose(1, 2).contiguous() if channel_last_out else h
    return (z, new_caches) if streaming else z


def run_decoder(ds: DecoderSpec, q: Tensor, caches: Optional[Sequence[Tensor]] = None,
                caches_out: Optional[Sequence[Tensor]] = None, opts: ExecOptions = _DEFAULT_OPTIONS):
    """q `[B,dim,F]` (channel-major) -> wav `[B,1,F*hop]`, and the new cache list if streaming (`caches_out` as in
    run_encoder)."""
    streaming = caches is not None
    opts = _effective(opts)
    caches = _contig(caches)
    new_caches: Optional[list] = [] if streaming else None

    def out(i):
        return caches_out[i] if caches_out is not None else None

    q = q.contiguous().float()
    if not streaming and not torch.compiler.is_compiling():
        chunks = _clip_chunks(q.shape[0], _decoder_clip_elems(ds, q.shape[2]))
        if len(chunks) > 1:
            return torch.cat([run_decoder(ds, q[lo:hi], None, None, opts) for lo, hi in chunks], dim=0)
    ci = 0
    if streaming and FUSE_STREAM and ops.dws_conv_stream_profitable(q.shape[2], ds.pre_dw_w.shape[1], 1):
        x, c = ops.dws_conv_stream(q, ds.pre_pw_wt, ds.pre_dw_w, ds.pre_dw_b, caches[0], hist_out=out(0))
        new_caches.append(c)
    elif streaming:
        h = ops.pw_conv(q, ds.pre_pw_wt)
        x, c = ops.dw_conv(h, ds.pre_dw_w, ds.pre_dw_b, hist=caches[0], want_hist=True, hist_out=out(0))
        new_caches.append(c)
    elif FUSE_DWS and ds.pre_dw_w.shape[1] == 5:
        x = ops.dws_conv(q, ds.pre_pw_wt, ds.pre_dw_w, ds.pre_dw_b)
    else:
        h = ops.pw_conv(q, ds.pre_pw_wt)
        x = ops.dw_conv(h, ds.pre_dw_w, ds.pre_dw_b)
    ci = 1
    for st in ds.stages:
        fused_up = FUSE_UPSAMPLE and (x.shape[2] * st.ratio) % 4 == 0
        if (streaming and FUSE_STREAM and FUSE_RESBLOCK and opts.stage_launches
                and (opts.decoder_stage_narrow if st.pw_wt.shape[1] <= FUSE_RESBLOCK_MAX_C else opts.wide_blocks)
                and _stage_fusable_blocks(st, x, True, opts.decoder_stage_narrow) > 0):
            # the whole stage — up-sampling layer and residual blocks — is one launch; the tensor between them never exists
            # (C = 384: the up-sampling layer and the FIRST block; the other two follow as launches of their own)
            nb = _stage_fusable_blocks(st, x, True, opts.decoder_stage_narrow)
            up_w = st.tr_w if st.taps is None else st.taps
            if (st is ds.stages[-1] and nb == len(st.blocks) and ds.post_w.dim() == 2 and ds.post_w.shape[0] == st.pw_wt.shape[1]
                    and ops.decoder_stage_post_supported(st.pw_wt.shape[1], x.shape[2] * st.ratio, nb, st.ratio, ds.post_w.shape[1])):
                # the LAST stage and the closing conv (streaming.py:639-648) in one launch: the stage's output — the hop's largest tensor —
                # is neither written nor read; the conv's cache is read at a stream's t = 0 and written by its last column group
                cp = ci + 1 + 2 * nb
                wav, cs_, c, cpost = ops.decoder_stage_post(
                    x, (up_w, st.up_lo, st.up_hi, st.pw_b, st.in_scale, st.ratio),
                    [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in st.blocks],
                    (ds.post_w, ds.post_b, ds.post_in_scale, ds.post_out_scale, ds.tanh),
                    [caches[ci + 1 + 2 * i: ci + 3 + 2 * i] for i in range(nb)], caches[ci], caches[cp],
                    [caches_out[ci + 1 + 2 * i: ci + 3 + 2 * i] for i in range(nb)] if caches_out is not None else None, out(ci), out(cp))
                new_caches.append(c)
                new_caches.extend(cs_)
                new_caches.append(cpost)
                return wav, new_caches
            x, cs_, c = ops.decoder_stage(
                x, (up_w, st.up_lo, st.up_hi, st.pw_b, st.in_scale, st.ratio),
                [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in st.blocks[:nb]],
                [caches[ci + 1 + 2 * i: ci + 3 + 2 * i] for i in range(nb)], caches[ci],
                [caches_out[ci + 1 + 2 * i: ci + 3 + 2 * i] for i in range(nb)] if caches_out is not None else None, out(ci))
            new_caches.append(c)
            new_caches.extend(cs_)
            ci += 1 + 2 * nb
            if nb < len(st.blocks):
                x = _stage_blocks(st.blocks[nb:], x, caches, ci, new_caches, caches_out, opts)
                ci += 2 * (len(st.blocks) - nb)
            continue
        elif streaming and FUSE_STREAM and (x.shape[2] * st.ratio) % 4 == 0:
            x, c = ops.up_conv(x, st.tr_w, st.pw_wt, st.pw_b, st.ratio, in_scale=st.in_scale, in_elu=True,
                               hist=caches[ci], want_hist=True, taps=st.taps, hist_out=out(ci))
            new_caches.append(c)
        elif streaming:
            u, c = ops.dw_convtr(x, st.tr_w, st.ratio, hist=caches[ci], want_hist=True,
                                 in_scale=st.in_scale, in_elu=True, hist_out=out(ci))
            new_caches.append(c)
            x = ops.pw_conv(u, st.pw_wt, st.pw_b)
        elif (not streaming and FUSE_RESBLOCK and FUSE_UPSAMPLE and opts.stage_launches
              and (opts.decoder_stage_narrow if st.pw_wt.shape[1] <= FUSE_RESBLOCK_MAX_C else opts.wide_blocks)
              and _stage_fusable_blocks(st, x, False, opts.decoder_stage_narrow) > 0):
            # (the widest stage, C = 768: its carry slots leave LDS room for ONE block behind the up-sampling phase; the other blocks follow)
            nb = _stage_fusable_blocks(st, x, False, opts.decoder_stage_narrow)
            blocks_ = [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in st.blocks[:nb]]
            if (st is ds.stages[-1] and nb == len(st.blocks) and ds.post_w.dim() == 2 and ds.post_w.shape[0] == st.pw_wt.shape[1]
                    and ops.decoder_stage_post_supported(st.pw_wt.shape[1], x.shape[2] * st.ratio, nb, st.ratio, ds.post_w.shape[1])):
                # the LAST stage and the closing conv (seanet.py:453-476) in one launch: the stage's [B, C, T] output — the step's largest
                # tensor — is neither written nor read
                return ops.decoder_stage_post(
                    x, (st.tr_w if st.taps is None else st.taps, st.up_lo, st.up_hi, st.pw_b, st.in_scale, st.ratio), blocks_,
                    (ds.post_w, ds.post_b, ds.post_in_scale, ds.post_out_scale, ds.tanh))
            x = ops.decoder_stage(
                x, (st.tr_w if st.taps is None else st.taps, st.up_lo, st.up_hi, st.pw_b, st.in_scale, st.ratio),
                [(rb.pw1_chain, rb.dw1_w, rb.dw1_b, rb.pw2_chain, rb.dw2_w, rb.dw2_b, rb.pre_scale, rb.out_scale) for rb in st.blocks[:nb]])
            if nb < len(st.blocks):
                x = _stage_blocks(st.blocks[nb:], x, None, 0, None, None, opts)
            ci += 1 + 2 * len(st.blocks)
            continue
        elif fused_up:
            # the up-sampled tensor only exists inside the GEMM's loader
            x = ops.up_conv(x, st.tr_w, st.pw_wt, st.pw_b, st.ratio, in_scale=st.in_scale, in_elu=True, taps=st.taps)
        else:
            u = ops.dw_convtr(x, st.tr_w, st.ratio, in_scale=st.in_scale, in_elu=True)
            x = ops.pw_conv(u, st.pw_wt, st.pw_b)
        ci += 1
        x = _stage_blocks(st.blocks, x, caches, ci, new_caches, caches_out, opts)
        ci += 2 * len(st.blocks)
    if streaming:
        wav, c = ops.conv_post(x, ds.post_w, ds.post_b, in_scale=ds.post_in_scale, in_elu=True,
                               out_scale=ds.post_out_scale, do_tanh=ds.tanh, hist=caches[ci], want_hist=True,
                               hist_out=out(ci))
        new_caches.append(c)
        return wav, new_caches
    return ops.conv_post(x, ds.post_w, ds.post_b, in_scale=ds.post_in_scale, in_elu=True,
                         out_scale=ds.post_out_scale, do_tanh=ds.tanh)


def encoder_cache_shapes(es: EncoderSpec) -> List[Tuple[int, int]]:
    """(channels, length) per cache, order of `Encoder.initialize_cache` (streaming.py:458-470)."""
    out = [(1, es.wav_cache_len)]
    for st in es.stages:
        for rb in st.blocks:
            c = rb.dw1_w.shape[0]
            out += [(c, rb.dw1_w.shape[1] - 1), (c, rb.dw2_w.shape[1] - 1)]
        out.append((st.down_dw_w.shape[0], st.down_dw_w.shape[1] - st.ratio))
    out.append((es.post_dw_w.shape[0], es.post_dw_w.shape[1] - 1))
    return out


def decoder_cache_shapes(ds: DecoderSpec) -> List[Tuple[int, int]]:
    """Order of `Decoder.initialize_cache` (streaming.py:599-607)."""
    out = [(ds.pre_dw_w.shape[0], ds.pre_dw_w.shape[1] - 1)]
    for st in ds.stages:
        out.append((st.tr_w.shape[0], 1))
        for rb in st.blocks:
            c = rb.dw1_w.shape[0]
            out += [(c, rb.dw1_w.shape[1] - 1), (c, rb.dw2_w.shape[1] - 1)]
    out.append((ds.post_w.shape[0], ds.post_w.shape[1] - 1))
    return out
