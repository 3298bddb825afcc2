// A stage's residual blocks in ONE launch (kernel: resblock_kernel.h, NB > 1): hilc_resblock_chain.
//
// The reference runs the blocks of a stage one after the other (seanet.py:316-330 `self.blocks[i]`, decoder: the
// SEANetResnetBlocks after each up-sampling layer, seanet.py:437-452; streaming.py:497-503, 633-639).  Here the 2 (encoder) or 3
// (decoder) blocks of a stage are one launch: per tile the blocks run back to back and a block's output stays in registers as
// the next block's input and shortcut — the activations between the blocks of a stage never reach HBM, and a streaming hop
// loses 2 of 3 launch boundaries per stage (at 1024 streams a launch is 5-10 tiles per workgroup: its ends are 10-20 % of it).
// Same products in the same order as hilc_resblock / hilc_resblock_stream: bit-identical (tests/test_gpu_ops.py).
#include "resblock_launch.h"

namespace {
constexpr bool chain_width(int C) { return C == 64 || C == 96 || C == 128 || C == 192; }
}

#ifdef HILC_DEBUG_STAMPS      // tools/stage_phase_times.py builds its own copy of the library with this; never in the product
extern "C" void hilc_debug_set_chain_stamp_buffer(unsigned long long* p) { g_dbg = p; }
#define HILC_CHAIN_DBG g_dbg
#else
#define HILC_CHAIN_DBG nullptr
#endif

extern "C" int hilc_resblock_chain_supported(int C, int T, int nblk, int streaming) {
  if (nblk < 2 || nblk > MAXBLK || T <= 0 || T % 4 != 0) return 0;
  // the instantiations hold the carry slots / tap tables of 2 blocks at the encoder's widths and of 3 at the decoder's
  // (offline C = 768: the carry slots of a second block do not fit beside the 32-column tile — 768 x (32 + 16) x 4 B + 36 KB of taps)
  const int max_blocks = (C == 96 || C == 192 || (streaming ? C == 768 : C == 384)) ? 3 : 2;
  if (nblk > max_blocks) return 0;
  if (chain_width(C)) return 1;
  if (!streaming) return C == 256 || C == 384 || C == 512;          // the wide blocks in the carry form (NARROW shapes)
  return (C == 512 || C == 768) && 32 % T == 0;
}

// packed pointwise weights of a chain launch: the 8-wave shapes split the rows in two classes also below C = 192
// (offline: C = 64 keeps four waves = one row class; the other widths as in the streaming form)
extern "C" int hilc_resblock_chain_row_classes_offline(int C) {
  static_assert(Cfg<64, false, false, 2, false>::RH == 1 && Cfg<96, false, false, 3, false>::RH == 1 &&
                Cfg<128, false, false, 2, true>::RH == 2 && Cfg<192, false, false, 3, false>::RH == 2 &&
                Cfg<256, false, false, 2, false>::RH == 4 && Cfg<384, false, false, 3, false>::RH == 4 &&
                Cfg<512, false, false, 2, false>::RH == 8, "packed layout");
  static_assert(Cfg<768, false, false, 1, false, -8>::RH == 8, "packed layout");
  if (C == 256 || C == 384) return 4;
  if (C == 512 || C == 768) return 8;
  return C == 128 || C == 192 ? 2 : (chain_width(C) ? 1 : 0);
}

extern "C" int hilc_resblock_chain_row_classes(int C) {
  static_assert(Cfg<64, true, true, 2, false>::RH == 1 && Cfg<96, true, true, 3, false>::RH == 1 &&
                Cfg<128, true, true, 2, true>::RH == 2 && Cfg<192, true, true, 3, false>::RH == 2 &&
                Cfg<512, true, false, 2, false>::RH == 8 && Cfg<768, true, false, 3, false>::RH == 8, "packed layout");
  // the hop's C = 256 / C = 384 stages: 32-column carry tiles, every wave one 32-row block (8 / 12 waves)
  static_assert(Cfg<256, true, true, 2, false, 5>::RH == 8 && Cfg<384, true, true, 3, false, -5>::RH == 12 && Cfg<512, true, false, 2, false, 8>::RH == 8, "packed layout");
  if (C == 256) return 8;
  if (C == 384) return 12;
  return C >= 512 ? 8 : ((C == 96 || C == 64) ? 1 : (chain_width(C) ? 2 : 0));
}

namespace {
int fill_blocks(ResArgs& a, const hilc_resblock_params* blocks, int nblk) {
  for (int i = 0; i < nblk; ++i) {
    const hilc_resblock_params& p = blocks[i];
    if (!p.w1t || !p.dw1_w || !p.dw1_b || !p.w2t || !p.dw2_w || !p.dw2_b) return HILC_ERR_NULL;
    if ((p.hist1 && p.hist1 == p.hist1_out) || (p.hist2 && p.hist2 == p.hist2_out)) return HILC_ERR_UNSUPPORTED;
    ResBlk& b = a.blk[i];
    b.w1t = p.w1t; b.dw1_w = p.dw1_w; b.dw1_b = p.dw1_b; b.w2t = p.w2t; b.dw2_w = p.dw2_w; b.dw2_b = p.dw2_b;
    b.hist1 = p.hist1; b.hist2 = p.hist2; b.hist1_out = p.hist1_out; b.hist2_out = p.hist2_out;
    b.pre_scale = p.pre_scale; b.out_scale = p.out_scale;
  }
  for (int i = nblk; i < MAXBLK; ++i) a.blk[i] = a.blk[0];
  return HILC_OK;
}
}  // namespace

extern "C" int hilc_resblock_chain(const float* x, float* y, const hilc_resblock_params* blocks, int nblk, int streaming,
                                   int B, int C, int T, void* stream) {
  if (!x || !y || !blocks) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (!hilc_resblock_chain_supported(C, T, nblk, streaming)) return HILC_ERR_UNSUPPORTED;
  if (x == y || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return HILC_ERR_UNSUPPORTED;
  if (streaming && (long)B * C * T * 4 >= (1L << 32)) return HILC_ERR_UNSUPPORTED;   // 32-bit flat column index / byte offsets
  ResArgs a{};
  a.x = x; a.y = y; a.T = T; a.tiles = 0; a.nblk = nblk; a.sched = nullptr; a.dbg = HILC_CHAIN_DBG;
  a.dn = ResDown{};
  a.up = ResUp{};
  a.post = ResPost{};
  if (const int rc = fill_blocks(a, blocks, nblk)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (!streaming) {        // offline: the carry form's contiguous runs (hilc_resblock), the blocks of the stage back to back per tile
    switch (C) {
      case 64: return launch_chain<64, false, 2, false>(a, B, s);
      case 96: return launch_chain<96, false, 3, false>(a, B, s);
      case 128: return launch_chain<128, false, 2, true>(a, B, s);
      case 192: return launch_chain<192, false, 3, false>(a, B, s);
      case 256: return launch_chain<256, false, 2, false>(a, B, s);
      case 384: return launch_chain<384, false, 3, false>(a, B, s);
      case 512: return launch_chain<512, false, 2, false>(a, B, s);
      default: return HILC_ERR_UNSUPPORTED;
    }
  }
  switch (C) {
    case 64: return launch_chain<64, true, 2, false>(a, B, s);         // four waves, two workgroups per CU (eight waves: 62 instead of 68 TF)
    case 96: return launch_chain<96, true, 3, false>(a, B, s);         // 3 row blocks do not split in two classes: 4 waves, two workgroups per CU
    case 128: return launch_chain<128, true, 2, true>(a, B, s);
    case 192: return launch_chain<192, true, 3, false>(a, B, s);
    case 512: return launch_chain<512, true, 2, false>(a, B, s);
    case 768: return launch_chain<768, true, 3, false>(a, B, s);
    default: return HILC_ERR_UNSUPPORTED;
  }
}

// ---- an ENCODER STAGE in one launch: its residual blocks and its down-sampling layer -------------------------------------------
// seanet.py:316-339 (`self.blocks[i]`, then `self.downsample[i]` = [Scale, ELU, 1x1 conv C -> 2C (no bias), depthwise conv k = 2r
// stride r]); streaming.py:497-511.  == hilc_resblock_chain followed by hilc_dws_conv(_stream) with the same arguments, bit for bit.
extern "C" int hilc_encoder_stage_supported(int C, int T, int nblk, int stride, int streaming) {
  if (nblk < 1 || nblk > 2 || T <= 0 || T % 4 != 0 || stride <= 0 || T % stride != 0) return 0;
  if ((C == 64 && stride == 2) || (C == 128 && stride == 4)) return 1;
  // the wide stages (narrow-tile shapes; the strided conv's outputs do not align with 4-column lanes there).  Offline: the carry form.
  // A hop (round 6): C = 256 on 32-column carry tiles (whole frames of 40 columns per stream: a stream's last r columns never straddle
  // two tiles), C = 512 on whole-stream tiles (T = 8, 16, 32)
  if (!streaming) return (C == 256 && stride == 5) || (C == 512 && stride == 8);
  return (C == 256 && stride == 5 && T % 40 == 0) || (C == 512 && stride == 8 && T % 8 == 0 && 32 % T == 0);
}

extern "C" int hilc_encoder_stage(const float* x, const hilc_resblock_params* blocks, int nblk, const hilc_down_params* down,
                                  int streaming, int B, int C, int T, void* stream) {
  if (!x || !blocks || !down) return HILC_ERR_NULL;
  if (!down->w_lo || !down->w_hi || !down->dw_w || !down->dw_b || !down->y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (!hilc_encoder_stage_supported(C, T, nblk, down->stride, streaming)) return HILC_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(down->y) & 15) ||
      (reinterpret_cast<uintptr_t>(down->res) & 15) || (reinterpret_cast<uintptr_t>(down->hist) & 15) ||
      (reinterpret_cast<uintptr_t>(down->hist_out) & 15))
    return HILC_ERR_UNSUPPORTED;
  if (down->hist && down->hist == down->hist_out) return HILC_ERR_UNSUPPORTED;
  if (down->res == down->y) return HILC_ERR_UNSUPPORTED;
  if (streaming && (long)B * 2 * C * T * 4 >= (1L << 32)) return HILC_ERR_UNSUPPORTED;
  ResArgs a{};
  a.x = x; a.y = down->y; a.T = T; a.tiles = 0; a.nblk = nblk; a.sched = nullptr; a.dbg = HILC_CHAIN_DBG;
  if (const int rc = fill_blocks(a, blocks, nblk)) return rc;
  a.up = ResUp{};
  a.post = ResPost{};
  ResDown& d = a.dn;
  d.w_lo = down->w_lo; d.w_hi = down->w_hi; d.dw_w = down->dw_w; d.dw_b = down->dw_b; d.hist = streaming ? down->hist : nullptr;
  d.hist_out = streaming ? down->hist_out : nullptr; d.res = down->res; d.y = down->y; d.in_scale = down->in_scale;
  hipStream_t s = (hipStream_t)stream;
  if (streaming) {
    switch (C) {
      case 64: return launch_chain<64, true, 2, false, 2>(a, B, s);
      case 128: return launch_chain<128, true, 2, true, 4>(a, B, s);
      case 256: return launch_chain<256, true, 2, false, 5>(a, B, s);
      default: return launch_chain<512, true, 2, false, 8>(a, B, s);
    }
  }
  switch (C) {
    case 64: return launch_chain<64, false, 2, false, 2>(a, B, s);
    case 128: return launch_chain<128, false, 2, true, 4>(a, B, s);
    case 256: return launch_chain<256, false, 2, false, 5>(a, B, s);
    default: return launch_chain<512, false, 2, false, 8>(a, B, s);
  }
}

// ---- the encoder's FIRST stage with its input computed in the launch (offline) ------------------------------------------------------------
// seanet.py:280-286 (first conv), 220-246 (SpecBlock of stage 0), 316-339 (blocks, down-sampling layer): == hilc_spec_block_conv_pre followed by
// hilc_encoder_stage (C = 64, r = 2), bit for bit; the [B][64][T] tensor between the two never exists (1.57 GB written and read at 256 clips).
// streaming = 1 (ABI 15; streaming.py:490-511): the same for a hop, with the waveform history `spec->hist` in front of every stream's t = 0 and the
// caches of hilc_encoder_stage(streaming).
extern "C" int hilc_encoder_stage0_supported(int T, int nblk, int stride, int n_fft, int hop, int pre_ksize, int streaming) {
  // a hop (ABI 15): runs of whole streams; T >= 128 so that a 128-column tile holds at most one stream start (two waveform pieces)
  return nblk >= 1 && nblk <= 2 && stride == 2 && n_fft == 64 && hop == 1 && pre_ksize == 5 && T > 0 && T % 4 == 0 && (!streaming || T >= 128);
}

extern "C" int hilc_encoder_stage0(const hilc_spec0_params* spec, const hilc_resblock_params* blocks, int nblk, const hilc_down_params* down,
                                   int streaming, int B, int T, void* stream) {
  if (!spec || !blocks || !down) return HILC_ERR_NULL;
  if (!spec->wav || !spec->dft_packed || !spec->nyq_sin || !spec->pw_packed || !spec->pre_w) return HILC_ERR_NULL;
  if (!down->w_lo || !down->w_hi || !down->dw_w || !down->dw_b || !down->y) return HILC_ERR_NULL;
  if (B <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (!hilc_encoder_stage0_supported(T, nblk, down->stride, spec->n_fft, spec->hop, spec->pre_ksize, streaming)) return HILC_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(down->y) & 15) || (reinterpret_cast<uintptr_t>(down->res) & 15) || down->res == down->y) return HILC_ERR_UNSUPPORTED;
  if (streaming) {
    if ((reinterpret_cast<uintptr_t>(down->hist) & 15) || (reinterpret_cast<uintptr_t>(down->hist_out) & 15)) return HILC_ERR_UNSUPPORTED;
    if (down->hist && down->hist == down->hist_out) return HILC_ERR_UNSUPPORTED;
    if ((long)B * 128 * T * 4 >= (1L << 32)) return HILC_ERR_UNSUPPORTED;            // 32-bit flat column index / byte offsets
    if (spec->hist != nullptr && spec->hist_len < spec->n_fft - 1) return HILC_ERR_SHAPE;
  }
  ResArgs a{};
  a.x = spec->wav; a.y = down->y; a.T = T; a.tiles = 0; a.nblk = nblk; a.sched = nullptr; a.dbg = HILC_CHAIN_DBG;
  if (const int rc = fill_blocks(a, blocks, nblk)) return rc;
  a.up = ResUp{};
  a.post = ResPost{};
  ResDown& d = a.dn;
  d.w_lo = down->w_lo; d.w_hi = down->w_hi; d.dw_w = down->dw_w; d.dw_b = down->dw_b;
  d.hist = streaming ? down->hist : nullptr; d.hist_out = streaming ? down->hist_out : nullptr;
  d.res = down->res; d.y = down->y; d.in_scale = down->in_scale;
  ResSpec0& sp = a.spec;
  sp.hist = streaming ? spec->hist : nullptr; sp.hist_len = streaming ? spec->hist_len : 0;
  sp.wav = spec->wav; sp.dft = spec->dft_packed; sp.nyq = spec->nyq_sin; sp.pw = spec->pw_packed; sp.bias = spec->bias;
  sp.pre_w = spec->pre_w; sp.pre_b = spec->pre_b; sp.pre_in_scale = spec->pre_in_scale; sp.mean = spec->mean; sp.stdv = spec->std;
  sp.out_scale = spec->out_scale; sp.normalize = spec->normalize;
  if (streaming) return launch_chain<64, true, 2, false, 2, false, true>(a, B, (hipStream_t)stream);
  return launch_chain<64, false, 2, false, 2, false, true>(a, B, (hipStream_t)stream);
}

// ---- a DECODER STAGE of a streaming hop in one launch: its up-sampling layer and its residual blocks -----------------------------
// seanet.py:431-452 (`[Scale, ELU, SConvTranspose1d (depthwise, k = 2r, stride r), 1x1 conv 2C -> C + bias]`, then the stage's three
// SEANetResnetBlocks); streaming.py:629-639 with the transposed conv's cache.  == hilc_up_conv_stream followed by hilc_resblock_chain,
// bit for bit.  Streaming: C = 768 (r = 8: a hop is 1-4 frames of 1536 channels per stream, whole streams per 32-column tile), and the
// carry-form stages C = 192 (r = 4) / C = 96 (r = 2), which also take the offline model (streaming = 0).
extern "C" int hilc_decoder_stage_supported(int C, int T, int nblk, int stride, int streaming) {
  if (nblk < 1 || nblk > 3 || T <= 0 || T % 4 != 0 || stride <= 0 || T % stride != 0) return 0;
  if (C == 768) return stride == 8 && (streaming ? 32 % T == 0 : nblk == 1);      // whole streams per 32-column tile; offline: carry form, up-sampling layer + FIRST block (LDS)
  // r = 5: up->tr_w = the EXPANDED tap table of hilc_up_conv_expand_taps.  Offline: carry form, the whole stage; a streaming hop
  // (round 6): 32-column carry tiles, runs of whole streams, the whole stage (rounds 4-5: 64-column halo tiles, the first block only)
  if (C == 384) return stride == 5;
  return (C == 192 && stride == 4) || (C == 96 && stride == 2);      // the carry form: streaming hops and the offline model
}

namespace {
int decoder_stage_entry(const hilc_up_params* up, const hilc_resblock_params* blocks, int nblk, float* y, const hilc_post_params* post, int streaming,
                        int B, int C, int T, void* stream);
}

extern "C" int hilc_decoder_stage(const hilc_up_params* up, const hilc_resblock_params* blocks, int nblk, float* y, int streaming,
                                  int B, int C, int T, void* stream) {
  return decoder_stage_entry(up, blocks, nblk, y, nullptr, streaming, B, C, T, stream);
}

// ---- the decoder's LAST stage AND its closing layer in one launch ----------------------------------------------------------------------
// seanet.py:453-476 (`[Scale, ELU, SConv1d(C, 1, k = 5)]`, then the model's out_scale / tanh) behind the stage of hilc_decoder_stage: the
// last block leaves ELU(in_scale * y) in the LDS tile and the launch stores the waveform `[B][1][T]` — the stage's `[B][C][T]` output is
// never written.  == hilc_decoder_stage followed by hilc_conv_post (k = 5), bit for bit (same row classes, same order of the partial sums).
// C = 96 with r = 2 and three blocks (the hil_speech / hil_music decoders' last stage): the offline model, and (ABI 15) a streaming hop —
// streaming.py:639-648 — whose closing conv takes its cache `post->hist` [B][C][4] (the previous hop's last 4 ACTIVATED columns, as
// hilc_conv_post's) at a stream's t = 0 and leaves `post->hist_out`.
extern "C" int hilc_decoder_stage_post_supported(int C, int T, int nblk, int stride, int ksize) {
  return C == 96 && stride == 2 && nblk == 3 && ksize == 5 && T > 0 && T % 4 == 0;
}

extern "C" int hilc_decoder_stage_post(const hilc_up_params* up, const hilc_resblock_params* blocks, int nblk, const hilc_post_params* post,
                                       int streaming, int B, int C, int T, void* stream) {
  if (!post || !post->w || !post->wav) return HILC_ERR_NULL;
  if (!up) return HILC_ERR_NULL;
  if (!hilc_decoder_stage_post_supported(C, T, nblk, up->stride, post->ksize)) return HILC_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(post->wav) & 15) || (reinterpret_cast<uintptr_t>(post->hist) & 15) ||
      (reinterpret_cast<uintptr_t>(post->hist_out) & 15))
    return HILC_ERR_UNSUPPORTED;
  if (streaming && post->hist && post->hist == post->hist_out) return HILC_ERR_UNSUPPORTED;
  return decoder_stage_entry(up, blocks, nblk, post->wav, post, streaming, B, C, T, stream);
}

namespace {
int decoder_stage_entry(const hilc_up_params* up, const hilc_resblock_params* blocks, int nblk, float* y, const hilc_post_params* post, int streaming,
                        int B, int C, int T, void* stream) {
  if (!up || !blocks || !y) return HILC_ERR_NULL;
  if (!up->x || !up->tr_w || !up->w_lo || !up->w_hi) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (!hilc_decoder_stage_supported(C, T, nblk, up->stride, streaming)) return HILC_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(up->tr_w) & 15)) return HILC_ERR_UNSUPPORTED;
  if (up->hist && up->hist == up->hist_out) return HILC_ERR_UNSUPPORTED;
  if (streaming && (long)B * C * T * 4 >= (1L << 32)) return HILC_ERR_UNSUPPORTED;
  ResArgs a{};
  a.x = up->x; a.y = y; a.T = T; a.tiles = 0; a.nblk = nblk; a.sched = nullptr; a.dbg = HILC_CHAIN_DBG;
  a.dn = ResDown{};
  if (const int rc = fill_blocks(a, blocks, nblk)) return rc;
  ResUp& u = a.up;
  u.xin = up->x; u.tr_w = up->tr_w; u.w_lo = up->w_lo; u.w_hi = up->w_hi; u.bias = up->bias; u.hist = up->hist;
  u.hist_out = streaming ? up->hist_out : nullptr; u.in_scale = up->in_scale;
  if (!streaming) u.hist = nullptr;
  hipStream_t s = (hipStream_t)stream;
  a.post = ResPost{};
  if (post != nullptr) {
    a.post.w = post->w; a.post.bias = post->bias; a.post.wav = post->wav; a.post.in_scale = post->in_scale;
    a.post.out_scale = post->out_scale; a.post.do_tanh = post->do_tanh;
    a.post.hist = streaming ? post->hist : nullptr;
    a.post.hist_out = streaming ? post->hist_out : nullptr;
    return streaming ? launch_chain<96, true, 3, false, -2, true>(a, B, s) : launch_chain<96, false, 3, false, -2, true>(a, B, s);
  }
  if (streaming) {
    switch (C) {
      case 768: return launch_chain<768, true, 3, false, -8>(a, B, s);
      case 384: return launch_chain<384, true, 3, false, -5>(a, B, s);
      case 192: return launch_chain<192, true, 3, false, -4>(a, B, s);
      default: return launch_chain<96, true, 3, false, -2>(a, B, s);
    }
  }
  if (C == 768) return launch_chain<768, false, 1, false, -8>(a, B, s);
  if (C == 384) return launch_chain<384, false, 3, false, -5>(a, B, s);
  return C == 192 ? launch_chain<192, false, 3, false, -4>(a, B, s) : launch_chain<96, false, 3, false, -2>(a, B, s);
}
}  // namespace
