// Entry points of the fused residual block, one block per launch (kernel: resblock_kernel.h).
#include "resblock_launch.h"

namespace {
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* wt, float* packed, int C, int RH) {
  const int idx = blockIdx.x * 256 + threadIdx.x;       // index into `packed`
  if (idx >= C * C) return;
  const int CB = C / 32, CBW = CB / RH, NQ = 2 * CBW, KT = C / 16;
  const int e = idx & 3, lane = (idx >> 2) & 63, w = idx >> 8;
  const int q = w % NQ, kt = (w / NQ) % KT, h = w / (NQ * KT);
  const int v = q * 4 + e, j = v / CBW, i = v % CBW;
  const int k = kt * 16 + 2 * j + (lane >> 5), m = 32 * (h * CBW + i) + (lane & 31);
  packed[idx] = wt[(long)k * C + m];
}

}  // namespace

namespace {
int resblock_entry(bool streaming, const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                   const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1, const float* hist2,
                   float* hist1_out, float* hist2_out, float* y, int* sched, int B, int C, int T, float pre_scale,
                   float out_scale, void* stream) {
  if (!x || !w1t || !dw1_w || !dw1_b || !w2t || !dw2_w || !dw2_b || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (x == y) return HILC_ERR_UNSUPPORTED;   // neighbouring tiles read each other's halo: not in place
  if (T % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
    return HILC_ERR_UNSUPPORTED;             // callers fall back to two hilc_dws_conv launches
  ResArgs a{};
  ResBlk& b0 = a.blk[0];
  a.x = x; b0.w1t = w1t; b0.dw1_w = dw1_w; b0.dw1_b = dw1_b; b0.w2t = w2t; b0.dw2_w = dw2_w; b0.dw2_b = dw2_b;
  a.y = y; a.T = T; a.tiles = 0; b0.pre_scale = pre_scale; b0.out_scale = out_scale;
  b0.hist1 = hist1; b0.hist2 = hist2; b0.hist1_out = hist1_out; b0.hist2_out = hist2_out;
  a.nblk = 1; a.run_tiles = 0;
  a.post = ResPost{};
  a.sched = sched;
#ifdef HILC_DEBUG_STAMPS
  a.dbg = g_dbg;
#else
  a.dbg = nullptr;
#endif
  if (streaming) {
    if ((hist1 && hist1 == hist1_out) || (hist2 && hist2 == hist2_out)) return HILC_ERR_UNSUPPORTED;   // first / last tiles of a clip race
    if ((long)B * C * T * 4 >= (1L << 32)) return HILC_ERR_UNSUPPORTED;   // 32-bit flat column index / byte offsets
    switch (C) {
      case 64: return launch_res<64, true>(a, B, (hipStream_t)stream);
      case 96: return launch_res<96, true>(a, B, (hipStream_t)stream);
      case 128: return launch_res<128, true>(a, B, (hipStream_t)stream);
      case 192: return launch_res<192, true>(a, B, (hipStream_t)stream);
      // the wide blocks of a hop (NARROW shapes): 64-column tiles on the flat column space, or whole-stream 32-column tiles
      case 256: return launch_res<256, true>(a, B, (hipStream_t)stream);
      case 384: return launch_res<384, true>(a, B, (hipStream_t)stream);
      case 512: return 32 % T == 0 ? launch_res<512, true>(a, B, (hipStream_t)stream) : HILC_ERR_UNSUPPORTED;
      case 768: return 32 % T == 0 ? launch_res<768, true>(a, B, (hipStream_t)stream) : HILC_ERR_UNSUPPORTED;
      default: return HILC_ERR_UNSUPPORTED;
    }
  }
  switch (C) {
    case 64: return launch_res<64, false>(a, B, (hipStream_t)stream);
    case 96: return launch_res<96, false>(a, B, (hipStream_t)stream);
    case 128: return launch_res<128, false>(a, B, (hipStream_t)stream);
    case 192: return launch_res<192, false>(a, B, (hipStream_t)stream);
    // the wide blocks (NARROW shapes, carry form): one launch instead of two hilc_dws_conv — the first half's output stays in LDS
    case 256: return launch_res<256, false>(a, B, (hipStream_t)stream);
    case 384: return launch_res<384, false>(a, B, (hipStream_t)stream);
    case 512: return launch_res<512, false>(a, B, (hipStream_t)stream);
    case 768: return launch_res<768, false>(a, B, (hipStream_t)stream);
    default: return HILC_ERR_UNSUPPORTED;
  }
}
}  // namespace

extern "C" int hilc_resblock_pack_weights(const float* wt, float* packed, int C, void* stream) {
  if (!wt || !packed) return HILC_ERR_NULL;
  if (!(C == 64 || C == 96 || C == 128 || C == 192 || C == 256 || C == 384 || C == 512 || C == 768)) return HILC_ERR_UNSUPPORTED;
  if (wt == packed) return HILC_ERR_UNSUPPORTED;
  const int RH = C >= 512 ? 8 : (C >= 256 ? 4 : (C == 192 ? 2 : 1));     // row classes of the one shape each width has
  static_assert(Cfg<256, true>::RH == 4 && Cfg<384, true>::RH == 4 && Cfg<512, true>::RH == 8 && Cfg<768, true>::RH == 8, "packed layout");
  static_assert(Cfg<256, false>::RH == 4 && Cfg<384, false>::RH == 4 && Cfg<512, false>::RH == 8 && Cfg<768, false>::RH == 8, "packed layout");
  static_assert(Cfg<64, false>::RH == 1 && Cfg<96, false>::RH == 1 && Cfg<128, false>::RH == 1 && Cfg<192, false>::RH == 2 &&
                Cfg<64, true>::RH == 1 && Cfg<96, true>::RH == 1 && Cfg<128, true>::RH == 1 && Cfg<192, true>::RH == 2, "packed layout");
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((C * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wt, packed, C, RH);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_resblock_pack_weights_rc(const float* wt, float* packed, int C, int row_classes, void* stream) {
  if (!wt || !packed) return HILC_ERR_NULL;
  if (C <= 0 || C % 32 != 0 || row_classes < 1 || (C / 32) % row_classes != 0) return HILC_ERR_UNSUPPORTED;
  if (wt == packed) return HILC_ERR_UNSUPPORTED;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((C * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wt, packed, C, row_classes);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_resblock(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                             const float* w2t, const float* dw2_w, const float* dw2_b, float* y, int B, int C,
                             int T, float pre_scale, float out_scale, void* stream) {
  return resblock_entry(false, x, w1t, dw1_w, dw1_b, w2t, dw2_w, dw2_b, nullptr, nullptr, nullptr, nullptr, y, nullptr,
                        B, C, T, pre_scale, out_scale, stream);
}

extern "C" int hilc_resblock_stream(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                                    const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1,
                                    const float* hist2, float* hist1_out, float* hist2_out, float* y, int B, int C,
                                    int T, float pre_scale, float out_scale, void* stream) {
  return resblock_entry(true, x, w1t, dw1_w, dw1_b, w2t, dw2_w, dw2_b, hist1, hist2, hist1_out, hist2_out, y, nullptr,
                        B, C, T, pre_scale, out_scale, stream);
}

extern "C" int hilc_resblock_balanced(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                                      const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1,
                                      const float* hist2, float* hist1_out, float* hist2_out, float* y, int* sched,
                                      int streaming, int B, int C, int T, float pre_scale, float out_scale,
                                      void* stream) {
  return resblock_entry(streaming != 0, x, w1t, dw1_w, dw1_b, w2t, dw2_w, dw2_b, hist1, hist2, hist1_out, hist2_out, y,
                        sched, B, C, T, pre_scale, out_scale, stream);
}

#ifdef HILC_DEBUG_STAMPS
extern "C" void hilc_debug_set_stamp_buffer(unsigned long long* p) { g_dbg = p; }
#endif

extern "C" int hilc_resblock_supported(int C, int T) {
  return (C == 64 || C == 96 || C == 128 || C == 192 || C == 256 || C == 384 || C == 512 || C == 768) && T > 0 && T % 4 == 0;
}

// widths and hop lengths hilc_resblock_stream / hilc_resblock_sched(streaming = 1) take: the offline widths, plus the wide
// blocks of a hop — C = 256 / 384 at any T % 4 == 0, C = 512 / 768 where whole streams tile 32 columns (T = 4, 8, 16, 32)
extern "C" int hilc_resblock_stream_supported(int C, int T) {
  if (T <= 0 || T % 4 != 0) return 0;
  if (C == 64 || C == 96 || C == 128 || C == 192 || C == 256 || C == 384) return 1;
  return (C == 512 || C == 768) && 32 % T == 0;
}
