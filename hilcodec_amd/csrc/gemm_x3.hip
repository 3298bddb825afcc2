// Entry points of the EXPERIMENTAL bf16x3 numerics mode (gemm_x3.h): decoder-side layers only, opt-in, never the
// default.  Same argument meaning as the fp32 entry points they mirror; `wsplit` replaces the fp32 `[K][M]` matrix.
#include "gemm_core.h"
#include "gemm_lin.h"
#include "gemm_epilogues.h"
#include "gemm_x3.h"

using namespace hilc;

namespace {
// wsplit[0][k][m] = bf16(w[k][m]),  wsplit[1][k][m] = bf16(w[k][m] - float(wsplit[0][k][m]))   (round to nearest even)
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ wt, unsigned short* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float w = wt[i];
  const __bf16 h = (__bf16)w;
  const __bf16 l = (__bf16)(w - (float)h);
  out[i] = __builtin_bit_cast(unsigned short, h);
  out[n + i] = __builtin_bit_cast(unsigned short, l);
}
}  // namespace

extern "C" int hilc_x3_split_weights(const float* wt, void* wsplit, int K, int M, void* stream) {
  if (!wt || !wsplit) return HILC_ERR_NULL;
  if (K <= 0 || M <= 0) return HILC_ERR_SHAPE;
  const long n = (long)K * M;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wt,
                     reinterpret_cast<unsigned short*>(wsplit), n);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_x3_supported(int K, int M, int T) { return K % 32 == 0 && M % 8 == 0 && T % 4 == 0; }

extern "C" int hilc_dws_conv_x3(const float* x, const void* wsplit, const float* dw_w, const float* dw_b, const float* res,
                                float* y, int B, int K, int M, int T, float in_scale, int in_elu, float out_scale,
                                int out_elu, void* stream) {
  if (!x || !wsplit || !dw_w || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (!hilc_x3_supported(K, M, T) || (reinterpret_cast<uintptr_t>(x) & 15) || !lin_ok(B, K, T)) return HILC_ERR_UNSUPPORTED;
  Dw5Epilogue ep;
  ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.res = res; ep.M = M; ep.T = T;
  ep.tiles = (T + Dw5Epilogue::STEP - 1) / Dw5Epilogue::STEP;
  ep.out_scale = out_scale; ep.out_elu = out_elu;
  ep.vec = (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (res == nullptr || (reinterpret_cast<uintptr_t>(res) & 15) == 0);
  TileCols cols;
  cols.K = K; cols.T = T; cols.tiles = ep.tiles; cols.step = Dw5Epilogue::STEP; cols.halo = 4;
  return launch_gemm_x3(reinterpret_cast<const unsigned short*>(wsplit), x, M, K, M, T, (long)B * ep.tiles, in_scale,
                        in_elu != 0, cols, ep, (hipStream_t)stream);
}

// streaming hop of a wide k5 / stride-1 layer: whole-clip tiles, caches in the epilogue (hilc_dws_conv_stream's T <= 128 form)
extern "C" int hilc_dws_conv_stream_x3(const float* x, const void* wsplit, const float* dw_w, const float* dw_b,
                                       const float* hist, float* hist_out, const float* res, float* y, int B, int K, int M,
                                       int T, float in_scale, int in_elu, float out_scale, int out_elu, void* stream) {
  if (!x || !wsplit || !dw_w || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (hist != nullptr && hist == hist_out) return HILC_ERR_UNSUPPORTED;
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) |
                    reinterpret_cast<uintptr_t>(hist) | reinterpret_cast<uintptr_t>(hist_out)) & 15) == 0;
  if (!hilc_x3_supported(K, M, T) || T > BN || !al || !lin_ok(B, K, T)) return HILC_ERR_UNSUPPORTED;
  const int cpt = BN / T;
  FlatCols cols;
  cols.K = K; cols.T = T; cols.tile_cols = cpt * T; cols.ncols = (long)B * T;
  Dw5SegEpilogue ep;
  ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.res = res; ep.hist = hist; ep.hist_out = hist_out; ep.B = B; ep.M = M;
  ep.T = T; ep.cpt = cpt; ep.out_scale = out_scale; ep.out_elu = out_elu;
  div_magic(T, ep.t_magic, ep.t_shift);
  return launch_gemm_x3(reinterpret_cast<const unsigned short*>(wsplit), x, M, K, M, T, ((long)B + cpt - 1) / cpt, in_scale,
                        in_elu != 0, cols, ep, (hipStream_t)stream);
}

namespace {
// new cache of the transposed conv: pro(x[b,k,Tin-1])  (causal_layers.py:168-188: cache = last input frame)
__global__ __launch_bounds__(256) void x3_up_hist_kernel(const float* x, float* hist_out, long n, int Tin, float in_scale) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) hist_out[e] = prologue(x[e * Tin + Tin - 1], in_scale, 1);
}
}  // namespace

extern "C" int hilc_up_conv_x3(const float* x, const float* hist, float* hist_out, const float* tr_w,
                               const float* tr_w_expanded, const void* wsplit, const float* bias, float* y, int B, int K,
                               int M, int Tin, int stride, float in_scale, void* stream) {
  if (!x || !tr_w || !wsplit || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || Tin <= 0 || stride <= 0) return HILC_ERR_SHAPE;
  const long Tout = (long)Tin * stride;
  const long ncols = (long)B * Tout;
  if (!hilc_x3_supported(K, M, (int)(Tout % 4 == 0 ? 4 : 1)) || ncols >= (1L << 31) || !lin_ok(B, K, Tin) ||
      (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(tr_w) & 15))
    return HILC_ERR_UNSUPPORTED;
  PwLdsEpilogue el;
  el.y = y; el.bias = bias; el.res = nullptr; el.M = M; el.T = (int)Tout; el.ncols = ncols; el.out_scale = 1.0f;
  div_magic((int)Tout, el.t_magic, el.t_shift);
  if (hist_out != nullptr) {
    if (hist_out == hist) return HILC_ERR_UNSUPPORTED;
    const long n = (long)B * K;
    HILC_CLEAR_ERROR();
    hipLaunchKernelGGL(x3_up_hist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, hist_out, n,
                       Tin, in_scale);
    HILC_CHECK_LAUNCH();
  }
  auto go = [&](auto bop) {
    bop.x = x; bop.w = decltype(bop)::kExpanded ? tr_w_expanded : tr_w; bop.hist = hist; bop.K = K; bop.Tin = Tin;
    bop.r = stride; bop.ncols = ncols; bop.in_scale = in_scale;
    return launch_x3(reinterpret_cast<const unsigned short*>(wsplit), M, K, M, (ncols + BN - 1) / BN, bop, el,
                     (hipStream_t)stream);
  };
  const bool exp_ok = tr_w_expanded != nullptr && (reinterpret_cast<uintptr_t>(tr_w_expanded) & 15) == 0;
  if (hist != nullptr) {       // streaming hop: the frame before the hop comes from the transposed conv's cache
    if (stride == 8) return go(UpB<8, true, true>{});
    if (stride == 4) return go(UpB<4, true, true>{});
    if (stride == 2) return go(UpB<2, true, true>{});
    if (exp_ok) return go(UpB<1, true, true>{});
    return HILC_ERR_UNSUPPORTED;
  }
  if (stride == 8) return go(UpB<8, true, false>{});
  if (stride == 4) return go(UpB<4, true, false>{});
  if (stride == 2) return go(UpB<2, true, false>{});
  if (exp_ok) return go(UpB<1, true, false>{});
  return HILC_ERR_UNSUPPORTED;
}
