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

extern "C" int hilc_up_conv_x3(const float* x, const float* tr_w, const float* tr_w_expanded, const void* wsplit,
                               const float* bias, float* y, int B, int K, int M, int Tin, int stride, float in_scale,
                               void* stream) {
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
  auto go = [&](auto bop) {
    bop.x = x; bop.w = decltype(bop)::kExpanded ? tr_w_expanded : tr_w; bop.hist = nullptr; bop.K = K; bop.Tin = Tin;
    bop.r = stride; bop.ncols = ncols; bop.in_scale = in_scale;
    return launch_x3(reinterpret_cast<const unsigned short*>(wsplit), M, K, M, (ncols + BN - 1) / BN, bop, el,
                     (hipStream_t)stream);
  };
  if (stride == 8) return go(UpB<8, true, false>{});
  if (stride == 4) return go(UpB<4, true, false>{});
  if (stride == 2) return go(UpB<2, true, false>{});
  if (tr_w_expanded != nullptr && (reinterpret_cast<uintptr_t>(tr_w_expanded) & 15) == 0) return go(UpB<1, true, false>{});
  return HILC_ERR_UNSUPPORTED;
}
