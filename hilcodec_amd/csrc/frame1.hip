// Single-frame layers of a streaming hop (T = 1 sample per stream and call): the 1x1 convolutions around the
// quantiser and of the last SpecBlock (StreamingEncoder / StreamingDecoder, streaming.py:395-517, with the hop of the
// slowest layers) and the depthwise taps that follow them.
//
// With T = 1 the activation `[B][K][1]` is a row-major [B x K] matrix: y[b][m] = sum_k wt[k][m] * pro(x[b][k]).  At
// 1024 streams this is a 128..1536 x 1024 x 128..1024 GEMM — 0.3 to 1 GFLOP, a few microseconds of MFMA time — so the
// launch is latency-bound, not throughput-bound: the tiled core (128-column tiles, K walked in LDS slices behind one
// barrier per slice) leaves most CUs idle and pays one L2 round trip per slice.  Here a workgroup owns a 32 x 32
// output tile, its four waves split K four ways, every wave feeds v_mfma_f32_32x32x2_f32 straight from global loads
// issued 16 k-pairs ahead (no LDS, no barrier in the loop), and the four partial tiles are added in wave order
// through LDS (fixed order: run-to-run deterministic).  The optional depthwise epilogue (k taps, stride 1) reads the
// k-1 cached pointwise outputs of the previous hops and writes the next cache (causal_layers.py:147-165).
#include "common.h"
#include "frame1.h"

namespace {

constexpr int UP = 16;   // k-pairs per group: 2 x 16 loads in flight per lane

__device__ __forceinline__ int acc_row1(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <bool DW>
__global__ __launch_bounds__(256) void frame1_kernel(Frame1Args a) {
  __shared__ float red[4][32][33];   // [wave][column b][row m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * 32;
  const long b0 = (long)blockIdx.y * 32;
  const int K = a.K, M = a.M;
  // rows / columns past the edge compute on clamped addresses and are never stored
  const int ml = m0 + l32 < M ? m0 + l32 : M - 1;
  const long bl = b0 + l32 < a.B ? b0 + l32 : a.B - 1;
  const int kw = (((K + 3) >> 2) + 1) & ~1;          // k range of a wave: even, so k-pairs never straddle waves
  const int kb = wave * kw;
  const int ke = kb + kw < K ? kb + kw : K;
  const float* ap = a.wt + (long)(kb + half) * M + ml;
  const float* bp = a.x + bl * K + kb + half;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int k = kb;
  for (; k + 2 * UP <= ke; k += 2 * UP) {
    float av[UP], bv[UP];
#pragma unroll
    for (int i = 0; i < UP; ++i) {
      av[i] = ap[(long)(2 * i) * M];
      bv[i] = bp[2 * i];
    }
#pragma unroll
    for (int i = 0; i < UP; ++i)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], prologue(bv[i], a.in_scale, a.in_elu), acc, 0, 0, 0);
    ap += (long)(2 * UP) * M;
    bp += 2 * UP;
  }
  for (; k < ke; k += 2) {                            // ragged end of the wave's range (K = 513: one odd sample)
    const bool ok = k + half < ke;
    const float av = ok ? ap[0] : 0.f;
    const float bv = ok ? prologue(bp[0], a.in_scale, a.in_elu) : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    ap += 2L * M;
    bp += 2;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][l32][acc_row1(r, lane)] = acc[r];
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = tid & 31, col = (tid >> 5) + 8 * s;
    const int m = m0 + row;
    const long b = b0 + col;
    if (m >= M || b >= a.B) continue;
    float h = __fadd_rn(__fadd_rn(__fadd_rn(red[0][col][row], red[1][col][row]), red[2][col][row]), red[3][col][row]);
    if (a.bias != nullptr) h = __fadd_rn(h, a.bias[m]);
    const long off = b * M + m;
    float v = h;
    if (DW) {
      // depthwise taps over [cache | this sample]; the new cache drops the oldest sample
      const int pad = a.ksize - 1;
      const float* w = a.dw_w + (long)m * a.ksize;
      const float* hp = a.hist != nullptr ? a.hist + off * pad : nullptr;
      float* ho = a.hist_out != nullptr ? a.hist_out + off * pad : nullptr;
      float acc1 = 0.f;
      for (int j = 0; j < pad; ++j) {
        const float c = hp != nullptr ? hp[j] : 0.f;
        acc1 = fmaf(w[j], c, acc1);
        if (ho != nullptr && j > 0) ho[j - 1] = c;
      }
      acc1 = fmaf(w[pad], h, acc1);
      if (ho != nullptr && pad > 0) ho[pad - 1] = h;
      if (a.dw_b != nullptr) acc1 = __fadd_rn(acc1, a.dw_b[m]);
      v = acc1;
    }
    // separate roundings, like the reference's `y.mul_(scale)` then `x.add_(y)`
    v = __fmul_rn(v, a.out_scale);
    if (a.res != nullptr) v = __fadd_rn(v, a.res[off]);
    if (a.out_elu) v = elu_fast(v);
    a.y[off] = v;
  }
}

// ---- the streaming encoder's single-frame TAIL in one launch (round 6) --------------------------------------------------------------
// streaming.py:512-517: `conv_post` = [ELU, depthwise conv k = 5 over [cache | frame], 1x1 conv K -> 128 + bias], then L2Norm over the 128
// channels.  As four launches (hilc_dw_conv + its cache update, hilc_pw_conv = frame1_kernel, hilc_l2norm) this was 77 us of a 4.46 ms hop at
// ~0 matrix utilisation.  Here a workgroup owns 32 streams and ALL 128 outputs: its four waves split K four ways exactly like frame1_kernel
// (same k ranges, same ascending k-pair order, the four partial tiles added in wave order) and feed the MFMAs of the four 32-row output tiles
// from ONE depthwise-convolved operand, which they compute on the fly — fmaf chain over the taps j = 0..4 on [cache | ELU(in_scale * x)], the
// expression of dw_generic_kernel — and whose new cache they store; the epilogue adds the bias and runs hilc_l2norm's arithmetic (one fmaf
// chain over the channels in ascending order, sqrt, max with eps, divide, multiply).  Same chains, same roundings: bit-identical to the
// four launches.
constexpr int TAIL_M = 128;
constexpr int TAIL_UP = 8;     // k-pairs per group (4 weight words + a cache quad + 5 taps per k-pair in flight)

__global__ __launch_bounds__(256) void encoder_tail_kernel(EncTailArgs a) {
  __shared__ float red[4][32][TAIL_M + 1];   // [wave][stream][channel]
  __shared__ float hs[32][TAIL_M + 1];       // summed, + bias
  __shared__ float dn[32];                   // per stream: max(||h||, eps)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
  const long b0 = (long)blockIdx.x * 32;
  const int K = a.K;
  const long bl = b0 + l32 < a.B ? b0 + l32 : a.B - 1;       // streams past the edge compute on a clamped address and store nothing
  const bool own = b0 + l32 < a.B;
  const int kw = (((K + 3) >> 2) + 1) & ~1;                  // frame1_kernel's split of K over the four waves
  const int kb = wave * kw;
  const int ke = kb + kw < K ? kb + kw : K;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // the operand of k: h = sum_j w[k][j] * v_j, v = [cache[b][k][0..3] | pro(x[b][k])]; the lane that computes it also stores the new cache
  for (int k = kb; k < ke; k += 2 * TAIL_UP) {
    float av[TAIL_UP][4], xv[TAIL_UP], w5[TAIL_UP][5];
    f32x4 cv[TAIL_UP];
#pragma unroll
    for (int i = 0; i < TAIL_UP; ++i) {
      const int kk = k + 2 * i + half;
      const bool ok = kk < ke;
      const int kc = ok ? kk : ke - 1;
      xv[i] = a.x[bl * K + kc];
      cv[i] = a.hist != nullptr ? *reinterpret_cast<const f32x4*>(a.hist + (bl * K + kc) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 5; ++j) w5[i][j] = a.dw_w[kc * 5 + j];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[i][m] = ok ? a.wt[(long)kc * TAIL_M + 32 * m + l32] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TAIL_UP; ++i) {
      const int kk = k + 2 * i + half;
      const bool ok = kk < ke;
      const float v4 = prologue(xv[i], a.in_scale, a.in_elu);
      float h = 0.f;
      h = fmaf(w5[i][0], cv[i].x, h);
      h = fmaf(w5[i][1], cv[i].y, h);
      h = fmaf(w5[i][2], cv[i].z, h);
      h = fmaf(w5[i][3], cv[i].w, h);
      h = fmaf(w5[i][4], v4, h);
      if (ok && own && a.hist_out != nullptr)
        *reinterpret_cast<f32x4*>(a.hist_out + (bl * K + kk) * 4) = f32x4{cv[i].y, cv[i].z, cv[i].w, v4};
      const float bop = ok ? h : 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][m], bop, acc[m], 0, 0, 0);
    }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][l32][32 * m + acc_row1(r, lane)] = acc[m][r];
  __syncthreads();
  // the four partial tiles in wave order (frame1_kernel's order), + bias
  for (int e = tid; e < 32 * TAIL_M; e += 256) {
    const int col = e / TAIL_M, m = e - col * TAIL_M;
    float h = __fadd_rn(__fadd_rn(__fadd_rn(red[0][col][m], red[1][col][m]), red[2][col][m]), red[3][col][m]);
    if (a.bias != nullptr) h = __fadd_rn(h, a.bias[m]);
    hs[col][m] = h;
  }
  __syncthreads();
  if (a.l2norm) {
    if (tid < 32) {                      // hilc_l2norm: ONE fmaf chain over the channels in ascending order
      float ss = 0.f;
      for (int m = 0; m < TAIL_M; ++m) ss = fmaf(hs[tid][m], hs[tid][m], ss);
      dn[tid] = fmaxf(sqrtf(ss), a.eps);
    }
    __syncthreads();
  }
  for (int e = tid; e < 32 * TAIL_M; e += 256) {
    const int col = e / TAIL_M, m = e - col * TAIL_M;
    const long b = b0 + col;
    if (b >= a.B) continue;
    const float v = hs[col][m];
    a.z[b * TAIL_M + m] = a.l2norm ? __fmul_rn(__fdiv_rn(v, dn[col]), a.scale) : v;
  }
}

}  // namespace

int launch_encoder_tail(const EncTailArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.K <= 0) return HILC_ERR_SHAPE;
  const long blocks = (a.B + 31) / 32;
  if (blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(encoder_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

int launch_frame1(const Frame1Args& a, hipStream_t stream) {
  if (a.B <= 0 || a.K <= 0 || a.M <= 0) return HILC_ERR_SHAPE;
  if (a.dw_w != nullptr && (a.ksize < 1 || a.ksize > 32)) return HILC_ERR_UNSUPPORTED;
  const unsigned gy = (unsigned)((a.B + 31) / 32);
  if (gy > 65535u) return HILC_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((a.M + 31) / 32), gy);
  HILC_CLEAR_ERROR();
  if (a.dw_w != nullptr)
    hipLaunchKernelGGL(frame1_kernel<true>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(frame1_kernel<false>, grid, dim3(256), 0, stream, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}
