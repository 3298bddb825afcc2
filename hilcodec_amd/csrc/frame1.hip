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

}  // namespace

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
