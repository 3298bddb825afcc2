// Linear-addressing flavour of the fp32-MFMA GEMM core (same tiling, LDS layout, MFMA order and epilogues as
// gemm_core.h) for B operands that are plain rows of a [.., K, T] tensor: pointwise convs whose 4-column groups
// never straddle clips (T % 4 == 0, 16-B aligned).
//
// Why a second core: PMC on the generic core (K = M = 768) shows MFMA-busy 72 % + VALU-active 19-23 % ~= 95 %
// of the time, i.e. for fp32 the matrix pipe and the vector ALUs do not overlap, and the generic K loop spends
// ~125 VALU + ~50 SALU instructions per wave per 32 MFMAs on 64-bit address arithmetic, pointer selects
// (compiled to exec-mask branches) and run-time loader flags.  Here every global address is
//     uniform base (SGPR, advanced once per K slice)  +  per-thread 32-bit byte offset (loop-invariant)
// so a load costs no VALU; the ragged K tail is handled by a second, pre-computed offset set (rows clamped to
// K-1) plus a zero scale for rows >= K (0 * finite == 0, and a non-finite sample only ever reaches its own column);
// ELU / halo-zeroing are compile-time.  Results are bit-identical to the generic core (same fmaf chain).
#pragma once
#include "gemm_core.h"

namespace hilc {

// Column mapping of one thread's 4-column group: element offset of (clip b, row 0, time t) and validity.
struct LinCol {
  unsigned off;   // b*K*T + t   (elements; the tensor is < 4 GiB, checked by the launcher)
  bool ok;        // the group exists (and, for halo tiles, lies inside [0, T))
};

// flattened (clip, t) columns, `tile_cols` per tile (BN, or whole clips for the streaming kernels)
struct FlatCols {
  int K, T, tile_cols;
  long ncols;
  static constexpr bool kZeroInvalid = false;   // invalid columns are never read back
  __device__ LinCol at(long ntile, int c) const {
    LinCol r;
    const long n = ntile * tile_cols + c;
    r.ok = c < tile_cols && n < ncols;
    const long b = r.ok ? n / T : 0;
    r.off = r.ok ? (unsigned)(b * (long)K * T + (n - b * T)) : 0u;
    return r;
  }
};

// per-clip tiles with a left halo (zero padding of the causal depthwise conv that follows)
struct TileCols {
  int K, T, tiles, step, halo;
  static constexpr bool kZeroInvalid = true;    // t < 0 must read as zero
  __device__ LinCol at(long ntile, int c) const {
    LinCol r;
    const long b = ntile / tiles;
    const int t = (int)(ntile - b * tiles) * step - halo + c;
    r.ok = t >= 0 && t < T;
    r.off = r.ok ? (unsigned)(b * (long)K * T + t) : 0u;
    return r;
  }
};

template <int MB, bool ELU, class Cols, class Epilogue>
__global__ __launch_bounds__(NT) void gemm_lin_kernel(const float* __restrict__ wt, const float* __restrict__ x, int M,
                                                      int K, int ldw, int T, long ntiles, int mtiles,
                                                      float in_scale, Cols cols, Epilogue ep) {
  constexpr int BM = 32 * MB;
  constexpr int AG = BK * BM / 4;
  constexpr int AP = (AG + NT - 1) / NT;
  constexpr int STG = 2 * BK * (BM + BN);
  constexpr int EPI = Epilogue::template lds_floats<MB>();
  constexpr int SM = STG > EPI ? STG : EPI;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
  float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);

  long id = blockIdx.x;   // XCD-aware tile order, as in gemm_core.h
  long grp = id / (8L * mtiles);
  int within = (int)(id - grp * 8L * mtiles);
  long ntile = grp * 8 + (within & 7);
  int mtile = within >> 3;
  if (ntile >= ntiles) return;
  const int m0 = mtile * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ktiles = (K + BK - 1) / BK;
  const int krem = K - (ktiles - 1) * BK;          // rows of the last slice (1..BK)

  // ---- loop-invariant per-thread byte offsets
  unsigned aoff[AP], aoff_last[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int g = tid + p * NT;
    const int kr = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
    int col = m0 + m4;
    col = col < ldw - 4 ? col : ldw - 4;           // rows >= M only feed accumulator rows that are never stored
    const int krl = kr < krem ? kr : krem - 1;
    aoff[p] = (unsigned)(kr * ldw + col) * 4u;
    aoff_last[p] = (unsigned)(krl * ldw + col) * 4u;
  }
  const LinCol lc = cols.at(ntile, (tid & 31) * 4);
  unsigned boff[BP], boff_last[BP];
  float sc_last[BP];
#pragma unroll
  for (int h = 0; h < BP; ++h) {
    const int r = (tid >> 5) + 8 * h;
    const int rl = r < krem ? r : krem - 1;
    boff[h] = lc.ok ? (lc.off + (unsigned)r * (unsigned)T) * 4u : 0u;
    boff_last[h] = lc.ok ? (lc.off + (unsigned)rl * (unsigned)T) * 4u : 0u;
    sc_last[h] = r < krem ? in_scale : 0.f;
  }
  const unsigned a_slice = (unsigned)BK * (unsigned)ldw * 4u;   // bytes per K slice
  const size_t b_slice = (size_t)BK * (size_t)T * 4u;

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  f32x4 ra[AP], rb[BP];
  auto fetch = [&](int kt, bool last) {
    const char* sa = reinterpret_cast<const char*>(wt) + (size_t)kt * a_slice;   // uniform
    const char* sb = reinterpret_cast<const char*>(x) + (size_t)kt * b_slice;
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const f32x4*>(sa + (last ? aoff_last[p] : aoff[p]));
#pragma unroll
    for (int h = 0; h < BP; ++h) rb[h] = *reinterpret_cast<const f32x4*>(sb + (last ? boff_last[h] : boff[h]));
  };
  auto stage = [&](int buf, bool last) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int g = tid + p * NT;
      if (AG % NT == 0 || g < AG) {
        const int k = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
        *reinterpret_cast<f32x4*>(&As[buf][k][m4]) = ra[p];
      }
    }
#pragma unroll
    for (int h = 0; h < BP; ++h) {
      f32x4 v = rb[h];
      if (Cols::kZeroInvalid) v = zero_unless(lc.ok, v);
      const float sc = last ? sc_last[h] : in_scale;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = v[e] * sc;
        v[e] = ELU ? elu_fast(s) : s;
      }
      *reinterpret_cast<f32x4*>(&Bs[buf][(tid >> 5) + 8 * h][(tid & 31) * 4]) = v;
    }
  };

  const int kh = lane >> 5, l31 = lane & 31;
  fetch(0, ktiles == 1);
  stage(0, ktiles == 1);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ktiles;
    const bool last = kt + 2 == ktiles;
    if (more) fetch(kt + 1, last);
    float av[2][MB], bv[2];
    bv[0] = Bs[buf][kh][wave * 32 + l31];
#pragma unroll
    for (int i = 0; i < MB; ++i) av[0][i] = As[buf][kh][i * 32 + l31];
#pragma unroll
    for (int j = 0; j < BK / 2; ++j) {
      const int cur = j & 1, nxt = cur ^ 1;
      if (j + 1 < BK / 2) {
        bv[nxt] = Bs[buf][2 * j + 2 + kh][wave * 32 + l31];
#pragma unroll
        for (int i = 0; i < MB; ++i) av[nxt][i] = As[buf][2 * j + 2 + kh][i * 32 + l31];
      }
#pragma unroll
      for (int i = 0; i < MB; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur], acc[i], 0, 0, 0);
    }
    if (more) {
      stage(buf ^ 1, last);
      __syncthreads();
    }
  }
  ep.template run<MB>(acc, smem, m0, ntile, wave, lane, tid);
}

// x must be 16-B aligned, T % 4 == 0, the tensor < 4 GiB (32-bit byte offsets); the caller checks.
template <class Cols, class Epilogue>
int launch_gemm_lin(const float* wt, const float* x, int M, int K, int ldw, int T, long ntiles, float in_scale,
                    bool in_elu, const Cols& cols, const Epilogue& ep, hipStream_t s) {
  int m32 = (M + 31) / 32;
  int MB;
  if (m32 % 4 == 0) MB = 4;
  else if (m32 % 3 == 0) MB = 3;
  else if (m32 < 4) MB = m32;
  else {
    int pad4 = (4 - m32 % 4) % 4, pad3 = (3 - m32 % 3) % 3;
    MB = pad3 < pad4 ? 3 : 4;
  }
  long groups = (ntiles + 7) / 8;
  while (MB > 1 && groups * 8 * ((m32 + MB - 1) / MB) < 128) --MB;
  if (const char* e = getenv("HILC_MB")) { int v = atoi(e); if (v >= 1 && v <= 4) MB = v; }   // tuning aid
  int mtiles = (m32 + MB - 1) / MB;
  long blocks = groups * 8 * mtiles;
  if (ntiles <= 0 || blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  dim3 grid((unsigned)blocks), block(NT);
  HILC_CLEAR_ERROR();
#define HILC_LIN(MBV, E) \
  hipLaunchKernelGGL((gemm_lin_kernel<MBV, E, Cols, Epilogue>), grid, block, 0, s, wt, x, M, K, ldw, T, ntiles, mtiles, in_scale, cols, ep)
  if (in_elu) {
    switch (MB) {
      case 1: HILC_LIN(1, true); break;
      case 2: HILC_LIN(2, true); break;
      case 3: HILC_LIN(3, true); break;
      default: HILC_LIN(4, true); break;
    }
  } else {
    switch (MB) {
      case 1: HILC_LIN(1, false); break;
      case 2: HILC_LIN(2, false); break;
      case 3: HILC_LIN(3, false); break;
      default: HILC_LIN(4, false); break;
    }
  }
#undef HILC_LIN
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

}  // namespace hilc
