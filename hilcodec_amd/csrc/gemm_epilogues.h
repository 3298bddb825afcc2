// Epilogues of the fp32-MFMA GEMM cores (gemm_core.h / gemm_lin.h / gemm_stream.h): they consume the accumulator tile
// of a workgroup (MB 32-row blocks x 128 columns, wave w owns columns [32w, 32w+32)) and may route it through LDS to
// apply a depthwise convolution along time before anything touches HBM.
#pragma once
#include "gemm_core.h"

namespace hilc {

// the linear-addressing cores use 32-bit byte offsets into x
static inline bool lin_ok(int B, int K, int T) { return (long)B * K * T * 4 < (1L << 32); }
// per-clip halo tiles (TileCols) clamp a column group to [0, T - 4]: they need T >= 4 on top
static inline bool lin_tile_ok(int B, int K, int T) { return T >= 4 && lin_ok(B, K, T); }

// n / d for n < 2^31 as __umulhi(n, magic) >> shift (Granlund-Montgomery): l = ceil(log2 d), magic = ceil(2^(31+l)/d); d >= 2
static inline void div_magic(int d, unsigned& m, unsigned& sh) {
  int l = 0;
  while ((1L << l) < d) ++l;
  if (l < 1) l = 1;
  m = (unsigned)(((1ULL << (31 + l)) + (unsigned long long)d - 1) / (unsigned long long)d);
  sh = (unsigned)(l - 1);
}

// ================================================================================================
// Epilogues
// ================================================================================================
struct PwEpilogue {
  float* y;
  const float* bias;
  const float* res;
  int M, T;
  long ncols;
  float out_scale;
  template <int MB> static constexpr int lds_floats() { return 0; }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float*, int m0, long ntile, int wave, int lane, int) const {
    long n = ntile * BN + wave * 32 + (lane & 31);
    if (n >= ncols) return;
    long b = n / T;
    long colbase = b * (long)M * T + (n - b * T);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = m0 + i * 32 + acc_row(r, lane);
        if (row < M) {
          long off = colbase + (long)row * T;
          float v = acc[i][r];
          // separate roundings, like the reference's `y.mul_(scale)` then `x.add_(y)` (no FMA contraction)
          if (bias != nullptr) v = __fadd_rn(v, bias[row]);
          v = __fmul_rn(v, out_scale);
          if (res != nullptr) v = __fadd_rn(v, res[off]);
          y[off] = v;
        }
      }
    }
  }
};

// rows come in (re, im) pairs: row 2k = cos_k, row 2k+1 = sin_k; regs (2j, 2j+1) of a lane hold a pair.
struct StftEpilogue {
  float* spec;
  int nbins, Tf;  // nbins = n_fft/2+1
  long ncols;
  float mean, stdv;
  int normalize;
  template <int MB> static constexpr int lds_floats() { return 0; }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float*, int m0, long ntile, int wave, int lane, int) const {
    const SpecFinish fin = SpecFinish::make(mean, stdv, normalize);
    long n = ntile * BN + wave * 32 + (lane & 31);
    if (n >= ncols) return;
    long b = n / Tf;
    long colbase = b * (long)nbins * Tf + (n - b * Tf);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        int row = m0 + i * 32 + acc_row(r, lane);
        int bin = row >> 1;
        if (bin < nbins) {
          float re = acc[i][r], im = acc[i][r + 1];
          // x.square().sum(dim=1).clamp_min(1e-12).sqrt()  (conv.py:357) — no FMA contraction
          const float v = fin(re, im);
          spec[colbase + (long)bin * Tf] = v;
        }
      }
    }
  }
};

// the same for per-clip column tiles (tile = frames [128*tix, +128) of clip b), used with StftSegB
struct StftClipEpilogue {
  float* spec;
  int nbins, Tf, tiles;
  float mean, stdv;
  int normalize;
  template <int MB> static constexpr int lds_floats() { return 0; }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float*, int m0, long ntile, int wave, int lane, int) const {
    const SpecFinish fin = SpecFinish::make(mean, stdv, normalize);
    const long b = ntile / tiles;
    const int f = (int)(ntile - b * tiles) * BN + wave * 32 + (lane & 31);
    if (f >= Tf) return;
    const long colbase = b * (long)nbins * Tf + f;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int row = m0 + i * 32 + acc_row(r, lane);
        const int bin = row >> 1;
        if (bin < nbins) {
          const float re = acc[i][r], im = acc[i][r + 1];
          const float v = fin(re, im);
          spec[colbase + (long)bin * Tf] = v;
        }
      }
    }
  }
};

constexpr int HS = 132;  // LDS row stride of the post-GEMM tile: 128 columns + 4, keeps float4 alignment

constexpr int CH = 2;    // the post-GEMM tile goes through LDS in chunks of CH 32-row blocks (33 KB)

// chunk c of the accumulators -> LDS rows [0, 32*nblk)
template <int MB>
__device__ __forceinline__ int acc_chunk_to_lds(const f32x16 (&acc)[MB], float* smem, int c, int wave, int lane) {
  __syncthreads();  // previous readers of the tile (staged K slices / previous chunk) are done
  const int nblk = (MB - c * CH) < CH ? (MB - c * CH) : CH;
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    if (i >= c * CH && i < c * CH + CH) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        smem[((i - c * CH) * 32 + acc_row(r, lane)) * HS + wave * 32 + (lane & 31)] = acc[i][r];
    }
  }
  __syncthreads();
  return nblk;
}

// Pointwise epilogue through LDS (T % 4 == 0, 16-B aligned y / res): the accumulator layout gives every lane 16
// scattered rows of one column (64 dword stores + 64 dword residual loads per lane and tile); transposed through
// LDS a thread owns 16 consecutive columns of one row: 4 x 16-B stores / loads.  Same arithmetic as PwEpilogue.
struct PwLdsEpilogue {
  float* y;
  const float* bias;
  const float* res;
  int M, T;
  long ncols;
  unsigned t_magic, t_shift;   // n / T for n < 2^31
  float out_scale;
  template <int MB> static constexpr int lds_floats() { return 32 * (MB < CH ? MB : CH) * HS; }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int tid) const {
#pragma unroll
    for (int ch = 0; ch < (MB + CH - 1) / CH; ++ch) {
      const int nblk = acc_chunk_to_lds<MB>(acc, smem, ch, wave, lane);
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int seg = tid + NT * s;
        const int row = seg >> 3, c0 = (seg & 7) * 16;
        const int m = m0 + ch * CH * 32 + row;
        if (row >= nblk * 32 || m >= M) continue;
        const float bv = bias != nullptr ? bias[m] : 0.f;
        const long n0 = ntile * BN + c0;
        unsigned b = __umulhi((unsigned)(n0 < ncols ? n0 : 0), t_magic) >> t_shift;
        int t = (int)((n0 < ncols ? n0 : 0) - (long)b * T);
        const float* hrow = smem + row * HS + c0;
        // offsets of the four column groups first, then ALL shortcut loads, then the stores: `res` may alias `y`, so a
        // shortcut load written after a store stays behind it (one exposed HBM round trip per group otherwise)
        long offs[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          offs[g] = ((long)b * M + m) * (long)T + t;
          t += 4;
          if (t >= T) { t = 0; ++b; }
        }
        f32x4 rq[4];
        if (res != nullptr) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (n0 + 4 * g < ncols) rq[g] = *reinterpret_cast<const f32x4*>(res + offs[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (n0 + 4 * g < ncols) {
            f32x4 v = *reinterpret_cast<const f32x4*>(hrow + 4 * g);
            // separate roundings, like the reference's `y.mul_(scale)` then `x.add_(y)` (no FMA contraction)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float a = v[e];
              if (bias != nullptr) a = __fadd_rn(a, bv);
              v[e] = __fmul_rn(a, out_scale);
            }
            if (res != nullptr) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], rq[g][e]);
            }
            *reinterpret_cast<f32x4*>(y + offs[g]) = v;
          }
        }
      }
    }
  }
};

// depthwise causal conv, k = 5, stride 1, on the GEMM tile: column c <-> time t0 + c, t0 = tix*124 - 4;
// output column c >= 4 reads columns c-4..c.  Thread = (row, 16-column segment).
struct Dw5Epilogue {
  float* y;
  const float* dw_w;   // [M][5]
  const float* dw_b;   // [M] or null
  const float* res;    // [B][M][T] or null (may alias y)
  int M, T, tiles;
  float out_scale;
  int out_elu;
  int vec;  // T % 4 == 0 and y/res 16-B aligned
  static constexpr int STEP = BN - 4;
  template <int MB> static constexpr int lds_floats() { return 32 * (MB < CH ? MB : CH) * HS; }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int tid) const {
    long b = ntile / tiles;
    int tix = (int)(ntile - b * tiles);
    int t0 = tix * STEP - 4;
#pragma unroll
    for (int ch = 0; ch < (MB + CH - 1) / CH; ++ch) {
    const int nblk = acc_chunk_to_lds<MB>(acc, smem, ch, wave, lane);
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      int seg = tid + NT * s;
      int row = seg >> 3, c0 = (seg & 7) * 16;
      int m = m0 + ch * CH * 32 + row;
      if (row >= nblk * 32 || m >= M) continue;
      float v[20];
      const float* hrow = smem + row * HS;
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        int c = c0 - 4 + 4 * g;
        float4 q = c >= 0 ? *reinterpret_cast<const float4*>(hrow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
      }
      float w[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) w[j] = dw_w[(long)m * 5 + j];
      float bias = dw_b ? dw_b[m] : 0.f;
      long rowoff = (b * M + m) * (long)T;
      // the shortcut rows of all four column groups are requested before the first store: `res` may alias `y`, so the
      // compiler keeps every later shortcut load behind the earlier stores — four exposed HBM round trips per segment
      // (the epilogue is ~20 % of a workgroup's life at K = 384, tools/lin_phase_times.py)
      float4 rq[4];
#ifndef HILC_DW5_RES_PREFETCH
#define HILC_DW5_RES_PREFETCH 1
#endif
      if (HILC_DW5_RES_PREFETCH && vec && res != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = c0 + 4 * g, t = t0 + c;
          if (c >= 4 && t < T) rq[g] = *reinterpret_cast<const float4*>(res + rowoff + t);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int c = c0 + 4 * g;
        int t = t0 + c;
        if (c < 4 || t >= T) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < 5; ++j) a = fmaf(w[j], v[4 * g + e + j], a);   // taps in order j = 0..4
          a = __fadd_rn(a, bias);
          a = __fmul_rn(a, out_scale);
          o[e] = a;
        }
        if (vec) {
          if (res != nullptr) {
            const float4 rr = HILC_DW5_RES_PREFETCH ? rq[g] : *reinterpret_cast<const float4*>(res + rowoff + t);
            o[0] = __fadd_rn(o[0], rr.x); o[1] = __fadd_rn(o[1], rr.y);
            o[2] = __fadd_rn(o[2], rr.z); o[3] = __fadd_rn(o[3], rr.w);
          }
          if (out_elu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = elu_fast(o[e]);
          }
          *reinterpret_cast<float4*>(y + rowoff + t) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (t + e < T) {
              float a = o[e];
              if (res != nullptr) a = __fadd_rn(a, res[rowoff + t + e]);
              if (out_elu) a = elu_fast(a);
              y[rowoff + t + e] = a;
            }
          }
        }
      }
    }
    }
  }
};

// The same depthwise k = 5, stride 1 epilogue for the wave-row GEMM form (gemm_lin.h: gemm_lin_wr_kernel): after
// wr_time_order() lane (c, h) holds columns 64 h .. 64 h + 63 of channel m0 + c in registers, so the five taps are FMAs on
// the accumulators with per-lane (= per-channel) tap registers; the only cross-lane traffic are the 4 columns in front of
// the upper half (v_permlane32_swap from the lower half's last 4).  No LDS, no barrier; same taps-in-order fmaf chain and
// the same separately rounded bias / scale / shortcut steps as Dw5Epilogue (bit-identical).  M % 128 == 0, T % 4 == 0,
// y / res 16-B aligned (the launcher checks); `res` may alias `y` (each lane reads its own addresses before it writes them).
template <bool RES, bool OELU>
struct Dw5RegEpilogue {
  float* y;
  const float* dw_w;   // [M][5]
  const float* dw_b;   // [M] or null
  const float* res;    // [B][M][T] (RES; may alias y)
  int M, T, tiles;
  float out_scale;
  static constexpr int STEP = BN - 4;
#ifndef HILC_WR_GB
#define HILC_WR_GB 4     // 4-column groups per batch (the shortcut loads of a batch are in flight while the previous batch computes)
#endif

  __device__ void run_wr(f32x16 (&acc)[4], int mrow0, long ntile, int lane) const {
    const long b = ntile / tiles;
    const int t0 = (int)(ntile - b * tiles) * STEP - 4;
    const bool FULL = t0 + BN <= T;      // uniform: every column of the tile lies inside [0, T)
    constexpr int GB = HILC_WR_GB;
    const int h = lane >> 5;
    const int m = mrow0 + (lane & 31);
    const int tb = t0 + 64 * h;                                       // time of this lane's column 0
    float* yrow = y + (b * M + m) * (long)T + tb;
    const float* rrow = res + (b * M + m) * (long)T + tb;
    float w[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) w[j] = dw_w[(long)m * 5 + j];
    const float bias = dw_b ? dw_b[m] : 0.f;
    // group G = columns 4G .. 4G+3 of the lane's half; the tile's first 4 columns (h == 0, G == 0) are halo only
    auto live = [&](int G) { return (G > 0 || h != 0) && (FULL || tb + 4 * G < T); };
    f32x4 rq[2][GB];
    auto load_res = [&](int batch) {
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const int G = batch * GB + g;
        if (live(G)) rq[batch & 1][g] = *reinterpret_cast<const f32x4*>(rrow + 4 * G);
      }
    };
    if (RES) load_res(0);
    wr_time_order(acc);
    float prev[4];                      // the 4 columns in front of the current group
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // lower half: zeros (its first outputs are discarded); upper half: the lower half's columns 60..63
      const float src = HILC_WR_V(acc, 60 + j);
      const auto p = __builtin_amdgcn_permlane32_swap(0u, __float_as_uint(src), false, false);
      const unsigned p0 = p[0];
      prev[j] = __uint_as_float(p0);
    }
#pragma unroll
    for (int batch = 0; batch < 16 / GB; ++batch) {
      if (RES && batch + 1 < 16 / GB) load_res(batch + 1);
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const int G = batch * GB + g;
        const float v[8] = {prev[0], prev[1], prev[2], prev[3], HILC_WR_V(acc, 4 * G), HILC_WR_V(acc, 4 * G + 1),
                            HILC_WR_V(acc, 4 * G + 2), HILC_WR_V(acc, 4 * G + 3)};
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < 5; ++j) a = fmaf(w[j], v[e + j], a);   // taps in order j = 0..4
          a = __fadd_rn(a, bias);
          o[e] = __fmul_rn(a, out_scale);
        }
        if (RES) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(o[e], rq[batch & 1][g][e]);
        }
        if (OELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = elu_fast(o[e]);
        }
        if (live(G)) *reinterpret_cast<f32x4*>(yrow + 4 * G) = o;
#pragma unroll
        for (int j = 0; j < 4; ++j) prev[j] = v[4 + j];
      }
    }
  }
};

// Down-sampling layers (depthwise k = 2r, stride r, r = 2 or 4) on the wave-row form of the GEMM tile: a lane holds one channel and
// 64 contiguous columns of it (lanes 32-63: columns 64-127), so output k of a half is
//     sum_j w[j] * U[-r + r*k + j],   j = 0 .. 2r-1 (in order),   U[q >= 0] = the lane's column q, U[q < 0] = the other half's last r
// for BOTH halves (the lower half's first H/r values of k are the tile's halo and are not stored): tile output i = k - H/r in the
// lower half, 64/r - H/r + k in the upper one.  Same products in the same order as DwStrideEpilogue, no LDS round trip.
// Offline only (no cache), M % 128 == 0, T % 4 == 0.  The launcher uses R = 4 (R = 2 measured slower than the LDS form: gemm.hip).
template <int R>
struct DwStrideRegEpilogue {
  float* y;
  const float* dw_w;   // [M][2R]
  const float* dw_b;
  int M, To, tiles, n_out;
  static constexpr int H = 4;
  static_assert(R == 2 || R == 4, "strides of the first two encoder stages");

  __device__ void run_wr(f32x16 (&acc)[4], int mrow0, long ntile, int lane) const {
    constexpr int NK = 64 / R, SH = H / R;
    const long b = ntile / tiles;
    const int o0 = (int)(ntile - b * tiles) * n_out;
    const int h = lane >> 5;
    const int m = mrow0 + (lane & 31);
    float w[2 * R];
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) w[j] = dw_w[(long)m * (2 * R) + j];
    const float bias = dw_b ? dw_b[m] : 0.f;
    wr_time_order(acc);
    float u[R + 64];                     // U[-R .. 63] at u[0 .. R+63]
#pragma unroll
    for (int j = 0; j < R; ++j) {        // upper half: the lower half's columns 64-R .. 63 (lower half: zeros, its first outputs are halo)
      const float src = HILC_WR_V(acc, 64 - R + j);
      const auto p = __builtin_amdgcn_permlane32_swap(0u, __float_as_uint(src), false, false);
      const unsigned p0 = p[0];
      u[j] = __uint_as_float(p0);
    }
#pragma unroll
    for (int q = 0; q < 64; ++q) u[R + q] = HILC_WR_V(acc, q);
    const int i0 = h ? NK - SH : -SH;    // tile output of k = 0
    float* const yrow = y + (b * M + m) * (long)To + o0 + i0;
#pragma unroll
    for (int k = 0; k < NK; k += 2) {
      float o[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) a = fmaf(w[j], u[R * (k + e) + j], a);
        o[e] = dw_b ? __fadd_rn(a, bias) : a;
      }
      // (R = 2: o0 and i0 are even, the pair is 8-B aligned; R = 4: scalar stores)
      const int i = i0 + k;
      if (R == 2) {
        if (i >= 0 && i < n_out && o0 + i + 1 < To) *reinterpret_cast<f32x2*>(yrow + k) = f32x2{o[0], o[1]};
        else if (i >= 0 && i < n_out && o0 + i < To) yrow[k] = o[0];
      } else {
        if (i >= 0 && i < n_out && o0 + i < To) yrow[k] = o[0];
        if (i + 1 >= 0 && i + 1 < n_out && o0 + i + 1 < To) yrow[k + 1] = o[1];
      }
    }
  }
};

// depthwise causal conv k = 2r, stride r (down-sampling): tile covers times [o0*r - H, +128) with
// H = round_up(r, 4); output o0 + i reads columns H - r + i*r + j, j < 2r.
struct DwStrideEpilogue {
  float* y;
  const float* dw_w;   // [M][2r]
  const float* dw_b;
  // streaming hop longer than one tile: cache [B][M][r] = the last r pointwise outputs of the previous hop (they
  // replace the zero padding in front of a clip's first tile), and its successor, written from the clip's last tile
  const float* hist = nullptr;
  float* hist_out = nullptr;
  int T = 0;           // input samples per clip (needed for hist_out only)
  int M, To, tiles, r, H, n_out;
  template <int MB> static constexpr int lds_floats() { return 32 * (MB < CH ? MB : CH) * HS; }

  // One wave = one row at a time (row = wave + 4*s is wave-uniform, so the 2r taps and the bias come through
  // scalar loads), lane = output i of the tile (n_out <= 64): no index division, coalesced stores, taps unrolled
  // for the strides the codec uses.
  template <int MB, int KR>
  __device__ void rows(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane) const {
    const long b = ntile / tiles;
    const int o0 = (int)(ntile - b * tiles) * n_out;
    const int k = KR > 0 ? KR : 2 * r;
    const int o = o0 + lane;
    const bool live = lane < n_out && o < To;
    const bool first = hist != nullptr && o0 == 0;          // uniform: a clip's first tile in a streaming hop
    const bool last = hist_out != nullptr && o0 + n_out >= To;
    // rows in groups of RG: the taps and biases of the whole group are requested (scalar loads) before the first row is
    // computed — one exposed scalar-memory round trip per group instead of one per row (at K = 64 / 128 this epilogue,
    // not the K loop, is most of the tile)
#ifdef HILC_DWS_RG
    constexpr int RG = HILC_DWS_RG;
#else
    constexpr int RG = KR > 0 && KR <= 8 ? 4 : 2;
#endif
#pragma unroll
    for (int ch = 0; ch < (MB + CH - 1) / CH; ++ch) {
      const int nblk = acc_chunk_to_lds<MB>(acc, smem, ch, wave, lane);
      for (int s0 = 0; s0 < 8 * nblk; s0 += RG) {
        float wv[RG][KR > 0 ? KR : 1];
        float bv[RG];
        int mrow[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const int row = __builtin_amdgcn_readfirstlane(wave + 4 * (s0 + g));
          const int m = m0 + ch * CH * 32 + row;
          mrow[g] = m;
          const int mc = m < M ? m : M - 1;     // uniform clamp: rows past M are computed on row M-1's taps and not stored
          if (KR > 0) {
#pragma unroll
            for (int j = 0; j < (KR > 0 ? KR : 1); ++j) wv[g][j] = dw_w[(long)mc * k + j];
          }
          bv[g] = dw_b ? dw_b[mc] : 0.f;
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const int row = __builtin_amdgcn_readfirstlane(wave + 4 * (s0 + g));
          const int m = mrow[g];
          if (s0 + g >= 8 * nblk || m >= M) continue;      // uniform
          const float* w = dw_w + (long)m * k;
          const float* h = smem + row * HS + (H - r) + (live ? lane : 0) * r;
          float a = 0.f;
          if (first) {                             // output 0's first r taps lie before t = 0: the cache, not the zero halo
            const bool from_cache = lane == 0;
            const float* hc = hist + (b * M + m) * (long)r;
            for (int j = 0; j < k; ++j) a = fmaf(w[j], (from_cache && j < r) ? hc[j] : h[j], a);
          } else if (KR == 8 && H == 4) {
            // stride 4: a lane's 8 operands are two aligned 16-B words, consecutive lanes are consecutive words — two conflict-free
            // ds_read_b128 instead of four ds_read2_b32 whose lanes sit 4 dwords apart (8 lanes per bank)
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(h), v1 = *reinterpret_cast<const f32x4*>(h + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaf(wv[g][j < KR ? j : 0], v[j], a);
          } else if (KR == 4 && H == 4) {
            const f32x2 v0 = *reinterpret_cast<const f32x2*>(h), v1 = *reinterpret_cast<const f32x2*>(h + 2);
            const float v[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
            for (int j = 0; j < 4; ++j) a = fmaf(wv[g][j < KR ? j : 0], v[j], a);
          } else if (KR > 0) {
#pragma unroll
            for (int j = 0; j < (KR > 0 ? KR : 1); ++j) a = fmaf(wv[g][j], h[j], a);
          } else {
            for (int j = 0; j < k; ++j) a = fmaf(w[j], h[j], a);
          }
          if (dw_b) a = __fadd_rn(a, bv[g]);
          if (live) y[(b * M + m) * (long)To + o] = a;
          if (last && lane < r) {                  // the clip's last tile: columns of t = T-r .. T-1
            const int c = (T - r + lane) - (o0 * r - H);
            hist_out[(b * M + m) * (long)r + lane] = smem[row * HS + c];
          }
        }
      }
    }
  }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int) const {
    switch (r) {                                 // uniform
      case 2: rows<MB, 4>(acc, smem, m0, ntile, wave, lane); break;
      case 4: rows<MB, 8>(acc, smem, m0, ntile, wave, lane); break;
      case 5: rows<MB, 10>(acc, smem, m0, ntile, wave, lane); break;
      case 8: rows<MB, 16>(acc, smem, m0, ntile, wave, lane); break;
      default: rows<MB, 0>(acc, smem, m0, ntile, wave, lane); break;
    }
  }
};

// The same down-sampling conv for a STREAMING hop longer than one tile (the encoder's first two stages: 320 / 160 samples
// per stream), on FLAT columns (FlatHaloCols): tile = flat inputs [ntile*n_out*r - H, +128) of the stream-major column
// space, outputs O = ntile*n_out + lane of the flat output space [B*To].  A stream's FIRST output takes its first r taps from
// that stream's cache (zeros without one) instead of the columns to its left, which belong to the previous stream: the cache
// words of the (at most NBND) stream starts of a tile are staged into a small LDS side buffer while the accumulators go to
// LDS (published by the same barrier), and a head lane reads the FIRST HALF of its 2r operands through a pointer into that
// buffer — every lane issues the same two vector reads.  The new cache = a stream's last r pointwise outputs, written by r lanes
// per row where a stream ends in the tile.  Same fmaf chain (taps j ascending) as DwStrideEpilogue: bit-identical outputs.
struct DwStrideFlatEpilogue {
  float* y;
  const float* dw_w;   // [M][2r]
  const float* dw_b;
  const float* hist = nullptr;    // [B][M][r]
  float* hist_out = nullptr;
  const float* res = nullptr;     // [B][M][To] added to the output (the next stage's SpecBlock branch); may not alias y
  int B, M, T, To, r, H, n_out;
  unsigned to_magic, to_shift;    // O / To for O < 2^31
  static constexpr int NBND = 2;  // stream starts per tile the side buffer holds (the launcher checks To >= n_out / NBND)
  static constexpr int HB = 8;    // side-buffer floats per (boundary, row): r <= 8 (4 KB: four workgroups per CU still fit)
  template <int MB> static constexpr int lds_floats() { return 32 * (MB < CH ? MB : CH) * HS + NBND * 32 * CH * HB; }

  template <int MB, int KR>
  __device__ void rows(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int tid) const {
    const int k = KR > 0 ? KR : 2 * r;
    float* const hbuf = smem + 32 * (MB < CH ? MB : CH) * HS;       // [NBND][rows of a chunk][HB]
    // everything below is 32-bit (the launcher checks B*To < 2^31 and B*M*To < 2^31): the kernel sits at its register budget
    const int O0 = (int)ntile * n_out;                    // first flat output of the tile (uniform)
    const int O = O0 + lane;
    const bool in_range = lane < n_out && O < B * To;
    const unsigned ob = in_range ? __umulhi((unsigned)O, to_magic) >> to_shift : 0u;     // stream of this lane's output
    const int o_in = O - (int)ob * To;
    const unsigned yoff = ob * (unsigned)(M * To) + (unsigned)o_in;                      // + m * To
    // stream boundaries b*To inside [O0, O0 + n_out]: b = hb0, hb0 + 1, ... (uniform); a head lane's boundary slot = ob - hb0
    const int hb0 = (int)(__umulhi((unsigned)(O0 + To - 1), to_magic) >> to_shift);
    const bool head = in_range && o_in == 0;
    const int hslot = head ? (int)ob - hb0 : 0;           // < NBND
    constexpr int RG = KR > 0 && KR <= 8 ? 4 : 2;
#pragma unroll
    for (int ch = 0; ch < (MB + CH - 1) / CH; ++ch) {
      // accumulators of the chunk -> LDS (acc_chunk_to_lds), and the cache words of the tile's stream starts -> side buffer
      __syncthreads();
      const int nblk = (MB - ch * CH) < CH ? (MB - ch * CH) : CH;
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        if (i >= ch * CH && i < ch * CH + CH) {
#pragma unroll
          for (int q = 0; q < 16; ++q)
            smem[((i - ch * CH) * 32 + acc_row(q, lane)) * HS + wave * 32 + (lane & 31)] = acc[i][q];
        }
      }
      for (int e = tid; e < NBND * 32 * CH * HB; e += NT) {
        const int j = e % HB, row = (e / HB) % (32 * CH), slot = e / (HB * 32 * CH);
        const int bq = hb0 + slot, m = m0 + ch * CH * 32 + row;
        float v = 0.f;
        if (hist != nullptr && j < r && bq < B && m < M && row < 32 * nblk) v = hist[(long)(bq * M + m) * r + j];
        hbuf[e] = v;
      }
      __syncthreads();
      for (int s0 = 0; s0 < 8 * nblk; s0 += RG) {
        float wv[RG][KR > 0 ? KR : 1];
        float bv[RG];
        int mrow[RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const int row = __builtin_amdgcn_readfirstlane(wave + 4 * (s0 + g));
          const int m = m0 + ch * CH * 32 + row;
          mrow[g] = m;
          const int mc = m < M ? m : M - 1;
          if (KR > 0) {
#pragma unroll
            for (int j = 0; j < (KR > 0 ? KR : 1); ++j) wv[g][j] = dw_w[(long)mc * k + j];
          }
          bv[g] = dw_b ? dw_b[mc] : 0.f;
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const int row = __builtin_amdgcn_readfirstlane(wave + 4 * (s0 + g));
          const int m = mrow[g];
          if (s0 + g >= 8 * nblk || m >= M) continue;      // uniform
          const float* w = dw_w + (long)m * k;
          const float* hhi = smem + row * HS + H + (in_range ? lane : 0) * r;         // second half: the output's own r columns
          const float* hlo = head ? hbuf + (hslot * 32 * CH + row) * HB : hhi - r;   // first half: the r columns before, or the cache
          float a = 0.f;
          if (KR == 8) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(hlo), v1 = *reinterpret_cast<const f32x4*>(hhi);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaf(wv[g][j < KR ? j : 0], v[j], a);
          } else if (KR == 4) {
            const f32x2 v0 = *reinterpret_cast<const f32x2*>(hlo), v1 = *reinterpret_cast<const f32x2*>(hhi);
            const float v[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
            for (int j = 0; j < 4; ++j) a = fmaf(wv[g][j < KR ? j : 0], v[j], a);
          } else if (KR > 0) {
#pragma unroll
            for (int j = 0; j < KR / 2; ++j) a = fmaf(wv[g][j], hlo[j], a);
#pragma unroll
            for (int j = 0; j < KR / 2; ++j) a = fmaf(wv[g][KR / 2 + j], hhi[j], a);
          } else {
            for (int j = 0; j < r; ++j) a = fmaf(w[j], hlo[j], a);
            for (int j = 0; j < r; ++j) a = fmaf(w[r + j], hhi[j], a);
          }
          if (dw_b) a = __fadd_rn(a, bv[g]);
          if (in_range) {
            const unsigned yo = yoff + (unsigned)(m * To);
            y[yo] = res != nullptr ? __fadd_rn(a, res[yo]) : a;
          }
          if (hist_out != nullptr && lane < r) {           // streams that END in this tile: their last r columns are the new cache
            for (int bq = hb0 > 1 ? hb0 : 1; (bq - 1) * To + To - 1 < O0 + n_out && bq <= B; ++bq) {      // uniform: 0-2 of them
              if (bq * To - 1 >= O0)
                hist_out[(long)((bq - 1) * M + m) * r + lane] = smem[row * HS + H + (bq * To - O0) * r - r + lane];
            }
          }
        }
      }
    }
  }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int tid) const {
    switch (r) {                                 // uniform
      case 2: rows<MB, 4>(acc, smem, m0, ntile, wave, lane, tid); break;
      case 4: rows<MB, 8>(acc, smem, m0, ntile, wave, lane, tid); break;
      case 5: rows<MB, 10>(acc, smem, m0, ntile, wave, lane, tid); break;
      case 8: rows<MB, 16>(acc, smem, m0, ntile, wave, lane, tid); break;
      default: rows<MB, 0>(acc, smem, m0, ntile, wave, lane, tid); break;
    }
  }
};

// Streaming hop, wide layers (T <= 128 samples per stream and call): a tile holds `cpt` WHOLE clips
// (columns q*T + t), so there is no halo — the samples before t = 0 come from the cache
// hist[b][m][pad] (pad = ksize - stride: the last `pad` pointwise outputs of the previous hop,
// causal_layers.py:147-165) and the new cache is written from the tile.  Generic k / stride (k5 s1 and
// k = 2r stride r); work item = (clip, row, output), output fastest: y / hist_out writes of a clip's
// rows are contiguous.
struct DwSegEpilogue {
  float* y;
  const float* dw_w;   // [M][k]
  const float* dw_b;
  const float* res;    // [B][M][To] or null (may alias y)
  const float* hist;   // [B][M][pad] or null (zeros)
  float* hist_out;     // [B][M][pad] or null
  int B, M, T, To, k, stride, pad, cpt;
  unsigned to_magic, to_shift, pad_magic, pad_shift;   // n / To, n / pad as __umulhi(n, magic) >> shift
  float out_scale;
  int out_elu;
  template <int MB> static constexpr int lds_floats() { return 32 * (MB < CH ? MB : CH) * HS; }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int tid) const {
    if (k == 16 && stride == 8) body<MB, 16>(acc, smem, m0, ntile, wave, lane, tid);        // uniform
    else if (k == 10 && stride == 5) body<MB, 10>(acc, smem, m0, ntile, wave, lane, tid);
    else body<MB, 0>(acc, smem, m0, ntile, wave, lane, tid);
  }

  template <int MB, int KS>
  __device__ void body(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int tid) const {
#pragma unroll
    for (int ch = 0; ch < (MB + CH - 1) / CH; ++ch) {
      const int nblk = acc_chunk_to_lds<MB>(acc, smem, ch, wave, lane);
      const int rows = 32 * nblk;                       // 32 or 64
      const int rsh = nblk == 1 ? 5 : 6;
      for (int idx = tid; idx < cpt * rows * To; idx += NT) {
        const int qr = To == 1 ? idx : (int)(__umulhi((unsigned)idx, to_magic) >> to_shift);   // idx / To
        const int o = idx - qr * To;
        const int q = qr >> rsh, row = qr & (rows - 1);
        const long b = ntile * cpt + q;
        const int m = m0 + ch * CH * 32 + row;
        if (m >= M || b >= B) continue;
        const float* h = smem + row * HS + q * T;
        const float* hp = hist != nullptr ? hist + (b * M + m) * (long)pad + pad : nullptr;   // hp[tt], tt < 0
        const float* w = dw_w + (long)m * k;
        float a = 0.f;
        if (KS > 0) {
          // the codec's down-sampling layers (k = 2 * stride = 16 or 10): taps and operands requested before the chain (same
          // chain, j ascending); only output 0 of a clip reaches into the cache, with its first k - stride taps
          constexpr int KK = KS > 0 ? KS : 2, PAD = KK / 2;
          float wv[KK], v[KK];
          if (KK % 4 == 0) {
#pragma unroll
            for (int j4 = 0; j4 < KK / 4; ++j4) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(w + 4 * j4);
              wv[4 * j4] = t.x; wv[4 * j4 + 1] = t.y; wv[4 * j4 + 2] = t.z; wv[4 * j4 + 3] = t.w;
            }
          } else {
#pragma unroll
            for (int j2 = 0; j2 < KK / 2; ++j2) {
              const f32x2 t = *reinterpret_cast<const f32x2*>(w + 2 * j2);
              wv[2 * j2] = t.x; wv[2 * j2 + 1] = t.y;
            }
          }
          const float* hs = h + o * PAD - PAD;          // stride == PAD
          if (o == 0) {
#pragma unroll
            for (int j = 0; j < PAD; ++j) v[j] = hp != nullptr ? hp[j - PAD] : 0.f;
          } else {
#pragma unroll
            for (int j = 0; j < PAD; ++j) v[j] = hs[j];
          }
#pragma unroll
          for (int j = PAD; j < KK; ++j) v[j] = hs[j];
#pragma unroll
          for (int j = 0; j < KK; ++j) a = fmaf(wv[j], v[j], a);
        } else {
          for (int j = 0; j < k; ++j) {
            const int tt = o * stride - pad + j;
            const float v = tt >= 0 ? h[tt] : (hp != nullptr ? hp[tt] : 0.f);
            a = fmaf(w[j], v, a);
          }
        }
        if (dw_b != nullptr) a = __fadd_rn(a, dw_b[m]);
        a = __fmul_rn(a, out_scale);
        const long off = (b * M + m) * (long)To + o;
        if (res != nullptr) a = __fadd_rn(a, res[off]);
        if (out_elu) a = elu_fast(a);
        y[off] = a;
      }
      if (hist_out != nullptr) {
        for (int idx = tid; idx < cpt * rows * pad; idx += NT) {
          const int qr = pad == 1 ? idx : (int)(__umulhi((unsigned)idx, pad_magic) >> pad_shift);   // idx / pad
          const int i = idx - qr * pad;
          const int q = qr >> rsh, row = qr & (rows - 1);
          const long b = ntile * cpt + q;
          const int m = m0 + ch * CH * 32 + row;
          if (m >= M || b >= B) continue;
          const int src = T - pad + i;          // last `pad` samples of [cache | this hop]
          const long ho = (b * M + m) * (long)pad;
          hist_out[ho + i] = src >= 0 ? smem[row * HS + q * T + src] : (hist != nullptr ? hist[ho + pad + src] : 0.f);
        }
      }
    }
  }
};

// k = 5, stride 1, T % 4 == 0 (so T >= 4 and a 4-column group never straddles clips): vector form of the above.
// Thread = (row, 16-column segment) like Dw5Epilogue; the 4 samples before a clip's t = 0 come from the cache.
struct Dw5SegEpilogue {
  float* y;
  const float* dw_w;   // [M][5]
  const float* dw_b;
  const float* res;
  const float* hist;   // [B][M][4] or null
  float* hist_out;     // [B][M][4] or null
  int B, M, T, cpt;
  unsigned t_magic, t_shift;   // n / T
  float out_scale;
  int out_elu;
  template <int MB> static constexpr int lds_floats() { return 32 * (MB < CH ? MB : CH) * HS; }

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], float* smem, int m0, long ntile, int wave, int lane, int tid) const {
#pragma unroll
    for (int ch = 0; ch < (MB + CH - 1) / CH; ++ch) {
      const int nblk = acc_chunk_to_lds<MB>(acc, smem, ch, wave, lane);
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int seg = tid + NT * s;
        const int row = seg >> 3, c0 = (seg & 7) * 16;
        const int m = m0 + ch * CH * 32 + row;
        if (row >= nblk * 32 || m >= M) continue;
        float v[20];
        const float* hrow = smem + row * HS;
#pragma unroll
        for (int g = 0; g < 5; ++g) {
          const int c = c0 - 4 + 4 * g;
          const f32x4 qv = c >= 0 ? *reinterpret_cast<const f32x4*>(hrow + c) : f32x4{0.f, 0.f, 0.f, 0.f};
          v[4 * g] = qv.x; v[4 * g + 1] = qv.y; v[4 * g + 2] = qv.z; v[4 * g + 3] = qv.w;
        }
        float w[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] = dw_w[(long)m * 5 + j];
        const float bias = dw_b ? dw_b[m] : 0.f;
        int q = (int)(__umulhi((unsigned)c0, t_magic) >> t_shift);   // clip of the first column
        int t = c0 - q * T;
        f32x4 rq[4];
        if (res != nullptr) {                 // all shortcut loads before the first store (see PwLdsEpilogue)
          int q2 = q, t2 = t;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const long b2 = ntile * cpt + q2;
            if ((c0 + 4 * g) < cpt * T && b2 < B) rq[g] = *reinterpret_cast<const f32x4*>(res + (b2 * M + m) * (long)T + t2);
            t2 += 4;
            if (t2 >= T) { t2 = 0; ++q2; }
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long b = ntile * cpt + q;
          const bool live = (c0 + 4 * g) < cpt * T && b < B;
          if (live) {
            const long bm = b * M + m;
            float win[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) win[e] = v[4 * g + e];
            if (t == 0) {                     // clip start: the previous 4 pointwise outputs are the cache
              f32x4 hv = f32x4{0.f, 0.f, 0.f, 0.f};
              if (hist != nullptr) hv = *reinterpret_cast<const f32x4*>(hist + bm * 4);
              win[0] = hv.x; win[1] = hv.y; win[2] = hv.z; win[3] = hv.w;
            }
            if (hist_out != nullptr && t == T - 4)
              *reinterpret_cast<f32x4*>(hist_out + bm * 4) = f32x4{win[4], win[5], win[6], win[7]};
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float a = 0.f;
#pragma unroll
              for (int j = 0; j < 5; ++j) a = fmaf(w[j], win[e + j], a);
              a = __fadd_rn(a, bias);
              o[e] = __fmul_rn(a, out_scale);
            }
            const long off = bm * (long)T + t;
            if (res != nullptr) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(o[e], rq[g][e]);
            }
            if (out_elu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = elu_fast(o[e]);
            }
            *reinterpret_cast<f32x4*>(y + off) = o;
          }
          t += 4;
          if (t >= T) { t = 0; ++q; }
        }
      }
    }
  }
};


}  // namespace hilc
