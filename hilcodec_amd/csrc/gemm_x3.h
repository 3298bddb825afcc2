// EXPERIMENTAL numerics mode, never the default: the GEMM of a pointwise / depthwise-separable / up-sampling layer on
// the bf16 matrix pipe with split operands ("bf16x3"),
//     W = W1 + W2,  X = X1 + X2   (bf16 parts, round to nearest even:  X1 = bf16(X), X2 = bf16(X - X1))
//     W X  ~=  W2 X1 + W1 X2 + W1 X1          (fp32 accumulation; the W2 X2 term, ~2^-16 relative, is dropped)
// i.e. operands carry 16 significant bits instead of 24 — between TF32 and fp32.  It exists because the fp32 matrix
// pipe of this chip is power-limited at ~100 TF on this workload while v_mfma_f32_32x32x16_bf16 does
// 16x the multiply-adds per cycle; it is only reachable through the *_x3 entry points, which the product calls for the
// DECODER only and only when the caller opts in (the encoder and the RVQ, hence every index, stay exact fp32).
//
// Same B-operand policies (gemm_lin.h) and epilogues (gemm_epilogues.h) as the fp32 core.  Two workgroup shapes: the
// fp32 core's (32*MB rows x 128 columns, 4 waves, wave w owns column block w) and, where M is a multiple of 192 or 256,
// an 8-wave form on 64*MB rows (gemm_x3w_kernel).  Differences from the fp32 core:
//   * a staged step is X3_KS 32x32x16 MFMA steps deep; global loads run X3_DEPTH staged steps ahead; K % 32 == 0
//   * the weights arrive pre-split (hilc_x3_split_weights: [2][K][M] bf16, once per checkpoint) and are copied to LDS;
//     the activations go through the layer's own prologue (Scale / ELU / transposed-conv taps) in fp32 and are split
//     while they are staged: 2.5 VALU per element (v_cvt_pk_bf16_f32, two masks, one packed subtract, one more cvt)
//   * both LDS images are the natural [k][column] bf16 rows (8-B writes of 4 columns); an MFMA operand needs 8
//     consecutive k of one column, which is what ds_read_b64_tr_b16 delivers from that image (a 4 x 16 transpose per
//     16 lanes).  Row stride 320 B: the two 16-lane groups served in one LDS cycle never share a bank.
#pragma once
#include <utility>

#include "gemm_lin.h"

namespace hilc {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

#ifndef HILC_X3_KS
#define HILC_X3_KS 1                 // 16-deep MFMA steps per staged step (barrier): 1 = 40 KB of LDS (4 workgroups per CU), 2 = 80 KB
#endif
#ifndef HILC_X3_DEPTH
#define HILC_X3_DEPTH 1              // staged steps between a global load and its use (measured: 2-4 are slower, the unrolled ring doubles the VGPRs)
#endif
#ifndef HILC_X3_MIN_WAVES
#define HILC_X3_MIN_WAVES 1          // waves per SIMD the register allocator must leave room for
#endif
constexpr int X3_MIN_WAVES = HILC_X3_MIN_WAVES;
constexpr int X3_KS = HILC_X3_KS;
constexpr int X3_DEPTH = HILC_X3_DEPTH;
constexpr int X3_BK = X3_KS * BK;    // rows per staged step
constexpr int X3_RS = 160;           // LDS row stride, bf16 elements
constexpr int X3_PART = X3_BK * X3_RS;   // elements of one operand part of one buffer
static_assert(BK == 16 && BP == 2, "two B slices of the fp32 core per step");

template <int I, int N, class F>
__device__ __forceinline__ void x3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    x3_static_for<I + 1, N>(f);
  }
}

// 4 fp32 -> their bf16 heads and the bf16 heads of the remainders, packed 4 x 16 bit each
__device__ __forceinline__ void x3_split4(f32x4 v, uint2& hi, uint2& lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
  const bf16x2 la = __builtin_convertvector(ra, bf16x2), lb = __builtin_convertvector(rb, bf16x2);
  hi.x = __builtin_bit_cast(unsigned, ha); hi.y = __builtin_bit_cast(unsigned, hb);
  lo.x = __builtin_bit_cast(unsigned, la); lo.y = __builtin_bit_cast(unsigned, lb);
}

template <int MB, class BOp, class Epilogue>
__global__ __launch_bounds__(NT, X3_MIN_WAVES) void gemm_x3_kernel(const unsigned short* __restrict__ wsplit, int M, int K, int ldw,
                                                     long ntiles, int mtiles, BOp bop, Epilogue ep) {
  constexpr int BM = 32 * MB;
  constexpr int AG = X3_BK * BM / 8;            // 16-B chunks of one weight part per step
  constexpr int AP = (AG + NT - 1) / NT;
  constexpr int STG = 2 * 2 * 2 * X3_PART / 2;  // floats: 2 buffers x (A, B) x 2 parts
  constexpr int EPI = Epilogue::template lds_floats<MB>();
  constexpr int SM = STG > EPI ? STG : EPI;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  unsigned short* const lds = reinterpret_cast<unsigned short*>(smem);
  // element offsets: [buf][operand][part][row][col]
  auto a_at = [](int buf, int part) { return ((buf * 2 + 0) * 2 + part) * X3_PART; };
  auto b_at = [](int buf, int part) { return ((buf * 2 + 1) * 2 + part) * X3_PART; };

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
  const int ksteps = K / X3_BK;
  const long part_stride = (long)K * ldw;       // elements between the two weight parts

  // weight chunks of this thread: row = g / (BM/8), 8 columns from m0 + (g % (BM/8)) * 8
  unsigned aoff[AP];
  int alds[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int g = tid + p * NT;
    const int kr = g / (BM / 8), m8 = (g % (BM / 8)) * 8;
    int col = m0 + m8;
    col = col < ldw - 8 ? col : ldw - 8;        // rows >= M only feed accumulator rows that are never stored
    aoff[p] = g < AG ? (unsigned)(kr * ldw + col) * 2u : 0u;
    alds[p] = kr * X3_RS + m8;
  }
  const unsigned a_step = (unsigned)X3_BK * (unsigned)ldw * 2u;   // bytes per K step
  const typename BOp::State bs = bop.init(ntile, tid, BK);

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // Global loads run DEPTH staged steps ahead of their use, in a ring of register sets (slot = step % DEPTH): at bf16
  // rates a 16-deep step is 12 MFMAs = 384 cycles per wave, far less than one HBM round trip, so a single step of
  // lead (the fp32 core's scheme) leaves the kernel latency-bound (measured: K = 384 ran at 23 % MFMA utilisation).
  constexpr int DEPTH = X3_DEPTH;
  f32x4 ra[DEPTH][2][AP];
  typename BOp::Raw rb[DEPTH][X3_KS][BP];
  auto fetch = [&](int kt, auto slot) {
    constexpr int S = decltype(slot)::value;
    const char* sa = reinterpret_cast<const char*>(wsplit) + (size_t)kt * a_step;   // uniform
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int p = 0; p < AP; ++p)
        ra[S][part][p] = *reinterpret_cast<const f32x4*>(sa + (size_t)part * part_stride * 2u + aoff[p]);
#pragma unroll
    for (int s = 0; s < X3_KS; ++s)
#pragma unroll
      for (int h = 0; h < BP; ++h) rb[S][s][h] = bop.fetch(bs, X3_KS * kt + s, false, h);
  };
  auto stage = [&](int buf, auto slot) {
    constexpr int S = decltype(slot)::value;
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int p = 0; p < AP; ++p) {
        const int g = tid + p * NT;
        if (AG % NT == 0 || g < AG) *reinterpret_cast<f32x4*>(lds + a_at(buf, part) + alds[p]) = ra[S][part][p];
      }
#pragma unroll
    for (int s = 0; s < X3_KS; ++s)
#pragma unroll
      for (int h = 0; h < BP; ++h) {
        uint2 hi, lo;
        x3_split4(bop.xform(bs, rb[S][s][h], false, h), hi, lo);
        const int e = (s * BK + (tid >> 5) + 8 * h) * X3_RS + (tid & 31) * 4;
        *reinterpret_cast<uint2*>(lds + b_at(buf, 0) + e) = hi;
        *reinterpret_cast<uint2*>(lds + b_at(buf, 1) + e) = lo;
      }
  };

  // transpose-read address of this lane inside a [k][column] image: 16-lane group g covers columns 16 (g & 1) ..+15 and
  // rows 8 (g >> 1) .. +3 (a second read, 4 rows down, completes the 8 k of the operand); lane p of the group points
  // at row p >> 2, columns 4 (p & 3) ..+3 and receives column p of the 4 x 16 block.
  const int g16 = lane >> 4, p16 = lane & 15;
  const int toff = (8 * (g16 >> 1) + (p16 >> 2)) * X3_RS + 16 * (g16 & 1) + 4 * (p16 & 3);
  auto operand = [&](int base, int ks, int col0) -> bf16x8 {
    const unsigned short* p = lds + base + ks * (BK * X3_RS) + toff + col0;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p + 4 * X3_RS));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < X3_KS; ++ks) {
      const bf16x8 b1 = operand(b_at(buf, 0), ks, wave * 32);
      const bf16x8 b2 = operand(b_at(buf, 1), ks, wave * 32);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const bf16x8 a1 = operand(a_at(buf, 0), ks, i * 32);
        const bf16x8 a2 = operand(a_at(buf, 1), ks, i * 32);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[i], 0, 0, 0);   // small terms first
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[i], 0, 0, 0);
      }
    }
  };

  x3_static_for<0, DEPTH>([&](auto d) {
    if (decltype(d)::value < ksteps) fetch(decltype(d)::value, d);
  });
  stage(0, std::integral_constant<int, 0>{});
  __syncthreads();
  for (int kt0 = 0; kt0 < ksteps; kt0 += DEPTH) {
    x3_static_for<0, DEPTH>([&](auto d) {
      constexpr int D = decltype(d)::value;
      const int kt = kt0 + D;
      if (kt >= ksteps) return;
      if (kt + DEPTH < ksteps) fetch(kt + DEPTH, d);   // slot D held step kt, which was staged one iteration ago
      compute(kt & 1);
      if (kt + 1 < ksteps) {
        stage((kt + 1) & 1, std::integral_constant<int, (D + 1) % DEPTH>{});
        __syncthreads();
      }
    });
  }
  if (EPI > 0) __syncthreads();     // the epilogue re-uses the staging buffers
  ep.template run<MB>(acc, smem, m0, ntile, wave, lane, tid);
}

// ---- 8-wave form: a workgroup = 64*MB rows x 128 columns; waves w and w + 4 share column block w & 3 and own the
// lower / upper MB row blocks.  Against the 4-wave form every staged activation (prologue + split: the dominant VALU
// work at bf16 rates) feeds twice as many MFMAs and every activation row tile is read from L2 half as often, at the
// same number of waves per SIMD (two workgroups of 8 waves per CU instead of four of 4).  The two wave groups run the
// (4-wave) epilogue side by side on their own half tile and their own LDS region; barriers inside it are hit by all
// 512 threads the same number of times.
constexpr int X3W_RSA = 288;         // weight-image row stride (bf16): 256 columns + 32, same bank argument as X3_RS
template <int MB, class BOp, class Epilogue>
__global__ __launch_bounds__(2 * NT) void gemm_x3w_kernel(const unsigned short* __restrict__ wsplit, int M, int K, int ldw,
                                                          long ntiles, int mtiles, BOp bop, Epilogue ep) {
  constexpr int BM = 64 * MB;
  constexpr int AG = BK * BM / 8;               // 16-B chunks of one weight part per step (<= 512)
  constexpr int APART = BK * X3W_RSA;           // elements of one weight part of one buffer
  constexpr int BPART = BK * X3_RS;
  constexpr int BUF = 2 * APART + 2 * BPART;    // elements of one buffer: [A1 | A2 | B1 | B2]
  constexpr int STG = 2 * BUF / 2;              // floats
  constexpr int EPI1 = Epilogue::template lds_floats<MB>();
  constexpr int SM = STG > 2 * EPI1 ? STG : 2 * EPI1;
  static_assert(AG <= 2 * NT, "one weight chunk per thread and part");
  __shared__ __attribute__((aligned(16))) float smem[SM];
  unsigned short* const lds = reinterpret_cast<unsigned short*>(smem);
  auto a_at = [](int buf, int part) { return buf * BUF + part * APART; };
  auto b_at = [](int buf, int part) { return buf * BUF + 2 * APART + part * BPART; };

  long id = blockIdx.x;
  long grp = id / (8L * mtiles);
  int within = (int)(id - grp * 8L * mtiles);
  long ntile = grp * 8 + (within & 7);
  int mtile = within >> 3;
  if (ntile >= ntiles) return;
  const int m0 = mtile * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp8 = wave >> 2;                   // 0: lower row blocks, 1: upper
  const int t256 = tid & (NT - 1);
  const int ksteps = K / BK;
  const long part_stride = (long)K * ldw;

  const int kr = tid / (BM / 8), m8 = (tid % (BM / 8)) * 8;
  int acol = m0 + m8;
  acol = acol < ldw - 8 ? acol : ldw - 8;
  const bool a_live = tid < AG;
  const unsigned aoff = a_live ? (unsigned)(kr * ldw + acol) * 2u : 0u;
  const int alds = kr * X3W_RSA + m8;
  const unsigned a_step = (unsigned)BK * (unsigned)ldw * 2u;
  const typename BOp::State bs = bop.init(ntile, t256, BK);   // this thread stages row (t256 >> 5) + 8 * grp8 of every step

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  f32x4 ra[2];
  typename BOp::Raw rb;
  auto fetch = [&](int kt) {
    const char* sa = reinterpret_cast<const char*>(wsplit) + (size_t)kt * a_step;
#pragma unroll
    for (int part = 0; part < 2; ++part)
      ra[part] = *reinterpret_cast<const f32x4*>(sa + (size_t)part * part_stride * 2u + aoff);
    rb = grp8 == 0 ? bop.fetch(bs, kt, false, 0) : bop.fetch(bs, kt, false, 1);
  };
  auto stage = [&](int buf) {
    if (a_live) {
#pragma unroll
      for (int part = 0; part < 2; ++part) *reinterpret_cast<f32x4*>(lds + a_at(buf, part) + alds) = ra[part];
    }
    uint2 hi, lo;
    x3_split4(grp8 == 0 ? bop.xform(bs, rb, false, 0) : bop.xform(bs, rb, false, 1), hi, lo);
    const int e = ((t256 >> 5) + 8 * grp8) * X3_RS + (t256 & 31) * 4;
    *reinterpret_cast<uint2*>(lds + b_at(buf, 0) + e) = hi;
    *reinterpret_cast<uint2*>(lds + b_at(buf, 1) + e) = lo;
  };
  const int g16 = lane >> 4, p16 = lane & 15;
  const int trow = 8 * (g16 >> 1) + (p16 >> 2), tcol = 16 * (g16 & 1) + 4 * (p16 & 3);
  auto operand = [&](int base, int rs, int col0) -> bf16x8 {
    const unsigned short* p = lds + base + trow * rs + tcol + col0;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p + 4 * rs));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  fetch(0);
  stage(0);
  __syncthreads();
  for (int kt = 0; kt < ksteps; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ksteps;
    if (more) fetch(kt + 1);
    const bf16x8 b1 = operand(b_at(buf, 0), X3_RS, (wave & 3) * 32);
    const bf16x8 b2 = operand(b_at(buf, 1), X3_RS, (wave & 3) * 32);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const bf16x8 a1 = operand(a_at(buf, 0), X3W_RSA, (grp8 * MB + i) * 32);
      const bf16x8 a2 = operand(a_at(buf, 1), X3W_RSA, (grp8 * MB + i) * 32);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[i], 0, 0, 0);
    }
    if (more) {
      stage(buf ^ 1);
      __syncthreads();
    }
  }
  __syncthreads();
  ep.template run<MB>(acc, smem + grp8 * EPI1, m0 + grp8 * 32 * MB, ntile, wave & 3, lane, t256);
}

#ifndef HILC_X3_WIDE
#define HILC_X3_WIDE 1
#endif

template <class BOp, class Epilogue>
int launch_x3(const unsigned short* wsplit, int M, int K, int ldw, long ntiles, const BOp& bop, const Epilogue& ep,
              hipStream_t s) {
  if (K % 32 != 0 || ldw % 8 != 0 || (reinterpret_cast<uintptr_t>(wsplit) & 15)) return HILC_ERR_UNSUPPORTED;
  const int m32 = (M + 31) / 32;
  const long groups = (ntiles + 7) / 8;
  if (ntiles <= 0) return HILC_ERR_SHAPE;
  if (HILC_X3_WIDE && (M % 256 == 0 || M % 192 == 0) && groups * 8 * (M / 192 + 1) >= 2L * device_cus()) {
    const int MBW = M % 256 == 0 ? 4 : 3;
    const int mt = M / (64 * MBW);
    const long nb = groups * 8 * mt;
    if (nb > 0x7fffffffL) return HILC_ERR_SHAPE;
    HILC_CLEAR_ERROR();
    if (MBW == 4)
      hipLaunchKernelGGL((gemm_x3w_kernel<4, BOp, Epilogue>), dim3((unsigned)nb), dim3(2 * NT), 0, s, wsplit, M, K, ldw, ntiles, mt, bop, ep);
    else
      hipLaunchKernelGGL((gemm_x3w_kernel<3, BOp, Epilogue>), dim3((unsigned)nb), dim3(2 * NT), 0, s, wsplit, M, K, ldw, ntiles, mt, bop, ep);
    HILC_CHECK_LAUNCH();
    return HILC_OK;
  }
  const int MB = pick_mb(m32, ntiles);
  int mtiles = (m32 + MB - 1) / MB;
  long blocks = groups * 8 * mtiles;
  if (blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  dim3 grid((unsigned)blocks), block(NT);
  HILC_CLEAR_ERROR();
  switch (MB) {
    case 1: hipLaunchKernelGGL((gemm_x3_kernel<1, BOp, Epilogue>), grid, block, 0, s, wsplit, M, K, ldw, ntiles, mtiles, bop, ep); break;
    case 2: hipLaunchKernelGGL((gemm_x3_kernel<2, BOp, Epilogue>), grid, block, 0, s, wsplit, M, K, ldw, ntiles, mtiles, bop, ep); break;
    case 3: hipLaunchKernelGGL((gemm_x3_kernel<3, BOp, Epilogue>), grid, block, 0, s, wsplit, M, K, ldw, ntiles, mtiles, bop, ep); break;
    default: hipLaunchKernelGGL((gemm_x3_kernel<4, BOp, Epilogue>), grid, block, 0, s, wsplit, M, K, ldw, ntiles, mtiles, bop, ep); break;
  }
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

template <class Cols, class Epilogue>
int launch_gemm_x3(const unsigned short* wsplit, const float* x, int M, int K, int ldw, int T, long ntiles,
                   float in_scale, bool in_elu, const Cols& cols, const Epilogue& ep, hipStream_t s) {
  if (in_elu) {
    RowsB<Cols, true> b;
    b.x = x; b.T = T; b.in_scale = in_scale; b.cols = cols;
    return launch_x3(wsplit, M, K, ldw, ntiles, b, ep, s);
  }
  RowsB<Cols, false> b;
  b.x = x; b.T = T; b.in_scale = in_scale; b.cols = cols;
  return launch_x3(wsplit, M, K, ldw, ntiles, b, ep, s);
}

}  // namespace hilc
