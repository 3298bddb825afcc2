#pragma once
// resblock_gemm.h — part 2 of 4 of the fused residual-block / stage kernel: the packed weight stream and the GEMM phases.
//
// Contract with the kernel (resblock_kernel.h):
//   * LDS: the B operand of every GEMM phase is the fp32 tile X[k][column], row stride K::XS floats, rows 0..C-1; a phase only READS
//     it.  acc_to_x WRITES the accumulators back as X[m][column]; the kernel puts a barrier between the two.
//   * registers: a phase owns its accumulators acc[CBW] (CBW x 16 VGPRs) and a WeightPipe (DEPTH x KP x CBW VGPRs) plus DEPTH x KP
//     B values; everything else a caller keeps live across a phase (the x rows xr[RW], column bookkeeping) must fit beside that in
//     the launch bound of Cfg (2 waves per SIMD for the 8-wave shapes).
//   * memory: weights arrive in MFMA lane order (hilc_resblock_pack_weights[_rc]); their loads are left in flight across
//     lds_barrier() (lgkmcnt only) — never __syncthreads() inside the tile loop.
#include "resblock_cfg.h"

namespace {

// The weight pointers go through an empty asm (LICM fence, see resblock_kernel) and come back without their
// address space: loads through them would be FLAT instructions (LDS-or-global check, both wait counters).  This
// type puts them back into the global address space -> global_load.
typedef const __attribute__((address_space(1))) float* gptr_t;
// the same for the laundered LDS row pointers: keep them 32-bit LDS pointers (ds_read / ds_write, not flat_load)
typedef __attribute__((address_space(3))) float* lptr_t;
typedef __attribute__((address_space(3))) f32x4* lvec_t;
typedef __attribute__((address_space(3))) f32x2* lvec2_t;

// Packed ("MFMA lane order") weights, produced by hilc_resblock_pack_weights from the k-major [K][C] matrix.  For
// the wave class h (row half, RH of them) and K slice kt the 8*CBW operands a lane feeds to the MFMAs sit in
// NQ = 2*CBW consecutive 16-B words per lane and a wave's 64 lanes read 1 KiB contiguous per load:
//   packed[(((h * C/16 + kt) * NQ + q) * 64 + lane) * 4 + e] = W[kt*16 + 2j + (lane >> 5)][32*(h*CBW + i) + (lane & 31)]
//   with q*4 + e = j*CBW + i   (j = k-pair of the slice, i = row block of the wave).
// DEPTH register sets: the weights of slice kt+DEPTH-1 are requested in the shadow of the MFMAs of slice kt; the
// first DEPTH-1 slices are requested by prefetch() BEFORE the element-wise phase that precedes the GEMM.
template <class K>
struct WeightPipe {
  static constexpr int CBW = K::CBW;
  static constexpr int DEPTH = K::DEPTH;
  static constexpr int KP = K::KP;              // k-pairs per register set (8 = one 16-deep slice, 4 = half of one)
  static constexpr int WPS = KP * CBW / 4;      // 16-B words per lane and set (consecutive in the packed array)
  static_assert(KP * CBW % 4 == 0 && 8 % KP == 0, "register set = whole 16-B words");
  float a[DEPTH][KP][CBW];
  // wset: UNIFORM pointer to the set's first word (scalar base + lane offset + immediate: no per-lane 64-bit adds)
  __device__ __forceinline__ void load_word(gptr_t wset, int slot, int q, int lane) {
    typedef const __attribute__((address_space(1))) f32x4* gvec_t;
    const f32x4 v = *(gvec_t)(wset + q * 256 + lane * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[slot][(q * 4 + e) / CBW][(q * 4 + e) % CBW] = v[e];
  }
  __device__ __forceinline__ void prefetch(const float* __restrict__ wt, int lane) {
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
      for (int q = 0; q < WPS; ++q) load_word((gptr_t)(wt + d * WPS * 256), d, q, lane);
  }
};

// (Round 5: the wide stages' packed matrices exceed one XCD's 4 MB L2 — C = 384: 4.7 MB per tile, C = 768: 2 x 2.36 MB per block — and every
// CU walks them cyclically, so PMC shows 3.5 GB read for a 0.47 GB input.  Requesting one stream per tile NON-TEMPORAL so that the rest stays
// resident was built and measured: MORE traffic (C = 768 blocks 1.89 -> 2.56 GB per launch) and a slower step (73.7 -> 74.0 - 74.3 ms, same
// box).  The misses are served by the 256 MB memory-side cache and hide behind DEPTH register sets; plain loads stay.)

// wt: this wave class's part of the packed matrix;  X: LDS tile;  colblk: the wave's 32-column block
template <class K, bool ZERO = true>
__device__ __forceinline__ void gemm_phase(const float* __restrict__ wt, const float* X, f32x16 (&acc)[K::CBW],
                                           WeightPipe<K>& wp, int colblk, int lane) {
  constexpr int C = K::CH, XS = K::XS;
  constexpr int CBW = K::CBW;
  constexpr int DEPTH = WeightPipe<K>::DEPTH, KP = WeightPipe<K>::KP, WPS = WeightPipe<K>::WPS;
  constexpr int NSETS = C / 2 / KP;
  const int kh = lane >> 5, l31 = lane & 31;
  // this lane's B column: X[(2p+kh)][32*colblk + l31], p = k-pair.  `xn` walks ahead of the MFMAs one register set at a
  // time and is laundered after every step: a DS instruction reaches 64 KB past its base register, the tile is up to
  // 136 KB, and left alone hipcc materialises one base register per far row and keeps them all alive (spilling them).
  lptr_t xn = (lptr_t)(X + kh * XS + colblk * 32 + l31);
  float b[DEPTH][KP];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) {
#pragma unroll
    for (int j = 0; j < KP; ++j) b[d][j] = xn[j * 2 * XS];
    xn += KP * 2 * XS;
    asm volatile("" : "+v"(xn));
  }
  // Issue order, pinned: the weight words of set s+DEPTH-1 are spread over the MFMAs of set s (one every fourth),
  // its LDS operand reads one per k-pair.
  constexpr int LD_EVERY = KP * CBW / WPS;           // = 4
  const float* wn = wt + (DEPTH - 1) * WPS * 256;    // uniform: first word of the set being fetched
#pragma unroll
  for (int s = 0; s < NSETS; ++s) {
    const int cur = s % DEPTH, nxt = (s + DEPTH - 1) % DEPTH;
    const int sn = s + DEPTH - 1;
    const bool more = sn < NSETS;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      if (more) b[nxt][j] = xn[j * 2 * XS];
#pragma unroll
      for (int i = 0; i < CBW; ++i) {
        const int n = j * CBW + i;                   // MFMA index inside the set
        if (more && n % LD_EVERY == 0) wp.load_word((gptr_t)wn, nxt, n / LD_EVERY, lane);
        if (ZERO && s == 0 && j == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp.a[cur][j][i], b[cur][j], acc[i], 0, 0, 0);
      }
    }
    if (more) {
      xn += KP * 2 * XS;
      wn += WPS * 256;
      asm volatile("" : "+v"(xn), "+s"(wn));
    }
    // pin the set: an empty asm that "updates" the accumulators and clobbers memory keeps this set's MFMAs above it and
    // the later sets' loads below it.  The builtins are pure, and left alone hipcc sinks a whole phase's MFMAs under all of
    // its operand loads (~300 spilled registers); pinned, they schedule as written, need fewer registers than the asm form
    // (C = 192: 202 instead of 219) and — unlike an asm MFMA (rounds 1-2) — carry their hazard information: split-bf16 phases
    // written with asm MFMAs fed by VALU conversions produced rare garbage tiles that no manual wait state fixed.
#pragma unroll
    for (int i = 0; i < CBW; ++i) asm volatile("" : "+v"(acc[i]) :: "memory");
  }
}

// The same phase ROLLED, for the wide channel counts of the NARROW stream shapes (C = 768: 96 register sets of 12 MFMAs —
// fully unrolled that is 25 KB of code per phase): a loop over groups of DEPTH sets, so that the register-set indices stay
// compile-time; issue order, products and k order are those of gemm_phase.
template <class K, bool ZERO = true>
__device__ __forceinline__ void gemm_phase_rolled(const float* __restrict__ wt, const float* X, f32x16 (&acc)[K::CBW],
                                                  WeightPipe<K>& wp, int colblk, int lane) {
  constexpr int C = K::CH, XS = K::XS;
  constexpr int CBW = K::CBW;
  constexpr int DEPTH = WeightPipe<K>::DEPTH, KP = WeightPipe<K>::KP, WPS = WeightPipe<K>::WPS;
  constexpr int NSETS = C / 2 / KP;
  static_assert(NSETS % DEPTH == 0 && NSETS >= 2 * DEPTH, "whole groups of register sets");
  constexpr int NG = NSETS / DEPTH;
  const int kh = lane >> 5, l31 = lane & 31;
  lptr_t xn = (lptr_t)(X + kh * XS + colblk * 32 + l31);
  float b[DEPTH][KP];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) {
#pragma unroll
    for (int j = 0; j < KP; ++j) b[d][j] = xn[j * 2 * XS];
    xn += KP * 2 * XS;
    asm volatile("" : "+v"(xn));
  }
  constexpr int LD_EVERY = KP * CBW / WPS;           // = 4
  const float* wn = wt + (DEPTH - 1) * WPS * 256;    // uniform: first word of the set being fetched
  if constexpr (ZERO) {
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }
  auto one_set = [&](auto dc, bool more) {
    constexpr int cur = decltype(dc)::value, nxt = (cur + DEPTH - 1) % DEPTH;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      if (more) b[nxt][j] = xn[j * 2 * XS];
#pragma unroll
      for (int i = 0; i < CBW; ++i) {
        const int n = j * CBW + i;
        if (more && n % LD_EVERY == 0) wp.load_word((gptr_t)wn, nxt, n / LD_EVERY, lane);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp.a[cur][j][i], b[cur][j], acc[i], 0, 0, 0);
      }
    }
    if (more) {
      xn += KP * 2 * XS;
      wn += WPS * 256;
      asm volatile("" : "+v"(xn), "+s"(wn));
    }
#pragma unroll
    for (int i = 0; i < CBW; ++i) asm volatile("" : "+v"(acc[i]) :: "memory");      // pin the set (see gemm_phase)
  };
  auto group = [&](bool last) {
    one_set(std::integral_constant<int, 0>{}, true);       // set s = g*DEPTH fetches set s + DEPTH - 1: inside this group
    if constexpr (DEPTH > 1) one_set(std::integral_constant<int, 1>{}, !last);
    if constexpr (DEPTH > 2) one_set(std::integral_constant<int, 2>{}, !last);
    if constexpr (DEPTH > 3) one_set(std::integral_constant<int, 3>{}, !last);
    static_assert(DEPTH <= 4, "group body");
  };
#pragma nounroll
  for (int g = 0; g < NG - 1; ++g) group(false);
  group(true);
}

template <class K>
__device__ __forceinline__ void acc_to_x(const f32x16 (&acc)[K::CBW], float* X, int rowblk0, int colblk, int lane) {
  constexpr int XS = K::XS;
  lptr_t xb = (lptr_t)(X + (rowblk0 * 32 + 4 * (lane >> 5)) * XS + colblk * 32 + (lane & 31));
#pragma unroll
  for (int i = 0; i < K::CBW; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xb[((r & 3) + 8 * (r >> 2)) * XS] = acc[i][r];     // acc_row(r, lane) without its lane term
    xb += 32 * XS;
    asm volatile("" : "+v"(xb));      // one base register per row block (see gemm_phase)
  }
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() drains vmcnt as well (it is a memory fence for global
// memory too), i.e. every barrier would wait for the weight words and the next tile's x rows that are deliberately
// kept in flight across it — measured as a 3-7 k cycle hole at the end of every tile.  All data exchanged between
// the waves here lives in LDS, and a wave's LDS operations complete in order: lgkmcnt(0) + s_barrier is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace
