#pragma once
// stream_gemm.h — one GEMM phase whose A operand (a packed weight matrix) streams from L2 in MFMA lane order and whose B operand is
// whatever the caller's functor reads from LDS: the DFT and the 1x1 conv of the one-launch SpecBlock (spec.hip) and of the stage-0
// phase of the fused encoder stage (resblock_kernel.h, Cfg::SPEC0).
#include "gemm_core.h"

namespace hilc {

typedef const __attribute__((address_space(1))) float* sg_gptr_t;
typedef const __attribute__((address_space(1))) f32x4* sg_gvec_t;

// One GEMM phase: acc[CB] (+)= A(packed, streamed from L2) * B, B element of k-pair P for this lane = bop(P).
template <int CB, int KP, int DEPTH, int NSETS, class BOp>
__device__ __forceinline__ void stream_gemm(const float* __restrict__ wt, f32x16 (&acc)[CB], int lane, BOp bop) {
  constexpr int WPS = KP * CB / 4;
  float a[DEPTH][KP][CB];
  float b[DEPTH][KP];
  auto load_word = [&](sg_gptr_t wset, int slot, int q) {
    const f32x4 v = *(sg_gvec_t)(wset + q * 256 + lane * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[slot][(q * 4 + e) / CB][(q * 4 + e) % CB] = v[e];
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) {
    if (d < NSETS) {
#pragma unroll
      for (int q = 0; q < WPS; ++q) load_word((sg_gptr_t)(wt + d * WPS * 256), d, q);
#pragma unroll
      for (int j = 0; j < KP; ++j) b[d][j] = bop(d * KP + j);
    }
  }
  const float* wn = wt + (DEPTH - 1) * WPS * 256;    // uniform: first word of the set being fetched
  constexpr int LD_EVERY = KP * CB / WPS;            // = 4
#pragma unroll
  for (int s = 0; s < NSETS; ++s) {
    const int cur = s % DEPTH, nxt = (s + DEPTH - 1) % DEPTH;
    const int sn = s + DEPTH - 1;
    const bool more = sn < NSETS;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      if (more) b[nxt][j] = bop(sn * KP + j);
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const int n = j * CB + i;
        if (more && n % LD_EVERY == 0) load_word((sg_gptr_t)wn, nxt, n / LD_EVERY);
        // builtin MFMAs pinned per register set (see resblock.hip: the pure builtins would otherwise be sunk under the
        // phase's later loads; asm MFMAs hide their hazards from the compiler)
        if (s == 0 && j == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][j][i], b[cur][j], acc[i], 0, 0, 0);
      }
    }
    if (more) {
      wn += WPS * 256;
      asm volatile("" : "+s"(wn));
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) asm volatile("" : "+v"(acc[i]) :: "memory");
  }
}

}  // namespace hilc
