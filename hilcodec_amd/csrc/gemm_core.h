// fp32-MFMA GEMM core of the HILCodec hot path (gfx950).
//
//   D[m][n] = sum_k  A[k][m] * B[k][n]          (A = folded weights, k-major; B = activations)
//
// One workgroup = 256 threads = 4 waves computes a (32*MB) x 128 tile; wave w owns columns
// [32w, 32w+32) and all MB 32-row blocks: MB f32x16 accumulators fed by v_mfma_f32_32x32x2_f32 (exact
// fp32 — bitwise an fmaf chain in k order k = 0..K-1, one rounding per product).
// K is streamed in BK=16 slices through double-buffered LDS with register prefetch: the global loads
// of slice t+1 are issued before the MFMAs of slice t and only *consumed* (prologue + ds_write)
// after them, so HBM/L2 latency hides under the matrix pipe.
//
// Loader  — produces the B operand.  `fetch` returns RAW values (so nothing waits on the load before
//           the MFMA block); `transform` (Scale/ELU prologue) runs at LDS-staging time.
// Epilogue — consumes the accumulators; may route them through LDS (the staging buffers are dead by
//            then) to apply a depthwise convolution along time before anything touches HBM.
#pragma once
#include <stdlib.h>

#include <atomic>

#include "common.h"

namespace hilc {

constexpr int BN = 128;
#ifndef HILC_BK
#define HILC_BK 16
#endif
constexpr int BK = HILC_BK;   // K-slice depth (multiple of 8)
constexpr int BP = BK / 8;     // B staging passes: 256 threads cover 8 rows x 128 columns per pass
constexpr int NT = 256;

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int MB, class Loader, class Epilogue>
__global__ __launch_bounds__(NT) void gemm_kernel(const float* __restrict__ wt, int M, int K, int ldw,
                                                  long ntiles, int mtiles, Loader ld, Epilogue ep) {
  constexpr int BM = 32 * MB;
  constexpr int AG = BK * BM / 4;         // float4 groups in an A slice
  constexpr int AP = (AG + NT - 1) / NT;  // per-thread passes over the A slice
  constexpr int STG = 2 * BK * (BM + BN);
  constexpr int EPI = Epilogue::template lds_floats<MB>();
  constexpr int SM = STG > EPI ? STG : EPI;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
  float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);

  // XCD-aware tile order: the `mtiles` row-tiles that share one B column-tile get block ids that
  // are congruent mod 8 (observed: block b runs on XCD b % 8), so the shared activations stay in
  // one XCD's L2.  Placement only changes speed, never results.
  long id = blockIdx.x;
  long grp = id / (8L * mtiles);
  int within = (int)(id - grp * 8L * mtiles);
  long ntile = grp * 8 + (within & 7);
  int mtile = within >> 3;
  if (ntile >= ntiles) return;
  const int m0 = mtile * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const typename Loader::State ls = ld.init(ntile, tid);

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // Straight-line staging code: every guard is a select on the address / value, never a branch, so
  // the K loop body stays one basic block and the scheduler can run the LDS reads ahead of the MFMAs.
  f32x4 ra[AP];
  typename Loader::Raw rb[BP];
  bool a_ok[AP];
  const float* a_ptr[AP];
  int a_k[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    int g = tid + p * NT;
    a_k[p] = g / (BM / 4);
    int m4 = (g % (BM / 4)) * 4;
    a_ok[p] = g < AG && (m0 + m4) < ldw;
    a_ptr[p] = wt + (long)a_k[p] * ldw + (a_ok[p] ? m0 + m4 : 0);
  }
  // fetch: issue the global loads only (RAW values; nothing here depends on the loaded data, so no
  // s_waitcnt lands in front of the MFMA block).  stage: zero-select, prologue, ds_write.
  bool ra_ok[AP];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      ra_ok[p] = a_ok[p] && (k0 + a_k[p]) < K;   // rows k >= K of A are zero: B may hold anything finite there
      ra[p] = *reinterpret_cast<const f32x4*>(ra_ok[p] ? a_ptr[p] + (long)k0 * ldw : wt);
    }
#pragma unroll
    for (int h = 0; h < BP; ++h) rb[h] = ld.fetch(ls, min(k0 + (tid >> 5) + 8 * h, K - 1));
  };
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      int g = tid + p * NT;
      if (AG % NT == 0 || g < AG) {
        int k = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
        *reinterpret_cast<f32x4*>(&As[buf][k][m4]) = zero_unless(ra_ok[p], ra[p]);
      }
    }
#pragma unroll
    for (int h = 0; h < BP; ++h)
      *reinterpret_cast<f32x4*>(&Bs[buf][(tid >> 5) + 8 * h][(tid & 31) * 4]) =
          ld.transform(ls, rb[h], min(k0 + (tid >> 5) + 8 * h, K - 1));
  };

  const int ktiles = (K + BK - 1) / BK;
  const int kh = lane >> 5, l31 = lane & 31;
  fetch(0);
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ktiles) fetch((kt + 1) * BK);
    // software-pipelined operand reads: k-pair j+1's LDS reads are in flight under k-pair j's MFMAs
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
    if (kt + 1 < ktiles) {
      stage(buf ^ 1, (kt + 1) * BK);
      __syncthreads();
    }
  }
  ep.template run<MB>(acc, smem, m0, ntile, wave, lane, tid);
}

// CUs of the current device (immutable per device, looked up once per device)
inline int device_cus() {
  constexpr int MAXDEV = 64;
  static std::atomic<int> cache[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int n = dev >= 0 && dev < MAXDEV ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    if (dev >= 0 && dev < MAXDEV) cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// Row-tile height (in 32-row MFMA blocks) of a launch with `ntiles` column tiles and m32 row blocks.  Big launches
// (offline: thousands of workgroups) want the tallest tile: every staged activation feeds MB MFMAs.  Small launches
// (a streaming hop: a few hundred workgroups on 256 CUs) are decided by load balance instead: the busiest CU
// serialises ceil(workgroups / CUs) tiles on its matrix pipe, so 384 tall tiles (2 on half of the CUs, 1 on the rest)
// lose a quarter of the chip against 768 tiles of half the height.  Cost = tiles on the busiest CU x (MB + fixed
// per-tile overhead); per-output arithmetic (k order) does not depend on MB, results are bit-identical.
inline int pick_mb(int m32, long ntiles) {
  long cus = device_cus();
  double fixed = 0.5;
#ifdef HILC_PICK_ENV      // tuning builds only (tools/): the launch heuristic's two constants from the environment
  if (const char* e = getenv("HILC_PICK_CUS")) cus = atol(e);
  if (const char* e = getenv("HILC_PICK_FIXED")) fixed = atof(e);
#endif
  const long groups = (ntiles + 7) / 8;
  int best = 1;
  double best_cost = 1e300;
  for (int mb = 1; mb <= 4; ++mb) {
    const long mtiles = (m32 + mb - 1) / mb;
    const long wgs = groups * 8 * mtiles;
    const double cost = (double)((wgs + cus - 1) / cus) * (mb + fixed);
    if (cost < best_cost * 0.999 || (cost <= best_cost * 1.001 && mb > best)) {   // ties: the taller tile
      best_cost = cost < best_cost ? cost : best_cost;
      best = mb;
    }
  }
  return best;
}

template <class Loader, class Epilogue>
int launch_gemm(const float* wt, int M, int K, int ldw, long ntiles, bool lds_epilogue, const Loader& ld,
                const Epilogue& ep, hipStream_t s) {
  (void)lds_epilogue;
  const int m32 = (M + 31) / 32;
  const int MB = pick_mb(m32, ntiles);
  const long groups = (ntiles + 7) / 8;
  int mtiles = (m32 + MB - 1) / MB;
  long blocks = groups * 8 * mtiles;
  if (ntiles <= 0 || blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  dim3 grid((unsigned)blocks), block(NT);
  HILC_CLEAR_ERROR();
  switch (MB) {
    case 1: hipLaunchKernelGGL((gemm_kernel<1, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<2, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
    case 3: hipLaunchKernelGGL((gemm_kernel<3, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
    default: hipLaunchKernelGGL((gemm_kernel<4, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
  }
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

}  // namespace hilc
