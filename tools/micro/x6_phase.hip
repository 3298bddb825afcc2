// Micro-benchmark (NOT on the product path): one GEMM PHASE of the fused stage kernel in split-bf16 arithmetic with THREE parts per
// operand and SIX products (x = h + m + l exactly; hh, hm, mh, hl, mm, lh kept: 24 significant bits per operand), as it would run
// inside resblock_kernel: the activation tile stays the fp32 [k][column] tile in LDS that the element-wise phases read and write
// (4 B per element: three bf16 planes would be 6 and not fit), every GEMM wave reads its columns' 8 consecutive k per 16-deep step
// (8 ds_read_b32 per column block), splits them in registers (11 VALU per pair) and feeds v_mfma_f32_32x32x16_bf16; the pre-split
// weights stream from L2 in MFMA lane order (48 B per lane, row block and step) DEPTH steps ahead; no barrier in the K loop; one
// persistent workgroup per CU.  The question: which wave mapping keeps the bf16 pipe busy?
//   G = 8: eight GEMM waves, wave = R row blocks x 1 column block  (today's fp32 mapping: weight stream 64 B/clk/CU at full rate)
//   G = 4: FOUR GEMM waves (one per SIMD; the workgroup's other four waves wait at the phase's closing barrier), wave = R row
//          blocks x 2 column blocks: half the weight bytes per flop, each weight register feeds two MFMAs
// Reports time per phase, MFMA utilisation (MFMA cycles / elapsed at 2.4 GHz) and the fp32-equivalent TFLOP/s.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/x6_phase.hip -o tools/micro/x6_phase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x4* gvec_t;
typedef __attribute__((address_space(3))) float* lptr_t;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 8 fp32 -> three bf16 parts each (round to nearest even; h + m + l == x exactly), as three 4-register MFMA operands
__device__ __forceinline__ void split8x3(const float (&v)[8], f32x4& ph, f32x4& pm, f32x4& pl) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f32x2 x = {v[2 * p], v[2 * p + 1]};
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const unsigned hu = __builtin_bit_cast(unsigned, h);
    const f32x2 r = {x[0] - __builtin_bit_cast(float, hu << 16), x[1] - __builtin_bit_cast(float, hu & 0xffff0000u)};
    const bf16x2 m = __builtin_convertvector(r, bf16x2);
    const unsigned mu = __builtin_bit_cast(unsigned, m);
    const f32x2 s = {r[0] - __builtin_bit_cast(float, mu << 16), r[1] - __builtin_bit_cast(float, mu & 0xffff0000u)};
    const bf16x2 l = __builtin_convertvector(s, bf16x2);
    ph[p] = __builtin_bit_cast(float, h);
    pm[p] = __builtin_bit_cast(float, m);
    pl[p] = __builtin_bit_cast(float, l);
  }
}

template <int C, int NCOL, int G, int R, int DEPTH, int NMAT, int SPLITMODE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void phase_kernel(const float* __restrict__ wpk, float* out, int iters) {
  constexpr int XS = NCOL + (C >= 768 ? 8 : 24);
  constexpr int Q = G == 4 ? 2 : 1;
  constexpr int NSTEP = C / 16;
  constexpr int NCC = NCOL / (32 * Q);           // column classes
  constexpr int NRC = G / NCC;                   // row classes
  static_assert(NCC * NRC == G && C == NRC * R * 32, "the GEMM waves tile the C x NCOL block");
  __shared__ __attribute__((aligned(16))) float X[C * XS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < C * XS; e += 512) X[e] = 0.01f + 1e-4f * (float)((e * 37) & 0xff);
  __syncthreads();
  f32x16 acc[R * Q];
  for (int i = 0; i < R * Q; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  constexpr long MATF = (long)C * C * 3 / 2;     // floats per packed matrix (3 bf16 planes)
  constexpr int WPS = R * 3;                     // 16-B words per lane and step
  if (wave < G) {
    const int cc = wave % NCC, rc = wave / NCC;
    const int kh = lane >> 5, l31 = lane & 31;
    for (int it = 0; it < iters; ++it) {
      // this wave's slice of matrix (it % NMAT): [rc][step][i][plane][lane] 16-B words
      const float* wt = wpk + (long)(it % NMAT) * MATF + (long)rc * (NSTEP * WPS * 256);
      asm volatile("" : "+s"(wt));
      f32x4 a[DEPTH][R][3];
#pragma unroll
      for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
        for (int q = 0; q < WPS; ++q) a[d][q / 3][q % 3] = *(gvec_t)(wt + (d * WPS + q) * 256 + lane * 4);
      const float* wn = wt + (DEPTH - 1) * WPS * 256;
      lptr_t xn = (lptr_t)(X + 8 * kh * XS + cc * Q * 32 + l31);
      float br[2][Q][8];
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) br[0][q][j] = xn[j * XS + 32 * q];
      f32x4 bp[2][Q][3];
      if constexpr (SPLITMODE == 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) split8x3(br[0][q], bp[0][q][0], bp[0][q][1], bp[0][q][2]);
      }
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int cur = s % DEPTH, nxt = (s + DEPTH - 1) % DEPTH;
        const bool more = s + DEPTH - 1 < NSTEP;
        if (s + 1 < NSTEP && !(ABL & 4)) {
          xn += 16 * XS;
          asm volatile("" : "+v"(xn));
#pragma unroll
          for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) br[(s + 1) & 1][q][j] = xn[j * XS + 32 * q];
        }
        if constexpr (SPLITMODE == 0) {            // split this step's operands right before its MFMAs
#pragma unroll
          for (int q = 0; q < Q; ++q) split8x3(br[s & 1][q], bp[s & 1][q][0], bp[s & 1][q][1], bp[s & 1][q][2]);
        }
        if (more && !(ABL & 1)) {
#pragma unroll
          for (int q = 0; q < WPS; ++q) a[nxt][q / 3][q % 3] = *(gvec_t)(wn + q * 256 + lane * 4);
        }
        // six products, small terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)   [weight part, activation part]
        constexpr int WP[6] = {2, 0, 1, 1, 0, 0}, XP[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < R; ++i)
#pragma unroll
            for (int q = 0; q < Q; ++q)
              acc[i * Q + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[cur][i][WP[t]]),
                                                                       __builtin_bit_cast(bf16x8, bp[s & 1][q][XP[t]]), acc[i * Q + q], 0, 0, 0);
        if constexpr (SPLITMODE == 1) {            // split the NEXT step's operands in the shadow of this step's MFMAs (scheduler's choice inside the set)
          if (s + 1 < NSTEP && !(ABL & 2)) {
#pragma unroll
            for (int q = 0; q < Q; ++q) split8x3(br[(s + 1) & 1][q], bp[(s + 1) & 1][q][0], bp[(s + 1) & 1][q][1], bp[(s + 1) & 1][q][2]);
          }
        }
        if (more) {
          wn += WPS * 256;
          asm volatile("" : "+s"(wn));
        }
#pragma unroll
        for (int i = 0; i < R * Q; ++i) asm volatile("" : "+v"(acc[i]) :: "memory");
      }
    }
  }
  __syncthreads();
  float sum = 0.f;
  for (int i = 0; i < R * Q; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  out[blockIdx.x * 512 + tid] = sum;
}

template <int C, int NCOL, int G, int R, int DEPTH, int NMAT, int SPLITMODE, int ABL = 0>
void run(const char* name, const float* w, float* out, int grid) {
  auto k = phase_kernel<C, NCOL, G, R, DEPTH, NMAT, SPLITMODE, ABL>;
  const int iters = 400;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, w, out, 20);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, w, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3 / iters;
  constexpr int Q = G == 4 ? 2 : 1;
  // MFMA cycles per SIMD and phase: G/4 waves per SIMD, each C/16 steps x R*Q*6 MFMAs x 32 cycles
  const double cyc = (G / 4.0) * (C / 16) * R * Q * 6 * 32.0;
  const double flop = 2.0 * C * C * NCOL * grid;
  printf("%-52s C=%d NCOL=%d grid=%3d %8.2f us/phase  MFMA util %.3f  fp32-equivalent %6.1f TF  vs fp32 MFMA at 98 %%: x%.2f\n", name, C, NCOL, grid, us,
         cyc / 2400.0 / us, flop / us * 1e-6, (cyc / 2400.0 * 16.0 / 6.0 / 0.98) / us);
}

int main() {
  float *w, *out;
  const size_t wbytes = 8ull * 768 * 768 * 6 + (1 << 20);
  CHECK(hipMalloc(&w, wbytes));
  std::vector<unsigned short> h(wbytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + ((i * 131) & 0x1ff));
  CHECK(hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&out, 256 * 512 * 4));
  // C = 192, 128-column tile (8 matrices = a 3-block stage with its up-sampling layer)
  run<192, 128, 8, 3, 2, 8, 0>("8 waves R3xQ1 depth2 split-before", w, out, 256);
  run<192, 128, 8, 3, 2, 8, 1>("8 waves R3xQ1 depth2 split-ahead", w, out, 256);
  run<192, 128, 8, 3, 3, 8, 1>("8 waves R3xQ1 depth3 split-ahead", w, out, 256);
  run<192, 128, 4, 3, 2, 8, 0>("4 waves R3xQ2 depth2 split-before", w, out, 256);
  run<192, 128, 4, 3, 2, 8, 1>("4 waves R3xQ2 depth2 split-ahead", w, out, 256);
  run<192, 128, 4, 3, 3, 8, 1>("4 waves R3xQ2 depth3 split-ahead", w, out, 256);
  run<192, 128, 4, 3, 2, 8, 1>("4 waves R3xQ2 depth2 split-ahead, 8 workgroups", w, out, 8);
  run<192, 128, 8, 3, 2, 8, 1>("8 waves R3xQ1 depth2 split-ahead, 8 workgroups", w, out, 8);
  // ablations (wrong results, timing only): 1 = no weight loads in the K loop, 2 = no split arithmetic, 4 = no LDS reads
  run<192, 128, 4, 3, 2, 8, 1, 1>("4 waves R3xQ2: no weight loads", w, out, 256);
  run<192, 128, 4, 3, 2, 8, 1, 2>("4 waves R3xQ2: no split", w, out, 256);
  run<192, 128, 4, 3, 2, 8, 1, 6>("4 waves R3xQ2: no split, no LDS reads", w, out, 256);
  run<192, 128, 4, 3, 2, 8, 1, 7>("4 waves R3xQ2: MFMAs only", w, out, 256);
  run<192, 128, 4, 3, 2, 8, 1, 7>("4 waves R3xQ2: MFMAs only, 8 workgroups", w, out, 8);
  run<192, 128, 8, 3, 2, 8, 1, 1>("8 waves R3xQ1: no weight loads", w, out, 256);
  run<192, 128, 8, 3, 2, 8, 1, 2>("8 waves R3xQ1: no split", w, out, 256);
  run<192, 128, 8, 3, 2, 8, 1, 7>("8 waves R3xQ1: MFMAs only", w, out, 256);
  // C = 384, 64-column tile
  run<384, 64, 8, 3, 2, 8, 1>("8 waves R3xQ1 depth2 split-ahead", w, out, 256);
  run<384, 64, 8, 3, 4, 8, 1>("8 waves R3xQ1 depth4 split-ahead", w, out, 256);
  run<384, 64, 4, 3, 2, 8, 1>("4 waves R3xQ2 depth2 split-ahead", w, out, 256);
  run<384, 64, 4, 3, 3, 8, 1>("4 waves R3xQ2 depth3 split-ahead", w, out, 256);
  // C = 768, 32-column tile: one column block, no column reuse possible
  run<768, 32, 8, 3, 3, 3, 1>("8 waves R3xQ1 depth3 split-ahead", w, out, 256);
  run<768, 32, 8, 3, 4, 3, 1>("8 waves R3xQ1 depth4 split-ahead", w, out, 256);
  // C = 96, 128-column tile: two 4-wave workgroups per CU today; here one workgroup
  run<96, 256, 4, 3, 2, 6, 1>("4 waves R3xQ2 depth2 split-ahead (256 columns)", w, out, 256);
  return 0;
}
