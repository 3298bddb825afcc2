// Prototype (NOT on the product path): fp32 GEMM emulated on the bf16 matrix pipe by operand splitting, next to the
// exact fp32-MFMA form, on one decoder-sized pointwise layer  Y[M x N] = W[M x K] * X[K x N]  (M = K = 768, N = 32768).
//   fp32      v_mfma_f32_32x32x2_f32            (what the product runs; 64 cycles per 32x32x2)
//   bf16x3    W = W1 + W2, X = X1 + X2 (bf16 parts):  W1X1 + W1X2 + W2X1          on v_mfma_f32_32x32x16_bf16
//   bf16x6    three parts each:  + W2X2 + W1X3 + W3X1                                (fp32 accumulation throughout)
// Each wave owns a 128-row x 32-column tile and streams BOTH operands from global memory in MFMA lane order (weights
// pre-split on the host — legitimate, they are constants; activations arrive as fp32 and are split in registers, which
// is what a real kernel would do at staging time).  No LDS sharing: every wave pulls its own weights, so the figure is
// a LOWER bound on what an LDS-staged kernel would reach.  Reports fp32-equivalent TFLOP/s (2*M*N*K / time) and the
// error against an fp64 host reference.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3_gemm.hip -o tools/micro/bf16x3_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline unsigned short f2bf(float f) {   // round to nearest even
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// ---- exact fp32: packed A [mt][kstep16][j(8 k-pairs)][i(4 blocks)] per lane as in the product's packed layout
__global__ __launch_bounds__(256) void gemm_f32(const float* __restrict__ wp, const float* __restrict__ x, float* y, int M,
                                                int K, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntile = blockIdx.x, mt = blockIdx.y;
  const int col = ntile * 128 + wave * 32 + (lane & 31), kh = lane >> 5;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* wt = wp + (long)mt * (K / 16) * 8 * 4 * 64;
  for (int ks = 0; ks < K / 16; ++ks) {
    float a[8][4];
    for (int q = 0; q < 8; ++q) {
      const f32x4 v = *(const f32x4*)(wt + ((long)(ks * 8 + q) * 64 + lane) * 4);
      for (int e = 0; e < 4; ++e) a[q][e] = v[e];     // word q = k-pair q, blocks 0..3
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float b = x[(long)(ks * 16 + 2 * j + kh) * N + col];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][i], b, acc[i], 0, 0, 0);
    }
  }
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      y[(long)row * N + col] = acc[i][r];
    }
}

// ---- bf16 split.  A parts: [part][mt][kstep16][i(4 blocks)][lane][8 bf16]: lane (row = lane & 31, k = 8*(lane >> 5) + 0..7)
// B: fp32 X, lane (col = lane & 31) needs k = 8*(lane >> 5) + 0..7 of the step: 8 strided dwords (coalesced across lanes)
template <int TERMS>
__global__ __launch_bounds__(256) void gemm_bf16(const u16x8* __restrict__ wparts, const float* __restrict__ x, float* y, int M,
                                                 int K, int N) {
  constexpr int NP = TERMS == 3 ? 2 : 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntile = blockIdx.x, mt = blockIdx.y;
  const int col = ntile * 128 + wave * 32 + (lane & 31), kh = lane >> 5;
  const long part_stride = (long)(M / 128) * (K / 16) * 4 * 64;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int ks = 0; ks < K / 16; ++ks) {
    // activations: 8 fp32 values of this lane, split into NP bf16 parts in registers
    float xv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = x[(long)(ks * 16 + 8 * kh + e) * N + col];
    bf16x8 bp[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      u16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        h[e] = f2bf(xv[e]);
        xv[e] -= bf2f(h[e]);
      }
      bp[p] = __builtin_bit_cast(bf16x8, h);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 ap[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p)
        ap[p] = __builtin_bit_cast(bf16x8, wparts[p * part_stride + (((long)mt * (K / 16) + ks) * 4 + i) * 64 + lane]);
      // smallest terms first
      if (TERMS == 6) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], acc[i], 0, 0, 0);
      }
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], acc[i], 0, 0, 0);
    }
  }
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      y[(long)row * N + col] = acc[i][r];
    }
}

static float frand(unsigned& s) {   // uniform (-1, 1)
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f;
}

int main() {
  const int M = 768, K = 768, N = 32768;
  std::vector<float> W((size_t)M * K), X((size_t)K * N);
  unsigned s = 12345u;
  for (auto& v : W) v = (frand(s) + frand(s) + frand(s)) * (1.0f / sqrtf((float)K));      // ~N(0, 1/K)
  for (auto& v : X) { float t = (frand(s) + frand(s) + frand(s)) * 0.8f; v = t > 0 ? t : expm1f(t); }   // ELU-like activations
  // packed operands
  std::vector<float> wp((size_t)M * K);
  for (int mt = 0; mt < M / 128; ++mt)
    for (int ks = 0; ks < K / 16; ++ks)
      for (int q = 0; q < 8; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e)
            wp[((((size_t)mt * (K / 16) + ks) * 8 + q) * 64 + lane) * 4 + e] =
                W[(size_t)(mt * 128 + 32 * e + (lane & 31)) * K + ks * 16 + 2 * q + (lane >> 5)];
  const size_t part = (size_t)(M / 128) * (K / 16) * 4 * 64 * 8;
  std::vector<unsigned short> wb(3 * part);
  for (int mt = 0; mt < M / 128; ++mt)
    for (int ks = 0; ks < K / 16; ++ks)
      for (int i = 0; i < 4; ++i)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            float r = W[(size_t)(mt * 128 + 32 * i + (lane & 31)) * K + ks * 16 + 8 * (lane >> 5) + e];
            for (int p = 0; p < 3; ++p) {
              const unsigned short h = f2bf(r);
              wb[p * part + (((((size_t)mt * (K / 16) + ks) * 4 + i) * 64 + lane) * 8) + e] = h;
              r -= bf2f(h);
            }
          }
  float *dwp, *dx, *dy;
  unsigned short* dwb;
  hipMalloc(&dwp, wp.size() * 4); hipMalloc(&dx, X.size() * 4); hipMalloc(&dy, (size_t)M * N * 4); hipMalloc(&dwb, wb.size() * 2);
  hipMemcpy(dwp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwb, wb.data(), wb.size() * 2, hipMemcpyHostToDevice);
  // fp64 reference on a sample of columns
  const int NS = 64;
  std::vector<double> ref((size_t)M * NS);
  for (int m = 0; m < M; ++m)
    for (int c = 0; c < NS; ++c) {
      double a = 0;
      const int col = c * (N / NS) + 7;
      for (int k = 0; k < K; ++k) a += (double)W[(size_t)m * K + k] * (double)X[(size_t)k * N + col];
      ref[(size_t)m * NS + c] = a;
    }
  std::vector<float> Y((size_t)M * N);
  dim3 grid(N / 128, M / 128), block(256);
  auto run = [&](const char* name, int which) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (which == 0) gemm_f32<<<grid, block>>>(dwp, dx, dy, M, K, N);
      else if (which == 3) gemm_bf16<3><<<grid, block>>>((const u16x8*)dwb, dx, dy, M, K, N);
      else gemm_bf16<6><<<grid, block>>>((const u16x8*)dwb, dx, dy, M, K, N);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, rms = 0, ymax = 0;
    for (int m = 0; m < M; ++m)
      for (int c = 0; c < NS; ++c) {
        const int col = c * (N / NS) + 7;
        const double d = fabs((double)Y[(size_t)m * N + col] - ref[(size_t)m * NS + c]);
        emax = d > emax ? d : emax;
        rms += d * d;
        ymax = fmax(ymax, fabs(ref[(size_t)m * NS + c]));
      }
    printf("{\"variant\": \"%s\", \"ms\": %.4f, \"fp32_equivalent_tflops\": %.1f, \"max_abs_err_vs_fp64\": %.3e, \"rms_err\": %.3e, \"max_abs_y\": %.2f}\n",
           name, best, 2.0 * M * N * K / best / 1e9, emax, sqrt(rms / (M * NS)), ymax);
  };
  run("fp32 v_mfma_f32_32x32x2_f32", 0);
  run("bf16x3 v_mfma_f32_32x32x16_bf16", 3);
  run("bf16x6 v_mfma_f32_32x32x16_bf16", 6);
  return 0;
}
