// Micro-benchmark: which part of a K-slice costs MFMA throughput on gfx950 (fp32 32x32x2, MB=4, 3 WG/CU)?
//   mode 1: MFMA + LDS operand reads               mode 3: + one __syncthreads per slice
//   mode 4: + 4 x ds_write_b128 per thread per slice (double buffer) + barrier
//   mode 5: + 4 x global float4 loads per thread per slice (streaming 16 KB/slice/WG), staged through regs
//   mode 6: mode 5 + 8 exp2-ELUs per thread per slice on the staged data
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* out, int iters, long stride) {
  constexpr int MB = 4;
  __shared__ __attribute__((aligned(16))) float As[2][16][128];
  __shared__ __attribute__((aligned(16))) float Bs[2][16][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 16 * 128; i += 256) { (&As[0][0][0])[i] = 1e-3f * i; (&Bs[0][0][0])[i] = 2e-3f * i; }
  __syncthreads();
  f32x16 acc[MB];
  for (int i = 0; i < MB; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int kh = lane >> 5, l31 = lane & 31;
  const float* gp = g + (long)blockIdx.x * stride + tid * 4;
  f32x4 r0 = {1, 2, 3, 4}, r1 = r0, r2 = r0, r3 = r0;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (MODE >= 5) {
      const float* p = gp + (long)(it & 63) * 4096;
      r0 = *(const f32x4*)(p); r1 = *(const f32x4*)(p + 1024); r2 = *(const f32x4*)(p + 2048); r3 = *(const f32x4*)(p + 3072);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float bv = Bs[buf][2 * j + kh][wave * 32 + l31], av[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) av[i] = As[buf][2 * j + kh][i * 32 + l31];
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
    }
    if (MODE >= 6) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r2[e] = r2[e] > 0 ? r2[e] : __builtin_amdgcn_exp2f(r2[e] * 1.4427f) - 1.f;
        r3[e] = r3[e] > 0 ? r3[e] : __builtin_amdgcn_exp2f(r3[e] * 1.4427f) - 1.f;
      }
    }
    if (MODE >= 4) {
      *(f32x4*)&As[buf ^ 1][tid >> 5][(tid & 31) * 4] = r0;
      *(f32x4*)&As[buf ^ 1][(tid >> 5) + 8][(tid & 31) * 4] = r1;
      *(f32x4*)&Bs[buf ^ 1][tid >> 5][(tid & 31) * 4] = r2;
      *(f32x4*)&Bs[buf ^ 1][(tid >> 5) + 8][(tid & 31) * 4] = r3;
    }
    if (MODE >= 3) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < MB; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(int bpc, int iters, const float* g, float* out) {
  int blocks = 256 * bpc;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(g, out, iters, 64L * 4096);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(g, out, iters, 64L * 4096);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 8 * 4 * 2.0 * 32 * 32 * 2;
  printf("mode=%d blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", MODE, bpc, ms, flops / ms / 1e9);
}

int main() {
  float *g, *out;
  size_t n = 768L * 64 * 4096 + 4096;
  hipMalloc(&g, n * 4);
  hipMemset(g, 0, n * 4);
  hipMalloc(&out, 768 * 256 * 4);
  for (int bpc = 2; bpc <= 3; ++bpc) {
    run<1>(bpc, 3000, g, out); run<3>(bpc, 3000, g, out); run<4>(bpc, 3000, g, out);
    run<5>(bpc, 3000, g, out); run<6>(bpc, 3000, g, out);
  }
  return 0;
}
