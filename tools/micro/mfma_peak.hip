// Micro-benchmark: fp32 MFMA issue ceiling on gfx950 under different operand sources.
//   mode 0: operands in registers (pure pipe rate)
//   mode 1: operands re-read from LDS before every MFMA group (ds_read2 + waitcnt, like the GEMM core)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MB, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float As[2][16][128];
  __shared__ float Bs[2][16][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 16 * 128; i += 256) { (&As[0][0][0])[i] = 1e-3f * i; (&Bs[0][0][0])[i] = 2e-3f * i; }
  __syncthreads();
  f32x16 acc[MB];
  for (int i = 0; i < MB; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = lane * 1e-3f, b = wave * 1e-2f;
  const int kh = lane >> 5, l31 = lane & 31;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float bv = b, av[MB];
      if (MODE == 1) bv = Bs[buf][2 * j + kh][wave * 32 + l31];
#pragma unroll
      for (int i = 0; i < MB; ++i) av[i] = MODE == 1 ? As[buf][2 * j + kh][(i * 32 + l31) & 127] : a + i;
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
    }
    if (MODE == 2) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < MB; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MB, int MODE>
void run(int blocks_per_cu, int iters) {
  float* out;
  int blocks = 256 * blocks_per_cu;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MB, MODE><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MB, MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 /*waves*/ * iters * 8 * MB * 2.0 * 32 * 32 * 2;
  printf("MB=%d mode=%d blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", MB, MODE, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int bpc = 1; bpc <= 4; ++bpc) run<4, 0>(bpc, 4000);
  for (int bpc = 1; bpc <= 4; ++bpc) run<4, 1>(bpc, 4000);
  for (int bpc = 1; bpc <= 4; ++bpc) run<2, 1>(bpc, 8000);
  return 0;
}
