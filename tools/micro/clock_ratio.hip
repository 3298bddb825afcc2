// What does s_memtime count?  One wave per CU spins on a dependent FMA chain (light load) or on fp32 MFMAs (heavy load)
// and reads s_memtime and s_memrealtime (constant 100 MHz) before and after.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/clock_ratio.hip -o /tmp/clock_ratio
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(unsigned long long* out, int iters, int heavy) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x * 1e-3f;
  f32x16 acc = {};
  for (int i = 0; i < iters; ++i) {
    if (heavy) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 64; ++j) a = fmaf(a, 1.0001f, 1e-7f);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
  if (a + acc[0] == 1.2345f) out[0] = 0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 4096 * 16);
  for (int heavy = 0; heavy < 2; ++heavy)
    for (int wg = 256; wg <= 4096; wg *= 16) {
      k<<<wg, 256>>>(d, 200000, heavy); hipDeviceSynchronize();
      unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("%s load, %4d workgroups: s_memtime %llu ticks in %llu ticks of the 100 MHz clock -> s_memtime runs at %.1f MHz\n",
             heavy ? "MFMA" : "FMA ", wg, h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
    }
  return 0;
}
