#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 160];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 160; i += 64) lds[i] = (unsigned short)i;   // value = element index: row = i / 160, col = i % 160
  __syncthreads();
  const int g = lane >> 4, p = lane & 15, r = p >> 2, c = p & 3;
  const int n0 = 16 * (g & 1), kh = g >> 1;
  const unsigned short* a = &lds[(8 * kh + r) * 160 + n0 + 4 * c];
  s16x4 v;
  unsigned addr = (unsigned)(uintptr_t)a;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (k=%d,n=%d)", h[l * 4 + j] / 160, h[l * 4 + j] % 160);
    printf("\n");
  }
  return 0;
}
