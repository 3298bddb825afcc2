// Micro-benchmark: what does ONE extra instruction cost next to a stream of fp32 MFMAs on gfx950?
// Every variant runs the same 16 independent-accumulator v_mfma_f32_32x32x2_f32 per iteration (4 accumulators,
// round-robin: a dependent MFMA is 4 issues = 256 cycles away) and adds fillers of one kind between them.
// Reported: cycles per MFMA per SIMD (ideal 64) at 1 / 2 / 3 waves per SIMD, and the marginal cost per filler.
// All instruction streams are inline asm so the compiler cannot move, merge or drop anything.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shadow.hip -o gpurun_out/mfma_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC, A, B) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

enum { BASE, FMA1, FMA4, FMA8, EXP1, EXP4, LDSR1, LDSR4, LDSR128_1, LDSW1, LDSW128_1, VMEM_Q, VMEM_1, SALU4, DEP_LDS,
       M16, PKFMA1, PKFMA4, PKMUL4, FMA_CLUSTER, VALU_ONLY_FMA, VALU_ONLY_PKFMA, VALU_ONLY_EXP, VALU_ONLY_MIX, NVAR };
static const char* kName[NVAR] = {"base", "+1 v_fma/mfma", "+4 v_fma/mfma", "+8 v_fma/mfma", "+1 v_exp/mfma",
                                  "+4 v_exp/mfma", "+1 ds_read_b32/mfma", "+4 ds_read_b32/mfma",
                                  "+1 ds_read_b128/mfma", "+1 ds_write_b32/mfma", "+1 ds_write_b128/mfma",
                                  "+1 global_load_b128 per 4 mfma", "+1 global_load_b128/mfma", "+4 s_add/mfma",
                                  "operands via ds_read (next group)", "16x16x4 (2 per slot)",
                                  "+1 v_pk_fma/mfma", "+4 v_pk_fma/mfma", "+4 v_pk_mul/mfma",
                                  "+64 v_fma clustered after 16 mfma (=4/mfma)", "NO mfma: 16 v_fma per slot",
                                  "NO mfma: 16 v_pk_fma per slot", "NO mfma: 16 v_exp per slot",
                                  "NO mfma: 12 v_fma + 4 v_exp per slot"};
static const int kFill[NVAR] = {0, 1, 4, 8, 1, 4, 1, 4, 1, 1, 1, 0 /*0.25*/, 1, 4, 1, 0, 1, 4, 4, 4, 16, 16, 16, 16};

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) lds[i] = 1e-3f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = lane * 1e-3f, b = 2e-3f * lane;
  float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  float lr[4] = {0.f, 0.f, 0.f, 0.f};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk[5] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}, {1.0001f, 0.9999f}};
  f32x4 lq = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};
  const unsigned laddr = (unsigned)(uintptr_t)(&lds[0]) + lane * 4;     // LDS byte address (low 32 bits of the flat ptr)
  const unsigned laddr16 = (unsigned)(uintptr_t)(&lds[0]) + lane * 16;
  const float* gp = g + (size_t)blockIdx.x * 4096 + tid * 4;
  int sa = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (V == M16) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(lq) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(gq) : "v"(a), "v"(b));
      } else if (V == DEP_LDS) {
        // operand of THIS mfma was read one group (4 mfma) earlier; issue the read for the next group now
        asm volatile("s_waitcnt lgkmcnt(3)");
        MFMA(acc[m & 3], lr[m & 3], b);
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(lr[m & 3]) : "v"(laddr), "n"(0));
      } else if (V == VALU_ONLY_FMA || V == VALU_ONLY_PKFMA || V == VALU_ONLY_EXP || V == VALU_ONLY_MIX) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          if (V == VALU_ONLY_FMA || (V == VALU_ONLY_MIX && (q & 3) != 3)) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q & 7]) : "v"(a));
          if (V == VALU_ONLY_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pk[q & 3]) : "v"(pk[4]));
          if (V == VALU_ONLY_EXP || (V == VALU_ONLY_MIX && (q & 3) == 3)) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
        }
      } else {
        MFMA(acc[m & 3], a, b);
      }
      if (V == PKFMA1 || V == PKFMA4) {
#pragma unroll
        for (int q = 0; q < kFill[V]; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pk[q]) : "v"(pk[4]));
      }
      if (V == PKMUL4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[q]) : "v"(pk[4]));
      }
      if (V == FMA_CLUSTER && m == 15) {
#pragma unroll
        for (int q = 0; q < 64; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q & 7]) : "v"(a));
      }
      if (V == FMA1 || V == FMA4 || V == FMA8) {
#pragma unroll
        for (int q = 0; q < kFill[V]; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q]) : "v"(a));
      }
      if (V == EXP1 || V == EXP4) {
#pragma unroll
        for (int q = 0; q < kFill[V]; ++q) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q]));
      }
      if (V == LDSR1 || V == LDSR4) {
#pragma unroll
        for (int q = 0; q < kFill[V]; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(lr[q]) : "v"(laddr));
      }
      if (V == LDSR128_1) asm volatile("ds_read_b128 %0, %1" : "=v"(lq) : "v"(laddr16));
      if (V == LDSW1) asm volatile("ds_write_b32 %0, %1" ::"v"(laddr), "v"(a));
      if (V == LDSW128_1) asm volatile("ds_write_b128 %0, %1" ::"v"(laddr16), "v"(lq));
      if (V == VMEM_1 || (V == VMEM_Q && (m & 3) == 0))
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gq) : "v"(gp));
      if (V == SALU4) asm volatile("s_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1\ns_add_u32 %0, %0, 1" : "+s"(sa));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  float s = sa;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int q = 0; q < 8; ++q) s += f[q];
  for (int q = 0; q < 4; ++q) s += lr[q] + lq[q] + gq[q] + pk[q][0] + pk[q][1];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
double run(int bpc, const float* g, float* out, double base) {
  const int iters = 2000, blocks = 256 * bpc;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<blocks, 256>>>(g, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<V><<<blocks, 256>>>(g, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // waves per SIMD = bpc; SIMD time per mfma = ms / (iters * 16 * bpc) ; in cycles at 2.4 GHz
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * bpc);
  const double fills = V == VMEM_Q ? 0.25 : (V == M16 ? 1 : kFill[V]);
  printf("%-36s waves/SIMD=%d  %8.3f ms  %6.1f cyc/mfma-slot", kName[V], bpc, ms, cyc);
  if (base > 0 && fills > 0) printf("   marginal %+6.1f cyc per filler", (cyc - base) / fills);
  printf("\n");
  return cyc;
}

template <int V>
void sweep(const float* g, float* out, const double* base) {
  for (int bpc = 1; bpc <= 3; ++bpc) run<V>(bpc, g, out, base[bpc]);
}

int main() {
  float *g, *out;
  hipMalloc(&g, (size_t)768 * 4096 * 4 + 65536);
  hipMemset(g, 0, (size_t)768 * 4096 * 4 + 65536);
  hipMalloc(&out, 768 * 256 * 4);
  double base[4] = {0, 0, 0, 0};
  for (int w = 0; w < 20; ++w) run<BASE>(3, g, out, 0);   // clock ramp
  for (int bpc = 1; bpc <= 3; ++bpc) base[bpc] = run<BASE>(bpc, g, out, 0);
  sweep<FMA1>(g, out, base); sweep<FMA4>(g, out, base); sweep<FMA8>(g, out, base);
  sweep<EXP1>(g, out, base); sweep<EXP4>(g, out, base);
  sweep<LDSR1>(g, out, base); sweep<LDSR4>(g, out, base); sweep<LDSR128_1>(g, out, base);
  sweep<LDSW1>(g, out, base); sweep<LDSW128_1>(g, out, base);
  sweep<VMEM_Q>(g, out, base); sweep<VMEM_1>(g, out, base);
  sweep<DEP_LDS>(g, out, base); sweep<M16>(g, out, base);
  run<BASE>(1, g, out, 0); run<BASE>(2, g, out, 0);   // warm clocks: the first lines above run on a cold chip
  sweep<PKFMA1>(g, out, base); sweep<PKFMA4>(g, out, base); sweep<PKMUL4>(g, out, base); sweep<FMA_CLUSTER>(g, out, base);
  sweep<VALU_ONLY_FMA>(g, out, base); sweep<VALU_ONLY_PKFMA>(g, out, base); sweep<VALU_ONLY_EXP>(g, out, base);
  sweep<VALU_ONLY_MIX>(g, out, base);
  return 0;
}
