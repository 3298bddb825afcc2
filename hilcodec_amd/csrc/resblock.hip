// Fully fused SEANet residual block for the narrow, long layers (C <= 192), gfx950.
//
//   y = x + out_scale * dw2( pw2( ELU( dw1( pw1( ELU(pre_scale * x) ) ) + b1 ) ) ) + b2 )      (seanet.py:129-148)
//
// For C in {64, 96} the un-fused block is HBM-bound (1x1 conv intensity = C/4 flop/B): this kernel
// reads x once and writes y once; everything between lives in one LDS tile.
//
// Workgroup = 256 threads = 4 waves, one clip, 120 output samples: the tile spans 128 columns,
// column c <-> time t0 + c with t0 = 120*tile - 8 (8 = left halo of two causal k=5 convs).
//   P0  x tile -> a1 = ELU(pre*x)                     -> LDS  X[k][c]     (zero outside [0,T))
//   P1  GEMM1  H1 = W1 * a1   (fp32 MFMA 32x32x2; wave w owns columns [32w,32w+32), all row blocks)
//   P2  accumulators -> LDS  X[m][c]
//   P3  a2 = ELU(dw1(H1)+b1) in place (a row is handled by one wave, so read-before-write holds
//       without a barrier); columns with t < 0 are forced to 0 (the second conv's zero padding)
//   P4  GEMM2  H2 = W2 * a2
//   P5  accumulators -> LDS
//   P6  y = (dw2(H2)+b2)*out_scale + x  -> HBM, 512-B contiguous per wave
// A operands (weights, k-major [K][C]) are read straight from global memory (L1/L2 resident, a few
// tens of KB) into VGPRs one 16-deep K slice ahead — no LDS staging, no barrier in the K loop.
// Summation order: k ascending (an fmaf chain), taps j = 0..4 — the same as the un-fused kernels.
#include <stdlib.h>

#include "gemm_core.h"

using namespace hilc;

namespace {

constexpr int XS = 128;   // LDS row stride (floats): every access is row-contiguous across lanes, no padding needed
constexpr int DWS = 12;   // per-row depthwise table in LDS: w1[5], b1, w2[5], b2
constexpr int TO = 120;   // output samples per tile

struct ResArgs {
  const float* x;
  const float* w1t;   // [C][C] k-major
  const float* dw1_w; // [C][5]
  const float* dw1_b; // [C]
  const float* w2t;
  const float* dw2_w;
  const float* dw2_b;
  float* y;
  int T, tiles;
  long total_tiles;
  float pre_scale, out_scale;
  unsigned long long* dbg;   // optional [blocks][8] s_memtime stamps (tools/res_phase_times.py)
};

unsigned long long* g_dbg = nullptr;

template <int C>
__device__ __forceinline__ void gemm_phase(const float* __restrict__ wt, const float* X, f32x16 (&acc)[C / 32],
                                           int wave, int lane) {
  constexpr int CB = C / 32;
  const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < CB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* wl = wt + (long)kh * C + l31;        // this lane's A column: wt[(2j+kh)*C + 32i + l31]
  const float* xl = X + kh * XS + wave * 32 + l31;  // this lane's B column: X[(2j+kh)][32w + l31]
  float a[2][8][CB], b[2][8];
  auto load = [&](int slot, int kt) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      b[slot][j] = xl[(kt * 16 + 2 * j) * XS];
#pragma unroll
      for (int i = 0; i < CB; ++i) a[slot][j][i] = wl[(long)(kt * 16 + 2 * j) * C + 32 * i];
    }
  };
  load(0, 0);
#pragma unroll
  for (int kt = 0; kt < C / 16; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < C / 16) load(cur ^ 1, kt + 1);
    __builtin_amdgcn_sched_barrier(0);   // keep the next slice's loads ahead of this slice's MFMAs
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < CB; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][j][i], b[cur][j], acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int C>
__device__ __forceinline__ void acc_to_x(const f32x16 (&acc)[C / 32], float* X, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < C / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) X[(i * 32 + acc_row(r, lane)) * XS + wave * 32 + (lane & 31)] = acc[i][r];
}

template <int C>
__global__ __launch_bounds__(256) void resblock_kernel(ResArgs a) {
  constexpr int CB = C / 32;
  __shared__ __attribute__((aligned(16))) float X[C * XS];
  __shared__ float DW[C * DWS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> scalar loads below
#define STAMP(i) do { if (a.dbg && tid == 0) a.dbg[stamp_tile * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  long stamp_tile = blockIdx.x;
  const int T = a.T;
  // depthwise taps / biases -> LDS once per workgroup (read back as half-wave broadcasts in P3 / P6)
  for (int e = threadIdx.x; e < C * DWS; e += 256) {
    const int m = e / DWS, j = e - m * DWS;
    float v;
    if (j < 5) v = a.dw1_w[m * 5 + j];
    else if (j == 5) v = a.dw1_b[m];
    else if (j < 11) v = a.dw2_w[m * 5 + (j - 6)];
    else v = a.dw2_b[m];
    DW[e] = v;
  }
  // persistent: this workgroup walks tiles blockIdx.x, +gridDim.x, ... ; while tile i is in its second
  // GEMM the x rows of tile i+gridDim.x are touched so that its P0 finds them in this XCD's L2.
  float touch = 0.f;
  for (long tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
  stamp_tile = tile;
  STAMP(0);
  const long b = tile / a.tiles;
  const int tix = (int)(tile - b * a.tiles);
  const int t0 = tix * TO - 8;
  // the weight loads are invariant across tiles; launder the pointers so LICM does not try to keep
  // every weight of both matrices in registers across the tile loop (it spills 8 KB/lane if it does)
  const float* w1t = a.w1t;
  const float* w2t = a.w2t;
  asm volatile("" : "+s"(w1t), "+s"(w2t));
  // element-wise phases: one half-wave = one row (row = 2*wave + (lane>>5) + 8*i), lane = 4 adjacent
  // columns: 16-B global accesses, 512 B contiguous per half-wave; a row is read and written by one
  // wave instruction, so the in-place update of P3 needs no barrier.
  constexpr int RW = C / 8;
  const int rsub = wave * 2 + (lane >> 5);
  const int c4 = (lane & 31) * 4;
  const int t = t0 + c4;                   // multiple of 4; T % 4 == 0: the group is entirely inside or outside
  const bool t_in = t >= 0 && t < T;
  const float* xb = a.x + b * (long)C * T;
  float* yb = a.y + b * (long)C * T;

  // ---- P0: every row's 16-B load in flight at once, then the prologue
  {
    f32x4 v[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
      v[i] = *reinterpret_cast<const f32x4*>(xb + (long)(rsub + 8 * i) * T + (t_in ? t : 0));
#pragma unroll
    for (int i = 0; i < RW; ++i)
      *reinterpret_cast<f32x4*>(&X[(rsub + 8 * i) * XS + c4]) = prologue4v(zero_unless(t_in, v[i]), a.pre_scale, 1);
  }
  __syncthreads();
  STAMP(1);

  f32x16 acc[CB];
  // ---- P1, P2
  gemm_phase<C>(w1t, X, acc, wave, lane);
  __syncthreads();
  STAMP(2);
  acc_to_x<C>(acc, X, wave, lane);
  __syncthreads();
  STAMP(3);

  // ---- P3: a2 = ELU(dw1(H1) + b1), zero for t < 0, in place
#pragma unroll 2
  for (int i = 0; i < RW; ++i) {
    const int m = rsub + 8 * i;
    float* row = &X[m * XS];
    const f32x4 cur = *reinterpret_cast<const f32x4*>(row + c4);
    const f32x4 prev = *reinterpret_cast<const f32x4*>(row + (c4 >= 4 ? c4 - 4 : 0));   // c4 == 0: discarded columns
    const float v[8] = {prev.x, prev.y, prev.z, prev.w, cur.x, cur.y, cur.z, cur.w};
    float w[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) w[j] = DW[m * DWS + j];
    const float bias = DW[m * DWS + 5];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) s = fmaf(w[j], v[e + j], s);
      s = elu_fast(__fadd_rn(s, bias));
      o[e] = (t >= 0) ? s : 0.f;
    }
    *reinterpret_cast<f32x4*>(row + c4) = o;
  }
  __syncthreads();
  STAMP(4);

  // ---- P4
  gemm_phase<C>(w2t, X, acc, wave, lane);
  // shortcut samples for P6, PF rows at a time: the first chunk is issued here and lands under the
  // next two barriers, chunk n+1 is issued before chunk n is consumed
  constexpr int PF = RW < 4 ? RW : 4;
  const bool out_ok = c4 >= 8 && t < T;
  f32x4 xs[2][PF];
  auto load_xs = [&](int slot, int i0) {
#pragma unroll
    for (int i = 0; i < PF; ++i)
      xs[slot][i] = *reinterpret_cast<const f32x4*>(xb + (long)(rsub + 8 * (i0 + i)) * T + (out_ok ? t : 0));
  };
  load_xs(0, 0);
  // L2 touch of the next tile's x rows: one dword per 128-B line, issued after the second GEMM
  // (nothing stays live across it), consumed by a never-true test at the end of P6
  constexpr int NTOUCH = (C * 4 + 255) / 256;
  float tv[NTOUCH];
  {
    const long nt = tile + gridDim.x;
    const bool have = nt < a.total_tiles;
    const long nb = have ? nt / a.tiles : b;
    const int nt0 = have ? (int)(nt - nb * a.tiles) * TO - 8 : t0;
    const float* nx = a.x + nb * (long)C * T;
#pragma unroll
    for (int i = 0; i < NTOUCH; ++i) {
      int e = tid + 256 * i;
      e = e < C * 4 ? e : C * 4 - 1;
      int tt = nt0 + (e & 3) * 32;
      tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
      tv[i] = nx[(long)(e >> 2) * T + tt];
    }
  }
  // ---- P5
  __syncthreads();
  STAMP(5);
  acc_to_x<C>(acc, X, wave, lane);
  __syncthreads();
  STAMP(6);

  // ---- P6: y = (dw2(H2) + b2) * out_scale + x  for columns >= 8, t < T
#pragma unroll
  for (int i0 = 0; i0 < RW; i0 += PF) {
    const int slot = (i0 / PF) & 1;
    if (i0 + PF < RW) load_xs(slot ^ 1, i0 + PF);
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int m = rsub + 8 * (i0 + i);
      const float* row = &X[m * XS];
      const f32x4 cur = *reinterpret_cast<const f32x4*>(row + c4);
      const f32x4 prev = *reinterpret_cast<const f32x4*>(row + (c4 >= 4 ? c4 - 4 : 0));
      const float v[8] = {prev.x, prev.y, prev.z, prev.w, cur.x, cur.y, cur.z, cur.w};
      float w[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) w[j] = DW[m * DWS + 6 + j];
      const float bias = DW[m * DWS + 11];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) s = fmaf(w[j], v[e + j], s);
        s = __fmul_rn(__fadd_rn(s, bias), a.out_scale);
        o[e] = __fadd_rn(s, xs[slot][i][e]);
      }
      if (out_ok) *reinterpret_cast<f32x4*>(yb + (long)m * T + t) = o;
    }
  }
#pragma unroll
  for (int i = 0; i < NTOUCH; ++i) touch += tv[i];
  STAMP(7);
  __syncthreads();   // the next tile's P0 overwrites X
  }
  if (touch == 1.2345678e-30f) a.y[0] = touch;   // keeps the touch loads alive; never true in practice
#undef STAMP
}

template <int C>
int launch_res(ResArgs a, int B, hipStream_t s) {
  a.total_tiles = (long)B * a.tiles;
  // persistent grid = exactly what can be resident (a surplus workgroup would only start after a
  // resident one has walked its whole tile list)
  static int per_cu = 0, n_cu = 0;
  if (per_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HILC_ERR_LAUNCH;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, resblock_kernel<C>, 256, 0) != hipSuccess || occ < 1)
      return HILC_ERR_LAUNCH;
    n_cu = prop.multiProcessorCount;
    per_cu = occ;
  }
  const long resident = (long)n_cu * per_cu;
  long blocks = a.total_tiles < resident ? a.total_tiles : resident;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(resblock_kernel<C>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

}  // namespace

extern "C" int hilc_resblock(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                             const float* w2t, const float* dw2_w, const float* dw2_b, float* y, int B, int C,
                             int T, float pre_scale, float out_scale, void* stream) {
  if (!x || !w1t || !dw1_w || !dw1_b || !w2t || !dw2_w || !dw2_b || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (x == y) return HILC_ERR_UNSUPPORTED;   // neighbouring tiles read each other's halo: not in place
  if (T % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
    return HILC_ERR_UNSUPPORTED;             // callers fall back to two hilc_dws_conv launches
  ResArgs a;
  a.x = x; a.w1t = w1t; a.dw1_w = dw1_w; a.dw1_b = dw1_b; a.w2t = w2t; a.dw2_w = dw2_w; a.dw2_b = dw2_b;
  a.y = y; a.T = T; a.tiles = (T + TO - 1) / TO; a.pre_scale = pre_scale; a.out_scale = out_scale;
  a.dbg = g_dbg;
  switch (C) {
    case 64: return launch_res<64>(a, B, (hipStream_t)stream);
    case 96: return launch_res<96>(a, B, (hipStream_t)stream);
    case 128: return launch_res<128>(a, B, (hipStream_t)stream);
    case 192: return launch_res<192>(a, B, (hipStream_t)stream);
    default: return HILC_ERR_UNSUPPORTED;
  }
}

extern "C" void hilc_debug_set_stamp_buffer(unsigned long long* p) { g_dbg = p; }

extern "C" int hilc_resblock_supported(int C, int T) {
  return (C == 64 || C == 96 || C == 128 || C == 192) && T % 4 == 0;
}
