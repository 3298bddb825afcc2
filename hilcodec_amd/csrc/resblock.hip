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
//
// STREAM instantiation (hilc_resblock_stream; streaming.py:195-276 with causal_layers.py:147-167 caches): the
// tile walks the FLAT column space (clip-major, b*T + t) so short hops (T = 160 / 320 per stream) still fill
// 120 of 128 columns; the 4 samples before a clip's t = 0 come from the caches hist1 / hist2 (= last 4
// pointwise outputs of the previous hop) instead of the LDS neighbours, and the lanes holding t = T-4..T-1
// store the new caches.  Per-column arithmetic is identical, so hop-by-hop output == offline output bit for bit.
#include <stdlib.h>

#include <atomic>

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
  int B;
  unsigned div_magic, div_shift;   // STREAM: n / T == __umulhi(n, div_magic) >> div_shift for n < 2^31
  const float* hist1;   // STREAM: [B][C][4] caches of the two depthwise convs (NULL = zeros), and their successors
  const float* hist2;
  float* hist1_out;
  float* hist2_out;
  // optional dynamic tile scheduler: two ints, zero at launch and zero again at exit.  The two workgroups that
  // share a CU do not share it fairly (the older one wins issue arbitration: lifetimes 4.7 M vs 7.0 M cycles for
  // the same 100 tiles at C = 96), so with static tile lists a third of the kernel runs at half occupancy; with
  // tickets the faster workgroup simply takes more tiles.
  int* sched;
  unsigned long long* dbg;   // optional [blocks][8] s_memtime stamps (tools/res_phase_times.py)
};

#ifdef HILC_DEBUG_STAMPS
unsigned long long* g_dbg = nullptr;   // tools/res_phase_times.py builds its own copy of the library with this
#endif

// Weight operands of one GEMM phase.  DEPTH register sets: the weights of slice kt+DEPTH-1 are requested in the
// shadow of the MFMAs of slice kt; the first DEPTH-1 slices are requested by prefetch() BEFORE the element-wise
// phase that precedes the GEMM (P0 / P3), so that their L2 round trip (~2.5 k cycles, otherwise exposed at the
// head of every GEMM phase) overlaps that phase.  C = 192 runs one workgroup per CU and gets two slices of lead.
// The weight pointers go through an empty asm (LICM fence, see resblock_kernel) and come back without their
// address space: loads through them would be FLAT instructions (LDS-or-global check, both wait counters).  This
// type puts them back into the global address space -> global_load.
typedef const __attribute__((address_space(1))) float* gptr_t;

// Packed ("MFMA lane order") weights, produced by hilc_resblock_pack_weights from the k-major [K][C] matrix:
//   packed[((kt * 2*CB + q) * 64 + lane) * 4 + e] = W[kt*16 + 2j + (lane >> 5)][32i + (lane & 31)],  q*4 + e = j*CB + i
// i.e. the 8*CB operands a lane feeds to the MFMAs of K slice kt sit in 2*CB consecutive 16-B words per lane and a
// wave's 64 lanes read 1 KiB contiguous per load: 2*CB vector loads per slice instead of 8*CB dword loads (every
// VMEM instruction in these loops costs ~16 cycles of MFMA issue; +4-6 % at C <= 128).
template <int C>
struct WeightPipe {
  static constexpr int CB = C / 32;
  static constexpr int DEPTH = C >= 192 ? 3 : 2;
  static constexpr int NQ = 2 * CB;            // 16-B words per lane and slice
  float a[DEPTH][8][CB];
  __device__ __forceinline__ void load_word(gptr_t wp, int slot, int kt, int q, int lane) {
    typedef const __attribute__((address_space(1))) f32x4* gvec_t;
    const f32x4 v = *(gvec_t)(wp + ((long)(kt * NQ + q) * 64 + lane) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[slot][(q * 4 + e) / CB][(q * 4 + e) % CB] = v[e];
  }
  __device__ __forceinline__ void prefetch(const float* __restrict__ wt, int lane) {
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
      for (int q = 0; q < NQ; ++q) load_word((gptr_t)wt, d, d, q, lane);
  }
};

template <int C>
__device__ __forceinline__ void gemm_phase(const float* __restrict__ wt, const float* X, f32x16 (&acc)[C / 32],
                                           WeightPipe<C>& wp, int wave, int lane) {
  constexpr int CB = C / 32;
  constexpr int DEPTH = WeightPipe<C>::DEPTH;
  const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < CB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* xl = X + kh * XS + wave * 32 + l31;  // this lane's B column: X[(2j+kh)][32w + l31]
  float b[DEPTH][8];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[d][j] = xl[(d * 16 + 2 * j) * XS];
  // Issue order, pinned: the 2*CB weight words of slice kt+DEPTH-1 are spread over the 8*CB MFMAs of slice kt (one
  // every fourth MFMA), its LDS operand reads one per k-pair.  Left to itself hipcc sinks the loads next to their
  // uses; sched_group_barrier lets the scheduler pick WHICH load fills a slot and it picks the consumer's own.
#pragma unroll
  for (int kt = 0; kt < C / 16; ++kt) {
    const int cur = kt % DEPTH, nxt = (kt + DEPTH - 1) % DEPTH;
    const int kn = kt + DEPTH - 1;
    const bool more = kn < C / 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (more) b[nxt][j] = xl[(kn * 16 + 2 * j) * XS];
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const int n = j * CB + i;                    // MFMA index inside the slice
        if (more && n % 4 == 0) wp.load_word((gptr_t)wt, nxt, kn, n / 4, lane);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp.a[cur][j][i], b[cur][j], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const float* wt, float* packed, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;       // index into `packed`
  if (idx >= C * C) return;
  const int CB = C / 32, NQ = 2 * CB;
  const int e = idx & 3, lane = (idx >> 2) & 63, w = idx >> 8;
  const int q = w % NQ, kt = w / NQ;
  const int v = q * 4 + e, j = v / CB, i = v % CB;
  const int k = kt * 16 + 2 * j + (lane >> 5), m = 32 * i + (lane & 31);
  packed[idx] = wt[(long)k * C + m];
}

template <int C>
__device__ __forceinline__ void acc_to_x(const f32x16 (&acc)[C / 32], float* X, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < C / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) X[(i * 32 + acc_row(r, lane)) * XS + wave * 32 + (lane & 31)] = acc[i][r];
}

template <int C, bool STREAM>
__global__ __launch_bounds__(256, (C == 128 ? 2 : 1)) void resblock_kernel(ResArgs a) {
  constexpr int CB = C / 32;
  __shared__ __attribute__((aligned(16))) float X[C * XS];
  __shared__ float DW[C * DWS];
  // STREAM, T >= 128 (at most one clip start per tile): that clip's two caches, staged before P0 so that P3 / P6
  // do not pay one exposed global-load latency per row for the single lane that needs them
  __shared__ __attribute__((aligned(16))) float HS[STREAM ? 2 * C * 4 : 4];
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
  __shared__ long s_next;
  long tile = blockIdx.x;
  while (tile < a.total_tiles) {
  stamp_tile = tile;
  STAMP(0);
  if (a.sched != nullptr && tid == 0) s_next = (long)gridDim.x + atomicAdd(a.sched, 1);   // read after P0's barrier
  // element-wise phases: one half-wave = one row (row = 2*wave + (lane>>5) + 8*i), lane = 4 adjacent
  // columns: 16-B global accesses, 512 B contiguous per half-wave; a row is read and written by one
  // wave instruction, so the in-place update of P3 needs no barrier.
  constexpr int RW = C / 8;
  const int rsub = wave * 2 + (lane >> 5);
  const int c4 = (lane & 31) * 4;
  // this lane's 4 columns: clip b, time t (a multiple of 4; T % 4 == 0: the group is entirely inside or outside)
  long b;
  int t;
  bool t_in;
  // STREAM: clips differ between lanes, so the lane carries one 32-bit BYTE offset against the scalar tensor
  // base (saddr + voffset form) — B*C*T*4 < 2^32, launcher-checked — plus the cache offset and two flags.
  // They are RE-DERIVED from a laundered copy of c4 at the start of P3 and P6 instead of staying live across
  // the GEMMs: the C = 128 kernel sits 12 VGPRs under the 2-waves/SIMD budget and hipcc gives up on that
  // occupancy for the whole kernel (+150 VGPRs) as soon as one region exceeds it.
  [[maybe_unused]] unsigned boff = 0;
  [[maybe_unused]] unsigned hoff = 0;          // element offset of this clip's [C][4] cache block
  [[maybe_unused]] bool clip_head = false;     // t == 0: the previous 4 samples live in the cache
  [[maybe_unused]] bool clip_tail = false;     // t == T-4 (an output column): these 4 samples are the new cache
  [[maybe_unused]] const unsigned row_b = (unsigned)T * 4u;
  auto lane_columns = [&]() {
    if constexpr (STREAM) {
      int c = c4;
      asm volatile("" : "+v"(c));
      const int flat = (int)tile * TO - 8 + c;
      t_in = flat >= 0 && flat < a.B * T;
      const unsigned ub = t_in ? __umulhi((unsigned)flat, a.div_magic) >> a.div_shift : 0u;   // flat / T
      b = ub;
      t = t_in ? flat - (int)ub * T : 0;
      boff = (ub * (unsigned)(C * T) + (unsigned)t) * 4u;
      hoff = ub * (unsigned)(C * 4);
      clip_head = t_in && t == 0;
      clip_tail = t_in && c >= 8 && t == T - 4;
    }
  };
  if constexpr (STREAM) {
    lane_columns();
  } else {
    b = tile / a.tiles;
    const int t0 = (int)(tile - b * a.tiles) * TO - 8;
    t = t0 + c4;
    t_in = t >= 0 && t < T;
  }
  const float* xb = a.x + (STREAM ? 0 : b) * (long)C * T;
  float* yb = a.y + (STREAM ? 0 : b) * (long)C * T;
  auto xrow = [&](int m, bool ok) -> const f32x4* {
    if constexpr (STREAM)
      return reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x) + (boff + (unsigned)m * row_b));
    else
      return reinterpret_cast<const f32x4*>(xb + (long)m * T + (ok ? t : 0));
  };
  auto yrow = [&](int m) -> f32x4* {
    if constexpr (STREAM)
      return reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.y) + (boff + (unsigned)m * row_b));
    else
      return reinterpret_cast<f32x4*>(yb + (long)m * T + t);
  };

  // the weight loads are invariant across tiles; launder the pointers so LICM does not try to keep
  // every weight of both matrices in registers across the tile loop (it spills 8 KB/lane if it does)
  const float* w1t = a.w1t;
  const float* w2t = a.w2t;
  asm volatile("" : "+s"(w1t), "+s"(w2t));
  [[maybe_unused]] const bool one_head = STREAM && T >= XS;
  if constexpr (STREAM) {
    if (one_head && __builtin_amdgcn_ballot_w64(clip_head) != 0) {   // wave-uniform
      if (clip_head) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const float* hp = which == 0 ? a.hist1 : a.hist2;
          f32x4 h[RW];
#pragma unroll
          for (int i = 0; i < RW; ++i) h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (hp != nullptr) {
#pragma unroll
            for (int i = 0; i < RW; ++i) h[i] = *reinterpret_cast<const f32x4*>(hp + hoff + (rsub + 8 * i) * 4);
          }
#pragma unroll
          for (int i = 0; i < RW; ++i) *reinterpret_cast<f32x4*>(&HS[(which * C + rsub + 8 * i) * 4]) = h[i];
        }
      }
    }
  }
  WeightPipe<C> wp;
  wp.prefetch(w1t, lane);                  // GEMM1's first weight slices travel while P0 runs
  // ---- P0: every row's 16-B load in flight at once, then the prologue
  {
    f32x4 v[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
      v[i] = *xrow(rsub + 8 * i, t_in);
#pragma unroll
    for (int i = 0; i < RW; ++i)
      *reinterpret_cast<f32x4*>(&X[(rsub + 8 * i) * XS + c4]) = prologue4v(zero_unless(t_in, v[i]), a.pre_scale, 1);
  }
  __syncthreads();
  STAMP(1);
  const long next = a.sched != nullptr ? s_next : tile + gridDim.x;

  f32x16 acc[CB];
  // ---- P1, P2
  gemm_phase<C>(w1t, X, acc, wp, wave, lane);
  __syncthreads();
  STAMP(2);
  acc_to_x<C>(acc, X, wave, lane);
  __syncthreads();
  STAMP(3);

  // ---- P3: a2 = ELU(dw1(H1) + b1), zero for t < 0, in place
  wp.prefetch(w2t, lane);                  // GEMM2's first weight slices travel while P3 runs
  lane_columns();
#pragma unroll 2
  for (int i = 0; i < RW; ++i) {
    const int m = rsub + 8 * i;
    float* row = &X[m * XS];
    const f32x4 cur = *reinterpret_cast<const f32x4*>(row + c4);
    f32x4 prev = *reinterpret_cast<const f32x4*>(row + (c4 >= 4 ? c4 - 4 : 0));   // c4 == 0: discarded columns
    if constexpr (STREAM) {
      if (clip_head) {
        if (one_head) {
          prev = *reinterpret_cast<const f32x4*>(&HS[m * 4]);   // written by this same lane before P0
        } else {
          prev = f32x4{0.f, 0.f, 0.f, 0.f};
          if (a.hist1 != nullptr) prev = *reinterpret_cast<const f32x4*>(a.hist1 + hoff + m * 4);
        }
      }
      if (clip_tail && a.hist1_out != nullptr) *reinterpret_cast<f32x4*>(a.hist1_out + hoff + m * 4) = cur;
    }
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
      o[e] = (STREAM || t >= 0) ? s : 0.f;   // STREAM: a clip's first samples never read their LDS neighbours
    }
    *reinterpret_cast<f32x4*>(row + c4) = o;
  }
  __syncthreads();
  STAMP(4);

  // ---- P4
  gemm_phase<C>(w2t, X, acc, wp, wave, lane);
  lane_columns();
  // shortcut samples for P6, PF rows at a time: the first chunk is issued here and lands under the
  // next two barriers, chunk n+1 is issued before chunk n is consumed
  constexpr int PF = RW < 4 ? RW : 4;
  const bool out_ok = STREAM ? (c4 >= 8 && t_in) : (c4 >= 8 && t < T);
  f32x4 xs[2][PF];
  auto load_xs = [&](int slot, int i0) {
#pragma unroll
    for (int i = 0; i < PF; ++i)
      xs[slot][i] = *xrow(rsub + 8 * (i0 + i), out_ok);
  };
  load_xs(0, 0);
  // L2 touch of the next tile's x rows: one dword per 128-B line, issued after the second GEMM
  // (nothing stays live across it), consumed by a never-true test at the end of P6
  constexpr int NTOUCH = (C * 4 + 255) / 256;
  float tv[NTOUCH];
  {
    const long nt = next < a.total_tiles ? next : tile;
    // (these loads also keep hipcc from hoisting P6's shortcut loads above the GEMM: without them, or with
    // per-lane clip indices here, the kernel needs ~130 more VGPRs)
    const bool have = next < a.total_tiles;
    long nb;
    int nt0;
    if constexpr (STREAM) {   // the next tile's first clip only: a tile that straddles clips is touched in part
      const int nf0 = (int)nt * TO - 8;
      const unsigned q = __umulhi((unsigned)(nf0 < 0 ? 0 : nf0), a.div_magic) >> a.div_shift;
      nb = q;
      nt0 = nf0 - (int)q * T;
    } else {
      nb = have ? nt / a.tiles : b;
      nt0 = (int)(nt - nb * a.tiles) * TO - 8;
    }
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
      f32x4 prev = *reinterpret_cast<const f32x4*>(row + (c4 >= 4 ? c4 - 4 : 0));
      if constexpr (STREAM) {
        if (clip_head) {
          if (one_head) {
            prev = *reinterpret_cast<const f32x4*>(&HS[(C + m) * 4]);
          } else {
            prev = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.hist2 != nullptr) prev = *reinterpret_cast<const f32x4*>(a.hist2 + hoff + m * 4);
          }
        }
        if (clip_tail && a.hist2_out != nullptr) *reinterpret_cast<f32x4*>(a.hist2_out + hoff + m * 4) = cur;
      }
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
      if (out_ok) *yrow(m) = o;
    }
  }
#pragma unroll
  for (int i = 0; i < NTOUCH; ++i) touch += tv[i];
  STAMP(7);
  __syncthreads();   // the next tile's P0 overwrites X
  tile = next;
  }
  if (a.sched != nullptr && tid == 0) {          // last workgroup out re-arms the scheduler for the next launch
    if (atomicAdd(a.sched + 1, 1) == (int)gridDim.x - 1) {
      a.sched[0] = 0;
      a.sched[1] = 0;
    }
  }
  if (touch == 1.2345678e-30f) a.y[0] = touch;   // keeps the touch loads alive; never true in practice
#undef STAMP
}

template <int C, bool STREAM>
int launch_res(ResArgs a, int B, hipStream_t s) {
  a.B = B;
  {  // division by the invariant T (Granlund-Montgomery, 31-bit dividends): l = ceil(log2 T), m = ceil(2^(31+l) / T)
    int l = 0;
    while ((1L << l) < a.T) ++l;
    if (l < 1) l = 1;
    const unsigned long long p = 1ULL << (31 + l);
    a.div_magic = (unsigned)((p + (unsigned long long)a.T - 1) / (unsigned long long)a.T);
    a.div_shift = (unsigned)(l - 1);
  }
  a.total_tiles = STREAM ? ((long)B * a.T + TO - 1) / TO : (long)B * a.tiles;
  // persistent grid = exactly what can be resident (a surplus workgroup would only start after a
  // resident one has walked its whole tile list)
  // immutable per-device facts, looked up once per device (a process may drive several GPUs)
  constexpr int MAXDEV = 64;
  static std::atomic<int> resident_cache[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return HILC_ERR_LAUNCH;
  int cached = dev >= 0 && dev < MAXDEV ? resident_cache[dev].load(std::memory_order_relaxed) : 0;
  if (cached == 0) {
    int n_cu = 0, occ = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1)
      return HILC_ERR_LAUNCH;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, resblock_kernel<C, STREAM>, 256, 0) != hipSuccess || occ < 1)
      return HILC_ERR_LAUNCH;
    cached = n_cu * occ;
    if (dev >= 0 && dev < MAXDEV) resident_cache[dev].store(cached, std::memory_order_relaxed);
  }
  const long resident = cached;
  long blocks = a.total_tiles < resident ? a.total_tiles : resident;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL((resblock_kernel<C, STREAM>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

}  // namespace

namespace {
int resblock_entry(bool streaming, const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                   const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1, const float* hist2,
                   float* hist1_out, float* hist2_out, float* y, int* sched, int B, int C, int T, float pre_scale,
                   float out_scale, void* stream) {
  if (!x || !w1t || !dw1_w || !dw1_b || !w2t || !dw2_w || !dw2_b || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (x == y) return HILC_ERR_UNSUPPORTED;   // neighbouring tiles read each other's halo: not in place
  if (T % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
    return HILC_ERR_UNSUPPORTED;             // callers fall back to two hilc_dws_conv launches
  ResArgs a;
  a.x = x; a.w1t = w1t; a.dw1_w = dw1_w; a.dw1_b = dw1_b; a.w2t = w2t; a.dw2_w = dw2_w; a.dw2_b = dw2_b;
  a.y = y; a.T = T; a.tiles = (T + TO - 1) / TO; a.pre_scale = pre_scale; a.out_scale = out_scale;
  a.hist1 = hist1; a.hist2 = hist2; a.hist1_out = hist1_out; a.hist2_out = hist2_out;
  a.sched = sched;
#ifdef HILC_DEBUG_STAMPS
  a.dbg = g_dbg;
#else
  a.dbg = nullptr;
#endif
  if (streaming) {
    if ((hist1 && hist1 == hist1_out) || (hist2 && hist2 == hist2_out)) return HILC_ERR_UNSUPPORTED;   // first / last tiles of a clip race
    if ((long)B * C * T * 4 >= (1L << 32)) return HILC_ERR_UNSUPPORTED;   // 32-bit flat column index / byte offsets
    switch (C) {
      case 64: return launch_res<64, true>(a, B, (hipStream_t)stream);
      case 96: return launch_res<96, true>(a, B, (hipStream_t)stream);
      case 128: return launch_res<128, true>(a, B, (hipStream_t)stream);
      case 192: return launch_res<192, true>(a, B, (hipStream_t)stream);
      default: return HILC_ERR_UNSUPPORTED;
    }
  }
  switch (C) {
    case 64: return launch_res<64, false>(a, B, (hipStream_t)stream);
    case 96: return launch_res<96, false>(a, B, (hipStream_t)stream);
    case 128: return launch_res<128, false>(a, B, (hipStream_t)stream);
    case 192: return launch_res<192, false>(a, B, (hipStream_t)stream);
    default: return HILC_ERR_UNSUPPORTED;
  }
}
}  // namespace

extern "C" int hilc_resblock_pack_weights(const float* wt, float* packed, int C, void* stream) {
  if (!wt || !packed) return HILC_ERR_NULL;
  if (!(C == 64 || C == 96 || C == 128 || C == 192)) return HILC_ERR_UNSUPPORTED;
  if (wt == packed) return HILC_ERR_UNSUPPORTED;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((C * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wt, packed, C);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_resblock(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                             const float* w2t, const float* dw2_w, const float* dw2_b, float* y, int B, int C,
                             int T, float pre_scale, float out_scale, void* stream) {
  return resblock_entry(false, x, w1t, dw1_w, dw1_b, w2t, dw2_w, dw2_b, nullptr, nullptr, nullptr, nullptr, y, nullptr,
                        B, C, T, pre_scale, out_scale, stream);
}

extern "C" int hilc_resblock_stream(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                                    const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1,
                                    const float* hist2, float* hist1_out, float* hist2_out, float* y, int B, int C,
                                    int T, float pre_scale, float out_scale, void* stream) {
  return resblock_entry(true, x, w1t, dw1_w, dw1_b, w2t, dw2_w, dw2_b, hist1, hist2, hist1_out, hist2_out, y, nullptr,
                        B, C, T, pre_scale, out_scale, stream);
}

extern "C" int hilc_resblock_balanced(const float* x, const float* w1t, const float* dw1_w, const float* dw1_b,
                                      const float* w2t, const float* dw2_w, const float* dw2_b, const float* hist1,
                                      const float* hist2, float* hist1_out, float* hist2_out, float* y, int* sched,
                                      int streaming, int B, int C, int T, float pre_scale, float out_scale,
                                      void* stream) {
  return resblock_entry(streaming != 0, x, w1t, dw1_w, dw1_b, w2t, dw2_w, dw2_b, hist1, hist2, hist1_out, hist2_out, y,
                        sched, B, C, T, pre_scale, out_scale, stream);
}

#ifdef HILC_DEBUG_STAMPS
extern "C" void hilc_debug_set_stamp_buffer(unsigned long long* p) { g_dbg = p; }
#endif

extern "C" int hilc_resblock_supported(int C, int T) {
  return (C == 64 || C == 96 || C == 128 || C == 192) && T % 4 == 0;
}
