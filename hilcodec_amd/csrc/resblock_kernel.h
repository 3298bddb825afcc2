#pragma once
// resblock_kernel.h — part 3 of 4: the kernel.  Phase map of one tile (details: resblock_cfg.h):
//   [U]   up-sampling layer (decoder stages, Cfg::UR > 0): operand built in the tile, two GEMMs, + bias -> x registers
//   per block of the chain:  P0 prologue -> tile | P1 GEMM1 | P2 acc -> tile | P3 dw1 + ELU in place | P4 GEMM2 | P5 acc -> tile |
//                            P6 dw2, scale, shortcut -> HBM / x registers (next block) / activated tile (Q follows)
//   [D]   down-sampling layer (encoder stages, Cfg::DR > 0): ELU -> tile, two GEMMs, strided depthwise conv (+ res) -> HBM
//   [Q]   closing conv of the decoder (Cfg::POST): taps over the activated tile, class sums through PR, tanh -> waveform
// Register / LDS contract between the phases:
//   * xr[RW] (f32x4 per row of the lane) is the ONLY tile-sized state in registers: the block's input and shortcut; P6 of a chain's
//     inner block overwrites it with the block's output; it is dead after the last block's P6 when D / Q follows (D re-uses it for the
//     next tile's rows).
//   * the LDS tile X[C][XS] has ONE owner at a time: P0/P3/U/D0/P6(Q) write operands, GEMM phases read them, acc_to_x overwrites them
//     with results; every hand-over is one lds_barrier().  Columns NCOL.. of a row are the carry slots (4 floats each): H1 / H2 of
//     every block, then the D phase's two halves or the Q phase's one — written by the lane of the tile's last column group, read by
//     the lane of column group 0 of the NEXT tile of the run, zeroed at a clip's first tile.
//   * DW / DWD (tap tables) are written once per workgroup (or per block: DW_RELOAD) and read-only inside a phase; PR belongs to Q.
#include "resblock_gemm.h"
#include "stream_gemm.h"

namespace {

// this lane's 4 columns of a tile: clip, time (a multiple of 4; T % 4 == 0: the group is entirely inside or outside)
struct Cols {          // (no padding bytes: the struct is copied, and hipcc keeps a copied struct's padding alive as private arrays
  long b;              //  that its backend then "promotes" into LDS — 18 bytes per thread, 2-9 KB of the tile's budget)
  int t;               // b: offline: clip
  int t_in;
  unsigned boff;       // STREAM: byte offset of (clip, row 0, t) against the tensor base (B*C*T*4 < 2^32, launcher-checked)
  unsigned hoff;       // STREAM: element offset of this clip's [C][4] cache block
  int head, tail;      // STREAM: t == 0 (previous 4 samples live in the cache) / t == T-4 on an output column
};
static_assert(sizeof(Cols) == 32, "no padding");

template <int C, bool STREAM, bool SCARRY = false, int NB = 1, bool W8 = false, int DRU = 0, bool POST = false, bool SPEC0 = false>
__global__ __launch_bounds__((Cfg<C, STREAM, SCARRY, NB, W8, DRU, POST, SPEC0>::NT), (Cfg<C, STREAM, SCARRY, NB, W8, DRU, POST, SPEC0>::MINW)) void resblock_kernel(ResArgs a) {
  using K = Cfg<C, STREAM, SCARRY, NB, W8, DRU, POST, SPEC0>;
  constexpr int DR = K::DR, UR = K::UR;
  using Pipe = WeightPipe<K>;
  constexpr int CBW = K::CBW, NW = K::NW, NT = K::NT, RW = K::RW, RB = K::RB, XS = K::XS, TO = K::TO, RSTEP = K::RSTEP;
  // 4 floats in front of the tile: the "previous 4 columns" read of column group 0 (discarded halo outputs) stays a
  // plain base + constant address instead of a select
  __shared__ __attribute__((aligned(16))) float Xbuf[4 + C * XS];
  // tap tables: all blocks of a chain resident (loaded once per workgroup) — except where the NARROW shapes' tables (C >= 256: 12 - 36 KB
  // per block) do not fit beside the tile (C = 384 x 3 offline, C = 768 x 3 in a hop): those reload the one table at the start of every block
  constexpr bool DW_RELOAD = NB > 1 && K::NARROW && (4 + C * XS + NB * C * DWS) * 4 > 156 * 1024;
  __shared__ __attribute__((aligned(16))) float DW[(DW_RELOAD ? 1 : NB) * C * DWS];
  constexpr bool DWIDE = DR == 5 || DR == 8;            // the wide stages' down-sampling phase: taps straight from global memory (no LDS left)
  __shared__ __attribute__((aligned(16))) float DWD[(DR > 0 && !DWIDE) ? 2 * C * DDS : 4];
  // STREAM, T >= 128 (at most one clip start per tile): that clip's two caches, staged before P0 so that P3 / P6
  // do not pay one exposed global-load latency per row for the single lane that needs them
  __shared__ __attribute__((aligned(16))) float HS[(STREAM && !K::NARROW) ? 2 * C * 4 : 4];
  // POST: the row classes' partial sums of the closing conv, [class][column]
  __shared__ __attribute__((aligned(16))) float PR[POST ? POST_CLASSES * K::NCOL : 4];
  static_assert(!POST || (K::RPI * K::NW == POST_CLASSES), "a row class of the closing conv = the rows one half-wave walks");
  // SPEC0: the tile's waveform segment (127 + 64 samples) and the same scaled for the first conv
  __shared__ __attribute__((aligned(16))) float SEG[SPEC0 ? (STREAM ? 256 : 192) : 4];     // (STREAM: two pieces, 63 + split and 63 + 128 - split samples)
  __shared__ __attribute__((aligned(16))) float SEGS[SPEC0 ? (STREAM ? 256 : 192) : 4];
  float* const X = Xbuf + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> scalar loads below
  constexpr int NCB = K::NCOL / 32;                      // column blocks of the tile
  const int colblk = wave % NCB;
  const int wclass = wave / NCB;                         // row class (0 unless RH == 2)
  const int rowblk0 = wclass * CBW;
#if defined(HILC_DEBUG_WAVE_STAMPS)
  // tools/narrow_phase_waits.py (round 6): EVERY wave stamps every barrier of a tile — end of its own issue (t0), its LDS operations drained
  // (t1, after s_waitcnt lgkmcnt(0)), the barrier passed (t2) — into dbg[((tile * NW + wave) * WS_MAXB + k) * 3 ...]; slot WS_MAXB - 1 holds
  // (blockIdx, barriers of the tile, tile start).  Never in the product.
  constexpr int WS_MAXB = 48;
  unsigned long long* ws_base = nullptr;
  int ws_k = 0;
#define STAMP(i) do { } while (0)
#define lds_barrier() do {                                                                                          \
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                 \
    const unsigned long long t2_ = __builtin_amdgcn_s_memtime();                                                   \
    if (ws_base != nullptr && lane == 0 && ws_k < WS_MAXB - 1) {                                                   \
      ws_base[ws_k * 3] = t0_; ws_base[ws_k * 3 + 1] = t1_; ws_base[ws_k * 3 + 2] = t2_;                           \
    }                                                                                                              \
    ++ws_k;                                                                                                        \
  } while (0)
#elif defined(HILC_DEBUG_STAMPS)
#define STAMP(i) do { if (a.dbg && tid == 0) a.dbg[stamp_tile * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  long stamp_tile = blockIdx.x;
#else
#define STAMP(i) do { } while (0)
#endif
  // (s_setprio per phase kind — element-wise high / GEMM low and the reverse — measured +-0 in round 3 and again on the stage kernels in round 6:
  //  profiles/r06_ab_setprio.txt, commit 8c897d4)
  const int T = a.T;
  const int nblk = NB == 1 ? 1 : a.nblk;                 // uniform; NB == 1: the block loop below folds away
  // depthwise taps / biases -> LDS once per workgroup (read back as half-wave broadcasts in P3 / P6)
  auto load_taps = [&](const ResBlk& bq, float* dst) {
    if constexpr (DW_RELOAD && !STREAM) {
      // (inside the tile loop) one load per element through a SELECTED address — four branches are four exec-mask regions with a load
      // and a wait each — and a compile-time trip count: all of a thread's loads are in flight together.  (Only here: in the streaming
      // instantiations, which sit at 256 registers, the loads in flight spill.)
      constexpr int NE = C * DWS;
      static_assert(NE % NT == 0, "whole rounds");
#pragma unroll
      for (int i = 0; i < NE / NT; ++i) {
        const int e = tid + i * NT;
        const int m = e / DWS, j = e - m * DWS;
        const float* src = j < 5 ? bq.dw1_w + (m * 5 + j) : (j == 5 ? bq.dw1_b + m : (j < 11 ? bq.dw2_w + (m * 5 + (j - 6)) : bq.dw2_b + m));
        dst[e] = *src;
      }
    } else if constexpr (DW_RELOAD) {
      // (the streaming reload kernels sit at 256 registers: the same selected address, three loads in flight)
      constexpr int NE = C * DWS;
      static_assert(NE % NT == 0, "whole rounds");
#pragma unroll 3
      for (int i = 0; i < NE / NT; ++i) {
        const int e = tid + i * NT;
        const int m = e / DWS, j = e - m * DWS;
        const float* src = j < 5 ? bq.dw1_w + (m * 5 + j) : (j == 5 ? bq.dw1_b + m : (j < 11 ? bq.dw2_w + (m * 5 + (j - 6)) : bq.dw2_b + m));
        dst[e] = *src;
      }
    }
  };
  if constexpr (!DW_RELOAD) {
    // All blocks' tables, ONE row per thread and round: a row's 12 words are 12 independent loads and every block's are requested before the
    // first LDS store — one global round trip per launch (two where C > NT) instead of two per block (an element per thread and round:
    // a hop is 5 - 10 tiles per workgroup, and the prologue's round trips were a visible part of a streaming stage launch).
    constexpr int RND = (C + NT - 1) / NT;
    if constexpr (NB == 1 && STREAM) {
      // (the one-block streaming shapes sit at their register limit — C = 128: 24 bytes of scratch with twelve more words live here: a row as
      //  three groups of four words, three round trips; these launches only run with ExecOptions.stage_launches off)
      const ResBlk& bq = a.blk[0];
#pragma unroll
      for (int it = 0; it < RND; ++it) {
        const int m = tid + it * NT;
        const int mc = (C % NT == 0 || m < C) ? m : C - 1;
#pragma unroll
        for (int w4 = 0; w4 < DWS / 4; ++w4) {
          float t4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 4 * w4 + e;
            t4[e] = j < 5 ? bq.dw1_w[mc * 5 + j] : (j == 5 ? bq.dw1_b[mc] : (j < 11 ? bq.dw2_w[mc * 5 + (j - 6)] : bq.dw2_b[mc]));
          }
          if (C % NT == 0 || m < C) *reinterpret_cast<f32x4*>(&DW[m * DWS + 4 * w4]) = f32x4{t4[0], t4[1], t4[2], t4[3]};
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    float tv[NB][RND][DWS];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      if (q < nblk) {
        const ResBlk& bq = a.blk[q];
#pragma unroll
        for (int it = 0; it < RND; ++it) {
          const int m = tid + it * NT;
          const int mc = (C % NT == 0 || m < C) ? m : C - 1;
#pragma unroll
          for (int j = 0; j < 5; ++j) tv[q][it][j] = bq.dw1_w[mc * 5 + j];
          tv[q][it][5] = bq.dw1_b[mc];
#pragma unroll
          for (int j = 0; j < 5; ++j) tv[q][it][6 + j] = bq.dw2_w[mc * 5 + j];
          tv[q][it][11] = bq.dw2_b[mc];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      if (q < nblk) {
#pragma unroll
        for (int it = 0; it < RND; ++it) {
          const int m = tid + it * NT;
          if (C % NT == 0 || m < C) {
#pragma unroll
            for (int w4 = 0; w4 < DWS / 4; ++w4)
              *reinterpret_cast<f32x4*>(&DW[(q * C + m) * DWS + 4 * w4]) =
                  f32x4{tv[q][it][4 * w4], tv[q][it][4 * w4 + 1], tv[q][it][4 * w4 + 2], tv[q][it][4 * w4 + 3]};
          }
        }
      }
    }
    }
  }
  if constexpr (DR > 0 && !DWIDE) {
    // the down-sampling layer's table the same way: one row (2r taps + bias) per thread, every load in flight before the first store
    constexpr int RND = (2 * C + NT - 1) / NT;
    float tw[RND][2 * DR + 1];
#pragma unroll
    for (int it = 0; it < RND; ++it) {
      const int m = tid + it * NT;
      const int mc = ((2 * C) % NT == 0 || m < 2 * C) ? m : 2 * C - 1;
#pragma unroll
      for (int j = 0; j < 2 * DR; ++j) tw[it][j] = a.dn.dw_w[mc * 2 * DR + j];
      tw[it][2 * DR] = a.dn.dw_b[mc];
    }
#pragma unroll
    for (int it = 0; it < RND; ++it) {
      const int m = tid + it * NT;
      if ((2 * C) % NT == 0 || m < 2 * C) {
#pragma unroll
        for (int j = 0; j < DDS; ++j) DWD[m * DDS + j] = j < 2 * DR ? tw[it][j] : (j == 8 ? tw[it][2 * DR] : 0.f);
      }
    }
  }
  if (tid < 4) Xbuf[tid] = 0.f;
  // element-wise phases: one half-wave = one row (row = 2*wave + (lane>>5) + 2*NW*i), lane = 4 adjacent
  // columns: 16-B global accesses, 512 B contiguous per half-wave; a row is read and written by one
  // wave instruction, so the in-place update of P3 needs no barrier.
  const int rsub = wave * K::RPI + (K::RPI > 1 ? lane / (K::NCOL / 4) : 0);
  const int c4 = (lane & (K::NCOL / 4 - 1)) * 4;
  [[maybe_unused]] const unsigned row_b = (unsigned)T * 4u;
  [[maybe_unused]] const bool one_head = STREAM && !K::NARROW && T >= K::NCOL;

  auto columns_of = [&](long tile) -> Cols {
    Cols s;
    s.boff = 0; s.hoff = 0; s.head = false; s.tail = false;
    if constexpr (STREAM) {
      const int flat = (int)tile * TO - K::HALO + c4;
      s.t_in = flat >= 0 && flat < a.B * T;
      const unsigned ub = s.t_in ? __umulhi((unsigned)flat, a.div_magic) >> a.div_shift : 0u;   // flat / T
      s.b = ub;
      s.t = s.t_in ? flat - (int)ub * T : 0;
      s.boff = (ub * (unsigned)(C * T) + (unsigned)s.t) * 4u;
      s.hoff = ub * (unsigned)(C * 4);
      s.head = s.t_in && s.t == 0;
      s.tail = s.t_in && c4 >= K::HALO && s.t == T - 4;
    } else {
      s.b = tile / a.tiles;
      s.t = (int)(tile - s.b * a.tiles) * TO + c4;
      s.t_in = s.t >= 0 && s.t < T;
    }
    return s;
  };
  auto xrow = [&](const Cols& s, int m) -> const f32x4* {     // rows outside [0, T) read a mapped dummy (zeroed in P0)
    if constexpr (STREAM)
      return reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x) + (s.boff + (unsigned)m * row_b));
    else
      return reinterpret_cast<const f32x4*>(a.x + s.b * (long)C * T + (long)m * T + (s.t_in ? s.t : 0));
  };
  auto yrow = [&](const Cols& s, int m) -> f32x4* {
    if constexpr (STREAM)
      return reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.y) + (s.boff + (unsigned)m * row_b));
    else
      return reinterpret_cast<f32x4*>(a.y + s.b * (long)C * T + (long)m * T + s.t);
  };

  // The 4 samples in front of this lane's column group: the tile's own columns, or — STREAM, at a stream's t = 0 — that
  // stream's cache.  Branch-free (address / value selects): a per-row `if (head)` is an exec-mask region per row that splits a
  // batch's loads — by the stamps P3 took 31.8 k cycles per tile at C = 128 against 7.5 k offline.
  // the current block's caches (set per block, below).  Scalars, not arrays: a private array that is written inside the tile loop
  // becomes an alloca, and the AMDGPU backend "promotes" it to LDS (9-18 bytes per thread: 2-9 KB of the tile's LDS budget)
  [[maybe_unused]] const float* hb0 = a.x;
  [[maybe_unused]] const float* hb1 = a.x;
  [[maybe_unused]] bool hv0 = false, hv1 = false;
  [[maybe_unused]] int blk_off = 0;                            // 8 * block: that block's carry slots behind a row
  Cols cs;
  // The 4 columns in front of a lane's column group are its left neighbour's — except for column group 0, whose "previous columns"
  // are the last 4 of the run's previous tile.  They live in two 4-float slots behind each row of the tile (H1 at X[m][NCOL..],
  // H2 at X[m][NCOL + 4..]): the lane of group 0 reads them through ONE base pointer chosen per phase (same row stride, same
  // immediate offsets as everybody else), the lane of the last group leaves them at a constant offset from its own pointer.
  // A clip's first tile finds zeros there (the convs' zero padding); STREAM: at a stream's t = 0 the cache overrides.
  const bool lane0 = c4 == 0, lane_last = c4 == K::NCOL - 4;
  auto prev_base = [&](int which) -> lptr_t {        // + i * RSTEP * XS = the 16-B word in front of row (rsub + RSTEP * i)'s group
    if constexpr (!K::CARRYMODE) return (lptr_t)(X + rsub * XS + c4 - 4);       // STREAM: group 0 = discarded halo columns, or a head
    else return lane0 ? (lptr_t)(X + rsub * XS + K::NCOL + which * 4 + blk_off) : (lptr_t)(X + rsub * XS + c4 - 4);
  };
  // prev[i] for the RB rows of the batch whose first row sits RW0 rows (in units of RSTEP) below rsub; pb = prev_base(which)
  auto prevs_of = [&](lptr_t pb, int m0, int which, f32x4 (&prev)[K::RB]) {
    if constexpr (!STREAM) {
#pragma unroll
      for (int i = 0; i < K::RB; ++i) prev[i] = *(lvec_t)(pb + i * RSTEP * XS);
    } else {
      bool staged = false;
      if constexpr (!K::NARROW) staged = one_head;      // wave-uniform: ONE branch per batch
      if (staged) {      // T >= tile width: the one cache block of the tile sits in HS, staged during P0
#pragma unroll
        for (int i = 0; i < K::RB; ++i) {
          const lptr_t pa = cs.head ? (lptr_t)(HS + (which * C + m0 + RSTEP * i) * 4) : pb + i * RSTEP * XS;
          prev[i] = *(lvec_t)pa;
        }
      } else {
        // every lane reads ITS stream's cache words (a valid address for every lane: hoff = 0 outside the tensor), heads keep them
        f32x4 hv[K::RB];
#pragma unroll
        for (int i = 0; i < K::RB; ++i)
          hv[i] = *reinterpret_cast<const f32x4*>((which ? hb1 : hb0) + ((which ? hv1 : hv0) ? cs.hoff + (unsigned)(m0 + RSTEP * i) * 4u : 0u));
#pragma unroll
        for (int i = 0; i < K::RB; ++i) {
          const f32x4 own = *(lvec_t)(pb + i * RSTEP * XS);
#pragma unroll
          for (int e = 0; e < 4; ++e) prev[i][e] = cs.head ? ((which ? hv1 : hv0) ? hv[i][e] : 0.f) : own[e];
        }
      }
    }
  };
  // xp = this lane's pointer to its group in the batch's first row: the last group's lane leaves the batch's carry words
  auto carry_out = [&](lptr_t xp, int which, const f32x4 (&cur)[K::RB]) {
    if constexpr (K::CARRYMODE) {
      if (lane_last) {
#pragma unroll
        for (int i = 0; i < K::RB; ++i) *(lvec_t)(xp + i * RSTEP * XS + 4 + which * 4 + blk_off) = cur[i];
      }
    }
  };

  // persistent: this workgroup walks tiles (static stride or tickets).  x registers: the tile's rows of this
  // lane, loaded one tile ahead; they are the GEMM input (through P0) AND the shortcut of P6.
  float touch = 0.f;
  constexpr int LPR = K::NCOL / 32;                 // 128-B lines per tile row
  // (STREAM: no touch loads — a hop's activations were written by the previous launch a few hundred microseconds ago, and the
  // registers of the address arithmetic are what the cache handling needs)
  // (UTOUCH, round 6: the offline carry form of the narrow decoder stages touches the next tile's INPUT FRAMES of the up-sampling phase —
  //  2C rows x NCOL / r frames — the same way: by the per-wave stamps the two operand builds of a C = 96 tile were 13 % of it, two exposed HBM
  //  round trips per half)
#ifdef HILC_RES_NO_UTOUCH      // A/B builds of tools/
  constexpr bool UTOUCH = false;
#else
  constexpr bool UTOUCH = !STREAM && UR > 0 && !K::NARROW && (K::NCOL / (UR > 0 ? UR : 1)) % 32 == 0;
#endif
  constexpr int LPRU = UTOUCH ? K::NCOL / UR / 32 : 1;     // 128-B lines of input frames per tile row
  constexpr int NTOUCH = STREAM ? 1 : (UTOUCH ? (2 * C * LPRU + NT - 1) / NT : (C * LPR + NT - 1) / NT);
  float tv[NTOUCH];                                 // L2 touch loads in flight across the tile boundary
#pragma unroll
  for (int i = 0; i < NTOUCH; ++i) tv[i] = 0.f;
  // Offline: this workgroup's run = tiles [run0, run1) of the launch, in order.  A run that starts inside a clip first walks the
  // tile in front of it as a WARM-UP (everything computed, nothing stored): its last 4 columns of H1 and H2 depend only on its own
  // columns 116..127, so the carry it leaves is exact whatever it was handed itself.
  // STREAM: tiles by static stride or by ticket (where several workgroups share a CU), every tile with its own halo.
  __shared__ long s_next;
  long run0 = blockIdx.x, run1 = a.total_tiles;
  long tile = blockIdx.x;
  if constexpr (K::CARRYMODE) {
    if (a.classes > 1) {
      // Workgroups that share a CU are not served equally: the one dispatched first (lower blockIdx: the grid is classes x CUs, a CU
      // holds one workgroup of every class) wins issue arbitration until it is done — with equal runs the first class finished at
      // 0.76 of the kernel and the last ran on alone (tools/res_wg_times.py).  Runs in proportion to the classes' speeds: a
      // "slot" u owns a contiguous range of tiles, its classes split it in dispatch order.
      const unsigned P = gridDim.x / (unsigned)a.classes, u = blockIdx.x % P, c = blockIdx.x / P;
      const long s0 = (long)u * a.total_tiles / P, len = (long)(u + 1) * a.total_tiles / P - s0;
      run0 = s0 + ((len * (long)a.cum[c]) >> 16);
      run1 = s0 + ((len * (long)a.cum[c + 1]) >> 16);
    } else if (a.run_tiles > 0) {         // chain launch on the streaming column space: runs of whole streams, the last may be short
      run0 = (long)blockIdx.x * a.run_tiles;
      run1 = run0 + a.run_tiles < a.total_tiles ? run0 + a.run_tiles : a.total_tiles;
    } else {
      run0 = (long)blockIdx.x * a.total_tiles / gridDim.x;
      run1 = ((long)blockIdx.x + 1) * a.total_tiles / gridDim.x;
    }
    const bool mid = STREAM ? (run0 * TO) % T != 0 : run0 % a.tiles != 0;     // does the run start inside a clip / stream?
    tile = (mid && run0 < run1) ? run0 - 1 : run0;
  }
  cs = columns_of(tile < run1 ? tile : 0);
  f32x4 xr[RW];
  if (UR == 0 && !SPEC0 && tile < run1) {
#pragma unroll
    for (int i = 0; i < RW; ++i) xr[i] = *xrow(cs, rsub + RSTEP * i);
  }
  lds_barrier();   // DW / pad visible
  while (tile < run1) {
    const bool warm = K::CARRYMODE && tile < run0;           // uniform
    constexpr bool TICKETS = !K::CARRYMODE && K::NW == 4;
    if constexpr (K::CARRYMODE && !STREAM) {
      // a clip's first tile: the zero padding in front of t = 0 is a zero carry (the end-of-tile barrier is behind us, P3 reads
      // it two barriers from here).  (STREAM: column group 0 of a stream's first tile is a head and takes the cache.)
      if (tile % a.tiles == 0) {
        constexpr int NSLOT = 2 * NB + K::DCAR / 2 + (POST ? 1 : 0);      // 4-float slots behind a row: H1 / H2 per block + the down-sampling phase's two halves (+ the closing conv's)
        for (int e = tid; e < NSLOT * C; e += NT)
          *reinterpret_cast<f32x4*>(X + (e / NSLOT) * XS + K::NCOL + (e % NSLOT) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (!K::CARRYMODE) {
      // tickets only where several workgroups share a CU (one per CU progresses evenly: static stride, no atomic round trip on
      // the critical path)
      if (TICKETS && a.sched != nullptr && tid == 0) s_next = (long)gridDim.x + atomicAdd(a.sched, 1);   // read after P0's barrier
    }
#if defined(HILC_DEBUG_WAVE_STAMPS)
    if (a.dbg != nullptr) {
      if (ws_base != nullptr && lane == 0) ws_base[(WS_MAXB - 1) * 3 + 1] = (unsigned long long)ws_k;      // barriers of the tile just finished
      ws_base = a.dbg + ((tile * NW + wave) * (long)WS_MAXB) * 3;
      if (lane == 0) { ws_base[(WS_MAXB - 1) * 3] = blockIdx.x; ws_base[(WS_MAXB - 1) * 3 + 2] = __builtin_amdgcn_s_memtime(); }
    }
    ws_k = 0;
#elif defined(HILC_DEBUG_STAMPS)
    stamp_tile = tile;
#endif
    STAMP(0);
    long next_tile = K::CARRYMODE ? tile + 1 : tile + gridDim.x;       // (tickets: s_next, read after P0's barrier)
    if constexpr (UR > 0) {
      // ---- U: x = W_up * u + bias; u[k][t] = w[k][t mod r] * a[k][t / r] + w[k][r + t mod r] * a[k][t / r - 1], a = ELU(in_scale * x_in),
      //      a[.][-1] = the stream's cache (activated).  Same expression and the same k order as hilc_up_conv: bit-identical.
      const ResUp& up = a.up;
      const int Tin = T / UR;
      const int q0 = cs.t / UR, p0 = cs.t - q0 * UR;             // input frame / phase of this lane's 4 columns (they share q0: r = 8)
      f32x16 acc[CBW];
#pragma unroll
      for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma nounroll
      for (int h = 0; h < 2; ++h) {
        const float* wu = (h ? up.w_hi : up.w_lo) + (long)wclass * (C * C / K::RH);
        asm volatile("" : "+s"(wu));
        {
          // the operands of UB rows first (one exposed round trip per batch, not one per row), then their arithmetic.
          // r = 8 / 4: the lane's 4 columns share one input frame q0 (phases p0 .. p0 + 3); r = 2: they are frames q0, q0, q0 + 1, q0 + 1
          // (phases 0, 1, 0, 1) — the same cases as UpB<R> in gemm_lin.h.
          constexpr int UB = RW % 6 == 0 ? 6 : RB;      // (round 6: all 12 rows of a C = 96 half in ONE batch measured +-0 — the build is issue-, not latency-bound)
          // r = 5 (any stride that does not divide 4 columns into whole frames): the EXPANDED tap table of hilc_up_conv_expand_taps,
          // `[2C][r][8]` — for the phase p0 = t mod r of a 4-column group its eight taps as two 16-B words — and three input frames per
          // row (q0 - 1, q0, q0 + 1): column e belongs to frame q0 + 1 where p0 + e >= r.  The expression of UpB<1> (gemm_lin.h).
          constexpr bool UEXP = UR == 5;
          lptr_t xp = (lptr_t)(X + rsub * XS + c4);
          const bool has_prev = cs.t_in && q0 >= 1;        // (offline: a group past the clip's end keeps its t)
          [[maybe_unused]] const bool has_next = cs.t_in && q0 + 1 < Tin;
#pragma unroll
          for (int i0 = 0; i0 < RW; i0 += UB) {
            float xc[UB], xq[UB];
            [[maybe_unused]] float xn[UB];
            f32x4 wa[UB];
            [[maybe_unused]] f32x4 wb[UB];
#pragma unroll
            for (int i = 0; i < UB; ++i) {
              const int k = h * C + rsub + RSTEP * (i0 + i);
              const long row = (long)cs.b * (2 * C) + k;
              const float* xi = up.xin + (cs.t_in ? row * Tin + q0 : 0);
              xc[i] = xi[0];
              const float* pp = has_prev ? xi - 1 : (up.hist != nullptr && cs.t_in ? up.hist + row : xi);
              xq[i] = pp[0];
              if constexpr (UR == 2) {
                xn[i] = xi[cs.t_in ? 1 : 0];                       // T % 4 == 0 and r = 2: frame q0 + 1 exists wherever the group does
                wa[i] = *reinterpret_cast<const f32x4*>(up.tr_w + k * 4);          // (w0, w1 | w2, w3)
              } else if constexpr (UEXP) {
                xn[i] = xi[has_next ? 1 : 0];
                const float* tw = up.tr_w + ((long)k * UR + p0) * 8;
                wa[i] = *reinterpret_cast<const f32x4*>(tw);
                wb[i] = *reinterpret_cast<const f32x4*>(tw + 4);
              } else {
                wa[i] = *reinterpret_cast<const f32x4*>(up.tr_w + k * (2 * UR) + p0);
                wb[i] = *reinterpret_cast<const f32x4*>(up.tr_w + k * (2 * UR) + UR + p0);
              }
            }
#pragma unroll
            for (int i = 0; i < UB; ++i) {
              const float a1 = prologue(xc[i], up.in_scale, 1);
              const float a0 = has_prev ? prologue(xq[i], up.in_scale, 1) : (up.hist != nullptr ? xq[i] : 0.f);   // the cache holds activated samples
              f32x4 u;
              float a_last = a1;
              if constexpr (UR == 2) {
                const float a2 = prologue(xn[i], up.in_scale, 1);
                u[0] = fmaf(wa[i].x, a1, wa[i].z * a0);
                u[1] = fmaf(wa[i].y, a1, wa[i].w * a0);
                u[2] = fmaf(wa[i].x, a2, wa[i].z * a1);
                u[3] = fmaf(wa[i].y, a2, wa[i].w * a1);
                a_last = a2;
              } else if constexpr (UEXP) {
                const float a2 = has_next ? prologue(xn[i], up.in_scale, 1) : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const bool nx = p0 + e >= UR;
                  u[e] = fmaf(wa[i][e], nx ? a2 : a1, wb[i][e] * (nx ? a1 : a0));
                }
                a_last = p0 + 3 >= UR ? a2 : a1;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = fmaf(wa[i][e], a1, wb[i][e] * a0);
              }
              *(lvec_t)(xp + i * RSTEP * XS) = zero_unless(cs.t_in, u);
              if constexpr (STREAM) {
                if (cs.tail && !warm && up.hist_out != nullptr)   // the stream's last group: its (last) input frame is the next hop's cache
                  up.hist_out[(long)cs.b * (2 * C) + h * C + rsub + RSTEP * (i0 + i)] = a_last;
              }
            }
            xp += UB * RSTEP * XS;
            asm volatile("" : "+v"(xp));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        Pipe wp;
        wp.prefetch(wu, lane);
        lds_barrier();
        if constexpr (K::NARROW) gemm_phase_rolled<K, false>(wu, X, acc, wp, colblk, lane);
        else gemm_phase<K, false>(wu, X, acc, wp, colblk, lane);
        lds_barrier();
      }
      acc_to_x<K>(acc, X, rowblk0, colblk, lane);
      lds_barrier();
      {
        lptr_t xp = (lptr_t)(X + rsub * XS + c4);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const float bv = up.bias != nullptr ? up.bias[rsub + RSTEP * i] : 0.f;
          f32x4 v = *(lvec_t)(xp + i * RSTEP * XS);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = up.bias != nullptr ? __fadd_rn(v[e], bv) : v[e];
          xr[i] = v;
        }
      }
      lds_barrier();       // block 0's P0 overwrites X
    }
    if constexpr (SPEC0) {
      // ---- S: x = conv_pre(wav) + out_scale * (W_pw * logspec(STFT_64(wav)) + bias) for the tile's 128 frames (hop 1: frame = sample).
      //      hilc_spec_block_conv_pre's phases S0 / A / B / C / D on this kernel's tile (wave = 32-column block, both row blocks), same
      //      fmaf chains and roundings: bit-identical; the result goes to the x registers instead of HBM.
      const ResSpec0& sp = a.spec;
      constexpr int N = 64, NBIN = 33, SROWS = 40, KP0 = 4, DEPTH0 = 2;
      // STREAM: frames are flat columns; a tile that holds a stream's t = 0 at local column `split` (T >= 128: at most one, and a run
      // starts on one: split = 0 never occurs with a predecessor) stages TWO pieces — the first stream's last samples, then the
      // second stream's 63 samples of history and its first 128 - split samples — and a column right of the split reads 63 further on.
      [[maybe_unused]] int split = K::NCOL;
      if constexpr (STREAM) {
        const int F0 = (int)tile * TO;
        const unsigned bA = __umulhi((unsigned)F0, a.div_magic) >> a.div_shift;
        const int tA0 = F0 - (int)bA * T;
        const int nxt = T - tA0;                               // columns until the next stream's t = 0
        split = nxt < K::NCOL ? nxt : K::NCOL;
        const int lenA = (N - 1) + split;
        for (int i = tid; i < 2 * (N - 1) + K::NCOL; i += NT) {
          const bool second = i >= lenA;
          const unsigned b = bA + (second ? 1u : 0u);
          const int t = second ? i - lenA - (N - 1) : tA0 - (N - 1) + i;
          float v = 0.f;
          if (b < (unsigned)a.B) {
            if (t >= 0) v = sp.wav[(long)b * T + t];
            else if (sp.hist != nullptr && t >= -sp.hist_len) v = sp.hist[(long)b * sp.hist_len + sp.hist_len + t];
          }
          SEG[i] = v;
          SEGS[i] = v * sp.pre_in_scale;
        }
        for (int i = tid; i < (SROWS - NBIN) * 128; i += NT) X[(NBIN + (i >> 7)) * XS + (i & 127)] = 0.f;   // zero rows of the conv's padded K
      } else {
      const int f0 = cs.t - c4;                              // the tile's first frame (offline: cs.t = f0 + c4 for every lane)
      {
        const int s0 = f0 - (N - 1);
        const float* wb = sp.wav + cs.b * (long)T;
        for (int i = tid; i < 127 + N; i += NT) {
          const int t = s0 + i;
          const float v = (t >= 0 && t < T) ? wb[t] : 0.f;
          SEG[i] = v;
          SEGS[i] = v * sp.pre_in_scale;                     // scaled once per sample, not once per use
        }
        for (int i = tid; i < (SROWS - NBIN) * 128; i += NT) X[(NBIN + (i >> 7)) * XS + (i & 127)] = 0.f;   // zero rows of the conv's padded K
      }
      }
      lds_barrier();
      const int kh = lane >> 5, col = colblk * 32 + (lane & 31);
      const int colx = col + ((STREAM && col >= split) ? N - 1 : 0);      // this column's window starts here in SEG
      f32x16 acc[CBW];
      {
        int off[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) off[p] = colx + 2 * p + kh;
        auto bop = [&](int P) -> float { return SEG[off[P & 7] + 16 * (P >> 3)]; };
        stream_gemm<CBW, KP0, DEPTH0, N / 2 / KP0>(sp.dft, acc, lane, bop);
      }
      float nyq_im = 0.f;
#pragma unroll 8
      for (int k = 0; k < N; ++k) nyq_im = fmaf(sp.nyq[k], SEG[colx + k], nyq_im);
      const SpecFinish finish = SpecFinish::make(sp.mean, sp.stdv, sp.normalize);
#pragma unroll
      for (int i = 0; i < CBW; ++i) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int row = i * 32 + acc_row(r, lane);          // even
          if (i == 0 && r == 0) {
            if (kh == 0) {                                    // rows 0, 1: the two real bins
              X[0 * XS + col] = finish(acc[0][0], 0.f);
              X[(N / 2) * XS + col] = finish(acc[0][1], nyq_im);
            } else {                                          // rows 4, 5: bin 2
              X[(row >> 1) * XS + col] = finish(acc[0][0], acc[0][1]);
            }
          } else {
            X[(row >> 1) * XS + col] = finish(acc[i][r], acc[i][r + 1]);
          }
        }
      }
      lds_barrier();
      {
        const float* sl = X + kh * XS + col;
        auto bop = [&](int P) -> float { return sl[2 * P * XS]; };
        stream_gemm<CBW, KP0, DEPTH0, SROWS / 2 / KP0>(sp.pw, acc, lane, bop);
      }
      lds_barrier();                                          // every wave is done reading the spectrogram rows
      acc_to_x<K>(acc, X, rowblk0, colblk, lane);
      lds_barrier();
      {
        lptr_t xp = (lptr_t)(X + rsub * XS + c4);
        // column c of the tile is time f0 + c = segment sample c + N - 1: the first conv's taps read samples c + N - 5 .. c + N - 1
        float sm[8];
        const int c4x = c4 + ((STREAM && c4 >= split) ? N - 1 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) sm[j] = SEGS[c4x + j + N - 5];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const int m = rsub + RSTEP * i;
          const float bv = sp.bias != nullptr ? sp.bias[m] : 0.f;
          const float pb = sp.pre_b != nullptr ? sp.pre_b[m] : 0.f;
          float w5[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) w5[j] = sp.pre_w[m * 5 + j];
          f32x4 v = *(lvec_t)(xp + i * RSTEP * XS);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float acc5 = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) acc5 = fmaf(w5[j], sm[e + j], acc5);
            const float rr = sp.pre_b != nullptr ? __fadd_rn(acc5, pb) : acc5;
            float t = v[e];
            if (sp.bias != nullptr) t = __fadd_rn(t, bv);
            t = __fmul_rn(t, sp.out_scale);
            v[e] = __fadd_rn(t, rr);                          // separate roundings: y.mul_(scale); x.add_(y)
          }
          xr[i] = v;
          if (i % 2 == 1) __builtin_amdgcn_sched_barrier(0);  // rows stay rows: hoisting every row's taps costs RW * 7 registers (spills)
        }
      }
      lds_barrier();       // block 0's P0 overwrites X
    }
#pragma nounroll
    for (int blk = 0; blk < nblk; ++blk) {
    const ResBlk& bp = a.blk[NB == 1 ? 0 : blk];
    const bool final_blk = NB == 1 || blk == nblk - 1;      // uniform
    const bool last_blk = DR == 0 && !POST && final_blk;    // the block whose output leaves the kernel (DR > 0 / POST: none, the D / Q phase follows)
    [[maybe_unused]] const bool post_blk = POST && final_blk;   // ... whose ACTIVATED output goes to the tile for the closing conv
    Cols cn = cs;                                           // the next tile's columns: needed by the chain's last block only
    bool have_next = false;
    if constexpr (NB > 1) blk_off = 8 * blk;
    const float* const DWb = DW + ((NB == 1 || DW_RELOAD) ? 0 : blk * C * DWS);
    if constexpr (DW_RELOAD) load_taps(bp, DW);              // the previous block's last reads are behind its closing barrier; P0's barrier publishes
    if constexpr (STREAM) {
      hb0 = bp.hist1 != nullptr ? bp.hist1 : a.x;
      hb1 = bp.hist2 != nullptr ? bp.hist2 : a.x;
      hv0 = bp.hist1 != nullptr;
      hv1 = bp.hist2 != nullptr;
    }
    // the weight loads are invariant across tiles; launder the pointers so LICM does not try to keep
    // every weight of both matrices in registers across the tile loop (it spills 8 KB/lane if it does)
    const float* w1t = bp.w1t + (long)wclass * (C * C / K::RH);
    const float* w2t = bp.w2t + (long)wclass * (C * C / K::RH);
    asm volatile("" : "+s"(w1t), "+s"(w2t));
    // STREAM, T >= tile width: the ONE stream start a tile can hold.  Its two [C][4] cache blocks are contiguous: one coalesced
    // 16-B load per thread, requested here and written to HS at the end of P0 (the round trip hides behind the prologue; P3 / P6
    // read HS barriers later).  (Round 3 until here: the few lanes that sit on t = 0 loaded their 2 * RW words themselves,
    // 32 masked loads in a row and their latency in front of P0 — 3-12 k cycles per tile by the stamps.)
    [[maybe_unused]] f32x4 hstage[(STREAM && !K::NARROW) ? (2 * C + NT - 1) / NT : 1];
    [[maybe_unused]] bool stage_hs = false;
    if constexpr (STREAM && !K::NARROW) {
      if (one_head) {
        const int first = (int)tile * TO - K::HALO, last = first + K::NCOL - 1;
        const unsigned bh = __umulhi((unsigned)last, a.div_magic) >> a.div_shift;      // stream of the tile's last column
        const int hcol = (int)bh * T;
        stage_hs = hcol >= first && (int)bh < a.B;                                     // uniform: its t = 0 lies in this tile
        if (stage_hs) {
#pragma unroll
          for (int q = 0; q < (2 * C + NT - 1) / NT; ++q) {
            const int e = tid + q * NT;
            const int which = e >= C ? 1 : 0, m = e - which * C;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < 2 * C && (which ? hv1 : hv0)) v = *reinterpret_cast<const f32x4*>((which ? hb1 : hb0) + bh * (unsigned)(C * 4) + (unsigned)m * 4u);
            hstage[q] = v;
          }
        }
      }
    }
    Pipe wp;
    wp.prefetch(w1t, lane);                  // GEMM1's first weight slices travel while P0 runs
    // ---- P0: the prologue on the x registers (they stay live: shortcut of P6)
    {
      lptr_t xp = (lptr_t)(X + rsub * XS + c4);   // walks down the tile RB rows at a time (laundered: see gemm_phase)
#pragma unroll
      for (int i0 = 0; i0 < RW; i0 += RB) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
          *(lvec_t)(xp + i * RSTEP * XS) = prologue4v(zero_unless(cs.t_in, xr[i0 + i]), bp.pre_scale, 1);
        xp += RB * RSTEP * XS;
        asm volatile("" : "+v"(xp));
      }
    }
    // last tile's touch loads are older than the x loads P0 just waited for: consuming them here costs no wait (at
    // the end of P6 it drained the queue, i.e. waited for the freshly issued x loads of the next tile)
#pragma unroll
    for (int i = 0; i < NTOUCH; ++i) touch += tv[i];
    if constexpr (STREAM && !K::NARROW) {
      if (stage_hs) {
#pragma unroll
        for (int q = 0; q < (2 * C + NT - 1) / NT; ++q) {
          const int e = tid + q * NT;
          if (e < 2 * C) *reinterpret_cast<f32x4*>(&HS[e * 4]) = hstage[q];
        }
      }
    }
    lds_barrier();
    STAMP(1);
    if (TICKETS && a.sched != nullptr) next_tile = s_next;

    f32x16 acc[CBW];
    // ---- P1, P2
    if constexpr (K::NARROW) gemm_phase_rolled<K>(w1t, X, acc, wp, colblk, lane);
    else gemm_phase<K>(w1t, X, acc, wp, colblk, lane);
    lds_barrier();
    STAMP(2);
    acc_to_x<K>(acc, X, rowblk0, colblk, lane);
    lds_barrier();
    STAMP(3);

    // ---- P3: a2 = ELU(dw1(H1) + b1), zero for t < 0, in place, RB rows at a time
    wp.prefetch(w2t, lane);                  // GEMM2's first weight slices travel while P3 runs
    lptr_t xp3 = (lptr_t)(X + rsub * XS + c4);
    lptr_t pb3 = prev_base(0);
#pragma unroll
    for (int i0 = 0; i0 < RW; i0 += RB) {
      f32x4 cur[RB], prev[RB], wa[RB];
      f32x2 wb[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int m = rsub + RSTEP * (i0 + i);
        const lptr_t row = xp3 + i * RSTEP * XS;
        cur[i] = *(lvec_t)(row);
        wa[i] = *reinterpret_cast<const f32x4*>(&DWb[m * DWS]);
        wb[i] = *reinterpret_cast<const f32x2*>(&DWb[m * DWS + 4]);
      }
      prevs_of(pb3, rsub + RSTEP * i0, 0, prev);
      carry_out(xp3, 0, cur);                          // H1's last 4 columns, before the row is overwritten in place
      if constexpr (STREAM) {
        if (cs.tail && !warm && bp.hist1_out != nullptr) {      // one exec-mask region per batch, not per row
#pragma unroll
          for (int i = 0; i < RB; ++i)
            *reinterpret_cast<f32x4*>(bp.hist1_out + cs.hoff + (rsub + RSTEP * (i0 + i)) * 4) = cur[i];
        }
      }
      f32x4 o[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const float v[8] = {prev[i].x, prev[i].y, prev[i].z, prev[i].w, cur[i].x, cur[i].y, cur[i].z, cur[i].w};
        const float w[5] = {wa[i].x, wa[i].y, wa[i].z, wa[i].w, wb[i].x};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 5; ++j) s = fmaf(w[j], v[e + j], s);
          o[i][e] = elu_fast(__fadd_rn(s, wb[i].y));
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i)
        *(lvec_t)(xp3 + i * RSTEP * XS) = o[i];
      xp3 += RB * RSTEP * XS;
      pb3 += RB * RSTEP * XS;
      asm volatile("" : "+v"(xp3), "+v"(pb3));
      __builtin_amdgcn_sched_barrier(0);   // batches stay batches: hoisting every row's reads costs RW * 14 registers
    }
    lds_barrier();
    STAMP(4);

    // ---- P4
    if constexpr (K::NARROW) gemm_phase_rolled<K>(w2t, X, acc, wp, colblk, lane);
    else gemm_phase<K>(w2t, X, acc, wp, colblk, lane);
    // L2 touch of the next tile's x rows: one dword per 128-B line, issued after the second GEMM, consumed by a
    // never-true test at the end of the kernel; the real loads of P6 then hit this XCD's L2
    have_next = next_tile < run1;
    if (last_blk) cn = columns_of(have_next ? next_tile : tile);
    if constexpr (!STREAM && UR == 0 && !SPEC0) {      // (UR > 0 / SPEC0: x is computed, not read — nothing to touch, and a.x has another shape)
      if (final_blk) {
      long nb;
      int nt0;
      if constexpr (STREAM) {   // the next tile's first clip only: a tile that straddles clips is touched in part
        const int nf0 = (int)(have_next ? next_tile : tile) * TO - K::HALO;
        const unsigned q = __umulhi((unsigned)(nf0 < 0 ? 0 : nf0), a.div_magic) >> a.div_shift;
        nb = q;
        nt0 = nf0 - (int)q * T;
      } else {
        const long nt = have_next ? next_tile : tile;
        nb = nt / a.tiles;
        nt0 = (int)(nt - nb * a.tiles) * TO;
      }
      const float* nx = a.x + nb * (long)C * T;
#pragma unroll
      for (int i = 0; i < NTOUCH; ++i) {
        int e = tid + NT * i;
        e = e < C * LPR ? e : C * LPR - 1;
        int tt = nt0 + (e % LPR) * 32;
        tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
        tv[i] = nx[(long)(e / LPR) * T + tt];
      }
      }
    }
    if constexpr (UTOUCH) {
      if (final_blk) {
        const long nt = have_next ? next_tile : tile;
        const long nb = nt / a.tiles;
        const int Tin = T / UR;
        const int nq0 = (int)(nt - nb * a.tiles) * TO / UR;            // the next tile's first input frame
        const float* nx = a.up.xin + nb * (long)(2 * C) * Tin;
#pragma unroll
        for (int i = 0; i < NTOUCH; ++i) {
          int e = tid + NT * i;
          e = e < 2 * C * LPRU ? e : 2 * C * LPRU - 1;
          int ff = nq0 + (e % LPRU) * 32;
          ff = ff > Tin - 1 ? Tin - 1 : ff;
          tv[i] = nx[(long)(e / LPRU) * Tin + ff];
        }
      }
    }
    // ---- P5
    lds_barrier();
    STAMP(5);
    acc_to_x<K>(acc, X, rowblk0, colblk, lane);
    lds_barrier();
    STAMP(6);

    // ---- P6: y = (dw2(H2) + b2) * out_scale + x  for t < T (not in a warm-up tile); then this batch's x registers take the
    //      next tile's rows
    const bool out_ok = !warm && (STREAM ? (c4 >= K::HALO && cs.t_in) : cs.t < T);
    lptr_t xp6 = (lptr_t)(X + rsub * XS + c4);
    lptr_t pb6 = prev_base(1);
#pragma unroll
    for (int i0 = 0; i0 < RW; i0 += RB) {
      f32x4 cur[RB], prev[RB], wc[RB];
      f32x2 wb[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int m = rsub + RSTEP * (i0 + i);
        const lptr_t row = xp6 + i * RSTEP * XS;
        cur[i] = *(lvec_t)(row);
        wb[i] = *reinterpret_cast<const f32x2*>(&DWb[m * DWS + 6]);   // w2_0, w2_1
        wc[i] = *reinterpret_cast<const f32x4*>(&DWb[m * DWS + 8]);   // w2_2, w2_3, w2_4, b2
      }
      prevs_of(pb6, rsub + RSTEP * i0, 1, prev);
      carry_out(xp6, 1, cur);
      if constexpr (STREAM) {
        if (cs.tail && !warm && bp.hist2_out != nullptr) {
#pragma unroll
          for (int i = 0; i < RB; ++i)
            *reinterpret_cast<f32x4*>(bp.hist2_out + cs.hoff + (rsub + RSTEP * (i0 + i)) * 4) = cur[i];
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int m = rsub + RSTEP * (i0 + i);
        const float v[8] = {prev[i].x, prev[i].y, prev[i].z, prev[i].w, cur[i].x, cur[i].y, cur[i].z, cur[i].w};
        const float w[5] = {wb[i].x, wb[i].y, wc[i].x, wc[i].y, wc[i].z};
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 5; ++j) s = fmaf(w[j], v[e + j], s);
          s = __fmul_rn(__fadd_rn(s, wc[i].w), bp.out_scale);
          o[e] = __fadd_rn(s, xr[i0 + i][e]);
        }
        if constexpr (NB == 1 && DR == 0 && !POST) {
          if (out_ok) *yrow(cs, m) = o;
        } else {
          if (last_blk) {
            if (out_ok) *yrow(cs, m) = o;
          } else if (post_blk) {
            // the closing conv's operand, in place (this batch's rows were read above; a row is one wave instruction: read before write)
            *(lvec_t)(xp6 + i * RSTEP * XS) = prologue4v(o, a.post.in_scale, 1);
          } else {
            xr[i0 + i] = o;                // the next block of the chain: its input and its shortcut (P0 zeroes what lies outside [0, T))
          }
        }
      }
      if (UR == 0 && !SPEC0 && last_blk && have_next) {
#pragma unroll
        for (int i = 0; i < RB; ++i) xr[i0 + i] = *xrow(cn, rsub + RSTEP * (i0 + i));
      }
      xp6 += RB * RSTEP * XS;
      pb6 += RB * RSTEP * XS;
      asm volatile("" : "+v"(xp6), "+v"(pb6));
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(7);
    lds_barrier();   // the next block's / tile's P0 overwrites X
    if (last_blk) cs = cn;
    }                // blocks of the chain
    if constexpr (POST) {
      // ---- Q: wav = post(sum_c sum_j w[c][j] * a[c][t - 4 + j] + bias), a = ELU(in_scale * y) left in the tile by the last block's P6
      //      (its closing barrier is behind us).  Row class q = the rows q, q + 8, ... one half-wave walks: an fmaf chain over (row, tap)
      //      ascending per class and column, then the classes in ascending order — hilc_conv_post's order.
      const bool have_next = next_tile < run1;
      const Cols cn = columns_of(have_next ? next_tile : tile);
      constexpr int PSLOT = K::NCOL + 8 * NB;                // the conv's carry slot behind a row: a's last 4 columns of the previous tile
      float part[4] = {0.f, 0.f, 0.f, 0.f};
      lptr_t xq = (lptr_t)(X + rsub * XS + c4);
      lptr_t pq = lane0 ? (lptr_t)(X + rsub * XS + PSLOT) : (lptr_t)(X + rsub * XS + c4 - 4);
      const float* wq = a.post.w + rsub * 5;
      // STREAM: a stream's first group takes the 4 columns in front of t = 0 from the conv's cache [B][C][4] (activated samples; zeros
      // without one), its last group leaves the next hop's.  Only tiles that hold a stream's t = 0 pay the cache loads (uniform test).
      [[maybe_unused]] bool any_head = false;
      [[maybe_unused]] const float* hq = a.post.w;
      [[maybe_unused]] float* hqo = nullptr;
      [[maybe_unused]] bool qtail = false;
      if constexpr (STREAM) {
        const int first = (int)tile * TO, last = first + K::NCOL - 1;
        const unsigned bl = __umulhi((unsigned)last, a.div_magic) >> a.div_shift;      // stream of the tile's last column
        any_head = (int)bl * T >= first || first % T == 0;                              // a stream starts inside this tile
        if (a.post.hist != nullptr) hq = a.post.hist + cs.hoff + rsub * 4;             // (hoff = 0 outside the tensor: a valid address for every lane)
        qtail = cs.tail && !warm && a.post.hist_out != nullptr;
        if (a.post.hist_out != nullptr) hqo = a.post.hist_out + cs.hoff + rsub * 4;
      }
      // (rolled: unrolled, hipcc hoists every batch's tap loads to the top of the phase — 60 registers, spills)
#pragma nounroll
      for (int i0 = 0; i0 < RW; i0 += RB) {
        f32x4 cur[RB], prev[RB];
        float w[RB][5];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          cur[i] = *(lvec_t)(xq + i * RSTEP * XS);
          prev[i] = *(lvec_t)(pq + i * RSTEP * XS);
#pragma unroll
          for (int j = 0; j < 5; ++j) w[i][j] = wq[i * RSTEP * 5 + j];
        }
        wq += RB * RSTEP * 5;
        if constexpr (STREAM) {
          if (any_head) {
            f32x4 hv[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i)
              hv[i] = a.post.hist != nullptr ? *reinterpret_cast<const f32x4*>(hq + i * RSTEP * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
              for (int e = 0; e < 4; ++e) prev[i][e] = cs.head ? hv[i][e] : prev[i][e];
          }
          if (qtail) {
#pragma unroll
            for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(hqo + i * RSTEP * 4) = cur[i];
          }
          hq += a.post.hist != nullptr ? RB * RSTEP * 4 : 0;
          hqo += RB * RSTEP * 4;
        }
        if (lane_last) {
#pragma unroll
          for (int i = 0; i < RB; ++i) *(lvec_t)(xq + i * RSTEP * XS + 4 + 8 * NB) = cur[i];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const float v[8] = {prev[i].x, prev[i].y, prev[i].z, prev[i].w, cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 5; ++j) part[e] = fmaf(w[i][j], v[e + j], part[e]);
        }
        xq += RB * RSTEP * XS;
        pq += RB * RSTEP * XS;
        asm volatile("" : "+v"(xq), "+v"(pq));
        __builtin_amdgcn_sched_barrier(0);
      }
      *reinterpret_cast<f32x4*>(&PR[rsub * K::NCOL + c4]) = f32x4{part[0], part[1], part[2], part[3]};
      lds_barrier();       // partial sums visible; every wave is done reading the tile (the next tile's first phase writes it)
      if (tid < K::NCOL) {
        float sacc = PR[tid];
#pragma unroll
        for (int q = 1; q < POST_CLASSES; ++q) sacc = __fadd_rn(sacc, PR[q * K::NCOL + tid]);
        if (a.post.bias != nullptr) sacc = __fadd_rn(sacc, a.post.bias[0]);
        sacc = __fmul_rn(sacc, a.post.out_scale);
        if (a.post.do_tanh) sacc = tanhf(sacc);
        if constexpr (STREAM) {
          const long flat = (long)tile * TO + tid;               // [B][1][T] is the flat column space itself
          if (!warm && flat < (long)a.B * T) a.post.wav[flat] = sacc;
        } else {
          const int t = cs.t - c4 + tid;                         // offline: cs.t = the tile's first column + c4
          if (!warm && t < T) a.post.wav[cs.b * (long)T + t] = sacc;
        }
      }
      cs = cn;             // (PR is next written seven barriers from here)
    }
    if constexpr (DR > 0) {
      // ---- D: the stage's down-sampling layer on the tile the last block left in the x registers
      const ResDown& dn = a.dn;
      const bool have_next = next_tile < run1;
      const Cols cn = columns_of(have_next ? next_tile : tile);
      const float* wd0 = dn.w_lo + (long)wclass * (C * C / K::RH);
      const float* wd1 = dn.w_hi + (long)wclass * (C * C / K::RH);
      asm volatile("" : "+s"(wd0), "+s"(wd1));
      Pipe wp;
      wp.prefetch(wd0, lane);
      auto d0 = [&]() {                            // D0: ELU(in_scale * y) -> LDS (P0 on the stage's output)
        lptr_t xp = (lptr_t)(X + rsub * XS + c4);
#pragma unroll
        for (int i0 = 0; i0 < RW; i0 += RB) {
#pragma unroll
          for (int i = 0; i < RB; ++i)
            *(lvec_t)(xp + i * RSTEP * XS) = prologue4v(zero_unless(cs.t_in, xr[i0 + i]), dn.in_scale, 1);
          xp += RB * RSTEP * XS;
          asm volatile("" : "+v"(xp));
        }
      };
      d0();
      if (!DWIDE && !SPEC0 && have_next) {         // the x registers are free (no shortcut here): the next tile's rows travel under both GEMMs
#pragma unroll
        for (int i = 0; i < RW; ++i) xr[i] = *xrow(cn, rsub + RSTEP * i);
      }
      lds_barrier();
      // C <= 192: both halves' accumulators live (one operand tile, two GEMMs back to back).  The wide stages have no registers for
      // that beside the strided conv's windows: one half at a time — GEMM, conv, the operand tile written AGAIN from the x registers
      // (one more ELU pass: ~1 % of the two GEMMs), GEMM, conv; the next tile's rows then travel under the second GEMM.
      f32x16 acc0[CBW];
      [[maybe_unused]] f32x16 acc1[DWIDE ? 1 : CBW];
      if constexpr (K::NARROW) gemm_phase_rolled<K>(wd0, X, acc0, wp, colblk, lane);
      else gemm_phase<K>(wd0, X, acc0, wp, colblk, lane);
      if constexpr (!DWIDE) {
        wp.prefetch(wd1, lane);
        if constexpr (K::NARROW) gemm_phase_rolled<K>(wd1, X, acc1, wp, colblk, lane);
        else gemm_phase<K>(wd1, X, acc1, wp, colblk, lane);
      }
      const bool out_ok = !warm && (STREAM ? cs.t_in : cs.t < T);
      [[maybe_unused]] const bool out_ok_tile = !warm;
      const int To = T / DR;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if constexpr (DWIDE) {
          if (h == 1) {
            lds_barrier();                         // the first half's conv has read the tile
            wp.prefetch(wd1, lane);
            d0();
            if (have_next) {
#pragma unroll
              for (int i = 0; i < RW; ++i) xr[i] = *xrow(cn, rsub + RSTEP * i);
            }
            lds_barrier();
            gemm_phase_rolled<K>(wd1, X, acc0, wp, colblk, lane);
          }
        }
        lds_barrier();
        if constexpr (DWIDE) acc_to_x<K>(acc0, X, rowblk0, colblk, lane);
        else acc_to_x<K>(h ? acc1 : acc0, X, rowblk0, colblk, lane);
        lds_barrier();
        if constexpr (DWIDE) {
          // The wide stages (r = 5 on 64-column tiles, r = 8 on 32-column tiles): outputs do not align with a lane's 4 columns.  Lane
          // j of a row's lanes computes the row's j-th output of this tile — output o reads the 2r columns [r o - r, r o + r) and
          // belongs to the tile that holds its LAST column — from the tile and, left of its first column, from the half's carry
          // slot (the previous tile's last DCAR columns; zeros at a clip's start = the conv's zero padding).  Taps and bias come
          // from global memory (the same few rows for every workgroup: L1 / L2), k ascending like DwStrideEpilogue: bit-identical.
          // STREAM (round 6: the hop's C = 256 stage on 32-column carry tiles, its C = 512 stage on whole-stream tiles): the same walk on the FLAT
          // column space — T % r == 0, so stream b's output o is the flat output b * To + o and reads the flat columns [r O - r, r O + r) —
          // except that the r samples in front of a stream's t = 0 are its CACHE `hist[b][2C][r]` (the previous hop's last r pointwise
          // outputs; zeros without one), not the previous stream's columns; the lane that holds a stream's last group stores the next one.
          constexpr int DC = K::DCAR;
          constexpr int DOFF = K::NCOL + 8 * NB;             // the two halves' carry slots behind a row (carry form)
          constexpr bool SLOTS = K::CARRYMODE;               // (whole-stream tiles: no slots — whatever lies left of column 0 is a cache)
          const int jl = c4 >> 2;
          const int t0 = STREAM ? (int)tile * TO : cs.t - c4;   // the tile's first column (offline: cs.t = t0 + c4 for every lane; STREAM: flat)
          const int o = t0 / DR + jl;                        // this lane's output (offline: of the clip; STREAM: flat)
          const int base = o * DR - DR - t0;                 // local column of the window's first sample: >= -(2 r - 1)
          [[maybe_unused]] unsigned sb = 0;                  // STREAM: the output's stream, and its index inside it
          int os = o;
          if constexpr (STREAM) {
            const bool in = o < a.B * To;
            sb = in ? __umulhi((unsigned)(o * DR), a.div_magic) >> a.div_shift : 0u;
            os = in ? o - (int)sb * To : 1;
          }
          [[maybe_unused]] const bool dhead = STREAM && os == 0;
          const bool act = out_ok_tile && (STREAM ? o < a.B * To : o < To) && o * DR + DR - 1 < t0 + K::NCOL;
          [[maybe_unused]] const int slot = DOFF + DC * h + DC;               // column -k of the tile lives at slot - k
          // one row per (rolled) iteration — unrolled, hipcc keeps every row's addresses and taps alive (31 spilled registers at C = 512) —
          // with the NEXT row's taps and bias (STREAM: and cache words) requested before this row's arithmetic
          lptr_t rowp = (lptr_t)(X + rsub * XS);
          const float* wrow = dn.dw_w + (long)(h * C + rsub) * (2 * DR);
          const float* brow = dn.dw_b + (h * C + rsub);
          float wn[2 * DR], bn;
          [[maybe_unused]] float hn[STREAM ? DR : 1];
          [[maybe_unused]] const float* hrow = dn.dw_w;      // STREAM: this lane's stream's cache row (a valid address for every lane)
          [[maybe_unused]] float* horow = nullptr;           // STREAM: where this lane's group — a stream's last — leaves the next cache
          [[maybe_unused]] const bool use_hist = STREAM && dn.hist != nullptr;
          if constexpr (STREAM) {
            if (use_hist) hrow = dn.hist + ((long)sb * (2 * C) + h * C + rsub) * DR;
            if (dn.hist_out != nullptr) horow = dn.hist_out + ((long)cs.b * (2 * C) + h * C + rsub) * DR;
          }
          [[maybe_unused]] const bool dtail = STREAM && cs.tail && !warm && dn.hist_out != nullptr;
          auto taps_of = [&](const float* wr, const float* br) {
            if constexpr (DR == 8) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) wn[4 * q + e] = w4[e];
              }
            } else {
#pragma unroll
              for (int q = 0; q < DR; ++q) {
                const f32x2 w2 = *reinterpret_cast<const f32x2*>(wr + 2 * q);
                wn[2 * q] = w2.x; wn[2 * q + 1] = w2.y;
              }
            }
            bn = br[0];
            if constexpr (STREAM) {
              if (use_hist) {                                 // uniform
                if constexpr (DR == 8) {
#pragma unroll
                  for (int q = 0; q < 2; ++q) {
                    const f32x4 h4 = *reinterpret_cast<const f32x4*>(hrow + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hn[4 * q + e] = h4[e];
                  }
                } else {
#pragma unroll
                  for (int j = 0; j < DR; ++j) hn[j] = hrow[j];
                }
              } else {
#pragma unroll
                for (int j = 0; j < DR; ++j) hn[j] = 0.f;
              }
            }
          };
          taps_of(wrow, brow);
          long yo = STREAM ? ((long)sb * (2 * C) + h * C + rsub) * (long)To + os : ((long)cs.b * (2 * C) + h * C + rsub) * (long)To + o;
#pragma nounroll
          for (int i0 = 0; i0 < RW; ++i0) {
            float v[2 * DR], w[2 * DR];
#pragma unroll
            for (int j = 0; j < 2 * DR; ++j) w[j] = wn[j];
            const float bb = bn;
            [[maybe_unused]] float hc[STREAM ? DR : 1];
            if constexpr (STREAM) {
#pragma unroll
              for (int j = 0; j < DR; ++j) hc[j] = hn[j];
            }
            wrow += RSTEP * (2 * DR);
            brow += RSTEP;
            if constexpr (STREAM) hrow += use_hist ? RSTEP * DR : 0;
            if (i0 + 1 < RW) taps_of(wrow, brow);
            if constexpr (DR == 8) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int col = base + 4 * q;
                const f32x4 x4 = *(lvec_t)(rowp + (col >= 0 ? col : (SLOTS ? slot + col : 0)));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = x4[e];
              }
            } else {
#pragma unroll
              for (int j = 0; j < 2 * DR; ++j) {
                const int col = base + j;
                v[j] = *(rowp + (col >= 0 ? col : (SLOTS ? slot + col : 0)));
              }
            }
            if constexpr (STREAM) {
#pragma unroll
              for (int j = 0; j < DR; ++j) v[j] = dhead ? hc[j] : v[j];
            }
            const f32x4 keep = *(lvec_t)(rowp + c4);          // this lane's own group: the tile's last DCAR columns become the carry
            if constexpr (STREAM) {
              // a stream's last group (T % 8 == 0: it never sits on column 0, and the r columns end inside this tile): the next hop's cache
              const f32x4 before = *(lvec_t)(rowp + (c4 >= 4 ? c4 - 4 : 0));
              if (dtail) {
                if constexpr (DR == 8) {
                  *reinterpret_cast<f32x4*>(horow) = before;
                  *reinterpret_cast<f32x4*>(horow + 4) = keep;
                } else {
                  horow[0] = before.w; horow[1] = keep.x; horow[2] = keep.y; horow[3] = keep.z; horow[4] = keep.w;
                }
              }
              horow += RSTEP * DR;
            }
            if constexpr (SLOTS) {
              if (c4 >= K::NCOL - DC)                          // (after every read of the slot: one wave instruction stream per row)
                *(lvec_t)(rowp + DOFF + DC * h + (c4 - (K::NCOL - DC))) = keep;
            }
            float s0 = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * DR; ++j) s0 = fmaf(w[j], v[j], s0);
            s0 = __fadd_rn(s0, bb);
            if (act) dn.y[yo] = dn.res != nullptr ? __fadd_rn(s0, dn.res[yo]) : s0;
            yo += (long)RSTEP * To;
            rowp += RSTEP * XS;
          }
        } else {
        // strided depthwise conv of the half's C rows: output (t / r) of row m reads columns [t - r, t + r) — the lane's own group and
        // the one in front of it (left neighbour, the half's carry slot, or the stream's cache), like P3
        lptr_t xpd = (lptr_t)(X + rsub * XS + c4);
        lptr_t pbd = lane0 ? (lptr_t)(X + rsub * XS + K::NCOL + 8 * NB + 4 * h) : (lptr_t)(X + rsub * XS + c4 - 4);
#pragma unroll
        for (int i0 = 0; i0 < RW; i0 += RB) {
          f32x4 cur[RB], prev[RB], wa[RB], wb[RB];
          float bb[RB];
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int m = h * C + rsub + RSTEP * (i0 + i);
            cur[i] = *(lvec_t)(xpd + i * RSTEP * XS);
            prev[i] = *(lvec_t)(pbd + i * RSTEP * XS);
            wa[i] = *reinterpret_cast<const f32x4*>(&DWD[m * DDS]);
            if constexpr (DR == 4) wb[i] = *reinterpret_cast<const f32x4*>(&DWD[m * DDS + 4]);
            bb[i] = DWD[m * DDS + 8];
          }
          if constexpr (STREAM) {
            // a stream's first group: the r samples in front of t = 0 are its cache (zeros without one)
            const unsigned hro = (unsigned)cs.b * (unsigned)(2 * C * DR) + (unsigned)(h * C + rsub + RSTEP * i0) * DR;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
              f32x4 hv = {0.f, 0.f, 0.f, 0.f};
              if (dn.hist != nullptr) {
                const float* hp = dn.hist + (cs.t_in ? hro + (unsigned)(RSTEP * i * DR) : 0u);
                if constexpr (DR == 4) hv = *reinterpret_cast<const f32x4*>(hp);
                else { const f32x2 h2 = *reinterpret_cast<const f32x2*>(hp); hv = f32x4{0.f, 0.f, h2.x, h2.y}; }
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) prev[i][e] = cs.head ? hv[e] : prev[i][e];
            }
          }
          if (lane_last) {                           // the half's last 4 columns: carry for the run's next tile
#pragma unroll
            for (int i = 0; i < RB; ++i) *(lvec_t)(xpd + i * RSTEP * XS + 4 + 8 * NB + 4 * h) = cur[i];
          }
          if constexpr (STREAM) {
            if (cs.tail && !warm && dn.hist_out != nullptr) {       // the stream's last r pointwise outputs: the next hop's cache
              const unsigned hro = (unsigned)cs.b * (unsigned)(2 * C * DR) + (unsigned)(h * C + rsub + RSTEP * i0) * DR;
#pragma unroll
              for (int i = 0; i < RB; ++i) {
                float* hp = dn.hist_out + hro + (unsigned)(RSTEP * i * DR);
                if constexpr (DR == 4) *reinterpret_cast<f32x4*>(hp) = cur[i];
                else *reinterpret_cast<f32x2*>(hp) = f32x2{cur[i].z, cur[i].w};
              }
            }
          }
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int m = h * C + rsub + RSTEP * (i0 + i);
            const long yo = ((long)cs.b * (2 * C) + m) * (long)To + (cs.t_in ? cs.t / DR : 0);
            if constexpr (DR == 4) {
              const float v[8] = {prev[i].x, prev[i].y, prev[i].z, prev[i].w, cur[i].x, cur[i].y, cur[i].z, cur[i].w};
              const float w[8] = {wa[i].x, wa[i].y, wa[i].z, wa[i].w, wb[i].x, wb[i].y, wb[i].z, wb[i].w};
              float s0 = 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) s0 = fmaf(w[j], v[j], s0);
              s0 = __fadd_rn(s0, bb[i]);
              if (out_ok) dn.y[yo] = dn.res != nullptr ? __fadd_rn(s0, dn.res[yo]) : s0;
            } else {
              const float v[6] = {prev[i].z, prev[i].w, cur[i].x, cur[i].y, cur[i].z, cur[i].w};
              const float w[4] = {wa[i].x, wa[i].y, wa[i].z, wa[i].w};
              f32x2 o2;
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                float s0 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) s0 = fmaf(w[j], v[2 * e + j], s0);
                o2[e] = __fadd_rn(s0, bb[i]);
              }
              if (out_ok) {
                if (dn.res != nullptr) {
                  const f32x2 r2 = *reinterpret_cast<const f32x2*>(dn.res + yo);
                  o2[0] = __fadd_rn(o2[0], r2.x);
                  o2[1] = __fadd_rn(o2[1], r2.y);
                }
                *reinterpret_cast<f32x2*>(dn.y + yo) = o2;
              }
            }
          }
          xpd += RB * RSTEP * XS;
          pbd += RB * RSTEP * XS;
          asm volatile("" : "+v"(xpd), "+v"(pbd));
          __builtin_amdgcn_sched_barrier(0);
        }
        }                // narrow / wide strided conv
      }
      lds_barrier();   // the next tile's P0 overwrites X
      cs = cn;
    }
    tile = next_tile;
  }
  if (!K::CARRYMODE && K::NW == 4 && a.sched != nullptr && tid == 0) {   // last workgroup out re-arms the scheduler for the next launch
    if (atomicAdd(a.sched + 1, 1) == (int)gridDim.x - 1) {
      a.sched[0] = 0;
      a.sched[1] = 0;
    }
  }
  if (touch == 1.2345678e-30f) a.y[0] = touch;   // keeps the touch loads alive; never true in practice
#if defined(HILC_DEBUG_WAVE_STAMPS)
  if (ws_base != nullptr && lane == 0) ws_base[(WS_MAXB - 1) * 3 + 1] = (unsigned long long)ws_k;
#undef lds_barrier
#endif
#undef STAMP
}

}  // namespace
