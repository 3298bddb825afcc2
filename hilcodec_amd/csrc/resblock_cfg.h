#pragma once
// resblock_cfg.h — shape of a launch (Cfg), its arguments (ResArgs) and the phase map of the fused residual-block / stage kernel.
// Part 1 of 4: resblock_cfg.h -> resblock_gemm.h (weight stream + GEMM phases) -> resblock_kernel.h (the kernel) -> resblock_launch.h
// (grids, run shares, launchers); resblock.hip (one block per launch) and resblock_chain.hip (whole stages) include the last.
//
// Fully fused SEANet residual block for the narrow, long layers (C <= 192), gfx950.
//
//   y = x + out_scale * dw2( pw2( ELU( dw1( pw1( ELU(pre_scale * x) ) ) + b1 ) ) ) + b2 )      (seanet.py:129-148)
//
// For C in {64, 96} the un-fused block is HBM-bound (1x1 conv intensity = C/4 flop/B): this kernel reads x ONCE and
// writes y once; everything between lives in registers and one LDS tile.
//
// Workgroup = NW waves walking a CONTIGUOUS run of tiles; a tile = NCOL columns of one clip (see Cfg), column c <-> time
// t0 + c with t0 = NCOL * (tile within the clip).  The two causal k = 5 convs need the 4 columns in front of a tile of H1 and
// of H2: the previous tile of the run leaves them in LDS (CARRY), a clip's first tile takes zeros (streaming: the caches).  A run
// that starts inside a clip walks the tile in front of it once without storing anything (warm-up).  No column is computed twice
// (rounds 1-3 recomputed an 8-column halo per tile: 6.25 % of the GEMM work).
//   P0  a1 = ELU(pre*x) from the x REGISTERS (loaded during the previous tile's P6)  -> LDS  X[k][c]
//   P1  GEMM1  H1 = W1 * a1   (fp32 MFMA 32x32x2; a wave owns one 32-column block and CBW row blocks)
//   P2  accumulators -> LDS  X[m][c]
//   P3  a2 = ELU(dw1(H1)+b1) in place (a row is handled by one wave instruction, so read-before-write holds
//       without a barrier, for the tile and for the carry)
//   P4  GEMM2  H2 = W2 * a2
//   P5  accumulators -> LDS
//   P6  y = (dw2(H2)+b2)*out_scale + x (the shortcut comes from the x registers: no re-read) -> HBM; as soon as a
//       row batch is stored its x registers are re-loaded with the NEXT tile's rows, so the loads travel under the
//       rest of P6 / the barrier and P0 never waits for HBM (the next tile's lines were touched into this XCD's L2
//       during GEMM2).
// Cost model behind this shape (profiles/r02_mfma_shadow_microbench.txt): next to fp32 MFMAs (64 cycles each) LDS
// traffic and sparse global loads are free, a VALU instruction costs ~2.8 cycles of the same pipe (4.9 when a SIMD
// hosts a single wave: one wave cannot issue VALU back to back), v_exp_f32 8.4.  Hence: (i) the element-wise phases
// process rows in BATCHES — all LDS reads of a batch, then the arithmetic, then the writes; the in-place row
// update used to serialise on one exposed LDS round trip per row (P3 / P6 ran at a third of their VALU rate);
// (ii) every LDS address is base + compile-time constant (a select in an address hides the no-alias fact from the
// scheduler); (iii) C = 192, whose 96 KB tile allows one workgroup per CU, runs 8 waves (two per SIMD, each owning
// half of the row blocks) so that the VALU phases issue at full rate.
// A operands (weights) are read straight from global memory (L1/L2 resident, a few tens of KB) into VGPRs one
// 16-deep K slice ahead — no LDS staging, no barrier in the K loop.
// Summation order: k ascending (an fmaf chain), taps j = 0..4 — the same as the un-fused kernels.
//
// STREAM instantiation (hilc_resblock_stream; streaming.py:195-276 with causal_layers.py:147-167 caches): the
// tile walks the FLAT column space (clip-major, b*T + t) so short hops (T = 160 / 320 per stream) still fill
// 120 of 128 columns; the 4 samples before a clip's t = 0 come from the caches hist1 / hist2 (= last 4
// pointwise outputs of the previous hop) instead of the LDS neighbours, and the lanes holding t = T-4..T-1
// store the new caches.  Per-column arithmetic is identical, so hop-by-hop output == offline output bit for bit.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "gemm_core.h"

using namespace hilc;

namespace {

constexpr int DWS = 12;   // per-row depthwise table in LDS: [w1_0..3 | w1_4, b1, w2_0, w2_1 | w2_2..4, b2]

// Shape of a workgroup: 128-column tiles for C <= 192 (8 waves at C = 192 and where W8_ says so, else 4: two or three workgroups per
// CU), the NARROW shapes for C >= 256.  (A 256-column lockstep shape with one 8-wave workgroup per CU was measured three times,
// rounds 2-4: 3-6 % slower at every width; tools/ history.)
// run shares of the dispatch classes of the offline carry form (two / three workgroups per CU)
// (tools/share_sweep2.sh on the -DHILC_RES_SHARE_ENV build: two classes 0.50 -> 2.47 / 2.02 ms at C = 96 / 128, 0.62-0.65 -> 2.40 /
// 1.93, 0.71 -> 2.44 / 1.99; three classes (C = 64) 1/3 each -> 1.49 ms, 0.44 / 0.31 / 0.25 -> 1.46)
// (round 4, stage launches, a box holding 2.39 GHz at 1.2 kW, tools/share_sweep_chain.sh: C = 96 stage 0.50 -> 9.16 ms, 0.56 -> 8.89, 0.60 - 0.62 -> 8.70,
// 0.64 -> 8.84, 0.68 -> 9.10, 0.72 -> 9.43; C = 64 stage 3.44 / 3.39 / 3.30 / 3.38 / 3.42 / 3.54)
#ifndef HILC_RES_SHARE2_0
#define HILC_RES_SHARE2_0 0.61
#endif
#ifndef HILC_RES_SHARE3_0
#define HILC_RES_SHARE3_0 0.44
#define HILC_RES_SHARE3_1 0.31
#endif
// NB_ > 1: a CHAIN — the NB_ consecutive residual blocks of one stage (seanet.py:316-330 / streaming.py: `blocks[s]`) in ONE
// launch: per tile the blocks run back to back, the output of block j stays in the x REGISTERS as the input (and shortcut) of
// block j + 1, only the last block stores; every block has its own carry slots, tap table, weights and (STREAM) caches.  Needs
// the carry form (contiguous runs) or whole-stream tiles (NARROW, C >= 512).  W8_: 8 waves also below C = 192 (one workgroup
// per CU: a streaming hop of 1024 streams is then 256 equal runs of 4 whole streams — no partly filled round).
// DR_ > 0: the stage's DOWN-SAMPLING layer (seanet.py:330-339: [Scale, ELU, 1x1 conv C -> 2C, depthwise k = 2r stride r]) as the
// last phase of the launch ("D"): the stage's output never reaches HBM — it goes from the last block's registers through ELU into
// the LDS tile, two GEMMs (the two halves of the 2C output rows) and the strided depthwise conv, which reads the tile like P3 does
// (previous 4 columns + own 4 columns per lane) with its own two carry slots.  DR_ = r in {2, 4}: the encoder's first two stages.
// POST_: the decoder's LAST layer (seanet.py:453-476: [Scale, ELU, conv k = 5 C -> 1 with bias] and the final out_scale / tanh, = hilc_conv_post)
// as the closing phase of the launch ("Q"): the last block leaves ELU(in_scale * y) in the LDS tile instead of storing y, every lane
// accumulates its rows' taps over its 4 columns (previous columns: left neighbour or a third carry slot), the row classes' partial sums
// meet in LDS in a fixed order — the order hilc_conv_post uses, so the two forms agree bit for bit — and 128 threads store the waveform:
// the stage's [B][C][T] output (2.36 GB at 256 clips) is neither written nor read.  Carry form, C = 96: offline, and (round 6) a streaming hop's runs
// of whole streams — there a stream's first column group takes the conv's cache (the previous hop's last 4 ACTIVATED columns) and its last one leaves the next.
// SPEC0_: the encoder's FIRST stage with its input computed in the launch ("S" phase, seanet.py:280-286, 220-246, 368-372): per tile the first
// conv (k = 5, 1 -> 64) and the stage's SpecBlock (STFT n_fft = 64 hop 1 -> log-magnitude -> 1x1 conv) of the waveform segment —
// hilc_spec_block_conv_pre's arithmetic — produce the x registers; the [B][64][T] tensor between that launch and the stage never exists.
// STREAM (round 6, runs of whole streams, T >= 128): a tile holds at most one stream start — its waveform segment is then two pieces, each
// with its own 63 samples of history (the stream's cache in front of t = 0).
template <int C, bool STREAM, bool SCARRY_ = false, int NB_ = 1, bool W8_ = false, int DR_ = 0, bool POST_ = false, bool SPEC0_ = false>
struct Cfg {
  static constexpr bool POST = POST_;
  static constexpr bool SPEC0 = SPEC0_;
  static_assert(!SPEC0_ || (C == 64 && (!STREAM || SCARRY_) && !W8_ && DR_ >= 0), "stage-0 input phase: the C = 64 encoder stage in the carry form (four waves, 128-column tiles)");
  static_assert(!POST_ || ((!STREAM || SCARRY_) && DR_ <= 0 && C <= 192 && !W8_), "closing conv: the carry form of a narrow stage (offline, or a hop's runs of whole streams)");
  static constexpr int NB = NB_;
  static constexpr int DR = DR_ > 0 ? DR_ : 0;
  // DR_ < 0: the stage's UP-SAMPLING layer (seanet.py:431-436: [Scale, ELU, depthwise transposed conv k = 2r stride r, 1x1 conv 2C -> C
  // with bias]) as the FIRST phase of the launch ("U", r = -DR_): the tile's x is not read but computed — the up-sampled operand of
  // the 2C rows is built in the LDS tile one half (C rows) at a time from the input frames, their cache and the 2r taps (two FMAs per
  // element, like the loader of hilc_up_conv), two GEMMs accumulate over the 2C rows in k order, + bias -> the x registers.  The
  // [B][C][T] tensor between the up-sampling layer and the first block never exists.  Whole-stream tiles only (NARROW, C >= 512).
  static constexpr int UR = DR_ < 0 ? -DR_ : 0;
  static_assert(DR_ <= 0 || (((DR_ == 2 || DR_ == 4) && (!STREAM || SCARRY_) && C <= 192) || ((!STREAM || SCARRY_) && DR_ == 5 && C == 256) || (!SCARRY_ && DR_ == 8 && C == 512)),
                "down-sampling phase: carry form, r = 2 / 4 (C <= 192) or the wide encoder stages (C = 256: r = 5, offline and the hop's 32-column carry form; C = 512: r = 8, offline and a hop's whole-stream tiles)");
  // carry columns of the down-sampling phase per half of its 2C rows: the strided conv reads k - r = r columns in front of its first
  // output's window; r = 2 / 4: the 4 in front of a lane's group; r = 8: 8; r = 5: up to 9 (a tile does not start on a multiple of 5) -> 12
  static constexpr int DCAR = DR_ <= 0 ? 0 : (DR_ == 5 ? 12 : (DR_ == 8 ? 8 : 4));
  static_assert(DR_ >= 0 || ((DR_ == -8 && C >= 512) || (DR_ == -5 && C == 384) || ((DR_ == -4 || DR_ == -2) && C <= 192 && (!STREAM || SCARRY_))),
                "up-sampling phase: 32-column tiles (r = 8: whole streams, or the offline carry form), the C = 384 stage (r = 5) or the carry form (r = 4 / 2)");
  static constexpr int CH = C;
  static constexpr int CB = C / 32;
  // NARROW (C >= 256: the wide blocks — of a streaming hop, 8 or 40 frames per stream, and since round 4 of the offline model): the
  // whole channel range of a 32- or 64-column tile in LDS, the eight waves split the ROW blocks (RH = 8 or 4 row classes).
  // STREAM: at 32 columns a tile is whole streams (T divides 32): every tile starts at a stream's t = 0, where the caches supply the
  // previous samples, so there is no halo to recompute; 64-column tiles walk the flat column space with an 8-column halo.
  // Offline: the carry form, like every other width — one workgroup per CU walks a contiguous run of a clip's tiles.
  static constexpr bool NARROW = C >= 256;
  // N32 (round 6): the STREAMING carry form of the two wide stages whose streams are 40 columns per frame of a hop (C = 256 encoder, C = 384
  // decoder).  64-column flat tiles with a halo could neither carry nor chain (a run of whole streams = 8 streams = 128 runs for 1 024
  // streams: half the chip), so those stages were 3 + 4 launches at 67 - 100 TF.  On 32-column tiles a run of whole streams is 5 tiles =
  // 4 streams: 1 024 streams = exactly one run per CU, no halo, no partly filled round, the whole stage — blocks and its down- / up-sampling
  // layer — one launch.  Every wave owns ONE 32-row block (C = 256: 8 waves, C = 384: 12 waves = three per SIMD).
  static constexpr bool N32 = STREAM && SCARRY_ && NARROW;
  static_assert(!N32 || C == 256 || C == 384, "32-column carry form: the C = 256 / C = 384 stages of a hop");
  static constexpr int NCOL = NARROW ? ((C >= 512 || N32) ? 32 : 64) : 128;      // tile width = LDS row stride (floats)
  static constexpr int XS = NCOL + ((!STREAM || SCARRY_) ? 8 * NB_ + 2 * DCAR + (POST_ ? 4 : 0) : 0);   // LDS row stride: the tile's columns (+ carry form: two 4-float slots, H1 and H2)
  // Offline (CARRYMODE): no halo — a workgroup walks a CONTIGUOUS run of tiles and carries the last 4 columns of both pointwise
  // outputs from one tile to the next in LDS, exactly what the streaming caches do from hop to hop.  (Until round 3 every tile
  // recomputed an 8-column left halo of the two causal k = 5 convs: 6.25 % of a 128-column tile.)  STREAM keeps the halo and the
  // strided / ticketed tile order: a hop is 2.5-5 tiles per workgroup, runs would rarely start on a stream's t = 0 and each start
  // inside a stream costs a warm-up tile (measured: 5.34 -> 6.26 ms per hop with two stream groups).
  // SCARRY: the carry form for a STREAMING launch whose geometry lets every run start on a stream's t = 0 (launch_res decides:
  // 1024 streams x 160 samples = 256 runs of exactly 5 tiles = 4 whole streams each, five rounds of tiles instead of six).
  static constexpr bool CARRYMODE = !STREAM || SCARRY_;
  static constexpr int HALO = CARRYMODE ? 0 : ((NARROW && C >= 512) ? 0 : 8);   // left halo of two causal k=5 convs, recomputed per tile
  static constexpr int TO = NCOL - HALO;             // output samples per tile
  static constexpr int NW = (N32 && C == 384) ? 12 : ((C >= 192 || W8_) ? 8 : 4);           // waves per workgroup
  static constexpr int NT = 64 * NW;
  static constexpr int RH = NW / (NCOL / 32);        // row classes: waves w and w + NCOL/32 share a column block
  static constexpr int CBW = CB / RH;                // 32-row MFMA blocks per wave
  static constexpr int RPI = 256 / NCOL;             // rows covered by one wave instruction of the element-wise phases
  static constexpr int RSTEP = RPI * NW;
  static constexpr int RW = C / RSTEP;               // rows per lane there
#ifndef HILC_RES_RB
#define HILC_RES_RB 4
#endif
  static constexpr int RB = HILC_RES_RB;             // rows per batch there
  // weight stream: DEPTH register sets of KP k-pairs each; the loads run DEPTH-1 sets (= (DEPTH-1)*KP*CBW MFMAs per
  // wave, twice that in wall time with two waves per SIMD) ahead of their use
#ifdef HILC_RES_KP
  static constexpr int KP = HILC_RES_KP;
  static constexpr int DEPTH = HILC_RES_DEPTH;
#else
  // (round 4, a box holding 2.39 GHz at 1.2 kW — tools/build_variants.py + variant_table.sh, one box: KP = 4 / DEPTH = 2 instead of 4 / 4
  //  at C = 192: 14.84 -> 14.64 ms per offline stage, instead of 8 / 2 at C = 96: 9.00 -> 8.89; the wide shapes lose with it: C = 768 5.30 -> 5.39)
  static constexpr int KP = C >= 96 ? 4 : 8;
  static constexpr int DEPTH = C >= 256 ? 4 : (C == 128 ? (STREAM ? 2 : 3) : 2);   // (STREAM, C = 128: the cache handling needs the third set's 20 registers)
#endif
#ifndef HILC_RES_MINW
#define HILC_RES_MINW 2
#endif
  static constexpr int MINW = NW >= 8 ? NW / 4 : HILC_RES_MINW;   // waves per SIMD the register budget must allow
  static_assert(NW % (NCOL / 32) == 0 && CB % RH == 0 && RW % RB == 0 && C % RSTEP == 0, "tile split");
  static_assert(NB_ >= 1 && NB_ <= 3 && (NB_ == 1 || CARRYMODE || (NARROW && C >= 512)), "a chain needs carries or whole-stream tiles");
};

struct ResBlk {       // one residual block's parameters
  const float* w1t;   // packed, see WeightPipe
  const float* dw1_w; // [C][5]
  const float* dw1_b; // [C]
  const float* w2t;
  const float* dw2_w;
  const float* dw2_b;
  const float* hist1;   // STREAM: [B][C][4] caches of the two depthwise convs (NULL = zeros), and their successors
  const float* hist2;
  float* hist1_out;
  float* hist2_out;
  float pre_scale, out_scale;
};

struct ResUp {        // the stage's up-sampling layer (UR > 0)
  const float* xin;   // [B][2C][T/r]
  const float* tr_w;  // [2C][2r] taps of the depthwise transposed conv
  const float* w_lo;  // rows [0, C) of the k-major [2C][C] pointwise weight, packed like a block's matrix
  const float* w_hi;  // rows [C, 2C)
  const float* bias;  // [C]
  const float* hist;  // [B][2C] the ACTIVATED last input frame of the previous hop (NULL = zeros)
  float* hist_out;
  float in_scale;
};

struct ResDown {      // the stage's down-sampling layer (DR > 0)
  const float* w_lo;  // packed like a block's matrix: columns [0, C) of the k-major [C][2C] pointwise weight
  const float* w_hi;  // columns [C, 2C)
  const float* dw_w;  // [2C][2r]
  const float* dw_b;  // [2C]
  const float* hist;  // STREAM: [B][2C][r] last r pointwise outputs of the previous hop (NULL = zeros)
  float* hist_out;
  const float* res;   // optional [B][2C][T/r]: added to the output (the next stage's SpecBlock branch)
  float* y;           // [B][2C][T/r]
  float in_scale;
};

struct ResPost {      // the decoder's last layer as the closing phase (POST)
  const float* w;     // [C][5]
  const float* bias;  // [1] or NULL
  float* wav;         // [B][1][T]
  const float* hist;  // STREAM: [B][C][4] the ACTIVATED last 4 columns of the previous hop (NULL = zeros), and their successor
  float* hist_out;
  float in_scale, out_scale;
  int do_tanh;
};

struct ResSpec0 {     // the encoder's first conv and first SpecBlock as the opening phase (SPEC0); arguments as hilc_spec_block_conv_pre
  const float* wav;   // [B][T]
  const float* dft;   // packed [64 x 64] DFT basis (hilc_spec_block_pack, which = 0)
  const float* nyq;   // [64] sin_{32} row of the basis
  const float* pw;    // packed [40 x 64] 1x1 conv weight (which = 1)
  const float* bias;  // [64] or NULL
  const float* pre_w; // [64][5]
  const float* pre_b; // [64] or NULL
  const float* hist;  // STREAM: [B][hist_len] waveform history (the samples before t = 0; NULL = zeros)
  int hist_len;
  float pre_in_scale, mean, stdv, out_scale;
  int normalize;
};

constexpr int POST_CLASSES = 8;   // row classes of the closing conv's reduction (= hilc_conv_post's: c mod 8), summed in ascending order

constexpr int MAXBLK = 3;
constexpr int DDS = 12;   // per-row table of the down-sampling taps in LDS: [w_0..3 | w_4..7 | b, -, -, -]

struct ResArgs {
  const float* x;
  ResBlk blk[MAXBLK];
  int nblk;
  ResDown dn;
  ResUp up;
  ResPost post;
  ResSpec0 spec;
  long run_tiles;     // chain launches on the streaming column space: tiles per run (whole streams), 0 = equal split of the grid
  float* y;
  int T, tiles;
  int classes;        // carry form: workgroups per CU (0 = equal runs) and the cumulative run shares of the dispatch classes, 16-bit fractions
  unsigned cum[5];
  long total_tiles;
  int B;
  unsigned div_magic, div_shift;   // STREAM: n / T == __umulhi(n, div_magic) >> div_shift for n < 2^31
  // optional dynamic tile scheduler: two ints, zero at launch and zero again at exit.  Co-resident workgroups do
  // not share a CU fairly (the older one wins issue arbitration), so with static tile lists part of the kernel runs
  // at reduced occupancy; with tickets the faster workgroup simply takes more tiles.
  int* sched;
  unsigned long long* dbg;   // optional [tiles][8] s_memtime stamps (HILC_DEBUG_STAMPS builds, tools/res_phase_times.py)
};

#ifdef HILC_DEBUG_STAMPS
unsigned long long* g_dbg = nullptr;   // tools/res_phase_times.py builds its own copy of the library with this
#endif

}  // namespace
