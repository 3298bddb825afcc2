// Linear-addressing flavour of the fp32-MFMA GEMM core (same tiling, LDS layout, MFMA order and epilogues as
// gemm_core.h) for B operands that are plain rows of a [.., K, T] tensor: pointwise convs whose 4-column groups
// never straddle clips (T % 4 == 0, 16-B aligned).
//
// Why a second core: PMC on the generic core (K = M = 768) shows MFMA-busy 72 % + VALU-active 19-23 % ~= 95 %
// of the time, i.e. for fp32 the matrix pipe and the vector ALUs do not overlap, and the generic K loop spends
// ~125 VALU + ~50 SALU instructions per wave per 32 MFMAs on 64-bit address arithmetic, pointer selects
// (compiled to exec-mask branches) and run-time loader flags.  Here every global address is
//     uniform base (SGPR, advanced once per K slice)  +  per-thread 32-bit byte offset (loop-invariant)
// so a load costs no VALU; the ragged K tail is handled by a second, pre-computed offset set (rows clamped to
// K-1) plus a zero scale for rows >= K (0 * finite == 0; a non-finite sample only ever reaches columns of its own clip);
// ELU / halo-zeroing are compile-time.  Results are bit-identical to the generic core (same fmaf chain).
#pragma once
#include "gemm_core.h"

namespace hilc {

// Column mapping of one thread's 4-column group: element offset of (clip b, row 0, time t) and validity.
struct LinCol {
  unsigned off;   // b*K*T + t   (elements; the tensor is < 4 GiB, checked by the launcher)
  bool ok;        // the group exists (and, for halo tiles, lies inside [0, T))
};

// flattened (clip, t) columns, `tile_cols` per tile (BN, or whole clips for the streaming kernels)
struct FlatCols {
  int K, T, tile_cols;
  long ncols;
  static constexpr bool kZeroInvalid = false;   // invalid columns are never read back
  __device__ LinCol at(long ntile, int c) const {
    LinCol r;
    const long n = ntile * tile_cols + c;
    r.ok = c < tile_cols && n < ncols;
    const long b = r.ok ? n / T : 0;
    r.off = r.ok ? (unsigned)(b * (long)K * T + (n - b * T)) : 0u;
    return r;
  }
};

// per-clip tiles with a left halo (zero padding of the causal depthwise conv that follows)
struct TileCols {
  int K, T, tiles, step, halo;
  static constexpr bool kZeroInvalid = true;    // t < 0 must read as zero
  __device__ LinCol at(long ntile, int c) const {
    LinCol r;
    const long b = ntile / tiles;
    const int t = (int)(ntile - b * tiles) * step - halo + c;
    r.ok = t >= 0 && t < T;
    // a group outside [0, T) is loaded from the nearest group of its OWN clip and multiplied by a zero scale (RowsB): a
    // non-finite sample then stays inside the clip it belongs to, as in the reference (pointing every invalid group at
    // x[0] let an Inf in clip 0's first samples turn the zero padding of EVERY clip's first tile into NaN)
    const int tc = t < 0 ? 0 : (t > T - 4 ? T - 4 : t);
    r.off = (unsigned)(b * (long)K * T + tc);
    return r;
  }
};

// flat (stream-major) columns with a left halo: tile covers flat columns [ntile*step - halo, +BN) of the [B*T] column space.
// For the strided depthwise epilogue of a streaming hop that is longer than one tile (DwStrideFlatEpilogue): per-stream tiles
// are 62 % (T = 160) / 86 % (T = 320) full, flat ones 97 %.  T % 4 == 0, step % 4 == 0, halo % 4 == 0: a 4-column group never
// straddles streams.  Columns before the first stream / past the last are never read back (the epilogue takes the cache or
// zeros at a stream's first output).
struct FlatHaloCols {
  int K, T, step, halo;
  long ncols;
  static constexpr bool kZeroInvalid = false;
  __device__ LinCol at(long ntile, int c) const {
    LinCol r;
    const long n = ntile * step - halo + c;
    r.ok = n >= 0 && n < ncols;
    const long b = r.ok ? n / T : 0;
    r.off = r.ok ? (unsigned)(b * (long)K * T + (n - b * T)) : 0u;
    return r;
  }
};

// ---- B-operand policies: everything loop-invariant lives in State, a fetch is "uniform base + 32-bit offset"
// rows of a [.., K, T] tensor through the Scale / ELU prologue
template <class Cols, bool ELU>
struct RowsB {
  static constexpr int kMinWaves = 4;   // register budget: four waves per SIMD (54 + 64 registers at MB = 4; the epilogues may use the rest)
  const float* x;
  int T;
  float in_scale;
  Cols cols;
  typedef f32x4 Raw;
  struct State {
    unsigned off[BP], off_last[BP];
    float sc_last[BP];
    float sc;       // in_scale, or 0 for a column group outside [0, T) (halo / ragged tile): 0 * finite == 0 and
                    // ELU(0) == 0, so the zero padding costs no select in the K loop
  };
  __device__ State init(long ntile, int tid, int krem) const {
    State s;
    const LinCol lc = cols.at(ntile, (tid & 31) * 4);
    s.sc = (lc.ok || !Cols::kZeroInvalid) ? in_scale : 0.f;
#pragma unroll
    for (int h = 0; h < BP; ++h) {
      const int r = (tid >> 5) + 8 * h;
      const int rl = r < krem ? r : krem - 1;
      const bool mapped = lc.ok || Cols::kZeroInvalid;     // halo groups read their own clip's edge group (zero scale)
      s.off[h] = mapped ? (lc.off + (unsigned)r * (unsigned)T) * 4u : 0u;
      s.off_last[h] = mapped ? (lc.off + (unsigned)rl * (unsigned)T) * 4u : 0u;
      s.sc_last[h] = r < krem ? s.sc : 0.f;
    }
    return s;
  }
  __device__ Raw fetch(const State& s, int kt, bool last, int h) const {
    const char* sb = reinterpret_cast<const char*>(x) + (size_t)kt * ((size_t)BK * (size_t)T * 4u);   // uniform
    return *reinterpret_cast<const f32x4*>(sb + (last ? s.off_last[h] : s.off[h]));
  }
  __device__ f32x4 xform(const State& s, Raw v, bool last, int h) const {
    const float sc = last ? s.sc_last[h] : s.sc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = v[e] * sc;
      v[e] = ELU ? elu_fast(t) : t;
    }
    return v;
  }
};

// Up-sampling front-end (see UpLoader in gemm.hip for the math): B = depthwise transposed conv (k = 2r, stride r)
// of pro(x), computed from x[q-1], x[q], x[q+1] and the 2r taps of the channel.  R = 8 / 4: the four columns of a
// group share one q (no selects); R = 2: columns map to q0,q0,q0+1,q0+1; R = 0: any stride, per-thread selects and
// eight scalar tap loads per row; R = 1: any stride with the EXPANDED tap table of hilc_up_conv_expand_taps
// (`[K][r][8]`: for each phase p0 = t mod r of a 4-column group its eight taps as two 16-B words) — two vector loads
// per row like R = 8 / 4 (every VMEM instruction in this loop costs ~16 cycles of MFMA issue).
template <int R, bool ELU, bool HIST>
struct UpB {
  // register budget: the tap arithmetic needs ~80 registers beside the 64 accumulators = three waves per SIMD.  Pinning
  // it to four (a few spilled registers) pays for the stride-2 and expanded-table (stride-5) forms: -5.5 % / -4 % on the
  // K = 192 and K = 768 decoder layers; neutral to slightly negative for strides 8 and 4, which keep three.
  static constexpr int kMinWaves = (R == 1 || R == 2) ? 4 : 1;
  const float* x;      // [B][K][Tin]
  const float* w;      // [K][2r], or [K][r][8] for R == 1
  const float* hist;   // [B][K] activated x[-1] (HIST)
  static constexpr bool kExpanded = R == 1;
  int K, Tin, r;
  long ncols;          // B * Tin * r
  float in_scale;
  struct Raw {
    f32x4 wa, wb;
    float xm, x0, xp;
  };
  struct State {
    unsigned xoff[BP], xoff_last[BP];   // byte offset of x[b][row][q0]
    unsigned woff[BP], woff_last[BP];   // byte offset of w[row][p0]
    unsigned hoff[BP], hoff_last[BP];   // byte offset of hist[b][row]
    float sc_last[BP];
    int p[4], dq[4];
    bool ok, has_prev, has_next;
  };
  __device__ State init(long ntile, int tid, int krem) const {
    State s;
    const long n = ntile * BN + (tid & 31) * 4;
    s.ok = n < ncols;
    const long Tout = (long)Tin * r;
    const long b = s.ok ? n / Tout : 0;
    const int t = s.ok ? (int)(n - b * Tout) : 0;
    const int q0 = t / r;
    s.has_prev = s.ok && q0 >= 1;
    s.has_next = s.ok && q0 + 1 < Tin;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = (t + e) / r;
      s.p[e] = (t + e) - q * r;
      s.dq[e] = q - q0;
    }
#pragma unroll
    for (int h = 0; h < BP; ++h) {
      const int rr = (tid >> 5) + 8 * h;
      const int rl = rr < krem ? rr : krem - 1;
      s.xoff[h] = (unsigned)((b * K + rr) * (long)Tin + q0) * 4u;
      s.xoff_last[h] = (unsigned)((b * K + rl) * (long)Tin + q0) * 4u;
      s.woff[h] = R == 1 ? (unsigned)((rr * r + s.p[0]) * 8) * 4u : (unsigned)(rr * 2 * r + s.p[0]) * 4u;
      s.woff_last[h] = R == 1 ? (unsigned)((rl * r + s.p[0]) * 8) * 4u : (unsigned)(rl * 2 * r + s.p[0]) * 4u;
      s.hoff[h] = (unsigned)(b * K + rr) * 4u;
      s.hoff_last[h] = (unsigned)(b * K + rl) * 4u;
      s.sc_last[h] = rr < krem ? 1.f : 0.f;
    }
    return s;
  }
  __device__ Raw fetch(const State& s, int kt, bool last, int h) const {
    const char* sx = reinterpret_cast<const char*>(x) + (size_t)kt * ((size_t)BK * (size_t)Tin * 4u);   // uniform
    const char* sw = reinterpret_cast<const char*>(w) + (size_t)kt * ((size_t)BK * (R == 1 ? 8u : 2u) * (size_t)r * 4u);
    const unsigned xo = last ? s.xoff_last[h] : s.xoff[h];
    const unsigned wo = last ? s.woff_last[h] : s.woff[h];
    Raw v;
    if (R == 8 || R == 4) {
      v.wa = *reinterpret_cast<const f32x4*>(sw + wo);
      v.wb = *reinterpret_cast<const f32x4*>(sw + wo + R * 4);
    } else if (R == 1) {
      v.wa = *reinterpret_cast<const f32x4*>(sw + wo);
      v.wb = *reinterpret_cast<const f32x4*>(sw + wo + 16);
    } else if (R == 2) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(sw + wo);     // p0 == 0: taps (w0,w1 | w2,w3)
      v.wa = f32x4{t4.x, t4.y, t4.x, t4.y};
      v.wb = f32x4{t4.z, t4.w, t4.z, t4.w};
    } else {
      const float* wr = reinterpret_cast<const float*>(sw + wo) - s.p[0];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.wa[e] = wr[s.p[e]];
        v.wb[e] = wr[s.p[e] + r];
      }
    }
    const float* xp = reinterpret_cast<const float*>(sx + xo);
    v.x0 = xp[0];
    v.xp = xp[s.has_next ? 1 : 0];
    if (HIST) {
      const char* sh = reinterpret_cast<const char*>(hist) + (size_t)kt * ((size_t)BK * 4u);
      const float* hp = reinterpret_cast<const float*>(sh + (last ? s.hoff_last[h] : s.hoff[h]));
      v.xm = *(s.has_prev ? xp - 1 : hp);
    } else {
      v.xm = xp[s.has_prev ? -1 : 0];
    }
    return v;
  }
  __device__ float act(float v) const {
    const float t = v * in_scale;
    return ELU ? elu_fast(t) : t;
  }
  __device__ f32x4 xform(const State& s, const Raw& v, bool last, int h) const {
    float a0 = s.has_prev ? act(v.xm) : (HIST && s.ok ? v.xm : 0.f);   // the cache holds activated samples
    float a1 = s.ok ? act(v.x0) : 0.f;
    float a2 = s.has_next ? act(v.xp) : 0.f;
    if (last) { a0 *= s.sc_last[h]; a1 *= s.sc_last[h]; a2 *= s.sc_last[h]; }   // rows >= K contribute zero
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool nx = (R == 8 || R == 4) ? false : (R == 2) ? (e >= 2) : (s.dq[e] != 0);
      const float cur = nx ? a2 : a1;
      const float prev = nx ? a1 : a0;
      o[e] = fmaf(v.wa[e], cur, v.wb[e] * prev);     // same expression as hilc_dw_convtr
    }
    return o;
  }
};

// Implicit im2col of the waveform for the causal STFT-as-GEMM (offline form, zero history).  Column = frame f of one
// clip (per-clip tiles of 128 frames), row k = sample k of the window, element = wav[b, f*hop - (n_fft-1) + k].
// The 127*hop + n_fft samples a tile touches are staged ONCE into LDS (zero outside [0, T)); a B element is then
// an LDS read at a loop-invariant offset + 16*kt — no per-element address arithmetic, no guards, and LDS reads
// instead of 8 scalar global loads per slice (a VMEM instruction costs ~16 cycles of MFMA issue here).
// Lanes read at a stride of 4*hop samples, so the segment is stored with one pad word per 16 samples
// (index u -> u + (u >> 4)): at most 2-way bank conflicts for every hop of the codec, and a K slice (16 samples) is
// still a constant step (17 words).  n_fft % 16 == 0;  SEG >= (127*hop + n_fft) * 17 / 16.
template <int SEG>
struct StftSegB {
  static constexpr int kMinWaves = 1;
  const float* wav;
  int T, Tf, n_fft, hop, tiles;   // tiles per clip = ceil(Tf / 128)
  typedef f32x4 Raw;
  struct State {
    const float* seg;
    int off[4][BP];
  };
  __device__ State init(long ntile, int tid, int) const {
    __shared__ float seg[SEG];
    const long b = ntile / tiles;
    const int f0 = (int)(ntile - b * tiles) * BN;
    const int s0 = f0 * hop - (n_fft - 1);
    const int len = (BN - 1) * hop + n_fft;
    const float* wb = wav + b * (long)T;
    for (int i = tid; i < len; i += NT) {
      const int t = s0 + i;
      seg[i + (i >> 4)] = (t >= 0 && t < T) ? wb[t] : 0.f;
    }
    __syncthreads();
    State s;
    s.seg = seg;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int h = 0; h < BP; ++h) {
        const int u = ((tid & 31) * 4 + e) * hop + (tid >> 5) + 8 * h;
        s.off[e][h] = u + (u >> 4);
      }
    return s;
  }
  __device__ Raw fetch(const State& s, int kt, bool, int h) const {
    const float* p = s.seg + kt * (BK + 1);   // uniform: 16 samples + their pad word
    Raw v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p[s.off[e][h]];
    return v;
  }
  __device__ f32x4 xform(const State&, Raw v, bool, int) const { return v; }
};

#ifdef HILC_DEBUG_STAMPS
__device__ unsigned long long* g_lin_dbg = nullptr;    // [workgroup][8]: 4 s_memtime stamps, HW_ID | XCC_ID << 32 (tools/lin_phase_times.py builds its own library)
#define LIN_STAMP(i) do { if (g_lin_dbg && threadIdx.x == 0) g_lin_dbg[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LIN_STAMP(i) do { } while (0)
#endif

template <int MB, class BOp, class Epilogue>
__global__ __launch_bounds__(NT, BOp::kMinWaves) void gemm_lin_kernel(const float* __restrict__ wt, int M, int K, int ldw, long ntiles,
                                                      int mtiles, BOp bop, Epilogue ep) {
  constexpr int BM = 32 * MB;
  constexpr int AG = BK * BM / 4;
  constexpr int AP = (AG + NT - 1) / NT;
  constexpr int STG = 2 * BK * (BM + BN);
  constexpr int EPI = Epilogue::template lds_floats<MB>();
  constexpr int SM = STG > EPI ? STG : EPI;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
  float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);

  long id = blockIdx.x;   // XCD-aware tile order, as in gemm_core.h
  long grp = id / (8L * mtiles);
  int within = (int)(id - grp * 8L * mtiles);
  long ntile = grp * 8 + (within & 7);
  int mtile = within >> 3;
  if (ntile >= ntiles) return;
  LIN_STAMP(0);
#ifdef HILC_DEBUG_STAMPS
  if (g_lin_dbg && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_lin_dbg[(long)blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  const int m0 = mtile * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ktiles = (K + BK - 1) / BK;
  const int krem = K - (ktiles - 1) * BK;          // rows of the last slice (1..BK)

  // ---- loop-invariant per-thread byte offsets of the weight slice
  unsigned aoff[AP], aoff_last[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int g = tid + p * NT;
    const int kr = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
    int col = m0 + m4;
    col = col < ldw - 4 ? col : ldw - 4;           // rows >= M only feed accumulator rows that are never stored
    const int krl = kr < krem ? kr : krem - 1;
    // threads beyond the slice (AG < NT, i.e. MB = 1 or 3) load nothing they keep: point them at row 0 of the slice,
    // their natural row index would run up to 16 rows past it (past the end of the matrix at the last slices)
    aoff[p] = g < AG ? (unsigned)(kr * ldw + col) * 4u : 0u;
    aoff_last[p] = g < AG ? (unsigned)(krl * ldw + col) * 4u : 0u;
  }
  const unsigned a_slice = (unsigned)BK * (unsigned)ldw * 4u;   // bytes per K slice
  const typename BOp::State bs = bop.init(ntile, tid, krem);

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  f32x4 ra[AP];
  typename BOp::Raw rb[BP];
  auto fetch = [&](int kt, bool last) {
    const char* sa = reinterpret_cast<const char*>(wt) + (size_t)kt * a_slice;   // uniform
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const f32x4*>(sa + (last ? aoff_last[p] : aoff[p]));
#pragma unroll
    for (int h = 0; h < BP; ++h) rb[h] = bop.fetch(bs, kt, last, h);
  };
  auto stage = [&](int buf, bool last) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int g = tid + p * NT;
      if (AG % NT == 0 || g < AG) {
        const int k = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
        *reinterpret_cast<f32x4*>(&As[buf][k][m4]) = ra[p];
      }
    }
#pragma unroll
    for (int h = 0; h < BP; ++h)
      *reinterpret_cast<f32x4*>(&Bs[buf][(tid >> 5) + 8 * h][(tid & 31) * 4]) = bop.xform(bs, rb[h], last, h);
  };

  const int kh = lane >> 5, l31 = lane & 31;
  fetch(0, ktiles == 1);
  stage(0, ktiles == 1);
  __syncthreads();
  LIN_STAMP(1);
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ktiles;
    const bool last = kt + 2 == ktiles;
    if (more) fetch(kt + 1, last);
    float av[2][MB], bv[2];
    bv[0] = Bs[buf][kh][wave * 32 + l31];
#pragma unroll
    for (int i = 0; i < MB; ++i) av[0][i] = As[buf][kh][i * 32 + l31];
#pragma unroll
    for (int j = 0; j < BK / 2; ++j) {
      const int cur = j & 1, nxt = cur ^ 1;
      if (j + 1 < BK / 2) {
        bv[nxt] = Bs[buf][2 * j + 2 + kh][wave * 32 + l31];
#pragma unroll
        for (int i = 0; i < MB; ++i) av[nxt][i] = As[buf][2 * j + 2 + kh][i * 32 + l31];
      }
#pragma unroll
      for (int i = 0; i < MB; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur], acc[i], 0, 0, 0);
    }
    if (more) {
      stage(buf ^ 1, last);
      __syncthreads();
    }
  }
  LIN_STAMP(2);
  ep.template run<MB>(acc, smem, m0, ntile, wave, lane, tid);
  LIN_STAMP(3);
}

// ---- "wave = row block" form of the same tile (MB = 4 only): wave w owns the 32 channels [32w, 32w+32) of the tile
// and ALL four 32-column blocks, and the MFMA operand roles are swapped — the activations go in as the A operand
// (M index = time), the weights as B (N index = channel) — so that D comes out as D[time][channel]: a lane holds one
// CHANNEL (lane & 31) and, after 32 v_permlane32_swap, 64 CONTIGUOUS time steps of it in its accumulator registers
// (lanes 0-31: columns 0-63 of the tile, lanes 32-63: columns 64-127; see wr_time_order).  A depthwise conv along time
// is then register arithmetic with per-lane taps, the output rows are 16-B stores straight from the accumulators: no
// accumulator -> LDS -> register round trip and no barrier in the epilogue (the LDS epilogues cost a K-independent
// ~51 k cycles per workgroup, 11-28 % of its life: profiles/r02_experiments.md).  The LDS traffic of the K loop is
// unchanged (per k-pair: 4 activation + 1 weight operand reads for 4 MFMAs), products and k order are the same, so the
// results are bit-identical to the column-block form.
//
// After wr_time_order():  V[i] (column i of the lane's 64-column half, i = 32 cb + 8 g + 4 q + e)  =  acc[cb + 2 q][4 g + e].
__device__ __forceinline__ void wr_time_order(f32x16 (&acc)[4]) {
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // v_permlane32_swap vdst, src0: lanes 32-63 of vdst <-> lanes 0-31 of src0
      // (scalar temporaries: this hipcc's __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index)
      const float lo = acc[cb][r], hi = acc[cb + 2][r];
      const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
      const unsigned p0 = p[0], p1 = p[1];
      acc[cb][r] = __uint_as_float(p0);
      acc[cb + 2][r] = __uint_as_float(p1);
    }
}
#define HILC_WR_V(acc, i) ((acc)[(((i) >> 5) & 1) + 2 * (((i) >> 2) & 1)][4 * (((i) >> 3) & 3) + ((i) & 3)])

template <class BOp, class Epilogue>
__global__ __launch_bounds__(NT, BOp::kMinWaves) void gemm_lin_wr_kernel(const float* __restrict__ wt, int M, int K, int ldw,
                                                                         long ntiles, int mtiles, BOp bop, Epilogue ep) {
  constexpr int MB = 4;
  constexpr int BM = 32 * MB;
  constexpr int AG = BK * BM / 4;
  constexpr int AP = (AG + NT - 1) / NT;
  constexpr int STG = 2 * BK * (BM + BN);
  __shared__ __attribute__((aligned(16))) float smem[STG];
  float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
  float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);

  long id = blockIdx.x;   // XCD-aware tile order, as in gemm_core.h
  long grp = id / (8L * mtiles);
  int within = (int)(id - grp * 8L * mtiles);
  long ntile = grp * 8 + (within & 7);
  int mtile = within >> 3;
  if (ntile >= ntiles) return;
  const int m0 = mtile * BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ktiles = (K + BK - 1) / BK;
  const int krem = K - (ktiles - 1) * BK;

  unsigned aoff[AP], aoff_last[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int g = tid + p * NT;
    const int kr = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
    int col = m0 + m4;
    col = col < ldw - 4 ? col : ldw - 4;
    const int krl = kr < krem ? kr : krem - 1;
    aoff[p] = (unsigned)(kr * ldw + col) * 4u;
    aoff_last[p] = (unsigned)(krl * ldw + col) * 4u;
  }
  const unsigned a_slice = (unsigned)BK * (unsigned)ldw * 4u;
  const typename BOp::State bs = bop.init(ntile, tid, krem);

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  f32x4 ra[AP];
  typename BOp::Raw rb[BP];
  auto fetch = [&](int kt, bool last) {
    const char* sa = reinterpret_cast<const char*>(wt) + (size_t)kt * a_slice;
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const f32x4*>(sa + (last ? aoff_last[p] : aoff[p]));
#pragma unroll
    for (int h = 0; h < BP; ++h) rb[h] = bop.fetch(bs, kt, last, h);
  };
  auto stage = [&](int buf, bool last) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int g = tid + p * NT;
      const int k = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
      *reinterpret_cast<f32x4*>(&As[buf][k][m4]) = ra[p];
    }
#pragma unroll
    for (int h = 0; h < BP; ++h)
      *reinterpret_cast<f32x4*>(&Bs[buf][(tid >> 5) + 8 * h][(tid & 31) * 4]) = bop.xform(bs, rb[h], last, h);
  };

  const int kh = lane >> 5, l31 = lane & 31;
  fetch(0, ktiles == 1);
  stage(0, ktiles == 1);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ktiles;
    const bool last = kt + 2 == ktiles;
    if (more) fetch(kt + 1, last);
    float wv[2], xv[2][MB];
    wv[0] = As[buf][kh][wave * 32 + l31];
#pragma unroll
    for (int i = 0; i < MB; ++i) xv[0][i] = Bs[buf][kh][i * 32 + l31];
#pragma unroll
    for (int j = 0; j < BK / 2; ++j) {
      const int cur = j & 1, nxt = cur ^ 1;
      if (j + 1 < BK / 2) {
        wv[nxt] = As[buf][2 * j + 2 + kh][wave * 32 + l31];
#pragma unroll
        for (int i = 0; i < MB; ++i) xv[nxt][i] = Bs[buf][2 * j + 2 + kh][i * 32 + l31];
      }
#pragma unroll
      for (int i = 0; i < MB; ++i)     // D[time][channel] += x[k][time] * w[k][channel]
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[cur][i], wv[cur], acc[i], 0, 0, 0);
    }
    if (more) {
      stage(buf ^ 1, last);
      __syncthreads();
    }
  }
  ep.run_wr(acc, m0 + wave * 32, ntile, lane);
}

template <class BOp, class Epilogue>
int launch_lin_wr(const float* wt, int M, int K, int ldw, long ntiles, const BOp& bop, const Epilogue& ep, hipStream_t s) {
  const int mtiles = (M + 127) / 128;
  const long groups = (ntiles + 7) / 8;
  const long blocks = groups * 8 * mtiles;
  if (ntiles <= 0 || blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL((gemm_lin_wr_kernel<BOp, Epilogue>), dim3((unsigned)blocks), dim3(NT), 0, s, wt, M, K, ldw, ntiles, mtiles, bop, ep);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

// the wave-row form pays where the column-block form would run 128-row tiles anyway
inline bool wr_shape(int M, long ntiles) { return M % 128 == 0 && pick_mb(M / 32, ntiles) == 4; }

template <class BOp, class Epilogue>
int launch_lin(const float* wt, int M, int K, int ldw, long ntiles, const BOp& bop, const Epilogue& ep, hipStream_t s) {
  const int m32 = (M + 31) / 32;
  const int MB = pick_mb(m32, ntiles);
  const long groups = (ntiles + 7) / 8;
  int mtiles = (m32 + MB - 1) / MB;
  long blocks = groups * 8 * mtiles;
  if (ntiles <= 0 || blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  dim3 grid((unsigned)blocks), block(NT);
  HILC_CLEAR_ERROR();
  switch (MB) {
    case 1: hipLaunchKernelGGL((gemm_lin_kernel<1, BOp, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, bop, ep); break;
    case 2: hipLaunchKernelGGL((gemm_lin_kernel<2, BOp, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, bop, ep); break;
    case 3: hipLaunchKernelGGL((gemm_lin_kernel<3, BOp, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, bop, ep); break;
    default: hipLaunchKernelGGL((gemm_lin_kernel<4, BOp, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, bop, ep); break;
  }
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

// x must be 16-B aligned, T % 4 == 0, the tensor < 4 GiB (32-bit byte offsets); the caller checks.
template <class Cols, class Epilogue>
int launch_gemm_lin(const float* wt, const float* x, int M, int K, int ldw, int T, long ntiles, float in_scale,
                    bool in_elu, const Cols& cols, const Epilogue& ep, hipStream_t s) {
  if (in_elu) {
    RowsB<Cols, true> b;
    b.x = x; b.T = T; b.in_scale = in_scale; b.cols = cols;
    return launch_lin(wt, M, K, ldw, ntiles, b, ep, s);
  }
  RowsB<Cols, false> b;
  b.x = x; b.T = T; b.in_scale = in_scale; b.cols = cols;
  return launch_lin(wt, M, K, ldw, ntiles, b, ep, s);
}

}  // namespace hilc
