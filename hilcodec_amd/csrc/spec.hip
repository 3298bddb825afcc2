// One-launch SpecBlock for the long encoder stages (n_fft = 64 / 128 / 256, where one workgroup can hold every bin of
// a 128-frame tile), gfx950:
//
//   y = x + out_scale * ( W * norm(log(max(|STFT(wav)|, 1e-5))) + bias )        (seanet.py:220-246, conv.py:329-358)
//
// Un-fused this is two launches (hilc_stft_logmag + hilc_pw_conv) with a [n_fft/2+1 x T] tensor through HBM between
// them, a DFT GEMM whose n_fft+2 rows pad to the next multiple of 32 (66 -> 96, 130 -> 160/192 rows: a third of the
// MFMAs of the two longest stages multiply zeros), and a 1x1 conv that is HBM-bound at K = 33 / 65.  Here, per tile:
//   S0  the 127*hop + n_fft waveform samples of the tile -> LDS once (implicit im2col: a B element of the DFT GEMM is an
//       LDS read at a per-lane offset + constant; one pad word per 16 samples keeps strided frames off one bank)
//   A   DFT GEMM with exactly n_fft rows: (cos_k, sin_k) pairs k = 1..n_fft/2-1 in adjacent accumulator registers of
//       a lane, and the two purely real bins (cos_0, cos_{n_fft/2}) share the pair slot of k = 0.  sin_0 is exactly
//       zero and dropped; sin_{n_fft/2} (|.| <= 1.4e-4: the fp32 rounding of sin(-pi n), kept for fidelity) is ONE row:
//       a scalar fmaf chain on the VALU for the lanes that own the Nyquist bin (n_fft FMAs per tile instead of a
//       padded 32-row MFMA block)
//   B   magnitude -> log -> normalise on the accumulators -> LDS tile S[bin][frame]
//   C   1x1 conv as a second GEMM, B operand = S
//   D   (acc + bias) * out_scale + x -> y through an LDS transpose (16-B coalesced loads / stores)
// Both A operands (DFT basis, conv weight) stream from L2 in packed "MFMA lane order" (hilc_spec_block_pack), as in
// the fused residual block: no LDS staging, no barrier inside either K loop.
// Arithmetic is the un-fused kernels' (same fmaf chains in k order, same separate roundings in the magnitude and the
// epilogue); only the summation grouping of nothing changes, so outputs are bit-identical to stft_logmag + pw_conv.
#include "gemm_core.h"
#include "stream_gemm.h"

using namespace hilc;

namespace {

typedef const __attribute__((address_space(1))) float* gptr_t;
typedef const __attribute__((address_space(1))) f32x4* gvec_t;
typedef __attribute__((address_space(3))) float* lptr_t;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int TF = 128;          // frames per tile
constexpr int ES = 132;          // row stride of the epilogue transpose tile

template <int N>
struct SpecCfg {
  static constexpr int HOP = N == 64 ? 1 : (N == 128 ? 2 : 8);   // the stage's hop is fixed with its n_fft (stage_hop below)
  // One pad word per 16 samples keeps the strided frames of a DFT operand read off one LDS bank.  At hop 1 consecutive lanes
  // read consecutive samples anyway: no padding, and every segment address of the kernel becomes base + immediate (the
  // stage-0 launch spent ~350 of its ~1 240 VALU instructions per tile and wave on `u + (u >> 4)`).
  static constexpr bool PAD = HOP > 1;
  static constexpr int CB = N / 32;                       // row blocks of both GEMMs (C == N)
  static constexpr int NB = N / 2 + 1;                    // bins
  static constexpr int KP = N <= 128 ? 4 : 2;             // k-pairs per weight register set
  static constexpr int DEPTH = 3;                         // register sets: the loads run two sets ahead
  static constexpr int SETS_A = N / 2 / KP;
  static constexpr int KPAIRS_C = ((NB + 1) / 2 + KP - 1) / KP * KP;   // conv K = NB rows, zero-padded to whole sets
  static constexpr int SETS_C = KPAIRS_C / KP;
  static constexpr int SROWS = 2 * KPAIRS_C;              // rows of the LDS spectrogram tile
  static constexpr int WPS = KP * CB / 4;                 // 16-B words per lane and set
};

struct SpecArgs {
  const float* wav;      // [B][T]
  const float* hist;     // streaming: [B][hist_len] waveform history (samples before t = 0), or null (zeros)
  int hist_len;
  const float* dft;      // packed [N x N] DFT basis (rows: see above)
  const float* nyq;      // [N] sin_{N/2} row of the basis
  const float* pw;       // packed [SROWS x N] conv weight (rows >= NB zero)
  const float* bias;     // [N] or null
  const float* x;        // [B][N][Tf] residual input (PRE: unused)
  const float* pre_w;    // PRE: first encoder conv `[N][5]` and its bias `[N]` — x is computed here, not read
  const float* pre_b;
  float pre_in_scale;
  float* y;              // [B][N][Tf]
  int B;                 // FLAT: clips / streams (tiles walk the flat frame space b * Tf + j)
  int T, Tf, hop, tiles;
  float mean, stdv, out_scale;
  int normalize;
};

// packed[((s * WPS + q) * 64 + lane) * 4 + e] = W[2 * (s*KP + j) + (lane >> 5)][32 * i + (lane & 31)],  q*4 + e = j*CB + i;
// rows >= K read as zero
__global__ __launch_bounds__(256) void spec_pack_kernel(const float* w, float* packed, int K, int M, int KP, int nsets) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int CB = M / 32, WPS = KP * CB / 4;
  if (idx >= nsets * WPS * 256) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, word = idx >> 8;
  const int q = word % WPS, s = word / WPS;
  const int v = q * 4 + e, j = v / CB, i = v % CB;
  const int k = 2 * (s * KP + j) + (lane >> 5), m = 32 * i + (lane & 31);
  packed[idx] = k < K ? w[(long)k * M + m] : 0.f;
}

template <bool PAD>
__device__ __forceinline__ int padded(int u) { return PAD ? u + (u >> 4) : u; }

// PRE (n_fft = 64, hop 1 only): the residual input is the first encoder conv of the SAME waveform tile,
// x[m][t] = sum_j pre_w[m][j] * (pre_in_scale * wav[t-4+j]) + pre_b[m]   (seanet.py:280-286; hilc_conv_pre's arithmetic),
// evaluated from the LDS segment in the epilogue: the [64 x T] tensor conv_pre would write and this kernel re-read
// (2 x 1.57 GB at B = 256) never exists.
// FLAT (round 6: the SpecBlock branches of a streaming hop, 40 - 320 frames per stream): a tile is 128 frames of the FLAT frame space
// b * Tf + j instead of 128 frames of one clip — at 40 frames per stream the per-clip tile was a third full (the hop took the two-launch
// path on the generic STFT loader instead: 59 TF), at 160 it was 62 % full.  A tile then touches up to FLAT_MAXP streams; each contributes its
// own piece of waveform ((frames - 1) * hop + n_fft samples, the first n_fft - 1 of them history in front of t = 0 where the piece starts a
// stream), the pieces sit back to back in `seg`, and a frame's window starts at its piece's base + (frame - first frame of the piece) * hop.
// Per-frame arithmetic is untouched: bit-identical to the per-clip form.
constexpr int FLAT_MAXP = 5;     // Tf >= 32: 128 frames touch at most 5 streams

template <int N, bool PRE, bool FLAT = false>
__global__ __launch_bounds__(256, 2) void spec_block_kernel(SpecArgs a) {
  using K = SpecCfg<N>;
  static_assert(!(PRE && FLAT), "the first conv rides on the per-clip form");
  constexpr int CB = K::CB, NB = K::NB;
  constexpr int HOP = K::HOP;
  constexpr bool PAD = K::PAD;
  constexpr int SEG_SAMPLES = FLAT ? TF * HOP + FLAT_MAXP * (N - HOP) : (TF - 1) * HOP + N;
  constexpr int SEG_MAX = (SEG_SAMPLES * (PAD ? 17 : 16)) / 16 + 2;
  constexpr int SE_FLOATS = K::SROWS * TF > 64 * ES ? K::SROWS * TF : 64 * ES;                      // S, later the epilogue tile
  __shared__ __attribute__((aligned(16))) float seg[SEG_MAX];
  __shared__ __attribute__((aligned(16))) float segs[PRE ? SEG_MAX : 4];   // PRE: pre_in_scale * segment (the first conv's input)
  __shared__ __attribute__((aligned(16))) float SE[SE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5, l31 = lane & 31;
  constexpr int hop = HOP;
  const long b = FLAT ? ((long)blockIdx.x * TF) / a.Tf : blockIdx.x / a.tiles;          // FLAT: the tile's first stream
  const int f0 = FLAT ? (int)((long)blockIdx.x * TF - b * a.Tf) : (int)(blockIdx.x - b * a.tiles) * TF;   // ... and its first frame in it
  [[maybe_unused]] const int nf0 = a.Tf - f0 < TF ? a.Tf - f0 : TF;                       // FLAT: frames of the first piece
  [[maybe_unused]] const int len0 = (nf0 - 1) * hop + N, lenf = (a.Tf - 1) * hop + N;     // samples of the first / of a whole-stream piece

  // ---- S0: waveform segment (zero outside [0, T)); padding rows of the spectrogram tile
  if constexpr (FLAT) {
    const int rest = TF - nf0, whole = rest / a.Tf, tail = rest - whole * a.Tf;
    const int total = len0 + whole * lenf + (tail > 0 ? (tail - 1) * hop + N : 0);
    for (int i = tid; i < total; i += 256) {
      int p = 0, idx = i, jstart = f0;
      if (i >= len0) {
        const int q = i - len0;
        p = 1 + q / lenf;
        idx = q - (p - 1) * lenf;
        jstart = 0;
      }
      const long bp = b + p;
      const int t = jstart * hop - (N - 1) + idx;
      float v = 0.f;
      if (bp < a.B) {
        if (t >= 0) { if (t < a.T) v = a.wav[bp * (long)a.T + t]; }
        else if (a.hist != nullptr && t >= -a.hist_len) v = a.hist[bp * (long)a.hist_len + a.hist_len + t];
      }
      seg[padded<PAD>(i)] = v;
    }
    for (int i = tid; i < (K::SROWS - NB) * TF; i += 256) SE[NB * TF + i] = 0.f;
  } else {
    const int s0 = f0 * hop - (N - 1);
    const int len = (TF - 1) * hop + N;
    const float* wb = a.wav + b * (long)a.T;
    const float* hb = a.hist != nullptr ? a.hist + b * (long)a.hist_len + a.hist_len : nullptr;   // hb[t], t < 0
    for (int i = tid; i < len; i += 256) {
      const int t = s0 + i;
      float v = 0.f;
      if (t >= 0) { if (t < a.T) v = wb[t]; }
      else if (hb != nullptr && t >= -a.hist_len) v = hb[t];
      seg[padded<PAD>(i)] = v;
      if constexpr (PRE) segs[padded<PAD>(i)] = v * a.pre_in_scale;     // scaled once per sample, not once per use
    }
    for (int i = tid; i < (K::SROWS - NB) * TF; i += 256) SE[NB * TF + i] = 0.f;
  }
  lds_barrier();

  const int col = wave * 32 + l31;                       // this lane's frame of the tile
  // first sample of this frame's window in `seg` (FLAT: inside its stream's piece)
  int u0 = col * hop;
  if constexpr (FLAT) {
    const int fr = f0 + col;                               // frame index counted from the first stream's frame 0
    const int p = fr / a.Tf, j = fr - p * a.Tf;
    u0 = p == 0 ? (j - f0) * hop : len0 + (p - 1) * lenf + j * hop;
  }
  f32x16 acc[CB];
  // ---- A: DFT.  B element of k-pair P (k = 2P + kh) = seg[padded(col*hop + k)]: 8 per-lane offsets per 16-sample
  //      slice, slices advance by the constant 17 words
  {
    int off[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) off[p] = padded<PAD>(u0 + 2 * p + kh);
    // (col*hop + 2p + kh) + 16*kt: padded() adds exactly 17*kt because the slice step is a multiple of 16
    auto bop = [&](int P) -> float { return seg[off[P & 7] + (PAD ? 17 : 16) * (P >> 3)]; };
    stream_gemm<CB, K::KP, K::DEPTH, K::SETS_A>(a.dft, acc, lane, bop);
  }
  // ---- Nyquist bin's imaginary part: scalar chain in k order (lanes 0..31 own rows 0/1 = cos_0 / cos_{N/2})
  float nyq_im = 0.f;
#pragma unroll 8
  for (int k = 0; k < N; ++k) nyq_im = fmaf(a.nyq[k], seg[padded<PAD>(u0 + k)], nyq_im);
  // ---- B: magnitude -> log -> normalise -> S[bin][frame]
  const SpecFinish finish = SpecFinish::make(a.mean, a.stdv, a.normalize);
  float* S = SE;
#pragma unroll
  for (int i = 0; i < CB; ++i) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int row = i * 32 + acc_row(r, lane);          // even
      if (i == 0 && r == 0) {
        if (kh == 0) {                                    // rows 0, 1: the two real bins
          S[0 * TF + col] = finish(acc[0][0], 0.f);
          S[(N / 2) * TF + col] = finish(acc[0][1], nyq_im);
        } else {                                          // rows 4, 5: bin 2
          S[(row >> 1) * TF + col] = finish(acc[0][0], acc[0][1]);
        }
      } else {
        S[(row >> 1) * TF + col] = finish(acc[i][r], acc[i][r + 1]);
      }
    }
  }
  lds_barrier();

  // ---- C: 1x1 conv, K = bins (zero-padded to whole register sets), B = S
  {
    const float* sl = S + kh * TF + col;
    auto bop = [&](int P) -> float { return sl[2 * P * TF]; };
    stream_gemm<CB, K::KP, K::DEPTH, K::SETS_C>(a.pw, acc, lane, bop);
  }
  lds_barrier();                                          // every wave is done reading S: it becomes the transpose tile

  // ---- D: epilogue, 64 rows at a time: thread = (row, 16-column segment), 2 per thread
  float* E = SE;
  const long ybase = b * (long)N * a.Tf;
#pragma unroll
  for (int ch = 0; ch < (CB + 1) / 2; ++ch) {
    if (ch > 0) lds_barrier();
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      if (i >= 2 * ch && i < 2 * ch + 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) E[((i - 2 * ch) * 32 + acc_row(r, lane)) * ES + col] = acc[i][r];
      }
    }
    lds_barrier();
    const int rows = (CB - 2 * ch) >= 2 ? 64 : 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int sg = tid + 256 * s;
      const int row = sg >> 3, c0 = (sg & 7) * 16;
      if (row >= rows) continue;
      const int m = ch * 64 + row;
      const float bv = a.bias != nullptr ? a.bias[m] : 0.f;
      // PRE: the row's five taps and bias once, before the stores (y may alias nothing here, but the compiler cannot know:
      // inside the group loop every store forced them to be loaded again — 91 VMEM reads per tile and wave)
      float w5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      float pb = 0.f;
      if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < 5; ++j) w5[j] = a.pre_w[m * 5 + j];
        pb = a.pre_b != nullptr ? a.pre_b[m] : 0.f;
      }
      const float* er = E + row * ES + c0;
      // where the four 4-frame groups of this 16-column segment live (Tf % 4 == 0: a group never straddles clips / streams)
      long offg[4];
      bool okg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (FLAT) {
          const int fr = f0 + c0 + 4 * g;                 // counted from the first stream's frame 0
          const int p = fr / a.Tf, j = fr - p * a.Tf;
          okg[g] = b + p < a.B;
          offg[g] = ((b + p) * (long)N + m) * a.Tf + j;
        } else {
          okg[g] = f0 + c0 + 4 * g < a.Tf;
          offg[g] = ybase + (long)m * a.Tf + f0 + c0 + 4 * g;
        }
      }
      f32x4 rq[4];
      if constexpr (!PRE) {       // all residual loads before the first store (x may alias y: later loads would wait behind it)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (a.x != nullptr && okg[g]) rq[g] = *reinterpret_cast<const f32x4*>(a.x + offg[g]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (okg[g]) {                                    // Tf % 4 == 0: whole groups
          f32x4 v = *reinterpret_cast<const f32x4*>(er + 4 * g);
          f32x4 rr;
          if constexpr (PRE) {
            // column c of the tile is time f0 + c (hop 1) = segment sample c + N - 1
            float sm[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sm[j] = segs[padded<PAD>(c0 + 4 * g + j + N - 5)];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float acc5 = 0.f;
#pragma unroll
              for (int j = 0; j < 5; ++j) acc5 = fmaf(w5[j], sm[e + j], acc5);
              rr[e] = a.pre_b != nullptr ? __fadd_rn(acc5, pb) : acc5;
            }
          } else {
            rr = rq[g];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = v[e];
            if (a.bias != nullptr) t = __fadd_rn(t, bv);
            t = __fmul_rn(t, a.out_scale);
            v[e] = (PRE || a.x != nullptr) ? __fadd_rn(t, rr[e]) : t;   // separate roundings: y.mul_(scale); x.add_(y)  (x NULL: the branch alone)
          }
          *reinterpret_cast<f32x4*>(a.y + offg[g]) = v;
        }
      }
    }
  }
}

// short clips — a streaming hop's 40 ... 320 frames per stream — tile the flat frame space (see FLAT above)
inline bool flat_tiles(int Tf) { return Tf >= 32 && Tf < 4 * TF && Tf % TF != 0; }

template <int N, bool PRE = false>
int launch_spec(const SpecArgs& a, int B, hipStream_t s) {
  const bool flat = !PRE && flat_tiles(a.Tf);
  const long blocks = flat ? ((long)B * a.Tf + TF - 1) / TF : (long)B * a.tiles;
  if (blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  HILC_CLEAR_ERROR();
  if constexpr (!PRE) {
    if (flat) {
      hipLaunchKernelGGL((spec_block_kernel<N, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
      HILC_CHECK_LAUNCH();
      return HILC_OK;
    }
  }
  hipLaunchKernelGGL((spec_block_kernel<N, PRE>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

int stage_hop(int n_fft) { return n_fft == 64 ? 1 : (n_fft == 128 ? 2 : (n_fft == 256 ? 8 : 0)); }

}  // namespace

extern "C" int hilc_spec_block_supported(int n_fft, int hop, int C, int T) {
  if (!(n_fft == 64 || n_fft == 128 || n_fft == 256)) return 0;
  if (hop != stage_hop(n_fft) || C != n_fft || T <= 0) return 0;
  const int Tf = (T - 1) / hop + 1;
  return Tf % 4 == 0;
}

extern "C" int hilc_spec_block_packed_floats(int n_fft, int which) {
  switch (n_fft) {
    case 64: return which == 0 ? 64 * 64 : SpecCfg<64>::SROWS * 64;
    case 128: return which == 0 ? 128 * 128 : SpecCfg<128>::SROWS * 128;
    case 256: return which == 0 ? 256 * 256 : SpecCfg<256>::SROWS * 256;
    default: return 0;
  }
}

extern "C" int hilc_spec_block_pack(const float* w, float* packed, int K, int n_fft, int which, void* stream) {
  if (!w || !packed) return HILC_ERR_NULL;
  if (!(n_fft == 64 || n_fft == 128 || n_fft == 256) || w == packed) return HILC_ERR_UNSUPPORTED;
  const int KP = n_fft == 64 ? SpecCfg<64>::KP : (n_fft == 128 ? SpecCfg<128>::KP : SpecCfg<256>::KP);
  const int floats = hilc_spec_block_packed_floats(n_fft, which);
  const int nsets = floats / n_fft / 2 / KP;
  if (which == 0 ? K != n_fft : K != n_fft / 2 + 1) return HILC_ERR_SHAPE;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(spec_pack_kernel, dim3((unsigned)((floats + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed,
                     K, n_fft, KP, nsets);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_spec_block(const float* wav, const float* hist, int hist_len, const float* dft_packed,
                               const float* nyq_sin, const float* pw_packed, const float* bias, const float* x, float* y,
                               int B, int T, int n_fft, int hop, float mean, float stdv, int normalize, float out_scale,
                               void* stream) {
  if (!wav || !dft_packed || !nyq_sin || !pw_packed || !y) return HILC_ERR_NULL;      // x NULL: y = out_scale * (W spec + bias)
  if (B <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (hist != nullptr && hist_len < n_fft - 1) return HILC_ERR_SHAPE;
  if (!hilc_spec_block_supported(n_fft, hop, n_fft, T)) return HILC_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return HILC_ERR_UNSUPPORTED;
  SpecArgs a;
  a.wav = wav; a.hist = hist; a.hist_len = hist_len;
  a.dft = dft_packed; a.nyq = nyq_sin; a.pw = pw_packed; a.bias = bias; a.x = x; a.y = y;
  a.B = B; a.T = T; a.Tf = (T - 1) / hop + 1; a.hop = hop; a.tiles = (a.Tf + TF - 1) / TF;
  a.mean = mean; a.stdv = stdv; a.out_scale = out_scale; a.normalize = normalize;
  a.pre_w = nullptr; a.pre_b = nullptr; a.pre_in_scale = 1.f;
  switch (n_fft) {
    case 64: return launch_spec<64>(a, B, (hipStream_t)stream);
    case 128: return launch_spec<128>(a, B, (hipStream_t)stream);
    default: return launch_spec<256>(a, B, (hipStream_t)stream);
  }
}

extern "C" int hilc_spec_block_conv_pre(const float* wav, const float* hist, int hist_len, const float* dft_packed,
                                        const float* nyq_sin, const float* pw_packed, const float* bias, const float* pre_w,
                                        const float* pre_b, float pre_in_scale, float* y, int B, int T, int n_fft, int hop,
                                        int pre_ksize, float mean, float stdv, int normalize, float out_scale, void* stream) {
  if (!wav || !dft_packed || !nyq_sin || !pw_packed || !pre_w || !y) return HILC_ERR_NULL;
  if (B <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (hist != nullptr && hist_len < n_fft - 1) return HILC_ERR_SHAPE;
  if (n_fft != 64 || hop != 1 || pre_ksize != 5 || !hilc_spec_block_supported(n_fft, hop, n_fft, T)) return HILC_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(y) & 15) return HILC_ERR_UNSUPPORTED;
  SpecArgs a;
  a.wav = wav; a.hist = hist; a.hist_len = hist_len;
  a.dft = dft_packed; a.nyq = nyq_sin; a.pw = pw_packed; a.bias = bias; a.x = nullptr; a.y = y;
  a.pre_w = pre_w; a.pre_b = pre_b; a.pre_in_scale = pre_in_scale;
  a.B = B; a.T = T; a.Tf = T; a.hop = 1; a.tiles = (a.Tf + TF - 1) / TF;
  a.mean = mean; a.stdv = stdv; a.out_scale = out_scale; a.normalize = normalize;
  return launch_spec<64, true>(a, B, (hipStream_t)stream);
}
