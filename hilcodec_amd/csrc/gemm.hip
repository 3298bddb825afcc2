// fp32-MFMA GEMM core for the HILCodec hot path (gfx950) and its two front-ends:
//   * hilc_pw_conv      — pointwise (1x1) convolution with fused Scale/ELU prologue and
//                         bias / scale / residual epilogue,
//   * hilc_stft_logmag  — strided-window DFT (implicit im2col of the waveform) with a fused
//                         magnitude -> log -> normalise epilogue.
//
// Tiling.  One workgroup = 256 threads = 4 waves computes a (32*MB) x 128 output tile; wave w
// owns columns [32w, 32w+32) and all MB row-blocks, so its accumulators are MB f32x16 registers
// fed by v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an fmaf chain in k order, k = 0..K-1).
// K is streamed in BK=16 slices, double-buffered in LDS with register prefetch:
//   A slice  wt[k][m]   (folded weights, pre-transposed on the host -> coalesced float4 rows)
//   B slice  X[k][n]    produced by a Loader functor (global activations, time contiguous)
// Operand fetch: lane l reads A[k=2j+(l>>5)][m=l&31], B[k=2j+(l>>5)][n=l&31] with ds_read_b32;
// each 32-lane half touches 32 consecutive dwords -> conflict-free without padding.
#include "common.h"

namespace {

constexpr int BN = 128;
constexpr int BK = 16;
constexpr int NT = 256;

template <int MB>
struct Smem {
  float A[2][BK][32 * MB];
  float B[2][BK][BN];
};

// ------------------------------------------------------------------------------------------------
// B-operand loaders.  Each thread owns one 4-column group (n4 = (tid & 31) * 4) and the two
// k rows (tid >> 5) and (tid >> 5) + 8 of every BK slice.
// ------------------------------------------------------------------------------------------------
struct PwLoader {
  const float* x;
  int K, T;
  long ncols;  // B*T, columns are the flattened (b, t) axis
  float in_scale;
  int in_elu;
  int vec;  // T % 4 == 0 -> a 4-column group never straddles two clips and is 16-B aligned
  struct State {
    long b0, b1, b2, b3;
    bool o0, o1, o2, o3;
  };

  __device__ State init(long n0, int tid) const {
    State s;
    long n = n0 + (tid & 31) * 4;
    auto col = [&](long nn, long& base, bool& ok) {
      ok = nn < ncols;
      long b = nn / T;
      base = b * (long)K * T + (nn - b * T);
    };
    col(n, s.b0, s.o0);
    s.b1 = s.b2 = s.b3 = 0; s.o1 = s.o2 = s.o3 = false;
    if (!vec) { col(n + 1, s.b1, s.o1); col(n + 2, s.b2, s.o2); col(n + 3, s.b3, s.o3); }
    return s;
  }
  __device__ float4 fetch(const State& s, int k) const {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) {
      if (vec) {
        if (s.o0) v = prologue4(*reinterpret_cast<const float4*>(x + s.b0 + (long)k * T), in_scale, in_elu);
      } else {
        if (s.o0) v.x = prologue(x[s.b0 + (long)k * T], in_scale, in_elu);
        if (s.o1) v.y = prologue(x[s.b1 + (long)k * T], in_scale, in_elu);
        if (s.o2) v.z = prologue(x[s.b2 + (long)k * T], in_scale, in_elu);
        if (s.o3) v.w = prologue(x[s.b3 + (long)k * T], in_scale, in_elu);
      }
    }
    return v;
  }
};

// Implicit im2col of the waveform: column = frame f of clip b, row k = sample k of the window,
// element = we[b, f*hop - (n_fft-1) + k] (history / zero for negative times).
struct StftLoader {
  const float* wav;
  const float* hist;
  int hist_len;
  int T, Tf, n_fft, hop;
  long ncols;  // B*Tf
  struct Col {
    long b;   // clip
    int t0;   // f*hop - (n_fft-1)
    bool ok;
  };
  struct State { Col c0, c1, c2, c3; };

  __device__ Col col(long nn) const {
    Col c;
    c.ok = nn < ncols;
    c.b = nn / Tf;
    int f = (int)(nn - c.b * Tf);
    c.t0 = f * hop - (n_fft - 1);
    return c;
  }
  __device__ State init(long n0, int tid) const {
    long n = n0 + (tid & 31) * 4;
    State s;
    s.c0 = col(n); s.c1 = col(n + 1); s.c2 = col(n + 2); s.c3 = col(n + 3);
    return s;
  }
  __device__ float at(const Col& c, int k) const {
    if (!c.ok) return 0.f;
    int t = c.t0 + k;
    if (t >= 0) return wav[c.b * (long)T + t];
    if (hist != nullptr) return hist[c.b * (long)hist_len + hist_len + t];
    return 0.f;
  }
  __device__ float4 fetch(const State& s, int k) const {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < n_fft) {
      v.x = at(s.c0, k);
      v.y = at(s.c1, k);
      v.z = at(s.c2, k);
      v.w = at(s.c3, k);
    }
    return v;
  }
};

// ------------------------------------------------------------------------------------------------
// Epilogues.  C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// ------------------------------------------------------------------------------------------------
struct PwEpilogue {
  float* y;
  const float* bias;
  const float* res;
  int M, T;
  long ncols;
  float out_scale;

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], int m0, long n0, int wave, int lane) const {
    long n = n0 + wave * 32 + (lane & 31);
    if (n >= ncols) return;
    long b = n / T;
    long colbase = b * (long)M * T + (n - b * T);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) {
          long off = colbase + (long)row * T;
          float v = acc[i][r];
          // separate roundings, like the reference's `y.mul_(scale)` then `x.add_(y)` (no FMA contraction)
          if (bias != nullptr) v = __fadd_rn(v, bias[row]);
          v = __fmul_rn(v, out_scale);
          if (res != nullptr) v = __fadd_rn(v, res[off]);
          y[off] = v;
        }
      }
    }
  }
};

// rows come in (re, im) pairs: row 2k = cos_k, row 2k+1 = sin_k; regs (2j, 2j+1) of a lane hold a pair.
struct StftEpilogue {
  float* spec;
  int nbins, Tf;  // nbins = n_fft/2+1
  long ncols;
  float mean, stdv;
  int normalize;

  template <int MB>
  __device__ void run(const f32x16 (&acc)[MB], int m0, long n0, int wave, int lane) const {
    long n = n0 + wave * 32 + (lane & 31);
    if (n >= ncols) return;
    long b = n / Tf;
    long colbase = b * (long)nbins * Tf + (n - b * Tf);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int bin = row >> 1;
        if (bin < nbins) {
          float re = acc[i][r], im = acc[i][r + 1];
          // x.square().sum(dim=1).clamp_min(1e-12).sqrt()  (conv.py:357) — no FMA contraction
          float p = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
          float v = sqrtf(fmaxf(p, 1e-12f));
          if (normalize != 2) v = logf(fmaxf(v, 1e-5f));  // seanet.py:228
          if (normalize == 1) v = __fdiv_rn(__fsub_rn(v, mean), stdv);  // seanet.py:236
          spec[colbase + (long)bin * Tf] = v;
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// The kernel
// ------------------------------------------------------------------------------------------------
template <int MB, class Loader, class Epilogue>
__global__ __launch_bounds__(NT) void gemm_kernel(const float* __restrict__ wt, int M, int K, int ldw,
                                                  long ntiles, int mtiles, Loader ld, Epilogue ep) {
  constexpr int BM = 32 * MB;
  constexpr int AG = BK * BM / 4;            // float4 groups in an A slice
  constexpr int AP = (AG + NT - 1) / NT;     // per-thread passes
  __shared__ Smem<MB> sm;

  // XCD-aware tile order: the `mtiles` row-tiles that share one B column-tile get block ids that
  // are congruent mod 8 (observed: block b runs on XCD b % 8), so the shared activations stay in
  // one XCD's L2.  Placement only changes speed, never results.
  long id = blockIdx.x;
  long grp = id / (8L * mtiles);
  int within = (int)(id - grp * 8L * mtiles);
  long ntile = grp * 8 + (within & 7);
  int mtile = within >> 3;
  if (ntile >= ntiles) return;
  const int m0 = mtile * BM;
  const long n0 = ntile * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const typename Loader::State ls = ld.init(n0, tid);

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 ra[AP], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      int g = tid + p * NT;
      int k = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
      ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g < AG && (k0 + k) < K && (m0 + m4) < ldw)
        ra[p] = *reinterpret_cast<const float4*>(wt + (long)(k0 + k) * ldw + m0 + m4);
    }
    rb[0] = ld.fetch(ls, k0 + (tid >> 5));
    rb[1] = ld.fetch(ls, k0 + (tid >> 5) + 8);
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      int g = tid + p * NT;
      if (g < AG) {
        int k = g / (BM / 4), m4 = (g % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(&sm.A[buf][k][m4]) = ra[p];
      }
    }
    *reinterpret_cast<float4*>(&sm.B[buf][tid >> 5][(tid & 31) * 4]) = rb[0];
    *reinterpret_cast<float4*>(&sm.B[buf][(tid >> 5) + 8][(tid & 31) * 4]) = rb[1];
  };

  const int ktiles = (K + BK - 1) / BK;
  const int live = min(MB, (M - m0 + 31) / 32);
  fetch(0);
  stage(0);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ktiles) fetch((kt + 1) * BK);
#pragma unroll
    for (int j = 0; j < BK / 2; ++j) {
      const int kk = 2 * j + (lane >> 5);
      const float bv = sm.B[buf][kk][wave * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        if (i < live) {  // wave-uniform: skips the dead 32-row blocks of a partial last row tile
          const float av = sm.A[buf][kk][i * 32 + (lane & 31)];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < ktiles) {
      stage(buf ^ 1);
      __syncthreads();
    }
  }
  ep.template run<MB>(acc, m0, n0, wave, lane);
}

template <class Loader, class Epilogue>
int launch_gemm(const float* wt, int M, int K, int ldw, long ncols, const Loader& ld, const Epilogue& ep,
                hipStream_t s) {
  // row-tile height (in 32-row MFMA blocks)
  int m32 = (M + 31) / 32;
  int MB;  // dead blocks of a partial last tile are skipped in the kernel, so prefer tall tiles
  if (m32 % 4 == 0) MB = 4;
  else if (m32 % 3 == 0) MB = 3;
  else if (m32 < 4) MB = m32;
  else MB = 4;
  int mtiles = (m32 + MB - 1) / MB;
  long ntiles = (ncols + BN - 1) / BN;
  long groups = (ntiles + 7) / 8;
  long blocks = groups * 8 * mtiles;
  if (blocks <= 0 || blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  dim3 grid((unsigned)blocks), block(NT);
  switch (MB) {
    case 1: HILC_CLEAR_ERROR(); hipLaunchKernelGGL((gemm_kernel<1, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
    case 2: HILC_CLEAR_ERROR(); hipLaunchKernelGGL((gemm_kernel<2, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
    case 3: HILC_CLEAR_ERROR(); hipLaunchKernelGGL((gemm_kernel<3, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
    default: HILC_CLEAR_ERROR(); hipLaunchKernelGGL((gemm_kernel<4, Loader, Epilogue>), grid, block, 0, s, wt, M, K, ldw, ntiles, mtiles, ld, ep); break;
  }
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

}  // namespace

extern "C" int hilc_pw_conv(const float* x, const float* wt, const float* bias, const float* res, float* y,
                            int B, int K, int M, int T, float in_scale, int in_elu, float out_scale,
                            void* stream) {
  if (!x || !wt || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (M % 4 != 0) return HILC_ERR_UNSUPPORTED;  // weight rows are read as float4
  long ncols = (long)B * T;
  PwLoader ld;
  ld.x = x; ld.K = K; ld.T = T; ld.ncols = ncols; ld.in_scale = in_scale; ld.in_elu = in_elu;
  ld.vec = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  PwEpilogue ep;
  ep.y = y; ep.bias = bias; ep.res = res; ep.M = M; ep.T = T; ep.ncols = ncols; ep.out_scale = out_scale;
  return launch_gemm(wt, M, K, M, ncols, ld, ep, (hipStream_t)stream);
}

extern "C" int hilc_stft_logmag(const float* wav, const float* hist, int hist_len, const float* basis_t,
                                float* spec, int B, int T, int n_fft, int hop, float mean, float stdv,
                                int normalize, void* stream) {
  if (!wav || !basis_t || !spec) return HILC_ERR_NULL;
  if (B <= 0 || T <= 0 || n_fft < 2 || (n_fft & 1) || hop <= 0) return HILC_ERR_SHAPE;
  if (hist != nullptr && hist_len < n_fft - 1) return HILC_ERR_SHAPE;
  int Tf = (T - 1) / hop + 1;
  int M = n_fft + 2;
  int m_pad = ((M + 31) / 32) * 32;
  long ncols = (long)B * Tf;
  StftLoader ld;
  ld.wav = wav; ld.hist = hist; ld.hist_len = hist_len; ld.T = T; ld.Tf = Tf; ld.n_fft = n_fft; ld.hop = hop;
  ld.ncols = ncols;
  StftEpilogue ep;
  ep.spec = spec; ep.nbins = n_fft / 2 + 1; ep.Tf = Tf; ep.ncols = ncols; ep.mean = mean; ep.stdv = stdv;
  ep.normalize = normalize;
  return launch_gemm(basis_t, M, n_fft, m_pad, ncols, ld, ep, (hipStream_t)stream);
}
