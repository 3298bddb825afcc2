// Front-ends of the fp32-MFMA GEMM core (gemm_core.h):
//   hilc_pw_conv      pointwise conv, fused Scale/ELU prologue, bias/scale/residual epilogue
//   hilc_dws_conv     pointwise conv -> depthwise causal conv (k5 s1, or k=2r stride r) fused through
//                     LDS: the [C x T] intermediate of a depthwise-separable block never reaches HBM
//   hilc_stft_logmag  strided-window DFT (implicit im2col of the waveform) -> |.| -> log -> normalise
#include "gemm_core.h"
#include "gemm_lin.h"
#include "gemm_epilogues.h"
#include "frame1.h"

using namespace hilc;

#ifndef HILC_WR_RES
#define HILC_WR_RES 0     // 1: wave-row form also for launches with a shortcut (measured 1.4-1.9x slower: A/B builds only)
#endif

namespace {

// ================================================================================================
// B-operand loaders.  Each thread owns one 4-column group (n4 = (tid & 31) * 4) and the two k rows
// (tid >> 5) and (tid >> 5) + 8 of every BK slice.
// ================================================================================================

// columns = flattened (clip, t) axis: no halo, tiles may straddle clips (plain pointwise conv).
struct PwFlatLoader {
  const float* x;
  int K, T;
  long ncols;  // B*T
  float in_scale;
  int in_elu;
  int vec;  // T % 4 == 0 and x 16-B aligned -> a 4-column group never straddles two clips
  int tile_cols = BN;   // columns of the flattened axis per tile (< BN: whole-clip tiles of the streaming path)
  typedef f32x4 Raw;
  struct State {
    long b0, b1, b2, b3;
    bool o0, o1, o2, o3;
  };
  __device__ State init(long ntile, int tid) const {
    State s;
    const int c = (tid & 31) * 4;
    long n = ntile * tile_cols + c;
    auto col = [&](long nn, long& base, bool& ok) {
      ok = nn < ncols && (int)(nn - ntile * tile_cols) < tile_cols;
      long b = nn / T;
      base = b * (long)K * T + (nn - b * T);
    };
    col(n, s.b0, s.o0);
    s.b1 = s.b2 = s.b3 = 0; s.o1 = s.o2 = s.o3 = false;
    if (!vec) { col(n + 1, s.b1, s.o1); col(n + 2, s.b2, s.o2); col(n + 3, s.b3, s.o3); }
    return s;
  }
  // branch-free: invalid columns read x[0] (always mapped) and are zeroed by a select
  __device__ f32x4 fetch(const State& s, int k) const {
    const long ko = (long)k * T;
    if (vec)   // uniform over the launch
      return *reinterpret_cast<const f32x4*>(x + (s.o0 ? s.b0 + ko : 0));
    f32x4 v;
    v.x = x[s.o0 ? s.b0 + ko : 0];
    v.y = x[s.o1 ? s.b1 + ko : 0];
    v.z = x[s.o2 ? s.b2 + ko : 0];
    v.w = x[s.o3 ? s.b3 + ko : 0];
    return v;
  }
  __device__ f32x4 transform(const State& s, f32x4 v, int) const {
    if (vec) return prologue4v(zero_unless(s.o0, v), in_scale, in_elu);   // pro(0) == 0
    v.x = s.o0 ? v.x : 0.f; v.y = s.o1 ? v.y : 0.f; v.z = s.o2 ? v.z : 0.f; v.w = s.o3 ? v.w : 0.f;
    return prologue4v(v, in_scale, in_elu);
  }
};

// per-clip column tiles with a left halo: tile covers times [tix*step - halo, +128) of clip b;
// samples outside [0, T) read as zero (causal zero padding / the reference's right "extra" padding).
struct PwTileLoader {
  const float* x;
  int K, T, tiles, step, halo;
  float in_scale;
  int in_elu;
  int vec;  // T % 4 == 0, step % 4 == 0, halo % 4 == 0, x aligned
  typedef f32x4 Raw;
  struct State {
    long base;  // (b*K)*T + t  (t may be negative; only dereferenced when valid)
    int t;
  };
  __device__ State init(long ntile, int tid) const {
    State s;
    long b = ntile / tiles;
    int tix = (int)(ntile - b * tiles);
    s.t = tix * step - halo + (tid & 31) * 4;
    s.base = b * (long)K * T + s.t;
    return s;
  }
  __device__ bool ok(const State& s, int e) const { return s.t + e >= 0 && s.t + e < T; }
  __device__ f32x4 fetch(const State& s, int k) const {
    const long off = s.base + (long)k * T;
    if (vec)   // uniform over the launch
      return *reinterpret_cast<const f32x4*>(x + (ok(s, 0) ? off : 0));
    f32x4 v;
    v.x = x[ok(s, 0) ? off : 0];
    v.y = x[ok(s, 1) ? off + 1 : 0];
    v.z = x[ok(s, 2) ? off + 2 : 0];
    v.w = x[ok(s, 3) ? off + 3 : 0];
    return v;
  }
  __device__ f32x4 transform(const State& s, f32x4 v, int) const {
    if (vec) return prologue4v(zero_unless(ok(s, 0), v), in_scale, in_elu);
    v.x = ok(s, 0) ? v.x : 0.f; v.y = ok(s, 1) ? v.y : 0.f; v.z = ok(s, 2) ? v.z : 0.f; v.w = ok(s, 3) ? v.w : 0.f;
    return prologue4v(v, in_scale, in_elu);
  }
};

// Implicit im2col of the waveform: column = frame f of clip b (flattened), row k = sample k of the
// window, element = we[b, f*hop - (n_fft-1) + k] (history / zero for negative times).
struct StftLoader {
  const float* wav;
  const float* hist;
  int hist_len;
  int T, Tf, n_fft, hop;
  long ncols;  // B*Tf
  typedef f32x4 Raw;
  struct Col {
    long b;
    int t0;  // f*hop - (n_fft-1)
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
  __device__ State init(long ntile, int tid) const {
    long n = ntile * BN + (tid & 31) * 4;
    State s;
    s.c0 = col(n); s.c1 = col(n + 1); s.c2 = col(n + 2); s.c3 = col(n + 3);
    return s;
  }
  // raw load; columns / times that do not exist read a mapped dummy address and are zeroed in `keep`
  __device__ float at(const Col& c, int k) const {
    const int t = c.t0 + k;
    if (hist != nullptr) {   // uniform over the launch (streaming): history for t < 0
      const float* p = t >= 0 ? wav + c.b * (long)T + t : hist + c.b * (long)hist_len + hist_len + t;
      return *(c.ok ? p : wav);
    }
    return wav[(c.ok && t >= 0) ? c.b * (long)T + t : 0];
  }
  __device__ f32x4 fetch(const State& s, int k) const {
    f32x4 v = {at(s.c0, k), at(s.c1, k), at(s.c2, k), at(s.c3, k)};
    return v;
  }
  __device__ bool keep(const Col& c, int k) const { return c.ok && (hist != nullptr || c.t0 + k >= 0); }
  __device__ f32x4 transform(const State& s, f32x4 v, int k) const {
    v.x = keep(s.c0, k) ? v.x : 0.f; v.y = keep(s.c1, k) ? v.y : 0.f;
    v.z = keep(s.c2, k) ? v.z : 0.f; v.w = keep(s.c3, k) ? v.w : 0.f;
    return v;
  }
};

// Up-sampling front-end: the B operand is the depthwise transposed conv (kernel 2r, stride r, causal,
// right-trimmed) of pro(x), computed on the fly — the [K x T*r] up-sampled tensor never exists in HBM.
//   u[k][q*r+p] = w[k][p] * a[k][q] + w[k][p+r] * a[k][q-1],   a = ELU(in_scale * x),  a[-1] = 0
// Columns = flattened (clip, output time); a 4-column group touches at most q0-1, q0, q0+1.
// R = 2 / 4 / 8: the 2R taps of a channel sit in one or two aligned 16-B rows and a 4-column group
// (t % 4 == 0) uses taps p0..p0+3 (+R) with p0 a multiple of 4 — two float4 loads instead of eight
// scalar gathers; R = 0: generic stride (e.g. 5).
template <int R, bool HIST = false>
struct UpLoader {
  const float* x;      // [B][K][Tin]
  const float* w;      // [K][2r]
  const float* hist;   // streaming: [B][K] = pro(x[b,k,-1]) of the previous hop (already activated), or NULL
  int K, Tin, r;
  long ncols;          // B * Tin * r   (Tin*r % 4 == 0)
  float in_scale;
  int in_elu;
  struct Raw {
    f32x4 wa, wb;
    float xv[3];
  };
  struct State {
    long xbase;        // b*K*Tin + q0
    long hbase;        // b*K
    int q0;
    int p[4];          // phase of column e
    int dq[4];         // q_e - q0  (0 or 1)
    bool ok;
  };
  __device__ State init(long ntile, int tid) const {
    State s;
    const long n = ntile * BN + (tid & 31) * 4;
    s.ok = n < ncols;
    const long Tout = (long)Tin * r;
    const long b = n / Tout;
    const int t = (int)(n - b * Tout);
    s.q0 = t / r;
    s.xbase = b * (long)K * Tin + s.q0;
    s.hbase = s.ok ? b * (long)K : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = (t + e) / r;
      s.p[e] = (t + e) - q * r;
      s.dq[e] = q - s.q0;
    }
    return s;
  }
  __device__ Raw fetch(const State& s, int k) const {
    Raw v;
    const float* wr = w + (long)k * 2 * r;
    if (R == 8 || R == 4) {            // p = p0..p0+3, p0 in {0, 4}
      v.wa = *reinterpret_cast<const f32x4*>(wr + s.p[0]);
      v.wb = *reinterpret_cast<const f32x4*>(wr + s.p[0] + R);
    } else if (R == 2) {               // p = 0,1,0,1: taps (w0,w1 | w2,w3) of one 16-B row
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(wr);
      v.wa = f32x4{t4.x, t4.y, t4.x, t4.y};
      v.wb = f32x4{t4.z, t4.w, t4.z, t4.w};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v.wa[e] = wr[s.p[e]];
        v.wb[e] = wr[s.p[e] + r];
      }
    }
    const long row = s.ok ? s.xbase + (long)k * Tin : 0;
    if (HIST) {        // branch-free: the frame before the hop comes from the cache
      const float* p0 = s.q0 >= 1 ? x + row - 1 : hist + s.hbase + k;
      v.xv[0] = *(s.ok ? p0 : x);
    } else {
      v.xv[0] = x[(s.ok && s.q0 >= 1) ? row - 1 : 0];
    }
    v.xv[1] = x[row];
    v.xv[2] = x[(s.ok && s.q0 + 1 < Tin) ? row + 1 : 0];
    return v;
  }
  __device__ f32x4 transform(const State& s, const Raw& v, int) const {
    float a[3];
    a[0] = (s.ok && s.q0 >= 1) ? prologue(v.xv[0], in_scale, in_elu) : 0.f;
    if (HIST) a[0] = (s.ok && s.q0 < 1) ? v.xv[0] : a[0];                   // the cache holds activated samples
    a[1] = s.ok ? prologue(v.xv[1], in_scale, in_elu) : 0.f;
    a[2] = (s.ok && s.q0 + 1 < Tin) ? prologue(v.xv[2], in_scale, in_elu) : 0.f;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float cur = s.dq[e] ? a[2] : a[1];
      const float prev = s.dq[e] ? a[1] : a[0];
      o[e] = fmaf(v.wa[e], cur, v.wb[e] * prev);     // same expression as hilc_dw_convtr
    }
    return o;
  }
};

}  // namespace

extern "C" int hilc_pw_conv(const float* x, const float* wt, const float* bias, const float* res, float* y,
                            int B, int K, int M, int T, float in_scale, int in_elu, float out_scale,
                            void* stream) {
  if (!x || !wt || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (M % 4 != 0) return HILC_ERR_UNSUPPORTED;  // weight rows are read as float4
  if (T == 1 && B <= 65535L * 32) {             // single-frame layers of a streaming hop: latency-bound, own kernel
    Frame1Args f{};
    f.x = x; f.wt = wt; f.bias = bias; f.res = res; f.y = y; f.B = B; f.K = K; f.M = M;
    f.in_scale = in_scale; f.in_elu = in_elu; f.out_scale = out_scale;
    return launch_frame1(f, (hipStream_t)stream);
  }
  long ncols = (long)B * T;
  PwFlatLoader ld;
  ld.x = x; ld.K = K; ld.T = T; ld.ncols = ncols; ld.in_scale = in_scale; ld.in_elu = in_elu;
  ld.vec = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  PwEpilogue ep;
  ep.y = y; ep.bias = bias; ep.res = res; ep.M = M; ep.T = T; ep.ncols = ncols; ep.out_scale = out_scale;
  if (ld.vec && lin_ok(B, K, T)) {
    FlatCols cols;
    cols.K = K; cols.T = T; cols.tile_cols = BN; cols.ncols = ncols;
    if (ncols < (1L << 31) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15) == 0) {
      PwLdsEpilogue el;
      el.y = y; el.bias = bias; el.res = res; el.M = M; el.T = T; el.ncols = ncols; el.out_scale = out_scale;
      div_magic(T, el.t_magic, el.t_shift);
      return launch_gemm_lin(wt, x, M, K, M, T, (ncols + BN - 1) / BN, in_scale, in_elu != 0, cols, el, (hipStream_t)stream);
    }
    return launch_gemm_lin(wt, x, M, K, M, T, (ncols + BN - 1) / BN, in_scale, in_elu != 0, cols, ep, (hipStream_t)stream);
  }
  return launch_gemm(wt, M, K, M, (ncols + BN - 1) / BN, false, ld, ep, (hipStream_t)stream);
}

namespace {
// new cache of the transposed conv: pro(x[b,k,Tin-1])  (causal_layers.py:168-188: cache = last input frame)
__global__ __launch_bounds__(256) void up_hist_kernel(const float* x, float* hist_out, long n, int Tin, float in_scale,
                                                      int in_elu) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) hist_out[e] = prologue(x[e * Tin + Tin - 1], in_scale, in_elu);
}
}  // namespace

namespace {
// expanded[k][p0][e] = w[k][(p0+e) mod r], expanded[k][p0][4+e] = w[k][(p0+e) mod r + r]   (e < 4)
__global__ __launch_bounds__(256) void expand_taps_kernel(const float* w, float* out, int K, int r) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= K * r * 8) return;
  const int e = idx & 3, hi = (idx >> 2) & 1, p0 = (idx >> 3) % r, k = (idx >> 3) / r;
  out[idx] = w[(long)k * 2 * r + (p0 + e) % r + hi * r];
}
}  // namespace

extern "C" int hilc_up_conv_expand_taps(const float* tr_w, float* expanded, int K, int stride, void* stream) {
  if (!tr_w || !expanded) return HILC_ERR_NULL;
  if (K <= 0 || stride <= 0) return HILC_ERR_SHAPE;
  if (tr_w == expanded) return HILC_ERR_UNSUPPORTED;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(expand_taps_kernel, dim3((unsigned)((K * stride * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     tr_w, expanded, K, stride);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

static int up_conv_entry(const float* x, const float* hist, float* hist_out, const float* tr_w, const float* tr_w_expanded,
                         const float* wt, const float* bias, float* y, int B, int K, int M, int Tin, int stride,
                         float in_scale, int in_elu, void* stream) {
  if (!x || !tr_w || !wt || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || Tin <= 0 || stride <= 0) return HILC_ERR_SHAPE;
  if (M % 4 != 0 || ((long)Tin * stride) % 4 != 0) return HILC_ERR_UNSUPPORTED;
  const long Tout = (long)Tin * stride;
  if (Tout > 0x7fffffffL) return HILC_ERR_SHAPE;
  const long ncols = (long)B * Tout;
  PwEpilogue ep;
  ep.y = y; ep.bias = bias; ep.res = nullptr; ep.M = M; ep.T = (int)Tout; ep.ncols = ncols; ep.out_scale = 1.0f;
  const bool w_aligned = (reinterpret_cast<uintptr_t>(tr_w) & 15) == 0;
  if (hist_out != nullptr) {
    if (hist_out == hist) return HILC_ERR_UNSUPPORTED;
    const long n = (long)B * K;
    HILC_CLEAR_ERROR();
    hipLaunchKernelGGL(up_hist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, hist_out, n,
                       Tin, in_scale, in_elu);
    HILC_CHECK_LAUNCH();
  }
  auto go = [&](auto ld) {
    ld.x = x; ld.w = tr_w; ld.hist = hist; ld.K = K; ld.Tin = Tin; ld.r = stride; ld.ncols = ncols; ld.in_scale = in_scale;
    ld.in_elu = in_elu;
    return launch_gemm(wt, M, K, M, (ncols + BN - 1) / BN, false, ld, ep, (hipStream_t)stream);
  };
  // linear-addressing core (gemm_lin.h): same arithmetic, far fewer VALU per K slice.  Instantiated for what the models launch (round 6 pruning:
  // 160 -> 32 kernels): ELU in front (`seanet.py:431-436`: every up-sampling layer has one), 16-B aligned taps and output (the allocator's), the
  // strides with a vector tap path (8 / 4 / 2) or the expanded table (5).  Everything else — no ELU, a misaligned view, another stride — takes
  // the generic core below: the same fmaf chains, bit-identical.
  const bool lds_epi = ncols < (1L << 31) && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
  if (lin_ok(B, K, Tin) && in_elu && lds_epi && w_aligned) {
    PwLdsEpilogue el;
    el.y = y; el.bias = bias; el.res = nullptr; el.M = M; el.T = (int)Tout; el.ncols = ncols; el.out_scale = 1.0f;
    div_magic((int)Tout, el.t_magic, el.t_shift);
    auto lin = [&](auto bop) {
      bop.x = x; bop.w = decltype(bop)::kExpanded ? tr_w_expanded : tr_w; bop.hist = hist; bop.K = K; bop.Tin = Tin; bop.r = stride; bop.ncols = ncols;
      bop.in_scale = in_scale;
      return launch_lin(wt, M, K, M, (ncols + BN - 1) / BN, bop, el, (hipStream_t)stream);
    };
#define HILC_UP(RV) do { return hist ? lin(UpB<RV, true, true>{}) : lin(UpB<RV, true, false>{}); } while (0)
    if (stride == 8) HILC_UP(8);
    if (stride == 4) HILC_UP(4);
    if (stride == 2) HILC_UP(2);
    if (tr_w_expanded != nullptr && (reinterpret_cast<uintptr_t>(tr_w_expanded) & 15) == 0) HILC_UP(1);
#undef HILC_UP
  }
  // (the generic core keeps ONE loader per cache mode — scalar tap loads, any stride: it is the fall-back of shapes no model launches)
  if (hist != nullptr) return go(UpLoader<0, true>{});
  return go(UpLoader<0>{});
}

extern "C" int hilc_up_conv_stream(const float* x, const float* hist, float* hist_out, const float* tr_w,
                                   const float* wt, const float* bias, float* y, int B, int K, int M, int Tin,
                                   int stride, float in_scale, int in_elu, void* stream) {
  return up_conv_entry(x, hist, hist_out, tr_w, nullptr, wt, bias, y, B, K, M, Tin, stride, in_scale, in_elu, stream);
}

extern "C" int hilc_up_conv_expanded(const float* x, const float* hist, float* hist_out, const float* tr_w,
                                     const float* tr_w_expanded, const float* wt, const float* bias, float* y, int B,
                                     int K, int M, int Tin, int stride, float in_scale, int in_elu, void* stream) {
  return up_conv_entry(x, hist, hist_out, tr_w, tr_w_expanded, wt, bias, y, B, K, M, Tin, stride, in_scale, in_elu, stream);
}

extern "C" int hilc_up_conv(const float* x, const float* tr_w, const float* wt, const float* bias, float* y,
                            int B, int K, int M, int Tin, int stride, float in_scale, int in_elu, void* stream) {
  return up_conv_entry(x, nullptr, nullptr, tr_w, nullptr, wt, bias, y, B, K, M, Tin, stride, in_scale, in_elu, stream);
}

extern "C" int hilc_dws_conv(const float* x, const float* wt, const float* dw_w, const float* dw_b,
                             const float* res, float* y, int B, int K, int M, int T, int ksize, int stride,
                             float in_scale, int in_elu, float out_scale, int out_elu, void* stream) {
  if (!x || !wt || !dw_w || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || T <= 0) return HILC_ERR_SHAPE;
  if (M % 4 != 0) return HILC_ERR_UNSUPPORTED;
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  PwTileLoader ld;
  ld.x = x; ld.K = K; ld.T = T; ld.in_scale = in_scale; ld.in_elu = in_elu;
  if (stride == 1) {
    if (ksize != 5) return HILC_ERR_UNSUPPORTED;
    Dw5Epilogue ep;
    ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.res = res; ep.M = M; ep.T = T;
    ep.tiles = (T + Dw5Epilogue::STEP - 1) / Dw5Epilogue::STEP;
    ep.out_scale = out_scale; ep.out_elu = out_elu;
    ep.vec = (T % 4 == 0) && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
             (res == nullptr || (reinterpret_cast<uintptr_t>(res) & 15) == 0);
    ld.tiles = ep.tiles; ld.step = Dw5Epilogue::STEP; ld.halo = 4;
    ld.vec = aligned && (T % 4 == 0);
    if (ld.vec && lin_tile_ok(B, K, T)) {
      TileCols cols;
      cols.K = K; cols.T = T; cols.tiles = ep.tiles; cols.step = ld.step; cols.halo = ld.halo;
#ifndef HILC_NO_WAVE_ROW
      // depthwise taps on the accumulator registers (gemm_lin.h: wave-row form).  Not for launches with a shortcut: there a
      // lane would fetch its 16-B pieces of 64 different rows per load (measured: K = 384 2.1 -> 3.0 ms, L1 thrash), the
      // LDS epilogue reads the shortcut as 64-B runs per lane
      if (ep.vec && (HILC_WR_RES || res == nullptr) && wr_shape(M, (long)B * ep.tiles)) {
        auto go = [&](auto er) {
          er.y = y; er.dw_w = dw_w; er.dw_b = dw_b; er.res = res; er.M = M; er.T = T; er.tiles = ep.tiles; er.out_scale = out_scale;
          if (in_elu) {
            RowsB<TileCols, true> bo;
            bo.x = x; bo.T = T; bo.in_scale = in_scale; bo.cols = cols;
            return launch_lin_wr(wt, M, K, M, (long)B * ep.tiles, bo, er, (hipStream_t)stream);
          }
          RowsB<TileCols, false> bo;
          bo.x = x; bo.T = T; bo.in_scale = in_scale; bo.cols = cols;
          return launch_lin_wr(wt, M, K, M, (long)B * ep.tiles, bo, er, (hipStream_t)stream);
        };
#if HILC_WR_RES
        if (res != nullptr) return out_elu ? go(Dw5RegEpilogue<true, true>{}) : go(Dw5RegEpilogue<true, false>{});
#endif
        return out_elu ? go(Dw5RegEpilogue<false, true>{}) : go(Dw5RegEpilogue<false, false>{});
      }
#endif
      return launch_gemm_lin(wt, x, M, K, M, T, (long)B * ep.tiles, in_scale, in_elu != 0, cols, ep, (hipStream_t)stream);
    }
    return launch_gemm(wt, M, K, M, (long)B * ep.tiles, true, ld, ep, (hipStream_t)stream);
  }
  if (ksize != 2 * stride || stride > 16 || res != nullptr || out_elu || out_scale != 1.0f) return HILC_ERR_UNSUPPORTED;
  DwStrideEpilogue ep;
  ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.M = M; ep.r = stride;
  ep.To = (T + stride - 1) / stride;
  ep.H = (stride + 3) / 4 * 4;
  ep.n_out = (BN - ep.H - stride) / stride + 1;
  while ((ep.n_out * stride) % 4 != 0) --ep.n_out;   // keep tile starts 16-B aligned
  ep.tiles = (ep.To + ep.n_out - 1) / ep.n_out;
  ld.tiles = ep.tiles; ld.step = ep.n_out * stride; ld.halo = ep.H;
  ld.vec = aligned && (T % 4 == 0);
  if (ld.vec && lin_tile_ok(B, K, T)) {
    TileCols cols;
    cols.K = K; cols.T = T; cols.tiles = ep.tiles; cols.step = ld.step; cols.halo = ld.halo;
#ifndef HILC_NO_WAVE_ROW
    // stride 4 (the second encoder stage): taps on the accumulator registers (wave-row form), as for k5 / stride 1 — 2.22 -> 2.05 ms
    // at K = 128 -> 256, T = 12000.  (Stride 2, K = 64 -> 128: 1.68 -> 1.71 ms — a lane's 32 outputs leave as sixteen 8-B stores
    // into 64 different rows per instruction; it keeps the LDS epilogue, whose stores are whole rows.)
    if (stride == 4 && ep.H == 4 && wr_shape(M, (long)B * ep.tiles)) {
      auto go = [&](auto er) {
        er.y = y; er.dw_w = dw_w; er.dw_b = dw_b; er.M = M; er.To = ep.To; er.tiles = ep.tiles; er.n_out = ep.n_out;
        if (in_elu) {
          RowsB<TileCols, true> bo;
          bo.x = x; bo.T = T; bo.in_scale = in_scale; bo.cols = cols;
          return launch_lin_wr(wt, M, K, M, (long)B * ep.tiles, bo, er, (hipStream_t)stream);
        }
        RowsB<TileCols, false> bo;
        bo.x = x; bo.T = T; bo.in_scale = in_scale; bo.cols = cols;
        return launch_lin_wr(wt, M, K, M, (long)B * ep.tiles, bo, er, (hipStream_t)stream);
      };
      return go(DwStrideRegEpilogue<4>{});
    }
#endif
    return launch_gemm_lin(wt, x, M, K, M, T, (long)B * ep.tiles, in_scale, in_elu != 0, cols, ep, (hipStream_t)stream);
  }
  return launch_gemm(wt, M, K, M, (long)B * ep.tiles, true, ld, ep, (hipStream_t)stream);
}

// which tile form hilc_dws_conv (ksize 5, stride 1, 16-B-lane path) takes for a shape: 1 = wave-row form with the depthwise taps
// on the accumulator registers, 0 = column-block form with the LDS epilogue (tests pin one against the other bit for bit)
extern "C" int hilc_dws_conv_wave_row(int B, int M, int T, int has_res) {
#ifdef HILC_NO_WAVE_ROW
  return 0;
#else
  if (B <= 0 || M <= 0 || T <= 0 || T % 4 != 0) return 0;
  const long tiles = (T + Dw5Epilogue::STEP - 1) / Dw5Epilogue::STEP;
  return ((HILC_WR_RES || !has_res) && wr_shape(M, (long)B * tiles)) ? 1 : 0;
#endif
}

extern "C" int hilc_dws_conv_stream(const float* x, const float* wt, const float* dw_w, const float* dw_b,
                                    const float* hist, float* hist_out, const float* res, float* y, int B, int K,
                                    int M, int T, int ksize, int stride, float in_scale, int in_elu,
                                    float out_scale, int out_elu, void* stream) {
  if (!x || !wt || !dw_w || !y) return HILC_ERR_NULL;
  if (B <= 0 || K <= 0 || M <= 0 || T <= 0 || ksize <= 0 || stride <= 0) return HILC_ERR_SHAPE;
  if (M % 4 != 0 || T % stride != 0 || ksize < stride || ksize > 32) return HILC_ERR_UNSUPPORTED;
  if (hist != nullptr && hist == hist_out) return HILC_ERR_UNSUPPORTED;
  const int pad = ksize - stride;
  if (pad == 0 && (hist != nullptr || hist_out != nullptr)) return HILC_ERR_SHAPE;
  if (T == 1 && stride == 1 && B <= 65535L * 32) {
    Frame1Args f{};
    f.x = x; f.wt = wt; f.res = res; f.y = y; f.dw_w = dw_w; f.dw_b = dw_b; f.hist = hist; f.hist_out = hist_out;
    f.B = B; f.K = K; f.M = M; f.ksize = ksize; f.in_scale = in_scale; f.in_elu = in_elu; f.out_scale = out_scale;
    f.out_elu = out_elu;
    return launch_frame1(f, (hipStream_t)stream);
  }
  if (T > BN) {
    // a hop longer than one tile (the encoder's first two down-sampling layers: 320 / 160 samples per stream): per-clip
    // tiles with a recomputed halo, exactly the offline kernel, plus the cache in front of the first tile and the new
    // cache out of the last one
    if (ksize != 2 * stride || stride > 16 || out_elu || out_scale != 1.0f) return HILC_ERR_UNSUPPORTED;
    if (T % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || !lin_ok(B, K, T)) return HILC_ERR_UNSUPPORTED;
    const int H = (stride + 3) / 4 * 4;
    int n_out = (BN - H - stride) / stride + 1;
    while ((n_out * stride) % 4 != 0) --n_out;
    const long nout_all = (long)B * (T / stride);
    if (nout_all + T < (1L << 31) && (long)B * M * (T / stride) < (1L << 31) && stride <= DwStrideFlatEpilogue::HB &&
        (T / stride) * DwStrideFlatEpilogue::NBND >= n_out + 1) {
      // flat columns (stream-major): a tile is 97 % full whatever the hop length (per-stream tiles: 62 % at T = 160, 86 % at 320)
      DwStrideFlatEpilogue ep;
      ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.hist = hist; ep.hist_out = hist_out; ep.res = res;
      ep.B = B; ep.M = M; ep.T = T; ep.To = T / stride; ep.r = stride; ep.H = H; ep.n_out = n_out;
      div_magic(ep.To, ep.to_magic, ep.to_shift);
      FlatHaloCols fc;
      fc.K = K; fc.T = T; fc.step = n_out * stride; fc.halo = H; fc.ncols = (long)B * T;
      const long ntiles = (nout_all + n_out - 1) / n_out;
      return launch_gemm_lin(wt, x, M, K, M, T, ntiles, in_scale, in_elu != 0, fc, ep, (hipStream_t)stream);
    }
    if (res != nullptr) return HILC_ERR_UNSUPPORTED;     // (per-stream tiles: the fallback for > 2^31 outputs has no shortcut)
    DwStrideEpilogue ep;
    ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.M = M; ep.r = stride; ep.hist = hist; ep.hist_out = hist_out; ep.T = T;
    ep.To = T / stride;
    ep.H = H;
    ep.n_out = n_out;
    ep.tiles = (ep.To + ep.n_out - 1) / ep.n_out;
    TileCols tc;
    tc.K = K; tc.T = T; tc.tiles = ep.tiles; tc.step = ep.n_out * stride; tc.halo = ep.H;
    return launch_gemm_lin(wt, x, M, K, M, T, (long)B * ep.tiles, in_scale, in_elu != 0, tc, ep, (hipStream_t)stream);
  }
  const int cpt = BN / T;
  PwFlatLoader ld;
  ld.x = x; ld.K = K; ld.T = T; ld.ncols = (long)B * T; ld.in_scale = in_scale; ld.in_elu = in_elu;
  ld.vec = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  ld.tile_cols = cpt * T;
  // n / d for n < 2^31 as __umulhi(n, magic) >> shift (Granlund-Montgomery): l = ceil(log2 d), magic = ceil(2^(31+l)/d)
  auto magic = [](int d, unsigned& m, unsigned& sh) {
    int l = 0;
    while ((1L << l) < d) ++l;
    if (l < 1) l = 1;
    m = (unsigned)(((1ULL << (31 + l)) + (unsigned long long)d - 1) / (unsigned long long)d);
    sh = (unsigned)(l - 1);
  };
  const long ntiles = ((long)B + cpt - 1) / cpt;
  const bool al = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(hist) |
                    reinterpret_cast<uintptr_t>(hist_out)) & 15) == 0;
  FlatCols cols;
  cols.K = K; cols.T = T; cols.tile_cols = cpt * T; cols.ncols = (long)B * T;
  const bool lin = ld.vec && lin_ok(B, K, T);
  if (ksize == 5 && stride == 1 && T % 4 == 0 && al) {
    Dw5SegEpilogue ep;
    ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.res = res; ep.hist = hist; ep.hist_out = hist_out; ep.B = B; ep.M = M;
    ep.T = T; ep.cpt = cpt; ep.out_scale = out_scale; ep.out_elu = out_elu;
    magic(T, ep.t_magic, ep.t_shift);
    if (lin) return launch_gemm_lin(wt, x, M, K, M, T, ntiles, in_scale, in_elu != 0, cols, ep, (hipStream_t)stream);
    return launch_gemm(wt, M, K, M, ntiles, true, ld, ep, (hipStream_t)stream);
  }
  DwSegEpilogue ep;
  ep.y = y; ep.dw_w = dw_w; ep.dw_b = dw_b; ep.res = res; ep.hist = hist; ep.hist_out = hist_out; ep.B = B; ep.M = M;
  ep.T = T; ep.To = T / stride; ep.k = ksize; ep.stride = stride; ep.pad = pad; ep.cpt = cpt;
  ep.out_scale = out_scale; ep.out_elu = out_elu;
  magic(ep.To, ep.to_magic, ep.to_shift);
  magic(pad > 0 ? pad : 1, ep.pad_magic, ep.pad_shift);
  if (lin) return launch_gemm_lin(wt, x, M, K, M, T, ntiles, in_scale, in_elu != 0, cols, ep, (hipStream_t)stream);
  return launch_gemm(wt, M, K, M, ntiles, true, ld, ep, (hipStream_t)stream);
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
  // offline, long clips: per-clip tiles whose waveform segment is staged once in LDS (gemm_lin.h, StftSegB)
  const int seg_len = ((BN - 1) * hop + n_fft) * 17 / 16 + 1;   // with one pad word per 16 samples
  if (hist == nullptr && n_fft % BK == 0 && Tf >= 4 * BN && seg_len <= 6144) {
    const int tiles = (Tf + BN - 1) / BN;
    StftClipEpilogue ec;
    ec.spec = spec; ec.nbins = n_fft / 2 + 1; ec.Tf = Tf; ec.tiles = tiles; ec.mean = mean; ec.stdv = stdv;
    ec.normalize = normalize;
    auto go = [&](auto bop) {
      bop.wav = wav; bop.T = T; bop.Tf = Tf; bop.n_fft = n_fft; bop.hop = hop; bop.tiles = tiles;
      return launch_lin(basis_t, M, n_fft, m_pad, (long)B * tiles, bop, ec, (hipStream_t)stream);
    };
    if (seg_len <= 512) return go(StftSegB<512>{});
    if (seg_len <= 1536) return go(StftSegB<1536>{});
    return go(StftSegB<6144>{});
  }
  return launch_gemm(basis_t, M, n_fft, m_pad, (ncols + BN - 1) / BN, false, ld, ep, (hipStream_t)stream);
}

#ifdef HILC_DEBUG_STAMPS
extern "C" int hilc_debug_set_lin_stamp_buffer(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(hilc::g_lin_dbg), &p, sizeof(p)) == hipSuccess ? HILC_OK : HILC_ERR_LAUNCH;
}
#endif
