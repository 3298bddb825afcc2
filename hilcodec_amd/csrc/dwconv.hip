// HBM-bound kernels of the HILCodec hot path (gfx950): depthwise causal conv (k5 s1 and the
// strided down-sampling k=2r), depthwise transposed conv (up-sampling), the Cin=1 / Cout=1 edge
// convs and L2Norm.  Layout `[B][C][T]`, time contiguous: consecutive lanes walk consecutive
// samples (coalesced), the k5 path moves float4 per lane.
#include "common.h"

namespace {

constexpr int MAXK = 16;

struct DwArgs {
  const float* x;
  const float* hist;
  const float* w;
  const float* bias;
  const float* res;
  float* y;
  int C, Cin, T, To, ksize, stride, pad, hist_len;
  float in_scale;
  int in_elu;
  float out_scale;
  int out_elu;
  int hist_raw;   // the history holds RAW input samples (the waveform cache of the first conv): they take the prologue
                  // too; 0 = it holds prologue'd samples (the depthwise caches)
};

// extended input: history for t < 0 (zero if none), prologue'd x for 0 <= t < T, zero beyond.
__device__ __forceinline__ float xe(const DwArgs& a, const float* xrow, const float* hrow, int t) {
  if (t >= 0) return t < a.T ? prologue(xrow[t], a.in_scale, a.in_elu) : 0.f;
  if (hrow != nullptr) return a.hist_raw ? prologue(hrow[a.hist_len + t], a.in_scale, a.in_elu) : hrow[a.hist_len + t];
  return 0.f;
}

__device__ __forceinline__ float dw_post(const DwArgs& a, float acc, int c, long off) {
  // separate roundings, like `y.mul_(scale).add_(shortcut)` (seanet.py:148) — no FMA contraction
  if (a.bias != nullptr) acc = __fadd_rn(acc, a.bias[c]);
  acc = __fmul_rn(acc, a.out_scale);
  if (a.res != nullptr) acc = __fadd_rn(acc, a.res[off]);
  if (a.out_elu) acc = elu_fast(acc);
  return acc;
}

// generic: one thread per output sample over the flattened (row = b*C + c, o) index, so that the
// few-sample rows of a streaming hop still fill whole wavefronts.
__global__ __launch_bounds__(256) void dw_generic_kernel(DwArgs a, long total) {
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  long row = e / a.To;
  int o = (int)(e - row * a.To);
  int c = (int)(row % a.C);
  long b = row / a.C;
  long inrow = a.Cin == 1 ? b : row;
  const float* xrow = a.x + inrow * (long)a.T;
  const float* hrow = a.hist ? a.hist + inrow * (long)a.hist_len : nullptr;
  const float* wr = a.w + (long)c * a.ksize;
  float acc = 0.f;
  int t0 = o * a.stride - a.pad;
  for (int j = 0; j < a.ksize; ++j) acc = fmaf(wr[j], xe(a, xrow, hrow, t0 + j), acc);
  long off = row * (long)a.To + o;
  a.y[off] = dw_post(a, acc, c, off);
}

// k == 5, stride 1, T % 4 == 0, 16-B aligned rows: 4 outputs per thread from two float4 loads;
// flattened (row, t/4) index.
__global__ __launch_bounds__(256) void dw_k5_vec_kernel(DwArgs a, long total) {
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int T4 = a.T >> 2;
  long row = e / T4;
  int t = (int)(e - row * T4) * 4;
  int c = (int)(row % a.C);
  long b = row / a.C;
  long inrow = a.Cin == 1 ? b : row;
  const float* xrow = a.x + inrow * (long)a.T;
  const float* hrow = a.hist ? a.hist + inrow * (long)a.hist_len : nullptr;
  float v[8];
  float4 cur = prologue4(*reinterpret_cast<const float4*>(xrow + t), a.in_scale, a.in_elu);
  v[4] = cur.x; v[5] = cur.y; v[6] = cur.z; v[7] = cur.w;
  if (t >= 4) {
    float4 prev = prologue4(*reinterpret_cast<const float4*>(xrow + t - 4), a.in_scale, a.in_elu);
    v[0] = prev.x; v[1] = prev.y; v[2] = prev.z; v[3] = prev.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = hrow ? hrow[a.hist_len - 4 + j] : 0.f;
      if (a.hist_raw && hrow) v[j] = prologue(v[j], a.in_scale, a.in_elu);
    }
  }
  float wk[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) wk[j] = a.w[(long)c * 5 + j];
  float out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float acc = 0.f;
    // output t+i reads inputs t+i-4 .. t+i  -> v[i+j], taps in order j = 0..4
#pragma unroll
    for (int j = 0; j < 5; ++j) acc = fmaf(wk[j], v[i + j], acc);
    long off = row * (long)a.T + t + i;
    out[i] = dw_post(a, acc, c, off);
  }
  *reinterpret_cast<float4*>(a.y + row * (long)a.T + t) = make_float4(out[0], out[1], out[2], out[3]);
}

// new cache = last `pad` samples of [hist | pro(x)]; one thread per (row, i)
__global__ void hist_out_kernel(const float* x, const float* hist, float* hist_out, long rows, int T, int pad,
                                int hist_len, float in_scale, int in_elu) {
  long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= rows * pad) return;
  long row = g / pad;
  int i = (int)(g - row * pad);
  int t = T - pad + i;
  float v;
  if (t >= 0) v = prologue(x[row * (long)T + t], in_scale, in_elu);
  else v = hist ? hist[row * (long)hist_len + hist_len + t] : 0.f;
  hist_out[row * (long)pad + i] = v;
}

struct TrArgs {
  const float* x;
  const float* hist;
  const float* w;
  float* y;
  int C, T, r;
  float in_scale;
  int in_elu;
};

__global__ __launch_bounds__(256) void dw_convtr_kernel(TrArgs a, long total) {
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  int To = a.T * a.r;
  long row = e / To;
  int n = (int)(e - row * To);
  int c = (int)(row % a.C);
  int q = n / a.r, p = n - q * a.r;
  const float* xrow = a.x + row * (long)a.T;
  float cur = prologue(xrow[q], a.in_scale, a.in_elu);
  float prev = q > 0 ? prologue(xrow[q - 1], a.in_scale, a.in_elu) : (a.hist ? a.hist[row] : 0.f);
  const float* wr = a.w + (long)c * 2 * a.r;
  a.y[row * (long)To + n] = fmaf(wr[p], cur, wr[p + a.r] * prev);
}

struct PostArgs {
  const float* x;
  const float* hist;
  const float* w;
  const float* bias;
  float* y;
  int C, T, ksize;
  float in_scale;
  int in_elu;
  float out_scale;
  int do_tanh;
  unsigned tblocks;   // blocks of 256 samples per clip; blockIdx.x = b * tblocks + tb (flat: no 65535 limit on either)
};

// Cout = 1: one thread per output sample, channels reduced sequentially (c = 0..C-1, then taps).
__global__ __launch_bounds__(256) void conv_post_kernel(PostArgs a) {
  const long b = blockIdx.x / a.tblocks;
  const unsigned tb = blockIdx.x - (unsigned)b * a.tblocks;
  int t = tb * 256 + threadIdx.x;
  if (t >= a.T) return;
  const int pad = a.ksize - 1;
  float acc = 0.f;
  for (int c = 0; c < a.C; ++c) {
    const float* xrow = a.x + (b * a.C + c) * (long)a.T;
    const float* hrow = a.hist ? a.hist + (b * a.C + c) * (long)pad : nullptr;
    const float* wr = a.w + (long)c * a.ksize;
    for (int j = 0; j < a.ksize; ++j) {
      int ti = t - pad + j;
      float v;
      if (ti >= 0) v = prologue(xrow[ti], a.in_scale, a.in_elu);
      else v = hrow ? hrow[pad + ti] : 0.f;
      acc = fmaf(wr[j], v, acc);
    }
  }
  if (a.bias) acc = __fadd_rn(acc, a.bias[0]);
  acc = __fmul_rn(acc, a.out_scale);
  if (a.do_tanh) acc = tanhf(acc);
  a.y[b * (long)a.T + t] = acc;
}

// Fast path (k = 5, T % 4 == 0; offline, or a streaming hop with its [B][C][4] cache): 64 lanes x 4 samples = 256 samples per block;
// the block's 8 waves split the channels in classes c mod 8 so 8x more loads are in flight, each lane moves 16 B, each sample's
// prologue (ELU) is evaluated once; partial sums meet in LDS.  The order of the reduction is c ascending within a class, then class
// 0..7 — fixed, and the same as the closing phase of hilc_decoder_stage_post (csrc/resblock_kernel.h, phase Q), whose half-waves walk
// exactly these classes: the two forms agree bit for bit.
__global__ __launch_bounds__(512) void conv_post_k5_kernel(PostArgs a) {
  constexpr int NCLS = 8;
  __shared__ float part[NCLS][256];
  const long b = blockIdx.x / a.tblocks;
  const unsigned tb = blockIdx.x - (unsigned)b * a.tblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = (tb * 64 + lane) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (t < a.T) {
    for (int c = wave; c < a.C; c += NCLS) {
      const float* xrow = a.x + (b * a.C + c) * (long)a.T;
      const f32x4 cur = prologue4v(*reinterpret_cast<const f32x4*>(xrow + t), a.in_scale, a.in_elu);
      f32x4 prev = {0.f, 0.f, 0.f, 0.f};
      if (t >= 4) prev = prologue4v(*reinterpret_cast<const f32x4*>(xrow + t - 4), a.in_scale, a.in_elu);
      else if (a.hist != nullptr) prev = *reinterpret_cast<const f32x4*>(a.hist + (b * a.C + c) * 4);   // cache: activated samples
      const float v[8] = {prev.x, prev.y, prev.z, prev.w, cur.x, cur.y, cur.z, cur.w};
      float w[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) w[j] = a.w[c * 5 + j];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[e] = fmaf(w[j], v[e + j], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[wave][lane * 4 + e] = acc[e];
  __syncthreads();
  const int tt = tb * 256 + threadIdx.x;
  if (threadIdx.x < 256 && tt < a.T) {
    float s = part[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < NCLS; ++q) s = __fadd_rn(s, part[q][threadIdx.x]);
    if (a.bias) s = __fadd_rn(s, a.bias[0]);
    s = __fmul_rn(s, a.out_scale);
    if (a.do_tanh) s = tanhf(s);
    a.y[b * (long)a.T + tt] = s;
  }
}

// One thread per frame (b, t), flat over B * T (a streaming hop has T = 1: one block per stream would run one thread per block).
// The sum of squares stays ONE fmaf chain over c ascending — the order every earlier build used, so z and the RVQ indices do not
// move — but the operands are requested 16 channels at a time before the chain consumes them: a frame costs C / 16 memory round
// trips instead of C (T = 1, C = 128, 1024 streams: 39 -> ~6 us).
__global__ __launch_bounds__(64) void l2norm_kernel(const float* x, float* y, int C, int T, float eps, float scale,
                                                    int channel_last_out, long frames) {
  const long f = (long)blockIdx.x * 64 + threadIdx.x;
  if (f >= frames) return;
  const long b = f / T;
  const int t = (int)(f - b * T);
  const float* xb = x + b * (long)C * T + t;
  constexpr int CH = 16;
  float ss = 0.f;
  int c0 = 0;
  for (; c0 + CH <= C; c0 += CH) {
    float v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = xb[(long)(c0 + i) * T];
#pragma unroll
    for (int i = 0; i < CH; ++i) ss = fmaf(v[i], v[i], ss);
  }
  for (; c0 < C; ++c0) {
    const float v = xb[(long)c0 * T];
    ss = fmaf(v, v, ss);
  }
  const float denom = fmaxf(sqrtf(ss), eps);  // F.normalize: x / max(||x||, eps)
  for (c0 = 0; c0 + CH <= C; c0 += CH) {
    float v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = xb[(long)(c0 + i) * T];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const float o = __fmul_rn(__fdiv_rn(v[i], denom), scale);
      if (channel_last_out) y[f * (long)C + c0 + i] = o;
      else y[b * (long)C * T + (long)(c0 + i) * T + t] = o;
    }
  }
  for (; c0 < C; ++c0) {
    const float o = __fmul_rn(__fdiv_rn(xb[(long)c0 * T], denom), scale);
    if (channel_last_out) y[f * (long)C + c0] = o;
    else y[b * (long)C * T + (long)c0 * T + t] = o;
  }
}

int launch_hist_out(const float* x, const float* hist, float* hist_out, long rows, int T, int pad, int hist_len,
                    float in_scale, int in_elu, hipStream_t s) {
  if (pad <= 0) return HILC_OK;
  long n = rows * pad;
  HILC_CLEAR_ERROR(); hipLaunchKernelGGL(hist_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, hist, hist_out, rows, T,
                     pad, hist_len, in_scale, in_elu);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

int launch_dw(DwArgs a, int B, hipStream_t s) {
  long rows = (long)B * a.C;
  if (rows > 0x7fffffffL) return HILC_ERR_SHAPE;
  bool vec = a.stride == 1 && a.ksize == 5 && a.T % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 && (a.hist == nullptr || a.hist_len >= 4);
  if (vec) {
    long total = rows * (a.T / 4);
    if ((total + 255) / 256 > 0x7fffffffL) return HILC_ERR_SHAPE;
    HILC_CLEAR_ERROR(); hipLaunchKernelGGL(dw_k5_vec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, total);
  } else {
    long total = rows * a.To;
    if ((total + 255) / 256 > 0x7fffffffL) return HILC_ERR_SHAPE;
    HILC_CLEAR_ERROR(); hipLaunchKernelGGL(dw_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, total);
  }
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

}  // namespace

extern "C" int hilc_dw_conv(const float* x, const float* hist, const float* w, const float* bias, const float* res,
                            float* y, float* hist_out, int B, int C, int T, int ksize, int stride, float in_scale,
                            int in_elu, float out_scale, int out_elu, void* stream) {
  if (!x || !w || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0 || ksize <= 0 || stride <= 0) return HILC_ERR_SHAPE;
  if (ksize > MAXK || ksize < stride) return HILC_ERR_UNSUPPORTED;
  DwArgs a;
  a.x = x; a.hist = hist; a.w = w; a.bias = bias; a.res = res; a.y = y;
  a.C = C; a.Cin = C; a.T = T; a.To = (T + stride - 1) / stride; a.ksize = ksize; a.stride = stride;
  a.pad = (ksize - 1) - (stride - 1); a.hist_len = a.pad;
  a.in_scale = in_scale; a.in_elu = in_elu; a.out_scale = out_scale; a.out_elu = out_elu; a.hist_raw = 0;
  int rc = launch_dw(a, B, (hipStream_t)stream);
  if (rc != HILC_OK) return rc;
  if (hist_out) return launch_hist_out(x, hist, hist_out, (long)B * C, T, a.pad, a.pad, in_scale, in_elu, (hipStream_t)stream);
  return HILC_OK;
}

extern "C" int hilc_conv_pre(const float* wav, const float* hist, int hist_len, const float* w, const float* bias,
                             float* y, int B, int C, int T, int ksize, float in_scale, void* stream) {
  if (!wav || !w || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0 || ksize <= 0) return HILC_ERR_SHAPE;
  if (ksize > MAXK) return HILC_ERR_UNSUPPORTED;
  if (hist && hist_len < ksize - 1) return HILC_ERR_SHAPE;
  DwArgs a;
  a.x = wav; a.hist = hist; a.w = w; a.bias = bias; a.res = nullptr; a.y = y;
  a.C = C; a.Cin = 1; a.T = T; a.To = T; a.ksize = ksize; a.stride = 1; a.pad = ksize - 1;
  a.hist_len = hist ? hist_len : a.pad;
  a.in_scale = in_scale; a.in_elu = 0; a.out_scale = 1.f; a.out_elu = 0; a.hist_raw = 1;
  return launch_dw(a, B, (hipStream_t)stream);
}

extern "C" int hilc_dw_convtr(const float* x, const float* hist, const float* w, float* y, float* hist_out, int B,
                              int C, int T, int stride, float in_scale, int in_elu, void* stream) {
  if (!x || !w || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0 || stride <= 0) return HILC_ERR_SHAPE;
  long rows = (long)B * C;
  if (rows > 0x7fffffffL) return HILC_ERR_SHAPE;
  TrArgs a;
  a.x = x; a.hist = hist; a.w = w; a.y = y; a.C = C; a.T = T; a.r = stride; a.in_scale = in_scale; a.in_elu = in_elu;
  long total = rows * (long)T * stride;
  if ((total + 255) / 256 > 0x7fffffffL) return HILC_ERR_SHAPE;
  HILC_CLEAR_ERROR(); hipLaunchKernelGGL(dw_convtr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, total);
  HILC_CHECK_LAUNCH();
  if (hist_out) return launch_hist_out(x, hist, hist_out, rows, T, 1, 1, in_scale, in_elu, (hipStream_t)stream);
  return HILC_OK;
}

extern "C" int hilc_conv_post(const float* x, const float* hist, const float* w, const float* bias, float* y,
                              float* hist_out, int B, int C, int T, int ksize, float in_scale, int in_elu,
                              float out_scale, int do_tanh, void* stream) {
  if (!x || !w || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0 || ksize <= 0) return HILC_ERR_SHAPE;
  PostArgs a;
  a.x = x; a.hist = hist; a.w = w; a.bias = bias; a.y = y; a.C = C; a.T = T; a.ksize = ksize;
  a.in_scale = in_scale; a.in_elu = in_elu; a.out_scale = out_scale; a.do_tanh = do_tanh;
  a.tblocks = (unsigned)ceil_div(T, 256);
  if ((long)B * a.tblocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  dim3 grid((unsigned)((long)B * a.tblocks));
  const bool fast = ksize == 5 && T % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (hist == nullptr || (reinterpret_cast<uintptr_t>(hist) & 15) == 0);
  HILC_CLEAR_ERROR();
  if (fast) hipLaunchKernelGGL(conv_post_k5_kernel, grid, dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  HILC_CHECK_LAUNCH();
  if (hist_out)
    return launch_hist_out(x, hist, hist_out, (long)B * C, T, ksize - 1, ksize - 1, in_scale, in_elu, (hipStream_t)stream);
  return HILC_OK;
}

extern "C" int hilc_tail(const float* x, const float* hist, float* out, long rows, int T, int pad, int hist_len,
                         void* stream) {
  if (!x || !out) return HILC_ERR_NULL;
  if (rows <= 0 || T <= 0 || pad <= 0) return HILC_ERR_SHAPE;
  if (hist && hist_len < pad - T) return HILC_ERR_SHAPE;
  return launch_hist_out(x, hist, out, rows, T, pad, hist_len, 1.0f, 0, (hipStream_t)stream);
}

extern "C" int hilc_l2norm(const float* x, float* y, int B, int C, int T, float eps, float scale,
                           int channel_last_out, void* stream) {
  if (!x || !y) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0) return HILC_ERR_SHAPE;
  const long frames = (long)B * T;
  const long blocks = (frames + 63) / 64;
  if (blocks > 0x7fffffffL) return HILC_ERR_SHAPE;
  HILC_CLEAR_ERROR(); hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, x, y, C, T, eps, scale, channel_last_out, frames);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}
