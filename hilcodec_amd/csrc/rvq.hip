// Residual vector quantiser for the HILCodec hot path (gfx950).
//
// hilc_rvq_encode: one workgroup (256 threads) owns FR=16 consecutive frames and walks the n
// stages in order, keeping the running residual and the running quantised sum in LDS.  Per stage
// every thread scores 4 of the K=1024 code vectors against all 16 frames with fp32 FMA chains in
// fixed channel order (c = 0..C-1), reading the codebook through its transposed copy `[C][K]`
// (consecutive lanes = consecutive codes = coalesced) and the residual as broadcast
// ds_read_b128 rows `[c][16 frames]`.  The distance is `|e|^2 - 2<r,e>` (same ordering of
// candidates as `-2 r@e^T + |e|^2`, models/hilcodec/vector_quantize.py:146-152); the arg-min keeps
// the LOWEST index among equal distances (torch CPU `min(dim)` behaviour) through a
// lexicographic (distance, index) wave64 shuffle reduction followed by a 4-wave LDS reduction.
// Batches of 8 192 frames and more score on the matrix pipe instead (rvq_encode_mfma_kernel below): the same chains, the same indices.
#include <stdlib.h>

#include "common.h"

namespace {

// frames per workgroup: 16 for large batches; 4 when 16 would leave most CUs idle (a streaming hop of 1024 frames
// = 64 workgroups otherwise) — the code scores are independent of FR and of the threads per workgroup, only the work split changes
constexpr int RS = 20;   // LDS row stride (floats) of the [c][frame] tiles: 16-B aligned, spreads banks
constexpr int CPT = 4;   // codes per thread (K / 256 for K = 1024)

struct RvqArgs {
  const float* z;
  const float* cb;    // [Nq][K][C]
  const float* cbt;   // [Nq][C][K]
  const float* norms; // [Nq][K]
  const int* n_clip;  // optional [B]: stages used by clip b (mixed-bitrate batches); NULL = n for every clip
  int64_t* indices;
  float* q;
  float* frame_err;
  int B, C, T, K, n;
  int channel_last, stage_major;
};

__device__ __forceinline__ long zoff(const RvqArgs& a, long g, int c) {
  if (a.channel_last) return g * a.C + c;
  long b = g / a.T;
  long t = g - b * a.T;
  return (b * a.C + c) * (long)a.T + t;
}

// NTH threads per workgroup, CPTK = K / NTH codes per thread: 256 x 4 for large batches; a streaming hop (1024 frames, FR = 4: one
// workgroup per CU) is bound by the L2 round trips of the code words, and 1024 x 1 puts four waves on every SIMD to hide them
template <int C, int FR, int NTH = 256>
__global__ __launch_bounds__(NTH) void rvq_encode_kernel(RvqArgs a) {
  constexpr int CPTK = 1024 / NTH, NWV = NTH / 64;
  static_assert(FR % 4 == 0 && FR <= RS, "frames are read as float4 rows");
  __shared__ __attribute__((aligned(16))) float res[C][RS];
  __shared__ __attribute__((aligned(16))) float qsum[C][RS];
  __shared__ float wbest[NWV][FR];
  __shared__ int widx[NWV][FR];
  __shared__ int sel[FR];
  __shared__ int nfr[FR];   // stages of each frame's clip

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const long nframes = (long)a.B * a.T;
  const long g0 = (long)blockIdx.x * FR;

  for (int e = tid; e < FR * C; e += NTH) {
    int f = e % FR, c = e / FR;
    long g = g0 + f;
    res[c][f] = g < nframes ? a.z[zoff(a, g, c)] : 0.f;
    qsum[c][f] = 0.f;
  }
  if (tid < FR) {
    long g = g0 + tid;
    int nf = a.n;
    if (a.n_clip != nullptr && g < nframes) {
      nf = a.n_clip[g / a.T];
      nf = nf < 1 ? 1 : (nf > a.n ? a.n : nf);
    }
    nfr[tid] = nf;
  }
  __syncthreads();

  for (int s = 0; s < a.n; ++s) {
    const float* cbt = a.cbt + (long)s * C * a.K;
    const float* cb = a.cb + (long)s * a.K * C;
    const float* nrm = a.norms + (long)s * a.K;

    float acc[CPTK][FR];
#pragma unroll
    for (int j = 0; j < CPTK; ++j)
#pragma unroll
      for (int f = 0; f < FR; ++f) acc[j][f] = 0.f;

    // Code words come from L2 (the transposed codebook, 512 KB per stage, is shared by every workgroup): UC channels'
    // worth of loads are issued before the first FMA of the group needs one.  With FR = 4 a CU holds one wave per
    // SIMD and nothing else hides the L2 round trip, so the group is deep (64 loads in flight per thread); with
    // FR = 16 the 64 accumulators leave room for 2 channels.
        constexpr int UC = FR <= 4 ? 16 : 2;          // (512 threads: 8 -> 113 us, 16 -> 96 us, 32 -> 132 us)
    static_assert(C % UC == 0, "channel groups");
    for (int c0 = 0; c0 < C; c0 += UC) {
      float e[UC][CPTK];
#pragma unroll
      for (int u = 0; u < UC; ++u)
#pragma unroll
        for (int j = 0; j < CPTK; ++j) e[u][j] = cbt[(long)(c0 + u) * a.K + tid + NTH * j];
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        float r[FR];
#pragma unroll
        for (int f4 = 0; f4 < FR; f4 += 4) {
          float4 v = *reinterpret_cast<const float4*>(&res[c0 + u][f4]);
          r[f4] = v.x; r[f4 + 1] = v.y; r[f4 + 2] = v.z; r[f4 + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < CPTK; ++j)
#pragma unroll
          for (int f = 0; f < FR; ++f) acc[j][f] = fmaf(r[f], e[u][j], acc[j][f]);
      }
    }

    float nk[CPTK];
#pragma unroll
    for (int j = 0; j < CPTK; ++j) nk[j] = nrm[tid + NTH * j];

#pragma unroll
    for (int f = 0; f < FR; ++f) {
      float best = fmaf(-2.f, acc[0][f], nk[0]);
      int bi = tid;
#pragma unroll
      for (int j = 1; j < CPTK; ++j) {
        float d = fmaf(-2.f, acc[j][f], nk[j]);
        if (d < best) { best = d; bi = tid + NTH * j; }   // ascending index order, strict <
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        float od = __shfl_xor(best, m, 64);
        int oi = __shfl_xor(bi, m, 64);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
      }
      if (lane == 0) { wbest[wave][f] = best; widx[wave][f] = bi; }
    }
    __syncthreads();
    if (tid < FR) {
      float best = wbest[0][tid];
      int bi = widx[0][tid];
#pragma unroll
      for (int w = 1; w < NWV; ++w) {
        float od = wbest[w][tid];
        int oi = widx[w][tid];
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
      }
      if (s >= nfr[tid]) bi = -1;   // this clip stops before stage s: no code, residual and sum untouched
      sel[tid] = bi;
      long g = g0 + tid;
      if (g < nframes) {
        long b = g / a.T, t = g - b * a.T;
        long off = a.stage_major ? ((long)s * a.B + b) * a.T + t : (b * a.n + s) * (long)a.T + t;
        a.indices[off] = bi;
      }
    }
    __syncthreads();
    // residual -= E[idx]; quantized_out += E[idx]   (vector_quantize.py:225-229)
    for (int e = tid; e < FR * C; e += NTH) {
      int c = e % C, f = e / C;
      const int k = sel[f];
      if (k >= 0) {
        float qv = cb[(long)k * C + c];
        res[c][f] = res[c][f] - qv;
        qsum[c][f] = qsum[c][f] + qv;
      }
    }
    __syncthreads();
  }

  for (int e = tid; e < FR * C; e += NTH) {
    int f = e % FR, c = e / FR;
    long g = g0 + f;
    if (g < nframes && a.q != nullptr) a.q[zoff(a, g, c)] = qsum[c][f];
  }
  if (a.frame_err != nullptr && tid < FR) {
    long g = g0 + tid;
    if (g < nframes) {
      float err = 0.f;
      for (int c = 0; c < C; ++c) {
        float d = a.z[zoff(a, g, c)] - qsum[c][tid];
        err = fmaf(d, d, err);
      }
      a.frame_err[g] = err;
    }
  }
}

// ---- large batches: the scores on the matrix pipe --------------------------------------------------------------------------------
// One workgroup (4 waves) owns 32 frames; per stage the score matrix <e_k, r_f> is a [1024 codes] x [32 frames] GEMM over the 128
// channels on v_mfma_f32_32x32x2_f32: wave w scores codes 256 w .. 256 w + 255 as eight 32-row tiles, tile i's row m = code
// 256 w + 8 m + i — so the A operand of a k-pair (channels 2p, 2p + 1) is two 16-B words per lane straight from the TRANSPOSED
// codebook `[C][K]` (a half-wave reads 1 KB contiguous), no packed copy — and the B operand is the residual tile in LDS.
// The fp32 MFMA accumulates exactly like an fmaf chain over k in ascending order, the products commute: every score has the bits
// of rvq_encode_kernel's `fmaf(r, e, acc)` chain, so distances, arg-mins and therefore indices are identical (tests/test_gpu_rvq.py).
// The arg-min visits a lane's 128 codes in ascending order (strict <), then combines lexicographically (distance, index) across the
// two half-waves and the four waves: the lowest index among equal distances, like torch's `min(dim)`.
template <int C>
__global__ __launch_bounds__(256, 2) void rvq_encode_mfma_kernel(RvqArgs a) {
  constexpr int FR = 32, NTH = 256, NWV = 4, XS = 33, DEPTH = 4;     // XS: row stride of the [c][frame] tiles (column accesses conflict-free)
  __shared__ float res[C * XS];
  __shared__ float qsum[C * XS];
  __shared__ float wbest[NWV][FR];
  __shared__ int widx[NWV][FR];
  __shared__ int sel[FR];
  __shared__ int nfr[FR];
  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nframes = (long)a.B * a.T;
  const long g0 = (long)blockIdx.x * FR;

  for (int e = tid; e < FR * C; e += NTH) {
    int f, c;
    if (a.channel_last) { c = e % C; f = e / C; } else { f = e % FR; c = e / FR; }      // coalesced either way
    const long g = g0 + f;
    res[c * XS + f] = g < nframes ? a.z[zoff(a, g, c)] : 0.f;
    qsum[c * XS + f] = 0.f;
  }
  if (tid < FR) {
    const long g = g0 + tid;
    int nf = a.n;
    if (a.n_clip != nullptr && g < nframes) {
      nf = a.n_clip[g / a.T];
      nf = nf < 1 ? 1 : (nf > a.n ? a.n : nf);
    }
    nfr[tid] = nf;
  }
  __syncthreads();

  typedef const __attribute__((address_space(1))) f32x4* gvec_t;
  for (int s = 0; s < a.n; ++s) {
    const float* cb = a.cb + (long)s * a.K * C;
    const float* nrm = a.norms + (long)s * a.K + 256 * wave;
    // this lane's eight codes of channel row 2p + h: cbt[s][2p + h][256 w + 8 l31 .. + 7]
    const float* ap = a.cbt + (long)s * C * a.K + (long)h * a.K + 256 * wave + 8 * l31;
    const float* bp = res + h * XS + l31;
    f32x16 acc[8];
    f32x4 wa[DEPTH][2];
    float b[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
      wa[d][0] = *(gvec_t)(ap + (long)(2 * d) * a.K);
      wa[d][1] = *(gvec_t)(ap + (long)(2 * d) * a.K + 4);
      b[d] = bp[2 * d * XS];
    }
#pragma unroll
    for (int p = 0; p < C / 2; ++p) {
      const int cur = p % DEPTH, nxt = (p + DEPTH - 1) % DEPTH;
      if (p + DEPTH - 1 < C / 2) {
        wa[nxt][0] = *(gvec_t)(ap + (long)(2 * (p + DEPTH - 1)) * a.K);
        wa[nxt][1] = *(gvec_t)(ap + (long)(2 * (p + DEPTH - 1)) * a.K + 4);
        b[nxt] = bp[2 * (p + DEPTH - 1) * XS];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (p == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][i >> 2][i & 3], b[cur], acc[i], 0, 0, 0);
      }
      // pin the k-pair (see gemm_phase in resblock_kernel.h: the builtins are pure and would be sunk below every later load)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(acc[i]) :: "memory");
    }

    // arg-min over this lane's 128 codes, ascending: MFMA row m = 8 (r >> 2) + (r & 3) + 4 h of tile i is code 256 w + 8 m + i
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 8 * (r >> 2) + (r & 3) + 4 * h;
      const f32x4 n0 = *(gvec_t)(nrm + 8 * m), n1 = *(gvec_t)(nrm + 8 * m + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = fmaf(-2.f, acc[i][r], i < 4 ? n0[i & 3] : n1[i & 3]);
        const int code = 256 * wave + 8 * m + i;
        if (r == 0 && i == 0) { best = d; bi = code; }
        else if (d < best) { best = d; bi = code; }          // ascending index order, strict <
      }
    }
    {
      const float od = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(bi, 32, 64);
      if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    if (h == 0) { wbest[wave][l31] = best; widx[wave][l31] = bi; }
    __syncthreads();
    if (tid < FR) {
      float bb = wbest[0][tid];
      int bj = widx[0][tid];
#pragma unroll
      for (int w = 1; w < NWV; ++w) {
        const float od = wbest[w][tid];
        const int oi = widx[w][tid];
        if (od < bb || (od == bb && oi < bj)) { bb = od; bj = oi; }
      }
      if (s >= nfr[tid]) bj = -1;   // this clip stops before stage s: no code, residual and sum untouched
      sel[tid] = bj;
      const long g = g0 + tid;
      if (g < nframes) {
        const long bq = g / a.T, t = g - bq * a.T;
        const long off = a.stage_major ? ((long)s * a.B + bq) * a.T + t : (bq * a.n + s) * (long)a.T + t;
        a.indices[off] = bj;
      }
    }
    __syncthreads();
    // residual -= E[idx]; quantized_out += E[idx]   (vector_quantize.py:225-229)
    for (int e = tid; e < FR * C; e += NTH) {
      const int c = e % C, f = e / C;
      const int k = sel[f];
      if (k >= 0) {
        const float qv = cb[(long)k * C + c];
        res[c * XS + f] = res[c * XS + f] - qv;
        qsum[c * XS + f] = qsum[c * XS + f] + qv;
      }
    }
    __syncthreads();
  }

  if (a.q != nullptr) {
    for (int e = tid; e < FR * C; e += NTH) {
      int f, c;
      if (a.channel_last) { c = e % C; f = e / C; } else { f = e % FR; c = e / FR; }
      const long g = g0 + f;
      if (g < nframes) a.q[zoff(a, g, c)] = qsum[c * XS + f];
    }
  }
  if (a.frame_err != nullptr && tid < FR) {
    const long g = g0 + tid;
    if (g < nframes) {
      float err = 0.f;
      for (int c = 0; c < C; ++c) {
        const float d = a.z[zoff(a, g, c)] - qsum[c * XS + tid];
        err = fmaf(d, d, err);
      }
      a.frame_err[g] = err;
    }
  }
}

__global__ __launch_bounds__(256) void mse_finalize_kernel(const float* frame_err, float* loss, int frames,
                                                           double count) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < frames; i += 256) s += (double)frame_err[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if (threadIdx.x < m) part[threadIdx.x] += part[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(part[0] / count);
}

struct DeqArgs {
  const int64_t* indices;
  const float* cb;
  const int* n_clip;   // optional [B]
  float* q;
  int B, C, T, K, n;
  int channel_last, stage_major;
};

__global__ __launch_bounds__(256) void rvq_decode_kernel(DeqArgs a) {
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  long total = (long)a.B * a.T * a.C;
  if (e >= total) return;
  long b, t;
  int c;
  if (a.channel_last) {  // e = (b*T + t)*C + c
    c = (int)(e % a.C);
    long g = e / a.C;
    b = g / a.T; t = g - b * a.T;
  } else {               // e = (b*C + c)*T + t
    t = e % a.T;
    long bc = e / a.T;
    c = (int)(bc % a.C);
    b = bc / a.C;
  }
  float acc = 0.f;
  int nb = a.n;
  if (a.n_clip != nullptr) { nb = a.n_clip[b]; nb = nb < 1 ? 1 : (nb > a.n ? a.n : nb); }
  for (int s = 0; s < nb; ++s) {
    long ioff = a.stage_major ? ((long)s * a.B + b) * a.T + t : (b * a.n + s) * (long)a.T + t;
    long k = a.indices[ioff];
    k = k < 0 ? 0 : (k >= a.K ? a.K - 1 : k);
    acc = acc + a.cb[((long)s * a.K + k) * a.C + c];
  }
  a.q[e] = acc;
}


// ---- training-side EMA statistics (models/hilcodec/vector_quantize.py:155-172) ----------------
// bucket[s] = [ num_curr (K) | embed_curr (K*C) ] with num_curr[k] = #frames whose stage-s code is k and
// embed_curr[k] = sum of those frames' stage-s INPUT residuals (z - sum_{j<s} E_j[idx_j], subtracted in stage
// order exactly like the encoder) — the reference's `cat([onehot.sum(0), (onehot.T @ flatten).view(-1)])`, laid
// out for ONE all-reduce over all stages.  One wave per (stage, code): it scans the stage's indices 64 frames
// at a time (coalesced), and adds the matching frames in ascending frame order, so the result is deterministic
// (no atomics).  Lane l owns channels l and l + 64.
struct EmaStatsArgs {
  const float* z;
  const float* cb;          // [Nq][K][C] (pre-update tables, the ones the indices were computed with)
  const int64_t* indices;
  float* bucket;            // [n][K + K*C]
  int B, C, T, K, n, rows;  // rows = stage dimension of `indices`
  int channel_last, stage_major;
};

__global__ __launch_bounds__(256) void rvq_ema_stats_kernel(EmaStatsArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 4 + wave;
  const int s = blockIdx.y;
  if (k >= a.K) return;
  const long nframes = (long)a.B * a.T;
  float acc0 = 0.f, acc1 = 0.f;
  long count = 0;
  for (long g0 = 0; g0 < nframes; g0 += 64) {
    const long g = g0 + lane;
    bool hit = false;
    if (g < nframes) {
      const long b = g / a.T, t = g - b * a.T;
      const long off = a.stage_major ? ((long)s * a.B + b) * a.T + t : (b * a.rows + s) * (long)a.T + t;
      hit = a.indices[off] == k;
    }
    unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    count += __builtin_popcountll(m);
    while (m) {
      const int f = __builtin_ctzll(m);
      m &= m - 1;
      const long gf = g0 + f;
      const long b = gf / a.T, t = gf - b * a.T;
      float r0, r1;
      if (a.channel_last) {
        r0 = a.z[gf * a.C + lane];
        r1 = a.z[gf * a.C + lane + 64];
      } else {
        r0 = a.z[(b * a.C + lane) * (long)a.T + t];
        r1 = a.z[(b * a.C + lane + 64) * (long)a.T + t];
      }
      for (int j = 0; j < s; ++j) {
        const long off = a.stage_major ? ((long)j * a.B + b) * a.T + t : (b * a.rows + j) * (long)a.T + t;
        const long kj = a.indices[off];
        const float* e = a.cb + ((long)j * a.K + kj) * a.C;
        r0 = r0 - e[lane];
        r1 = r1 - e[lane + 64];
      }
      acc0 = acc0 + r0;
      acc1 = acc1 + r1;
    }
  }
  float* out = a.bucket + (long)s * (a.K + (long)a.K * a.C);
  if (lane == 0) out[k] = (float)count;
  out[a.K + (long)k * a.C + lane] = acc0;
  out[a.K + (long)k * a.C + lane + 64] = acc1;
}

// ema_num = ema_num*decay + num*(1-decay); ema_embed likewise; embed = ema_embed / ema_num   (ema_inplace +
// vector_quantize.py:165-169).  One thread per (stage, code, channel) reads the OLD ema_num; a second, stream-
// ordered launch then updates ema_num itself (one thread per code) — writing it here would race with the other
// channels' reads.
struct EmaUpdateArgs {
  float* embed;
  float* ema_num;
  float* ema_embed;
  const float* bucket;
  float decay, alpha;
  int K, C, n;
};

__global__ __launch_bounds__(256) void rvq_ema_update_kernel(EmaUpdateArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)a.K * a.C;
  if (e >= per * a.n) return;
  const int s = (int)(e / per);
  const long kc = e - s * per;
  const int k = (int)(kc / a.C);
  const float* bk = a.bucket + (long)s * (a.K + per);
  const float en = fmaf(bk[k], a.alpha, __fmul_rn(a.ema_num[(long)s * a.K + k], a.decay));
  const float ee = fmaf(bk[a.K + kc], a.alpha, __fmul_rn(a.ema_embed[e], a.decay));
  a.ema_embed[e] = ee;
  a.embed[e] = __fdiv_rn(ee, en);
}

__global__ __launch_bounds__(256) void rvq_ema_num_kernel(EmaUpdateArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)a.K * a.n) return;
  const int s = (int)(e / a.K), k = (int)(e - (long)s * a.K);
  const float* bk = a.bucket + (long)s * (a.K + (long)a.K * a.C);
  a.ema_num[e] = fmaf(bk[k], a.alpha, __fmul_rn(a.ema_num[e], a.decay));
}

}  // namespace

extern "C" int hilc_rvq_encode_mixed(const float* z, const float* codebooks, const float* codebooks_t,
                                     const float* norms, const int* n_per_clip, int64_t* indices, float* q,
                                     float* frame_err, int B, int C, int T, int K, int Nq, int n, int channel_last,
                                     int stage_major, int flags, void* stream) {
  if (!z || !codebooks || !codebooks_t || !norms || !indices) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || Nq <= 0) return HILC_ERR_SHAPE;
  if (n < 1 || n > Nq) return HILC_ERR_RANGE;
  if (C != 128 || K != 256 * CPT || (flags & ~HILC_RVQ_VALU_ONLY) != 0) return HILC_ERR_UNSUPPORTED;
  RvqArgs a;
  a.z = z; a.cb = codebooks; a.cbt = codebooks_t; a.norms = norms; a.n_clip = n_per_clip; a.indices = indices; a.q = q;
  a.frame_err = frame_err; a.B = B; a.C = C; a.T = T; a.K = K; a.n = n;
  a.channel_last = channel_last; a.stage_major = stage_major;
  long nframes = (long)B * T;
  HILC_CLEAR_ERROR();
  // small batches (a streaming hop: 1024 frames): 4 frames per workgroup = one workgroup per CU, 512 threads x 2 codes = two waves per
  // SIMD to hide the L2 round trips of the code words (256 x 4: 143 us, 512 x 2: 96 us, 1024 x 1: 107 us; 8 / 16 frames per
  // workgroup: 240 / 369 us — profiles/r04_experiments.md)
  // large batches: the score GEMM on the matrix pipe, 32 frames per workgroup (two workgroups per CU)
  // flags & HILC_RVQ_VALU_ONLY keeps large batches on the VALU form (16 frames per workgroup): the switch a caller has if a future
  // part's fp32 MFMA should ever stop accumulating as a sequential fmaf chain over ascending k — which is what makes the two
  // forms, hence offline and streaming indices, identical; tests/test_gpu_rvq.py pins that on every box it runs on, and runs
  // this branch too.  (An argument, not an environment variable: the library keeps no process-global switches.)
  const bool valu_only = (flags & HILC_RVQ_VALU_ONLY) != 0;
  if (nframes >= 32 * 256 && !valu_only)
    hipLaunchKernelGGL((rvq_encode_mfma_kernel<128>), dim3((unsigned)((nframes + 31) / 32)), dim3(256), 0, (hipStream_t)stream, a);
  else if (nframes <= 16 * 512)
    hipLaunchKernelGGL((rvq_encode_kernel<128, 4, 512>), dim3((unsigned)((nframes + 3) / 4)), dim3(512), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((rvq_encode_kernel<128, 16>), dim3((unsigned)((nframes + 15) / 16)), dim3(256), 0, (hipStream_t)stream, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_rvq_encode(const float* z, const float* codebooks, const float* codebooks_t,
                               const float* norms, int64_t* indices, float* q, float* frame_err, int B, int C,
                               int T, int K, int Nq, int n, int channel_last, int stage_major, int flags, void* stream) {
  return hilc_rvq_encode_mixed(z, codebooks, codebooks_t, norms, nullptr, indices, q, frame_err, B, C, T, K, Nq, n,
                               channel_last, stage_major, flags, stream);
}

extern "C" int hilc_mse_finalize(const float* frame_err, float* loss, int frames, double count, void* stream) {
  if (!frame_err || !loss) return HILC_ERR_NULL;
  if (frames <= 0 || count <= 0) return HILC_ERR_SHAPE;
  HILC_CLEAR_ERROR(); hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, frame_err, loss, frames, count);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_rvq_decode_mixed(const int64_t* indices, const float* codebooks, const int* n_per_clip, float* q,
                                     int B, int C, int T, int K, int Nq, int n, int channel_last, int stage_major,
                                     void* stream) {
  if (!indices || !codebooks || !q) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || Nq <= 0) return HILC_ERR_SHAPE;
  if (n < 1 || n > Nq) return HILC_ERR_RANGE;
  DeqArgs a;
  a.indices = indices; a.cb = codebooks; a.n_clip = n_per_clip; a.q = q; a.B = B; a.C = C; a.T = T; a.K = K; a.n = n;
  a.channel_last = channel_last; a.stage_major = stage_major;
  long total = (long)B * T * C;
  HILC_CLEAR_ERROR(); hipLaunchKernelGGL(rvq_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_rvq_decode(const int64_t* indices, const float* codebooks, float* q, int B, int C, int T, int K,
                               int Nq, int n, int channel_last, int stage_major, void* stream) {
  return hilc_rvq_decode_mixed(indices, codebooks, nullptr, q, B, C, T, K, Nq, n, channel_last, stage_major, stream);
}

extern "C" int hilc_rvq_ema_stats(const float* z, const float* codebooks, const int64_t* indices, float* bucket,
                                  int B, int C, int T, int K, int n, int index_rows, int channel_last,
                                  int stage_major, void* stream) {
  if (!z || !codebooks || !indices || !bucket) return HILC_ERR_NULL;
  if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || n <= 0 || index_rows < n) return HILC_ERR_SHAPE;
  if (C != 128) return HILC_ERR_UNSUPPORTED;
  EmaStatsArgs a;
  a.z = z; a.cb = codebooks; a.indices = indices; a.bucket = bucket; a.B = B; a.C = C; a.T = T; a.K = K; a.n = n;
  a.rows = index_rows; a.channel_last = channel_last; a.stage_major = stage_major;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(rvq_ema_stats_kernel, dim3((unsigned)((K + 3) / 4), (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

extern "C" int hilc_rvq_ema_update(float* embed, float* ema_num, float* ema_embed, const float* bucket, double decay,
                                   int K, int C, int n, void* stream) {
  if (!embed || !ema_num || !ema_embed || !bucket) return HILC_ERR_NULL;
  if (K <= 0 || C <= 0 || n <= 0) return HILC_ERR_SHAPE;
  EmaUpdateArgs a;
  a.embed = embed; a.ema_num = ema_num; a.ema_embed = ema_embed; a.bucket = bucket;
  a.decay = (float)decay; a.alpha = (float)(1.0 - decay); a.K = K; a.C = C; a.n = n;
  const long total = (long)n * K * C;
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL(rvq_ema_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HILC_CHECK_LAUNCH();
  hipLaunchKernelGGL(rvq_ema_num_kernel, dim3((unsigned)(((long)n * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

thread_local int hilc_last_hip_error_code = 0;

extern "C" const char* hilc_last_hip_error(void) {
  return hipGetErrorString((hipError_t)hilc_last_hip_error_code);
}

extern "C" int hilc_abi_version(void) { return HILC_ABI_VERSION; }

extern "C" const char* hilc_error_string(int code) {
  switch (code) {
    case HILC_OK: return "ok";
    case HILC_ERR_SHAPE: return "bad shape (a dimension is <= 0 or inconsistent)";
    case HILC_ERR_NULL: return "required pointer is NULL";
    case HILC_ERR_LAUNCH: return "HIP kernel launch failed";
    case HILC_ERR_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
    case HILC_ERR_RANGE: return "'n' must be in range of 1 <= n <= num_quantizers";
    default: return "unknown error";
  }
}
