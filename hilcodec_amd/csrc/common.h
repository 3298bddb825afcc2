// Shared device helpers for the HILCodec gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hilcodec_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// hipGetLastError() also reports (and clears) errors left behind by unrelated earlier runtime calls
// on this thread (e.g. the host framework's own probing), so every launcher clears it first.
#define HILC_CLEAR_ERROR() (void)hipGetLastError()

extern thread_local int hilc_last_hip_error_code;   // rvq.hip

#define HILC_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) {                                  \
      hilc_last_hip_error_code = (int)e__;                    \
      return HILC_ERR_LAUNCH;                                 \
    }                                                         \
  } while (0)

// Hot-path ELU: max(x, 2^(min(x,0)*log2 e) - 1) with the hardware v_exp_f32 (1 ulp): 3 VALU + the transcendental
// instead of ~32 + a divergent branch for expm1f.  For x > 0 the exponential term is exactly 0 and the max returns x
// itself; for x <= 0, e^x - 1 >= x.  |elu_fast - expm1| <= 1.2e-7 ABSOLUTE (the rounding of e ~ 1, i.e. one fp32 ulp
// of an O(1) activation; tests/test_gpu_ops.py::test_elu_fast_error bounds it on a dense grid) — relative accuracy
// near 0- is given up, which a following dot product cannot see.  Justified by data, not taste: the full-size parity
// census (profiles/r02_parity_census_final.json) finds 0 index flips in 67 200 argmins with this form AND with expm1f.
// (max instead of compare + select: in the VALU-bound phases every instruction is ~2.8 cycles of the pipe the fp32
// MFMAs need, profiles/r02_mfma_shadow_microbench.txt.)
__device__ __forceinline__ float elu_fast(float x) {
#ifdef HILC_ELU_EXPM1
  return x > 0.0f ? x : expm1f(x);   // A/B build for the parity census (tools/census_run.sh): torch's own ELU form
#endif
  // min(x, 0) is not an instruction: exp2(x * log2 e) >= 1 exactly when x >= 0, so clamping the EXPONENTIAL to [0, 1]
  // (med3 with 0 and 1 folds into the clamp bit of v_exp_f32: free) gives the same bits as exp2(min(x, 0) * log2 e)
  // for every x (x <= 0: the clamp is the identity; x > 0: both are exactly 1) — 3 VALU + the transcendental
  const float e = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x * 1.44269504088896341f), 0.f, 1.f);
  return fmaxf(x, e - 1.0f);
}

// optional "scale then ELU" prologue applied to a conv input sample
__device__ __forceinline__ float prologue(float x, float scale, int do_elu) {
  float v = x * scale;
  return do_elu ? elu_fast(v) : v;
}

__device__ __forceinline__ float4 prologue4(float4 v, float scale, int do_elu) {
  v.x = prologue(v.x, scale, do_elu);
  v.y = prologue(v.y, scale, do_elu);
  v.z = prologue(v.z, scale, do_elu);
  v.w = prologue(v.w, scale, do_elu);
  return v;
}

// native 4-vector flavour (selects on HIP's struct float4 go through scratch memory; these do not)
__device__ __forceinline__ f32x4 prologue4v(f32x4 v, float scale, int do_elu) {
  v.x = prologue(v.x, scale, do_elu);
  v.y = prologue(v.y, scale, do_elu);
  v.z = prologue(v.z, scale, do_elu);
  v.w = prologue(v.w, scale, do_elu);
  return v;
}
__device__ __forceinline__ f32x4 zero_unless(bool ok, f32x4 v) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return ok ? v : z;
}

// SpecBlock front half on one (re, im) pair: |.| -> log -> normalise (conv.py:357, seanet.py:228,236).
//   reference:  m = sqrt(max(re^2 + im^2, 1e-12));  v = (log(max(m, 1e-5)) - mean) / std
// evaluated as  v = c1 * log2(max(re^2 + im^2, 1e-10)) + c0,  c1 = ln2 / (2 std), c0 = -mean / std   (the two clamps
// collapse: max(sqrt(max(p, 1e-12)), 1e-5) == sqrt(max(p, 1e-10))), i.e. 5 VALU + one v_log_f32 instead of a correctly
// rounded sqrt, a full logf and an IEEE division (~50 VALU + 3 transcendentals, which cost as much matrix-pipe time as
// a third of the DFT of the n_fft = 64 stage).  The clamp keeps the argument in the normal range, so the raw
// v_log_f32 (1 ulp on log2) needs no denormal scaling; the result differs from the step-by-step form by a few 1e-7
// absolute on values in [-12, 3] — inside the reference's own rounding and far inside the 2e-4 spectrogram bar
// (tests/test_gpu_ops.py::test_stft_vs_oracle, golden SpecBlock KATs).  The squares stay separately rounded.
// mode 0: plain log-magnitude (c1 = ln2 / 2, c0 = 0: the streaming model's merged normalisation), 1: normalised,
// 2: magnitude only (CausalSTFT as a layer): sqrt(max(p, 1e-12)) on v_sqrt_f32.
struct SpecFinish {
  float c1, c0;
  int mode;
  __host__ __device__ static SpecFinish make(float mean, float stdv, int normalize) {
    SpecFinish f;
    f.mode = normalize;
    const float half_ln2 = 0.34657359027997264f;
    f.c1 = normalize == 1 ? half_ln2 / stdv : half_ln2;
    f.c0 = normalize == 1 ? -mean / stdv : 0.f;
    return f;
  }
  __device__ __forceinline__ float operator()(float re, float im) const {
    const float p = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
    if (mode == 2) return __builtin_amdgcn_sqrtf(fmaxf(p, 1e-12f));
    return fmaf(c1, __builtin_amdgcn_logf(fmaxf(p, 1e-10f)), c0);
  }
};

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
