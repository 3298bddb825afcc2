// Shared device helpers for the HILCodec gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hilcodec_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// ELU(alpha=1) exactly as the reference computes it on CPU: x > 0 ? x : expm1(x)
// (torch's CPU ELU is bit-identical to expm1, SURVEY.md §7 "Transcendentals").
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }

// optional "scale then ELU" prologue applied to a conv input sample
__device__ __forceinline__ float prologue(float x, float scale, int do_elu) {
  float v = x * scale;
  return do_elu ? elu1(v) : v;
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

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
