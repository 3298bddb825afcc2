// Single-frame (T = 1) pointwise / depthwise-separable layer of a streaming hop: see frame1.hip.
#pragma once
#include <hip/hip_runtime.h>

struct Frame1Args {
  const float* x;       // [B][K]
  const float* wt;      // [K][M]
  const float* bias;    // [M] or null: added to the pointwise output
  const float* res;     // [B][M] or null
  float* y;             // [B][M]
  const float* dw_w;    // [M][ksize] or null (pointwise layer only)
  const float* dw_b;    // [M] or null
  const float* hist;    // [B][M][ksize-1] or null (zeros)
  float* hist_out;      // [B][M][ksize-1] or null; must not alias hist
  long B;
  int K, M, ksize;
  float in_scale;
  int in_elu;
  float out_scale;
  int out_elu;
};

// HILC_OK / HILC_ERR_*; B up to 65535 * 32 streams
int launch_frame1(const Frame1Args& a, hipStream_t stream);
