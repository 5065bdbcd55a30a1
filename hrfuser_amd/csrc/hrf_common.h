// Device helpers shared by every HRFuser kernel (gfx950 / wave64).
#pragma once
#include "hrf_rt.h"

// input-transform modes of the conv/dw loaders: value read from HBM is the RAW producer output,
// the BatchNorm affine (+activation) or LayerNorm is applied on load, never materialised.
enum {
  HRF_TF_NONE = 0,
  HRF_TF_AFFINE = 1,        // v*scale[c] + shift[c]
  HRF_TF_AFFINE_RELU = 2,   // relu(v*scale[c] + shift[c])
  HRF_TF_AFFINE_GELU = 3,   // gelu(v*scale[c] + shift[c])   (exact erf GELU)
  HRF_TF_LN = 4             // (v - mean[row])*rstd[row]*scale[c] + shift[c]   (1x1 only)
};
enum { HRF_ACT_NONE = 0, HRF_ACT_RELU = 1, HRF_ACT_GELU = 2 };

// erf(x) by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, fp32-rounding class): ~15 VALU ops and one
// v_exp_f32 instead of the ~60-instruction libm erff.  GELU is evaluated on load by every consumer
// of a BN+GELU tensor (CrossFFN, 189 sites per forward), so its instruction count sets both the
// VALU time and the I-cache footprint of the unrolled loaders.
__device__ __forceinline__ float hrf_erf(float x) {
#ifndef HRF_FAST_ERF   // default: libm-accurate erff (keeps fp32 ReLU-mask flips as rare as the reference)
  return erff(x);
#endif
  const float ax = fabsf(x);
  const float t = 1.0f / fmaf(0.3275911f, ax, 1.0f);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-ax * ax);
  return x < 0.f ? -e : e;
}
__device__ __forceinline__ float hrf_gelu(float x) {
  return 0.5f * x * (1.0f + hrf_erf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float hrf_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + hrf_erf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float hrf_act(int act, float u) {
  return act == HRF_ACT_RELU ? fmaxf(u, 0.f) : (act == HRF_ACT_GELU ? hrf_gelu(u) : u);
}
__device__ __forceinline__ float hrf_act_grad(int act, float u) {
  return act == HRF_ACT_RELU ? (u > 0.f ? 1.f : 0.f) : (act == HRF_ACT_GELU ? hrf_gelu_grad(u) : 1.f);
}
// apply an AFFINE* transform (mode 1..3)
__device__ __forceinline__ float hrf_tf_affine(int mode, float v, float sc, float sh) {
  float u = fmaf(v, sc, sh);
  return mode == HRF_TF_AFFINE_RELU ? fmaxf(u, 0.f) : (mode == HRF_TF_AFFINE_GELU ? hrf_gelu(u) : u);
}
__device__ __forceinline__ int hrf_tf_act(int mode) {
  return mode == HRF_TF_AFFINE_RELU ? HRF_ACT_RELU : (mode == HRF_TF_AFFINE_GELU ? HRF_ACT_GELU : HRF_ACT_NONE);
}

__device__ __forceinline__ float hrf_wave_sum(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}
