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

// erf(x), branch-free, <= 2.5 ulp (max abs error 1.3e-7): |x| < 0.9: x*P7(x^2); else 1 - 2^Q9(|x|)
// with Q9 a fit of log2(erfc) on [0.9, 4] (erf == 1 in fp32 beyond 3.92).  Coefficients fitted by
// Chebyshev least squares (tools/fit_erf.py).  ~22 VALU ops + one v_exp_f32 and
// NO divergent branch - the libm erff is a two-branch routine that both bloats the unrolled loaders
// (GELU is evaluated on load at 189 sites per forward) and serialises the two paths per wave.
// The resulting GELU is closer to the exact value (4.3e-7) than torch's own fp32 GELU (1.2e-6).
__device__ __forceinline__ float hrf_exp2(float x) {
#ifdef HRF_EMUL
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float hrf_erf(float x) {
#ifdef HRF_LIBM_ERF
  return erff(x);
#endif
  const float ax = fabsf(x), t = ax * ax;
  float ps = -1.0492395631445106e-05f;
  ps = fmaf(ps, t, 0.00011508714669616893f);
  ps = fmaf(ps, t, -0.0008512076456099749f);
  ps = fmaf(ps, t, 0.005222628358751535f);
  ps = fmaf(ps, t, -0.026865895837545395f);
  ps = fmaf(ps, t, 0.11283788830041885f);
  ps = fmaf(ps, t, -0.37612637877464294f);
  ps = fmaf(ps, t, 1.128379225730896f);
  ps *= ax;
  const float ac = fminf(ax, 4.0f);
  float q = -4.564023825537333e-08f;
  q = fmaf(q, ac, 3.329453193146037e-06f);
  q = fmaf(q, ac, -7.52767373342067e-05f);
  q = fmaf(q, ac, 0.000906358181964606f);
  q = fmaf(q, ac, -0.00701051764190197f);
  q = fmaf(q, ac, 0.038408491760492325f);
  q = fmaf(q, ac, -0.1591843068599701f);
  q = fmaf(q, ac, -0.9112436771392822f);
  q = fmaf(q, ac, -1.6307036876678467f);
  q = fmaf(q, ac, 0.00048264043289236724f);
  const float pl = 1.0f - hrf_exp2(q);
  const float r = ax < 0.9f ? ps : pl;
  return x < 0.f ? -r : r;
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
