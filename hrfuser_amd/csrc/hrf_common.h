// Device helpers shared by every HRFuser kernel (gfx950 / wave64).
#pragma once
#include <type_traits>
#include "hrf_rt.h"
#include "../../include/hrfuser_hip.h"
#include "../../include/hrfuser_hip_debug.h"

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

// erf(z), branch-free: Abramowitz-Stegun 7.1.26, erf(|z|) = 1 - t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2) with
// t = 1 / (1 + p |z|); in fp32 |error| <= 6.1e-7 absolute (GELU: 4.7e-7, its derivative 3.3e-7 - the accuracy of the previous
// two-polynomial fit and closer to the exact value than torch's own fp32 GELU, 1.2e-6).  ~11 VALU operations + one reciprocal
// and one exp2 (the two-polynomial form was ~22 + one exp2): GELU is evaluated ON LOAD at every consumer of a CrossFFN tensor
// (~1.2 G evaluations per HRFuser-T step), so its length is step time; no libm erff either (a two-branch routine that bloats
// the unrolled loaders and serialises both paths per wave).  E returns exp(-z^2): with z = x / sqrt(2) that is the Gaussian
// of GELU's derivative, which therefore costs no second exponential.
__device__ __forceinline__ float hrf_exp2(float x) {
#ifdef HRF_EMUL
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float hrf_rcp(float x) {
#ifdef HRF_EMUL
  return 1.0f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float hrf_erf_e(float z, float& E) {
#ifdef HRF_LIBM_ERF
  E = expf(-z * z);
  return erff(z);
#endif
  const float az = fabsf(z);
  const float t = hrf_rcp(fmaf(0.3275911f, az, 1.0f));
  E = hrf_exp2(-1.4426950408889634f * az * az);
  float p = 1.061405429f;
  p = fmaf(p, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = fmaf(-(p * t), E, 1.0f);
  return copysignf(r, z);
}
__device__ __forceinline__ float hrf_erf(float x) { float E; return hrf_erf_e(x, E); }
__device__ __forceinline__ float hrf_gelu(float x) {
  return 0.5f * x * (1.0f + hrf_erf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float hrf_gelu_grad(float x) {
  float E;
  const float cdf = 0.5f * (1.0f + hrf_erf_e(x * 0.70710678118654752440f, E));
  return cdf + x * (0.39894228040143267794f * E);
}
__device__ __forceinline__ float hrf_act(int act, float u) {
  return act == HRF_ACT_RELU ? fmaxf(u, 0.f) : (act == HRF_ACT_GELU ? hrf_gelu(u) : u);
}
__device__ __forceinline__ float hrf_act_grad(int act, float u) {
  return act == HRF_ACT_RELU ? (u > 0.f ? 1.f : 0.f) : (act == HRF_ACT_GELU ? hrf_gelu_grad(u) : 1.f);
}
// act(u) and act'(u) from one evaluation (the erf / exp of GELU are shared)
__device__ __forceinline__ void hrf_act_both(int act, float u, float& val, float& grad) {
  if (act == HRF_ACT_GELU) {
    float E;
    const float cdf = 0.5f * (1.0f + hrf_erf_e(u * 0.70710678118654752440f, E));
    val = u * cdf;
    grad = cdf + u * (0.39894228040143267794f * E);
  } else if (act == HRF_ACT_RELU) {
    val = fmaxf(u, 0.f);
    grad = u > 0.f ? 1.f : 0.f;
  } else {
    val = u;
    grad = 1.f;
  }
}
// apply an AFFINE* transform (mode 1..3)
__device__ __forceinline__ float hrf_tf_affine(int mode, float v, float sc, float sh) {
  float u = fmaf(v, sc, sh);
  return mode == HRF_TF_AFFINE_RELU ? fmaxf(u, 0.f) : (mode == HRF_TF_AFFINE_GELU ? hrf_gelu(u) : u);
}
// ONE uniform branch per activation / transform kind around a whole (unrolled) loop: `f` is a generic lambda taking the kind as
// a std::integral_constant, so its body is straight-line code per kind.  With the kind tested PER ELEMENT (the functions above
// called with a run-time kind inside an unrolled loop) the compiler keeps a scalar branch per element and every GELU chain
// (rcp -> 5 dependent FMAs -> exp2 -> ...) runs on its own, none overlapping the next: dw4_fwd 12.85 -> 11.8 us,
// lin_bwd_data<5,...> had 200+ branches for 20 epilogue elements.
#define HRF_KIND_INLINE __attribute__((always_inline))
template <class F>
__device__ __forceinline__ void hrf_with_act(int act, F&& f) {
  if (act == HRF_ACT_GELU) f(std::integral_constant<int, HRF_ACT_GELU>{});
  else if (act == HRF_ACT_RELU) f(std::integral_constant<int, HRF_ACT_RELU>{});
  else f(std::integral_constant<int, HRF_ACT_NONE>{});
}
template <class F>
__device__ __forceinline__ void hrf_with_tf(int mode, F&& f) {
  if (mode == HRF_TF_AFFINE_GELU) f(std::integral_constant<int, HRF_TF_AFFINE_GELU>{});
  else if (mode == HRF_TF_AFFINE_RELU) f(std::integral_constant<int, HRF_TF_AFFINE_RELU>{});
  else if (mode == HRF_TF_AFFINE) f(std::integral_constant<int, HRF_TF_AFFINE>{});
  else f(std::integral_constant<int, HRF_TF_NONE>{});
}
__device__ __forceinline__ int hrf_tf_act(int mode) {
  return mode == HRF_TF_AFFINE_RELU ? HRF_ACT_RELU : (mode == HRF_TF_AFFINE_GELU ? HRF_ACT_GELU : HRF_ACT_NONE);
}

// ------------------------------------------------------------------ BatchNorm finalize arithmetic (shared)
// One implementation for the stand-alone finalize kernels, their packed (SyncBN) forms and the on-load prologues, kept
// SHORT on purpose: in a consumer prologue this is a dependent chain in front of the kernel's own work.  No fp64
// division or square root (30-40 dependent instructions each): the moments are scaled by 1/count (computed while the
// moment loads are in flight) and 1/sqrt(var+eps) is the fp32 hardware estimate + one Newton step (<= 1.5e-7 relative).
__device__ __forceinline__ float hrf_rsqrt_nr(float v) {
  const float r = rsqrtf(v);
  return r * fmaf(-0.5f * v * r, r, 1.5f);
}
__device__ __forceinline__ void hrf_bn_solve(double s1, double s2, double inv_count, float eps, float g, float b,
                                             float& sc, float& sh, float& meanf, float& invstd, double& var) {
  const double mean = s1 * inv_count;
  var = s2 * inv_count - mean * mean;                      // biased variance (train-mode normalisation)
  if (var < 0.0) var = 0.0;
  invstd = hrf_rsqrt_nr((float)(var + (double)eps));
  meanf = (float)mean;
  sc = g * invstd;
  sh = b - meanf * sc;
}
// running statistics (torch: running_var uses the UNBIASED variance)
__device__ __forceinline__ void hrf_bn_running(float* rm, float* rv, int c, float momentum, float meanf, double var, double count) {
  const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
  rm[c] = (1.f - momentum) * rm[c] + momentum * meanf;
  rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
}
// dy = cA*du + cB*y + cC from the global (sum du, sum du*y); sduy = sum du*yhat
__device__ __forceinline__ void hrf_bn_bwd_solve(double sdu, double sdux, double mu, double is, double g, double inv_count,
                                                 int train, float& a, float& b2, float& c2) {
  const double gi = g * is;
  a = (float)gi;
  if (train) {
    const double bm = (sdux - mu * sdu) * is * inv_count, gib = gi * is * bm;
    b2 = (float)(-gib);
    c2 = (float)(gib * mu - gi * (sdu * inv_count));
  } else {                                                 // frozen statistics: dy = gamma*invstd*du
    b2 = 0.f; c2 = 0.f;
  }
}

// ------------------------------------------------------------------ BatchNorm finalize on load (consumer side)
// See hrf_bn_fin_t in include/hrfuser_hip.h.  Every thread of the block calls these in the kernel prologue, followed by
// a __syncthreads(); the per-channel results land in LDS (sSc/sSh resp. sA/sB/sC, >= f.C floats each) and the kernel then
// reads its transform coefficients from there instead of from the global scale/shift arrays.  `writer` selects the one
// block of the grid that also publishes the results (and the running statistics / parameter gradients) to memory.
// (c_begin, c_count): the channel slice this block needs (results at sSc[c - c_begin]); default = all channels.
__device__ __forceinline__ void hrf_bn_fin_onload(const hrf_bn_fin_t& f, float* sSc, float* sSh, int tid, int nthreads,
                                                  bool writer, int c_begin = 0, int c_count = 1 << 30) {
  const int C = f.C;
  const int c_end = min(C, c_begin + c_count);
  sSc -= c_begin; sSh -= c_begin;
  const int K = f.copies > 0 ? f.copies : HRF_STAT_COPIES;
  const double count = f.count_ptr != nullptr ? *f.count_ptr : f.count;   // SyncBN: the all-reduced count of the ranks
  const double inv_count = 1.0 / count;                    // (independent of the loads below: overlaps their latency)
  // one thread per channel (consecutive lanes = consecutive channels: a wave's load touches the fewest cache lines - every
  // block of the grid reads the same ~2*C*16 doubles, and the request count on those hot lines is what this prologue
  // costs: a 16-lanes-per-channel split was 2-3x slower, tools/bench_fin.py)
  for (int c = c_begin + tid; c < c_end; c += nthreads) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < HRF_STAT_COPIES; ++k) {         // unconditional loads (a load under `if` gets its own round trip)
      const size_t o = (size_t)(k < K ? k : 0) * 2 * C + c;
      const double v1 = f.stats[o], v2 = f.stats[o + C];
      s1 += k < K ? v1 : 0.0;
      s2 += k < K ? v2 : 0.0;
    }
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    float sc, sh, meanf, invstd;
    double var;
    hrf_bn_solve(s1, s2, inv_count, f.eps, g, b, sc, sh, meanf, invstd, var);
    sSc[c] = sc;
    sSh[c] = sh;
    if (writer && f.write) {
      f.scale[c] = sc; f.shift[c] = sh; f.mean[c] = meanf; f.invstd[c] = invstd;
      if (f.update_running) hrf_bn_running(f.running_mean, f.running_var, c, f.momentum, meanf, var, count);
    }
  }
}

__device__ __forceinline__ void hrf_bn_bfin_onload(const hrf_bn_bfin_t& f, float* sA, float* sB, float* sC, int tid,
                                                   int nthreads, bool writer, int c_begin = 0, int c_count = 1 << 30) {
  const int C = f.C;
  const int c_end = min(C, c_begin + c_count);
  sA -= c_begin; sB -= c_begin; sC -= c_begin;
  const int K = f.copies > 0 ? f.copies : HRF_STAT_COPIES;
  const double inv_count = 1.0 / (f.count_ptr != nullptr ? *f.count_ptr : f.count);
  for (int c = c_begin + tid; c < c_end; c += nthreads) {
    double sdu = 0.0, sdux = 0.0;
#pragma unroll
    for (int k = 0; k < HRF_STAT_COPIES; ++k) {         // unconditional loads (a load under `if` gets its own round trip)
      const size_t o = (size_t)(k < K ? k : 0) * 2 * C + c;
      const double v1 = f.gstats[o], v2 = f.gstats[o + C];
      sdu += k < K ? v1 : 0.0;
      sdux += k < K ? v2 : 0.0;
    }
    const double mu = f.mean[c], is = f.invstd[c], g = f.gamma ? f.gamma[c] : 1.f;
    float a, b2, c2;
    hrf_bn_bwd_solve(sdu, sdux, mu, is, g, inv_count, f.train, a, b2, c2);
    sA[c] = a; sB[c] = b2; sC[c] = c2;
    if (writer && f.write) {
      f.cA[c] = a; f.cB[c] = b2; f.cC[c] = c2;
      // parameter gradients from the rank-LOCAL moments (data-parallel gradients are summed afterwards)
      double ldu = sdu, ldux = sdux, ps = f.pgrad_scale != 0.f ? (double)f.pgrad_scale : 1.0;
      if (f.gstats_local != nullptr) { ldu = f.gstats_local[c]; ldux = f.gstats_local[C + c]; ps = 1.0; }
      if (f.dgamma) f.dgamma[c] += (float)(ps * (ldux - mu * ldu) * is);
      if (f.dbeta) f.dbeta[c] += (float)(ps * ldu);
    }
  }
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
