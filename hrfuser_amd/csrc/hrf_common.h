// Device helpers shared by every HRFuser kernel (gfx950 / wave64).
#pragma once
#include "hrf_rt.h"
#include "../../include/hrfuser_hip.h"

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

// ------------------------------------------------------------------ fused BatchNorm finalize
// Ordering without device-scope fences: a `__threadfence()` (agent-scope release) writes back the whole
// XCD L2 on gfx950 - measured +30 us per producer launch.  Device-scope ATOMICS execute at the memory side,
// so it is enough that (1) every thread waits for the completion of its own moment atomics (workgroup
// fence = s_waitcnt) before the block's ticket increment and (2) the finalising block reads the moments
// with agent-scope (cache-bypassing) loads.
#ifdef HRF_EMUL
#define HRF_LOAD_COHERENT(p) (*(p))
#define HRF_FENCE_WG() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#else
#define HRF_LOAD_COHERENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define HRF_FENCE_WG() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#endif

// one channel of hrf_bn_finalize (sums the replicated moment copies)
__device__ __forceinline__ void hrf_bn_fin_channel(const double* stats, const hrf_bn_fin_t& f, int c) {
  const int C = f.C;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < HRF_STAT_COPIES; ++k) {
    s1 += HRF_LOAD_COHERENT(&stats[(size_t)k * 2 * C + c]);
    s2 += HRF_LOAD_COHERENT(&stats[(size_t)k * 2 * C + C + c]);
  }
  const double mean = s1 / f.count;
  double var = s2 / f.count - mean * mean;               // biased variance (train-mode normalisation)
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
  const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
  const float sc = g * invstd;
  f.scale[c] = sc;
  f.shift[c] = b - (float)mean * sc;
  f.mean[c] = (float)mean;
  f.invstd[c] = invstd;
  if (f.update_running) {                                // torch: running_var uses the UNBIASED var
    const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
    f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
    f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
  }
}

// one channel of hrf_bn_bwd_finalize (single-rank form: local == global moments)
__device__ __forceinline__ void hrf_bn_bfin_channel(const double* gstats, const hrf_bn_bfin_t& f, int c) {
  const int C = f.C;
  double sdu = 0.0, sdux = 0.0;
#pragma unroll
  for (int k = 0; k < HRF_STAT_COPIES; ++k) {
    sdu += HRF_LOAD_COHERENT(&gstats[(size_t)k * 2 * C + c]);
    sdux += HRF_LOAD_COHERENT(&gstats[(size_t)k * 2 * C + C + c]);
  }
  const double mu = f.mean[c], is = f.invstd[c], g = f.gamma ? f.gamma[c] : 1.f;
  const double sduy = (sdux - mu * sdu) * is;            // sum du * yhat
  if (f.dgamma) f.dgamma[c] += (float)sduy;
  if (f.dbeta) f.dbeta[c] += (float)sdu;
  if (f.train) {
    const double a = sdu / f.count, b = sduy / f.count;
    f.cA[c] = (float)(g * is);
    f.cB[c] = (float)(-g * is * is * b);
    f.cC[c] = (float)(-g * is * a + g * is * is * b * mu);
  } else {
    f.cA[c] = (float)(g * is); f.cB[c] = 0.f; f.cC[c] = 0.f;
  }
}

// Two-level ticket: returns true in exactly one block of the grid - the one that arrives last, after every
// other block's moment atomics are visible.  `copy_blocks` = gridDim.x (the dimension that selects the
// moment copy), `per_copy_mult` = number of blocks sharing one blockIdx.x (gridDim.y * gridDim.z).
// Must be called by ALL threads of ALL blocks after the block's own moment atomics.
__device__ __forceinline__ bool hrf_last_block(unsigned* ticket, unsigned copy_blocks, unsigned per_copy_mult) {
  __shared__ int s_last;
  HRF_FENCE_WG();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned c = blockIdx.x % HRF_STAT_COPIES;
    const unsigned ngrp = copy_blocks < HRF_STAT_COPIES ? copy_blocks : HRF_STAT_COPIES;
    const unsigned gsz = ((copy_blocks - c + HRF_STAT_COPIES - 1) / HRF_STAT_COPIES) * per_copy_mult;
    int last = 0;
    if (atomicAdd(&ticket[c], 1u) == gsz - 1) {
      ticket[c] = 0;                                       // nobody else touches this group ticket any more
      if (atomicAdd(&ticket[HRF_STAT_COPIES], 1u) == ngrp - 1) { ticket[HRF_STAT_COPIES] = 0; last = 1; }
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

__device__ __forceinline__ void hrf_bn_fin_fused(const hrf_bn_fin_t& f, const double* stats, unsigned nthreads,
                                                 unsigned copy_blocks, unsigned per_copy_mult) {
  if (!hrf_last_block(f.ticket, copy_blocks, per_copy_mult)) return;
  for (int c = threadIdx.x; c < f.C; c += nthreads) hrf_bn_fin_channel(stats, f, c);
}

__device__ __forceinline__ void hrf_bn_bfin_fused(const hrf_bn_bfin_t& f, const double* gstats, unsigned nthreads,
                                                  unsigned copy_blocks, unsigned per_copy_mult) {
  if (!hrf_last_block(f.ticket, copy_blocks, per_copy_mult)) return;
  for (int c = threadIdx.x; c < f.C; c += nthreads) hrf_bn_bfin_channel(gstats, f, c);
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
