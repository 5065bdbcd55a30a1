// HBM-bound elementwise / reduction kernels of the HRFuser backbone (gfx950, wave64, NHWC fp32).
//
//   bn_finalize / bn_bwd_finalize   per-channel BatchNorm bookkeeping from atomically-reduced sums
//   ln_stats / ln_bwd               LayerNorm row statistics and backward (channel-last rows)
//   affine_act_res / act_bwd        BN-apply + activation + residual materialisation and its adjoint
//   fuse_sum / bilinear_up_bwd      HRModule cross-resolution exchange (hrnet.py:184-207)
//   adamw                           fused flat-buffer AdamW step
//
// Reference ops replaced: F.batch_norm, F.layer_norm, F.relu/gelu, torch.add, F.interpolate
// (bilinear, align_corners=False) call sites listed in SURVEY.md 2.1a.
#include <cstring>
#include <type_traits>
#include "hrf_common.h"
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

// ------------------------------------------------------------------------------- BatchNorm
__global__ void bn_finalize_kernel(const double* stats, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, double count, float eps,
                                   float momentum, int update_running, float* scale, float* shift,
                                   float* mean_out, float* invstd_out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < HRF_STAT_COPIES; ++k) { s1 += stats[(size_t)k * 2 * C + c]; s2 += stats[(size_t)k * 2 * C + C + c]; }
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  float sc, sh, meanf, invstd;
  double var;
  hrf_bn_solve(s1, s2, 1.0 / count, eps, g, b, sc, sh, meanf, invstd, var);
  scale[c] = sc;
  shift[c] = sh;
  mean_out[c] = meanf;
  invstd_out[c] = invstd;
  if (update_running) hrf_bn_running(running_mean, running_var, c, momentum, meanf, var, count);
}

__global__ void bn_bwd_finalize_kernel(const double* gstats, const double* gstats_local, const float* gamma,
                                       const float* mean, const float* invstd, double count, int train,
                                       float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sdu = 0.0, sdux = 0.0, ldu = 0.0, ldux = 0.0;
#pragma unroll
  for (int k = 0; k < HRF_STAT_COPIES; ++k) {
    sdu += gstats[(size_t)k * 2 * C + c]; sdux += gstats[(size_t)k * 2 * C + C + c];
    if (gstats_local) { ldu += gstats_local[(size_t)k * 2 * C + c]; ldux += gstats_local[(size_t)k * 2 * C + C + c]; }
  }
  if (!gstats_local) { ldu = sdu; ldux = sdux; }
  const double mu = mean[c], is = invstd[c], g = gamma ? gamma[c] : 1.f;
  // parameter grads use the rank-LOCAL moments (data-parallel grads are averaged afterwards);
  // the dy coefficients use the (SyncBN: all-reduced) global moments.
  if (dgamma) dgamma[c] += (float)((ldux - mu * ldu) * is);
  if (dbeta) dbeta[c] += (float)ldu;
  float a, b2, c2;
  hrf_bn_bwd_solve(sdu, sdux, mu, is, g, 1.0 / count, train, a, b2, c2);
  cA[c] = a; cB[c] = b2; cC[c] = c2;
}

// ---- SyncBN exchange of SEVERAL independent BatchNorms at once (data-parallel training).  The replicated moments of up
// to PK_MAX layers are folded into ONE packed fp64 buffer (2*C doubles per layer, layer after layer), the host all-reduces
// that buffer with ONE collective, and one launch finalises all of them from the packed sums.
constexpr int PK_MAX = 8;
struct BnPackArgs { const double* src[PK_MAX]; int C[PK_MAX]; int off[PK_MAX]; double rows[PK_MAX]; int roff[PK_MAX]; };
__global__ __launch_bounds__(256) void bn_pack_kernel(BnPackArgs a, double* packed) {
  const int e = blockIdx.y, C2 = 2 * a.C[e];
  if (a.roff[e] >= 0 && blockIdx.x == 0 && threadIdx.x == 0) packed[a.roff[e]] = a.rows[e];   // this rank's sample count
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C2; c += gridDim.x * 256) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < HRF_STAT_COPIES; ++k) s += a.src[e][(size_t)k * C2 + c];
    packed[a.off[e] + c] = s;
  }
}
struct BnFinPackArgs { hrf_bn_fin_t f[PK_MAX]; int off[PK_MAX]; };
__global__ __launch_bounds__(256) void bn_finalize_packed_kernel(BnFinPackArgs a, const double* packed) {
  const hrf_bn_fin_t& f = a.f[blockIdx.y];
  const double* st = packed + a.off[blockIdx.y];
  for (int c = blockIdx.x * 256 + threadIdx.x; c < f.C; c += gridDim.x * 256) {
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    float sc, sh, meanf, invstd;
    double var;
    const double count = f.count_ptr != nullptr ? *f.count_ptr : f.count;
    hrf_bn_solve(st[c], st[f.C + c], 1.0 / count, f.eps, g, b, sc, sh, meanf, invstd, var);
    f.scale[c] = sc; f.shift[c] = sh; f.mean[c] = meanf; f.invstd[c] = invstd;
    if (f.update_running) hrf_bn_running(f.running_mean, f.running_var, c, f.momentum, meanf, var, count);
  }
}
struct BnBFinPackArgs { hrf_bn_bfin_t f[PK_MAX]; int off[PK_MAX]; };
__global__ __launch_bounds__(256) void bn_bwd_finalize_packed_kernel(BnBFinPackArgs a, const double* packed, const double* packed_local) {
  const hrf_bn_bfin_t& f = a.f[blockIdx.y];
  const double* gs = packed + a.off[blockIdx.y];
  const double* ls = packed_local != nullptr ? packed_local + a.off[blockIdx.y] : gs;
  // parameter grads from the rank-LOCAL moments (data-parallel grads are summed afterwards) or, without a local copy,
  // pgrad_scale (= 1/world) of the global value; dy coefficients from the all-reduced sums
  const double ps = packed_local != nullptr ? 1.0 : (f.pgrad_scale != 0.f ? (double)f.pgrad_scale : 1.0);
  for (int c = blockIdx.x * 256 + threadIdx.x; c < f.C; c += gridDim.x * 256) {
    const double sdu = gs[c], sdux = gs[f.C + c], ldu = ls[c], ldux = ls[f.C + c];
    const double mu = f.mean[c], is = f.invstd[c], g = f.gamma ? f.gamma[c] : 1.f;
    if (f.dgamma) f.dgamma[c] += (float)(ps * (ldux - mu * ldu) * is);
    if (f.dbeta) f.dbeta[c] += (float)(ps * ldu);
    float a, b2, c2;
    hrf_bn_bwd_solve(sdu, sdux, mu, is, g, 1.0 / (f.count_ptr != nullptr ? *f.count_ptr : f.count), f.train, a, b2, c2);
    f.cA[c] = a; f.cB[c] = b2; f.cC[c] = c2;
  }
}

// ------------------------------------------------------------------------------- LayerNorm
// 16 lanes cooperate on one row (C = 18..624), 4 rows per wave, 16 rows per 256-thread block.
struct LnStatsArgs {
  const float* x;
  int rows;
  int C;
  float eps;
  float* rowstat;
};
__global__ __launch_bounds__(256) void ln_stats_kernel(HrfGroup<LnStatsArgs> grp) {
  const LnStatsArgs& pa_ = grp.sel();
  const float* x = pa_.x;
  int rows = pa_.rows;
  int C = pa_.C;
  float eps = pa_.eps;
  float* rowstat = pa_.rowstat;
  const int sub = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool rv = row < rows;
  const float* xr = x + (long)(rv ? row : 0) * C;
  float s = 0.f;
  for (int c = sub; c < C; c += 16) s += rv ? xr[c] : 0.f;
  s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
  const float mean = s / (float)C;
  float v = 0.f;
  for (int c = sub; c < C; c += 16) { const float d = rv ? xr[c] - mean : 0.f; v = fmaf(d, d, v); }
  v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
  if (rv && sub == 0) {
    rowstat[2 * row] = mean;
    rowstat[2 * row + 1] = 1.0f / sqrtf(v / (float)C + eps);
  }
}

// dx (+)= rstd*(g - mean_c(g) - xhat*mean_c(g*xhat)),  g = da*gamma;  dgamma += sum da*xhat; dbeta += sum da
// 16 lanes per row; lane `sub` owns channels sub, sub+16, ... (NCH of them) for every row it visits,
// so the per-channel sums live in registers across the row loop (one LDS atomic per channel per
// thread at the very end) and each element of da / x is loaded exactly once, all loads of a row
// issued before the first use.
struct LnBwdArgs {
  const float* da;
  const float* x;
  const float* rowstat;
  const float* gamma;
  int rows;
  int C;
  float* dx;
  int accumulate;
  float* dgamma;
  float* dbeta;
  long copy_stride;
};
// LPR lanes per row (16: rows of up to 80 channels, 16 rows per pass; 64: wider rows, 4 rows per pass - at 312 / 624 channels the
// 16-lane form had 60 / 120 dword loads per lane and 60 resp. 15 blocks on the 24x40 and 12x20 grids of HRFuser-B: 20 - 37 us for
// 2 - 7 MB), NCH channels per lane.  The parameter gradients of a block meet through wave shuffles and plain LDS stores (the
// first version merged its 16 row slots with same-address LDS atomics).
template <int NCH, int LPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(HrfGroup<LnBwdArgs> grp) {
  const LnBwdArgs& pa_ = grp.sel();
  const float* da = pa_.da;
  const float* x = pa_.x;
  const float* rowstat = pa_.rowstat;
  const float* gamma = pa_.gamma;
  int rows = pa_.rows;
  int C = pa_.C;
  float* dx = pa_.dx;
  int accumulate = pa_.accumulate;
  float* dgamma = pa_.dgamma;
  float* dbeta = pa_.dbeta;
  long copy_stride = pa_.copy_stride;
  constexpr int RPB = 256 / LPR, CW = LPR * NCH;
  __shared__ float sacc[4 * 2 * CW];
  const int sub = threadIdx.x % LPR, slot = threadIdx.x / LPR, wave = threadIdx.x >> 6;
  float gam[NCH], ag[NCH], ab[NCH];
  bool cv[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = sub + LPR * k;
    cv[k] = c < C;
    gam[k] = gamma[cv[k] ? c : 0];
    ag[k] = 0.f; ab[k] = 0.f;
  }
  const int nrb = (rows + RPB - 1) / RPB;
  const float invC = 1.0f / (float)C;
  for (int rbk = blockIdx.x; rbk < nrb; rbk += gridDim.x) {
    const int row = rbk * RPB + slot;
    const bool rv = row < rows;
    const int rc = rv ? row : 0;
    const long base = (long)rc * C;
    const float mean = rowstat[2 * rc], rstd = rowstat[2 * rc + 1];
    float d[NCH], xh[NCH], pv[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const long idx = (rv && cv[k]) ? base + sub + LPR * k : 0;
      d[k] = da[idx];
      xh[k] = x[idx];
      pv[k] = accumulate ? dx[idx] : 0.f;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const bool ok = rv && cv[k];
      d[k] = ok ? d[k] : 0.f;
      xh[k] = ok ? (xh[k] - mean) * rstd : 0.f;
      const float g = d[k] * gam[k];
      s1 += g; s2 = fmaf(g, xh[k], s2);
    }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float m1 = s1 * invC, m2 = s2 * invC;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float v = rstd * (d[k] * gam[k] - m1 - xh[k] * m2);
      if (rv && cv[k]) dx[base + sub + LPR * k] = pv[k] + v;
      ag[k] = fmaf(d[k], xh[k], ag[k]);
      ab[k] += d[k];
    }
  }
  // row slots of one wave (LPR = 16: lanes sub, sub + 16, sub + 32, sub + 48), then the four waves through LDS
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { ag[k] += __shfl_xor(ag[k], o); ab[k] += __shfl_xor(ab[k], o); }
    if ((threadIdx.x & 63) < LPR) {
      sacc[(wave * 2 + 0) * CW + sub + LPR * k] = ag[k];
      sacc[(wave * 2 + 1) * CW + sub + LPR * k] = ab[k];
    }
  }
  __syncthreads();
  const long cp = (long)(blockIdx.x % HRF_STAT_COPIES) * copy_stride;
  for (int i = threadIdx.x; i < C; i += 256) {
    hrf_atomic_add(&dgamma[cp + i], (sacc[i] + sacc[2 * CW + i]) + (sacc[4 * CW + i] + sacc[6 * CW + i]));
    hrf_atomic_add(&dbeta[cp + i], (sacc[CW + i] + sacc[3 * CW + i]) + (sacc[5 * CW + i] + sacc[7 * CW + i]));
  }
}

// ------------------------------------------------------------------------------- GroupNorm (norm_cfg type 'GN')
// mmcv build_norm_layer(dict(type='GN', num_groups=G), C) -> nn.GroupNorm (hrnet.py:338-339, resnet.py:161-164,
// hrformer.py:269): statistics per (sample, group) over H*W*(C/G) elements - no cross-sample and no cross-rank exchange.
// No reference config uses it, so it gets a plain three-launch form per direction instead of the on-load machinery of
// BatchNorm: (1) per-(sample, channel) moments with the reduction structure of act_bwd_kernel (a thread keeps one channel),
// fp64 atomics into [B][2][C]; (2) every block of the apply kernel folds the C/G channel moments of its sample's groups
// in its prologue and normalises its chunk of pixels.  The output is the PRE-activation value gamma*xhat + beta; the
// consumers apply ReLU / GELU on load through a unit affine (runtime.conv_bn).
__global__ __launch_bounds__(256) void gn_moments_kernel(const float* v, const float* w, long rows_per_sample, int C,
                                                         double* out, long chunk) {
  HRF_DYN_SMEM(float, sacc);                              // [2][C]
  const int n = blockIdx.y;
  const long p0 = (long)blockIdx.x * chunk, p1 = min(rows_per_sample, p0 + chunk);
  const float* vb = v + (long)n * rows_per_sample * C;
  const float* wb = w != nullptr ? w + (long)n * rows_per_sample * C : vb;
  for (int i = threadIdx.x; i < 2 * C; i += 256) sacc[i] = 0.f;
  __syncthreads();
  if (C <= 256) {
    const int R = 256 / C, c = threadIdx.x % C, r0 = threadIdx.x / C;
    if (r0 < R) {
      float s1 = 0.f, s2 = 0.f;
      for (long p = p0 + r0; p < p1; p += R) { const float a = vb[p * C + c]; s1 += a; s2 = fmaf(a, wb[p * C + c], s2); }
      hrf_atomic_add(&sacc[c], s1);
      hrf_atomic_add(&sacc[C + c], s2);
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      float s1 = 0.f, s2 = 0.f;
      for (long p = p0; p < p1; ++p) { const float a = vb[p * C + c]; s1 += a; s2 = fmaf(a, wb[p * C + c], s2); }
      sacc[c] = s1; sacc[C + c] = s2;
    }
  }
  __syncthreads();
  double* o = out + (long)n * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) hrf_atomic_add(&o[i], (double)sacc[i]);
}

// (mean, rstd) of group g of this block's sample from the per-channel moments -> sStat[2*g], sStat[2*g+1]
__device__ __forceinline__ void gn_group_stats(const double* mom, int C, int G, long rows_per_sample, float eps, float* sStat) {
  const int cg = C / G;
  const double inv = 1.0 / ((double)rows_per_sample * (double)cg);
  for (int g = threadIdx.x; g < G; g += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < cg; ++k) { s1 += mom[g * cg + k]; s2 += mom[C + g * cg + k]; }
    const double mean = s1 * inv;
    double var = s2 * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    sStat[2 * g] = (float)mean;
    sStat[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* raw, const double* mom, const float* gamma, const float* beta,
                                                       float eps, long rows_per_sample, int C, int G, float* y, float* stat, long chunk) {
  HRF_DYN_SMEM(float, sStat);                             // [G][2]
  const int n = blockIdx.y;
  gn_group_stats(mom + (long)n * 2 * C, C, G, rows_per_sample, eps, sStat);
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < 2 * G; i += 256) stat[(long)n * 2 * G + i] = sStat[i];
  const int cg = C / G;
  const long e0 = (long)blockIdx.x * chunk * C, e1 = min(rows_per_sample, ((long)blockIdx.x + 1) * chunk) * C;
  const long base = (long)n * rows_per_sample * C;
  for (long e = e0 + threadIdx.x; e < e1; e += 256) {
    const int c = (int)(e % C), g = c / cg;
    y[base + e] = fmaf((raw[base + e] - sStat[2 * g]) * sStat[2 * g + 1], gamma[c], beta[c]);
  }
}

// draw = rstd * (du*gamma - (S1 + xhat*S2) / m),  S1 = sum_group du*gamma,  S2 = sum_group du*gamma*xhat,  m = H*W*C/G;
// gmom = per-(sample, channel) (sum du, sum du*raw).  Block (0, 0) adds the parameter gradients of all samples.
__global__ __launch_bounds__(256) void gn_bwd_kernel(const float* du, const float* raw, const float* stat, const double* gmom,
                                                     const float* gamma, int B, long rows_per_sample, int C, int G, float* draw,
                                                     float* dgamma, float* dbeta, long chunk) {
  HRF_DYN_SMEM(float, sS);                                // [G][4]: mean, rstd, S1/m, S2/m
  const int n = blockIdx.y, cg = C / G;
  const double* gm = gmom + (long)n * 2 * C;
  const double invm = 1.0 / ((double)rows_per_sample * (double)cg);
  for (int g = threadIdx.x; g < G; g += 256) {
    const double mean = stat[(long)n * 2 * G + 2 * g], rstd = stat[(long)n * 2 * G + 2 * g + 1];
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < cg; ++k) {
      const int c = g * cg + k;
      const double a = gm[c], b = gm[C + c], ga = gamma[c];
      s1 += ga * a;
      s2 += ga * rstd * (b - mean * a);
    }
    sS[4 * g] = (float)mean; sS[4 * g + 1] = (float)rstd; sS[4 * g + 2] = (float)(s1 * invm); sS[4 * g + 3] = (float)(s2 * invm);
  }
  __syncthreads();
  if (blockIdx.x == 0 && n == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      const int g = c / cg;
      double dg = 0.0, db = 0.0;
      for (int b = 0; b < B; ++b) {
        const double mean = stat[(long)b * 2 * G + 2 * g], rstd = stat[(long)b * 2 * G + 2 * g + 1];
        const double a = gmom[(long)b * 2 * C + c], bb = gmom[(long)b * 2 * C + C + c];
        dg += rstd * (bb - mean * a);
        db += a;
      }
      if (dgamma != nullptr) dgamma[c] += (float)dg;
      if (dbeta != nullptr) dbeta[c] += (float)db;
    }
  }
  const long e0 = (long)blockIdx.x * chunk * C, e1 = min(rows_per_sample, ((long)blockIdx.x + 1) * chunk) * C;
  const long base = (long)n * rows_per_sample * C;
  for (long e = e0 + threadIdx.x; e < e1; e += 256) {
    const int c = (int)(e % C), g = c / cg;
    const float xh = (raw[base + e] - sS[4 * g]) * sS[4 * g + 1];
    draw[base + e] = sS[4 * g + 1] * (du[base + e] * gamma[c] - sS[4 * g + 2] - xh * sS[4 * g + 3]);
  }
}

// VW consecutive floats as one global access (VW = 4 | 2; like hrf_ld4, natural alignment is not needed on gfx950)
#ifdef HRF_EMUL
template <int VW> inline void hrf_ldv(const float* p, float* v) { std::memcpy(v, p, VW * sizeof(float)); }
template <int VW> inline void hrf_stv(float* p, const float* v) { std::memcpy(p, v, VW * sizeof(float)); }
#else
template <int VW> struct HrfVec { typedef float T __attribute__((ext_vector_type(VW), aligned(4))); };
template <int VW> __device__ __forceinline__ void hrf_ldv(const float* p, float* v) {
  const typename HrfVec<VW>::T t = *reinterpret_cast<const typename HrfVec<VW>::T*>(p);
#pragma unroll
  for (int e = 0; e < VW; ++e) v[e] = t[e];
}
template <int VW> __device__ __forceinline__ void hrf_stv(float* p, const float* v) {
  typename HrfVec<VW>::T t;
#pragma unroll
  for (int e = 0; e < VW; ++e) t[e] = v[e];
  *reinterpret_cast<typename HrfVec<VW>::T*>(p) = t;
}
#endif

// ------------------------------------------------------------------------------- BN apply + act + residual
// act_first=1: out = res + rowscale*act(sc1*y1+sh1)            (CrossFFN tail: x + GELU(BN(h3)))
// act_first=0: out = act(sc1*y1+sh1 [+ res] [+ sc2*y2+sh2])    (Bottleneck tail / transition ReLU)
struct AffineActResArgs {
  const float* y1;
  const float* sc1;
  const float* sh1;
  const float* y2;
  const float* sc2;
  const float* sh2;
  const float* res;
  const float* rowscale;
  int rows_per_sample;
  int act;
  int act_first;
  float* out;
  long total;
  int C;
  hrf_bn_fin_t fin1;
  hrf_bn_fin_t fin2;
  int vec4;
};
__global__ __launch_bounds__(256) void affine_act_res_kernel(HrfGroup<AffineActResArgs> grp) {
  const AffineActResArgs& pa_ = grp.sel();
  const float* y1 = pa_.y1;
  const float* sc1 = pa_.sc1;
  const float* sh1 = pa_.sh1;
  const float* y2 = pa_.y2;
  const float* sc2 = pa_.sc2;
  const float* sh2 = pa_.sh2;
  const float* res = pa_.res;
  const float* rowscale = pa_.rowscale;
  int rows_per_sample = pa_.rows_per_sample;
  int act = pa_.act;
  int act_first = pa_.act_first;
  float* out = pa_.out;
  long total = pa_.total;
  int C = pa_.C;
  hrf_bn_fin_t fin1 = pa_.fin1;
  hrf_bn_fin_t fin2 = pa_.fin2;
  // BatchNorm(s) of the inputs finalised on load (hrf_bn_fin_t)
  __shared__ float sFin[4 * HRF_FIN_MAXC];
  // (round 6) the coefficients ALWAYS sit in LDS for C <= HRF_FIN_MAXC - finalised on load, or copied: a pointer that is LDS on one
  // path and global on another is a GENERIC pointer, every sc[c] / sh[c] a flat_load (eval mode, explicit coefficients: 16 flat
  // dword loads per 4 elements beside the three 16-byte streams - 43 us for the 256-channel residual merge that takes 19 us in
  // training).  The loop bodies below are instantiated once per address space.
  const bool lds_coef = C <= HRF_FIN_MAXC;
  if (fin1.stats != nullptr) hrf_bn_fin_onload(fin1, sFin, sFin + HRF_FIN_MAXC, threadIdx.x, 256, blockIdx.x == 0);
  else if (lds_coef) for (int c = threadIdx.x; c < C; c += 256) { sFin[c] = sc1[c]; sFin[HRF_FIN_MAXC + c] = sh1[c]; }
  if (fin2.stats != nullptr) hrf_bn_fin_onload(fin2, sFin + 2 * HRF_FIN_MAXC, sFin + 3 * HRF_FIN_MAXC, threadIdx.x, 256, blockIdx.x == 0);
  else if (lds_coef && y2 != nullptr) for (int c = threadIdx.x; c < C; c += 256) { sFin[2 * HRF_FIN_MAXC + c] = sc2[c]; sFin[3 * HRF_FIN_MAXC + c] = sh2[c]; }
  if (lds_coef) __syncthreads();
  // vector paths (C % 4 == 0: 16 bytes, C % 2 == 0: 8 bytes - the 18 / 78-channel rows): a thread owns VW consecutive channels of
  // a row; the channel group advances by a constant per iteration (no 64-bit modulo per element - the dword loop below spends
  // more instructions on `i % C` than on the arithmetic - and 1/VW of the memory instructions)
  auto vbody = [&](auto vw, const float* sc1, const float* sh1, const float* sc2, const float* sh2) HRF_KIND_INLINE {
    constexpr int VW = decltype(vw)::value;
    const long nv = total / VW, stride = (long)gridDim.x * 256;
    const int CV = C / VW, dc = (int)(stride % CV);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int cv = (int)(i % CV);
    for (; i < nv; i += stride) {
      const int c = VW * cv;
      float yv[VW], rv[VW], y2v[VW], o[VW];
      hrf_ldv<VW>(y1 + VW * i, yv);
#pragma unroll
      for (int e = 0; e < VW; ++e) { rv[e] = 0.f; y2v[e] = 0.f; }
      if (res) hrf_ldv<VW>(res + VW * i, rv);
      if (y2) hrf_ldv<VW>(y2 + VW * i, y2v);
      const float rsc = (act_first && rowscale) ? rowscale[((VW * i) / C) / rows_per_sample] : 1.f;
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        float v = fmaf(yv[e], sc1[c + e], sh1[c + e]);
        if (act_first) {
          v = hrf_act(act, v) * rsc;
          if (res) v += rv[e];
        } else {
          if (res) v += rv[e];
          if (y2) v += fmaf(y2v[e], sc2[c + e], sh2[c + e]);
          v = hrf_act(act, v);
        }
        o[e] = v;
      }
      hrf_stv<VW>(out + VW * i, o);
      cv += dc; if (cv >= CV) cv -= CV;
    }
  };
  if (pa_.vec4 == 4) {
    if (lds_coef) vbody(std::integral_constant<int, 4>{}, sFin, sFin + HRF_FIN_MAXC, sFin + 2 * HRF_FIN_MAXC, sFin + 3 * HRF_FIN_MAXC);
    else vbody(std::integral_constant<int, 4>{}, sc1, sh1, sc2, sh2);
    return;
  }
  if (pa_.vec4 == 2) {
    if (lds_coef) vbody(std::integral_constant<int, 2>{}, sFin, sFin + HRF_FIN_MAXC, sFin + 2 * HRF_FIN_MAXC, sFin + 3 * HRF_FIN_MAXC);
    else vbody(std::integral_constant<int, 2>{}, sc1, sh1, sc2, sh2);
    return;
  }
  if (lds_coef) { sc1 = sFin; sh1 = sFin + HRF_FIN_MAXC; sc2 = sFin + 2 * HRF_FIN_MAXC; sh2 = sFin + 3 * HRF_FIN_MAXC; }   // (odd C: the dword path keeps generic pointers)
  auto body = [&](auto i) {
    const int c = (int)(i % C);
    float v = fmaf(y1[i], sc1[c], sh1[c]);
    if (act_first) {
      v = hrf_act(act, v);
      if (rowscale) v *= rowscale[(i / C) / rows_per_sample];
      if (res) v += res[i];
    } else {
      if (res) v += res[i];
      if (y2) v += fmaf(y2[i], sc2[c], sh2[c]);
      v = hrf_act(act, v);
    }
    out[i] = v;
  };
  if (total <= 0x7fffffffL) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) body(i);
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) body(i);
  }
}

// out = res + rowscale*act(sc1*y1+sh1) (CrossFFN tail) AND the LayerNorm row statistics of `out` for the
// next block's norm1: 16 lanes per row, lane `sub` owns channels sub, sub+16, ...
struct FfnTailArgs {
  const float* y1;
  const float* sc1;
  const float* sh1;
  const float* res;
  const float* rowscale;
  int rows_per_sample;
  int act;
  float* out;
  int rows;
  int C;
  float eps;
  float* rowstat;
  hrf_bn_fin_t fin1;
};
template <int NCH>
__global__ __launch_bounds__(256) void ffn_tail_kernel(HrfGroup<FfnTailArgs> grp) {
  const FfnTailArgs& pa_ = grp.sel();
  const float* y1 = pa_.y1;
  const float* sc1 = pa_.sc1;
  const float* sh1 = pa_.sh1;
  const float* res = pa_.res;
  const float* rowscale = pa_.rowscale;
  int rows_per_sample = pa_.rows_per_sample;
  int act = pa_.act;
  float* out = pa_.out;
  int rows = pa_.rows;
  int C = pa_.C;
  float eps = pa_.eps;
  float* rowstat = pa_.rowstat;
  hrf_bn_fin_t fin1 = pa_.fin1;
  __shared__ float sFin[2 * HRF_FIN_MAXC];
  if (fin1.stats != nullptr) {                             // BatchNorm of y1 finalised on load (hrf_bn_fin_t)
    hrf_bn_fin_onload(fin1, sFin, sFin + HRF_FIN_MAXC, threadIdx.x, 256, blockIdx.x == 0);
    __syncthreads();
    sc1 = sFin; sh1 = sFin + HRF_FIN_MAXC;
  }
  const int sub = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool rv = row < rows;
  const long base = (long)(rv ? row : 0) * C;
  const float rs = rowscale ? rowscale[(rv ? row : 0) / rows_per_sample] : 1.f;
  float yv[NCH], rsd[NCH], v[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = sub + 16 * k;
    const long idx = (rv && c < C) ? base + c : 0;
    yv[k] = y1[idx];
    rsd[k] = res ? res[idx] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = sub + 16 * k;
    const bool ok = rv && c < C;
    const int cc = c < C ? c : 0;
    const float t = rsd[k] + rs * hrf_act(act, fmaf(yv[k], sc1[cc], sh1[cc]));
    v[k] = ok ? t : 0.f;
    if (ok) out[base + c] = t;
    s += v[k];
  }
  s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) { const float d = (rv && sub + 16 * k < C) ? v[k] - mean : 0.f; q = fmaf(d, d, q); }
  q += __shfl_xor(q, 8); q += __shfl_xor(q, 4); q += __shfl_xor(q, 2); q += __shfl_xor(q, 1);
  if (rv && sub == 0) { rowstat[2 * (long)row] = mean; rowstat[2 * (long)row + 1] = 1.0f / sqrtf(q / (float)C + eps); }
}

// g = dout * act'(.)  written once; per-channel (sum g, sum g*y_k) for up to three BatchNorms fed by g.
//   mode 0: mask = out > 0            (ReLU applied last: uses the saved output)
//   mode 1: g = dout*rowscale*gelu'(sc*y1+sh)   (GELU applied first)
//   mode 2: g = dout                  (no activation)
// A thread keeps ONE channel for its whole life (C <= 256: 256/C rows per pass, thread = (row, c);
// wider rows: channels c, c+256, c+512 of one row per pass), so the moments accumulate in registers
// and LDS/global atomics are paid once per thread / block, not per element.
struct ActBwdArgs {
  const float* dout;
  const float* out;
  const float* y1;
  const float* sc;
  const float* sh;
  const float* rowscale;
  int rows_per_sample;
  int mode;
  float* g;
  const float* y2;
  const float* y3;
  double* st1;
  double* st2;
  double* st3;
  long rows;
  int C;
};
__global__ __launch_bounds__(256) void act_bwd_kernel(HrfGroup<ActBwdArgs> grp) {
  const ActBwdArgs& pa_ = grp.sel();
  const float* dout = pa_.dout;
  const float* out = pa_.out;
  const float* y1 = pa_.y1;
  const float* sc = pa_.sc;
  const float* sh = pa_.sh;
  const float* rowscale = pa_.rowscale;
  int rows_per_sample = pa_.rows_per_sample;
  int mode = pa_.mode;
  float* g = pa_.g;
  const float* y2 = pa_.y2;
  const float* y3 = pa_.y3;
  double* st1 = pa_.st1;
  double* st2 = pa_.st2;
  double* st3 = pa_.st3;
  long rows = pa_.rows;
  int C = pa_.C;
  HRF_DYN_SMEM(float, sacc);                              // [4*C]: sum g, sum g*y1, sum g*y2, sum g*y3
  for (int i = threadIdx.x; i < 4 * C; i += 256) sacc[i] = 0.f;
  const int cw = C <= 256 ? C : 256, R = C <= 256 ? 256 / C : 1;
  const int r = threadIdx.x / cw, c = threadIdx.x - r * cw;
  const bool active = r < R;
  const int nj = (C + 255) / 256;                          // <= 3 (C <= 768)
  float a0[3], a1[3], a2[3], a3[3], scj[3], shj[3];
  bool vj[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    a0[j] = a1[j] = a2[j] = a3[j] = 0.f;
    const int cj = c + 256 * j;
    vj[j] = active && j < nj && cj < C;
    scj[j] = 1.f; shj[j] = 0.f;
    if (mode == 1) { scj[j] = sc[vj[j] ? cj : 0]; shj[j] = sh[vj[j] ? cj : 0]; }
  }
  const bool need1 = mode == 1 || st1 != nullptr;
#pragma unroll 2
  for (long row0 = (long)blockIdx.x * R; row0 < rows; row0 += (long)gridDim.x * R) {
    const long row = row0 + r;
    const bool rv = row < rows;
    float rs = 1.f;
    if (rowscale) rs = rowscale[(rv ? row : 0) / rows_per_sample];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < nj) {
        const bool ok = rv && vj[j];
        const long idx = ok ? row * C + c + 256 * j : 0;
        float v = dout[idx];
        const float ov = mode == 0 ? out[idx] : 1.f;
        const float y1v = need1 ? y1[idx] : 0.f;
        const float y2v = st2 ? y2[idx] : 0.f;
        const float y3v = st3 ? y3[idx] : 0.f;
        if (mode == 0) v = ov > 0.f ? v : 0.f;
        else if (mode == 1) v *= hrf_gelu_grad(fmaf(y1v, scj[j], shj[j])) * rs;
        v = ok ? v : 0.f;
        if (ok) g[idx] = v;
        a0[j] += v;
        a1[j] = fmaf(v, y1v, a1[j]);
        a2[j] = fmaf(v, y2v, a2[j]);
        a3[j] = fmaf(v, y3v, a3[j]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (vj[j]) {
      const int cj = c + 256 * j;
      hrf_atomic_add(&sacc[cj], a0[j]);
      if (st1) hrf_atomic_add(&sacc[C + cj], a1[j]);
      if (st2) hrf_atomic_add(&sacc[2 * C + cj], a2[j]);
      if (st3) hrf_atomic_add(&sacc[3 * C + cj], a3[j]);
    }
  }
  __syncthreads();
  const size_t cp = (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * C;
  for (int cc = threadIdx.x; cc < C; cc += 256) {
    const double s = (double)sacc[cc];
    if (st1) { hrf_atomic_add(&st1[cp + cc], s); hrf_atomic_add(&st1[cp + C + cc], (double)sacc[C + cc]); }
    if (st2) { hrf_atomic_add(&st2[cp + cc], s); hrf_atomic_add(&st2[cp + C + cc], (double)sacc[2 * C + cc]); }
    if (st3) { hrf_atomic_add(&st3[cp + cc], s); hrf_atomic_add(&st3[cp + C + cc], (double)sacc[3 * C + cc]); }
  }
}

// Vector variants (VW = 4 for C % 4 == 0, VW = 2 for C % 2 == 0 - the 18 / 78-channel rows; C / VW <= 256): a thread keeps VW
// consecutive channels for its whole life, 256 * VW / C rows per pass - 1/VW of the memory instructions of the dword kernel, which
// left the 256-channel stem launches at 3 TB/s.
template <int VW>
__global__ __launch_bounds__(256) void act_bwd_vec_kernel(HrfGroup<ActBwdArgs> grp) {
  const ActBwdArgs& pa_ = grp.sel();
  const float* dout = pa_.dout;
  const float* out = pa_.out;
  const float* y1 = pa_.y1;
  const float* rowscale = pa_.rowscale;
  const int rows_per_sample = pa_.rows_per_sample, mode = pa_.mode, C = pa_.C;
  float* g = pa_.g;
  const float* y2 = pa_.y2;
  const float* y3 = pa_.y3;
  double* st1 = pa_.st1;
  double* st2 = pa_.st2;
  double* st3 = pa_.st3;
  const long rows = pa_.rows;
  // [R + 1][4*C]: per row group (sum g, sum g*y1, sum g*y2, sum g*y3), then their totals.  (The row groups used to meet in ONE
  // [4*C] array through LDS float atomics - R lanes on every address, 4 * VW atomics per thread: ~1.5 us at the end of every
  // launch of a latency-bound chain; plain stores + one pass that adds the R rows)
  HRF_DYN_SMEM(float, sacc);
  const int cw = C / VW, R = 256 / cw;
  const int r = threadIdx.x / cw, c = VW * (threadIdx.x - r * cw);
  const bool active = r < R;
  float a0[VW], a1[VW], a2[VW], a3[VW], scv[VW], shv[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) {
    a0[e] = a1[e] = a2[e] = a3[e] = 0.f;
    scv[e] = 1.f; shv[e] = 0.f;
    if (mode == 1) { scv[e] = pa_.sc[active ? c + e : 0]; shv[e] = pa_.sh[active ? c + e : 0]; }
  }
  const bool need1 = mode == 1 || st1 != nullptr;
#pragma unroll 2
  for (long row0 = (long)blockIdx.x * R; row0 < rows; row0 += (long)gridDim.x * R) {
    const long row = row0 + r;
    const bool ok = active && row < rows;
    float rs = 1.f;
    if (rowscale) rs = rowscale[(ok ? row : 0) / rows_per_sample];
    const long idx = ok ? row * C + c : 0;
    float v[VW], ov[VW], y1v[VW], y2v[VW], y3v[VW];
    hrf_ldv<VW>(dout + idx, v);
#pragma unroll
    for (int e = 0; e < VW; ++e) { ov[e] = 1.f; y1v[e] = 0.f; y2v[e] = 0.f; y3v[e] = 0.f; }
    if (mode == 0) hrf_ldv<VW>(out + idx, ov);
    if (need1) hrf_ldv<VW>(y1 + idx, y1v);
    if (st2) hrf_ldv<VW>(y2 + idx, y2v);
    if (st3) hrf_ldv<VW>(y3 + idx, y3v);
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      float t = v[e];
      if (mode == 0) t = ov[e] > 0.f ? t : 0.f;
      else if (mode == 1) t *= hrf_gelu_grad(fmaf(y1v[e], scv[e], shv[e])) * rs;
      t = ok ? t : 0.f;
      v[e] = t;
      a0[e] += t;
      a1[e] = fmaf(t, y1v[e], a1[e]);
      a2[e] = fmaf(t, y2v[e], a2[e]);
      a3[e] = fmaf(t, y3v[e], a3[e]);
    }
    if (ok) hrf_stv<VW>(g + idx, v);
  }
  if (active) {
    float* pr = sacc + (size_t)r * 4 * C + c;
#pragma unroll
    for (int e = 0; e < VW; ++e) { pr[e] = a0[e]; pr[C + e] = a1[e]; pr[2 * C + e] = a2[e]; pr[3 * C + e] = a3[e]; }
  }
  __syncthreads();
  float* stot = sacc + (size_t)R * 4 * C;
  for (int i = threadIdx.x; i < 4 * C; i += 256) {
    float t = 0.f;
    for (int rr = 0; rr < R; ++rr) t += sacc[(size_t)rr * 4 * C + i];
    stot[i] = t;
  }
  __syncthreads();
  sacc = stot;
  const size_t cp = (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * C;
  for (int cc = threadIdx.x; cc < C; cc += 256) {
    const double s = (double)sacc[cc];
    if (st1) { hrf_atomic_add(&st1[cp + cc], s); hrf_atomic_add(&st1[cp + C + cc], (double)sacc[C + cc]); }
    if (st2) { hrf_atomic_add(&st2[cp + cc], s); hrf_atomic_add(&st2[cp + C + cc], (double)sacc[2 * C + cc]); }
    if (st3) { hrf_atomic_add(&st3[cp + cc], s); hrf_atomic_add(&st3[cp + C + cc], (double)sacc[3 * C + cc]); }
  }
}

// out = res + res2 + y * mask * mscale * rowscale[b]   (Dropout / DropPath arithmetic; the Bernoulli
// draws come from torch's graph-safe Philox generator, only the arithmetic runs here)
struct ScaleAddArgs {
  const float* y;
  const float* mask;
  float mscale;
  const float* rowscale;
  int rows_per_sample;
  const float* res;
  const float* res2;
  float* out;
  long total;
  int C;
};
__global__ __launch_bounds__(256) void scale_add_kernel(HrfGroup<ScaleAddArgs> grp) {
  const ScaleAddArgs& pa_ = grp.sel();
  const float* y = pa_.y;
  const float* mask = pa_.mask;
  float mscale = pa_.mscale;
  const float* rowscale = pa_.rowscale;
  int rows_per_sample = pa_.rows_per_sample;
  const float* res = pa_.res;
  const float* res2 = pa_.res2;
  float* out = pa_.out;
  long total = pa_.total;
  int C = pa_.C;
  // (index arithmetic in 32 bits whenever the tensor allows it: a 64-bit division is ~100 instructions on this ISA)
  auto body = [&](auto i) {
    float v = y[i] * mscale;
    if (mask) v *= mask[i];
    if (rowscale) v *= rowscale[(i / C) / rows_per_sample];
    if (res) v += res[i];
    if (res2) v += res2[i];
    out[i] = v;
  };
  if (total <= 0x7fffffffL) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) body(i);
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) body(i);
  }
}

// ------------------------------------------------------------------------------- cross-resolution exchange
struct FuseTerm { int type; const float* p; const float* sc; const float* sh; int Hs, Ws; };
struct FuseArgs { FuseTerm t[4]; float* out; int B, H, W, C; hrf_bn_fin_t fin[4]; };

__device__ __forceinline__ void bil_src(int dst, int in, int out, int& i0, int& i1, float& w1) {
  // F.interpolate(mode='bilinear', align_corners=False): src = (dst+0.5)*in/out - 0.5, clamped at 0
  const float scale = (float)in / (float)out;
  float s = ((float)dst + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void fuse_sum_kernel(HrfGroup<FuseArgs> grp) {
  const FuseArgs& a = grp.sel();
  const long total = (long)a.B * a.H * a.W * a.C;
  // BatchNorms of the conv-produced terms finalised on load (hrf_bn_fin_t; C <= HRF_FIN_MAXC / 2 per term)
  __shared__ float sFin[4 * HRF_FIN_MAXC];
  bool any = false;
  const float* tsc[4];
  const float* tsh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    tsc[k] = a.t[k].sc; tsh[k] = a.t[k].sh;
    if (a.fin[k].stats != nullptr) {
      float* base = sFin + k * HRF_FIN_MAXC;
      hrf_bn_fin_onload(a.fin[k], base, base + HRF_FIN_MAXC / 2, threadIdx.x, 256, blockIdx.x == 0);
      tsc[k] = base; tsh[k] = base + HRF_FIN_MAXC / 2;
      any = true;
    }
  }
  if (any) __syncthreads();
  // (index arithmetic in 32 bits whenever the tensor allows it: four 64-bit divisions per element cost more than the sum)
  auto body = [&](auto i) {
    const int c = (int)(i % a.C);
    const auto pix = i / a.C;
    const auto prow = pix / a.W;
    const int x = (int)(pix - prow * a.W), b = (int)(prow / a.H), y = (int)(prow - (decltype(prow))b * a.H);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const FuseTerm& t = a.t[k];
      if (t.type == 1) acc += t.p[i];
      else if (t.type == 2) acc += fmaf(t.p[i], tsc[k][c], tsh[k][c]);
      else if (t.type == 3) {
        int y0, y1, x0, x1; float wy, wx;
        bil_src(y, t.Hs, a.H, y0, y1, wy);
        bil_src(x, t.Ws, a.W, x0, x1, wx);
        const float* base = t.p + (long)b * t.Hs * t.Ws * a.C + c;
        const float v00 = base[((long)y0 * t.Ws + x0) * a.C], v01 = base[((long)y0 * t.Ws + x1) * a.C];
        const float v10 = base[((long)y1 * t.Ws + x0) * a.C], v11 = base[((long)y1 * t.Ws + x1) * a.C];
        // same association as ATen upsample_bilinear2d: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
        const float top = (1.f - wx) * v00 + wx * v01, bot = (1.f - wx) * v10 + wx * v11;
        const float v = (1.f - wy) * top + wy * bot;
        acc += fmaf(v, tsc[k][c], tsh[k][c]);
      } else if (t.type == 4) {
        // nn.Upsample(scale_factor = H / Hs, mode='nearest') (hrnet.py:135-146): src = floor(dst * Hs / H)
        const int ys = y / (a.H / t.Hs), xs = x / (a.W / t.Ws);
        const float v = t.p[(((long)b * t.Hs + ys) * t.Ws + xs) * a.C + c];
        acc += fmaf(v, tsc[k][c], tsh[k][c]);
      }
    }
    a.out[i] = fmaxf(acc, 0.f);
  };
  if (total <= 0x7fffffffL) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) body(i);
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) body(i);
  }
}

// adjoint of the nearest up-sampling by an integer factor k: du[low pixel] = sum of its k x k block of g, plus the
// (sum du, sum du*ylow) moments of the BatchNorm in front of it; one thread = one (low-res pixel, channel) at a time
__global__ __launch_bounds__(256) void nearest_up_bwd_kernel(const float* g, int ldG, int goff, int B, int H, int W, int C,
                                                             const float* ylow, int Hs, int Ws, float* du, double* stats) {
  HRF_DYN_SMEM(float, sacc);                              // [R][2*C]
  const int cw = C <= 256 ? C : 256, R = C <= 256 ? 256 / C : 1;
  const int r = threadIdx.x / cw, c0 = threadIdx.x - r * cw;
  const bool active = r < R;
  const long npix = (long)B * Hs * Ws;
  const int ky = H / Hs, kx = W / Ws;
  float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
  for (long q0 = (long)blockIdx.x * R; q0 < npix; q0 += (long)gridDim.x * R) {
    const long q = q0 + r;
    if (!active || q >= npix) continue;
    const int qx = (int)(q % Ws), qy = (int)((q / Ws) % Hs), b = (int)(q / ((long)Ws * Hs));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = c0 + 256 * j;
      if (c >= C) break;
      float acc = 0.f;
      for (int y = qy * ky; y < (qy + 1) * ky; ++y) {
        const float* grow = g + (((long)b * H + y) * W + (long)qx * kx) * ldG + goff + c;
        for (int x = 0; x < kx; ++x) acc += grow[(long)x * ldG];
      }
      const long i = q * C + c;
      du[i] = acc;
      a1[j] += acc;
      if (ylow != nullptr) a2[j] = fmaf(acc, ylow[i], a2[j]);
    }
  }
  if (stats == nullptr) return;
  // the R row groups of the block meet through plain stores ([R][2*C]) + one pass (LDS float atomics with R lanes per address
  // cost ~1.5 us at the end of every launch: act_bwd_vec_kernel)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = c0 + 256 * j;
    if (active && c < C) { sacc[(size_t)r * 2 * C + c] = a1[j]; sacc[(size_t)r * 2 * C + C + c] = a2[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int rr = 0; rr < R; ++rr) t += sacc[(size_t)rr * 2 * C + i];
    const size_t cp = (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * C;
    hrf_atomic_add(&stats[cp + i], (double)t);
  }
}

// adjoint of the bilinear up-sampling written as a gather over the low-res grid (no atomics on the
// tensor), plus the (sum, sum*ylow) moments for the BatchNorm that precedes the up-sampling.
// A thread keeps ONE channel for its whole life (C <= 256: 256/C low-res pixels per pass), so the
// moments accumulate in registers (the first version paid two LDS atomics per element: 31 us).
struct BilUpBwdArgs {
  const float* g;
  int ldG;
  int goff;
  int B;
  int H;
  int W;
  int C;
  const float* ylow;
  int Hs;
  int Ws;
  float* du;
  double* stats;
};
__global__ __launch_bounds__(256) void bilinear_up_bwd_kernel(HrfGroup<BilUpBwdArgs> grp) {
  const BilUpBwdArgs& pa_ = grp.sel();
  const float* g = pa_.g;
  int ldG = pa_.ldG;
  int goff = pa_.goff;
  int B = pa_.B;
  int H = pa_.H;
  int W = pa_.W;
  int C = pa_.C;
  const float* ylow = pa_.ylow;
  int Hs = pa_.Hs;
  int Ws = pa_.Ws;
  float* du = pa_.du;
  double* stats = pa_.stats;
  HRF_DYN_SMEM(float, sacc);                              // [R][2*C]
  const int cw = C <= 256 ? C : 256, R = C <= 256 ? 256 / C : 1;
  const int r = threadIdx.x / cw, c0 = threadIdx.x - r * cw;
  const bool active = r < R;
  const long npix = (long)B * Hs * Ws;
  const float ry = (float)H / (float)Hs, rx = (float)W / (float)Ws;
  float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
  for (long q0 = (long)blockIdx.x * R; q0 < npix; q0 += (long)gridDim.x * R) {
    const long q = q0 + r;
    if (!active || q >= npix) continue;
    const int qx = (int)(q % Ws), qy = (int)((q / Ws) % Hs), b = (int)(q / ((long)Ws * Hs));
    // candidate hi-res rows whose i0 or i1 can equal qy: src in (qy-1, qy+1)
    int ylo = (int)floorf(((float)qy - 1.f + 0.5f) * ry - 0.5f) - 1, yhi = (int)ceilf(((float)qy + 1.f + 0.5f) * ry - 0.5f) + 1;
    int xlo = (int)floorf(((float)qx - 1.f + 0.5f) * rx - 0.5f) - 1, xhi = (int)ceilf(((float)qx + 1.f + 0.5f) * rx - 0.5f) + 1;
    ylo = max(ylo, 0); yhi = min(yhi, H - 1); xlo = max(xlo, 0); xhi = min(xhi, W - 1);
    // column weights of the window once per output pixel (<= 24 columns covers up-sampling factors up to 10):
    // the inner loops below are loads and fmas only, 8 independent loads in flight
    const bool narrow = xhi - xlo < 24;
    float wv[24];
#pragma unroll
    for (int u = 0; u < 24; ++u) {
      const int x = min(xlo + u, xhi);
      int x0, x1; float wx;
      bil_src(x, Ws, W, x0, x1, wx);
      wv[u] = xlo + u <= xhi ? (x0 == qx ? 1.f - wx : 0.f) + (x1 == qx ? wx : 0.f) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = c0 + 256 * j;
      if (c >= C) break;
      float acc = 0.f;
      if (narrow) {
        for (int y = ylo; y <= yhi; ++y) {
          int y0, y1; float wy;
          bil_src(y, Hs, H, y0, y1, wy);
          const float cy = (y0 == qy ? 1.f - wy : 0.f) + (y1 == qy ? wy : 0.f);
          if (cy == 0.f) continue;
          const float* grow = g + (((long)b * H + y) * W) * ldG + goff + c;
          float part = 0.f;
#pragma unroll
          for (int bq = 0; bq < 3; ++bq) {
            if (xlo + 8 * bq > xhi) break;
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = grow[(long)min(xlo + 8 * bq + u, xhi) * ldG];
#pragma unroll
            for (int u = 0; u < 8; ++u) part = fmaf(wv[8 * bq + u], v[u], part);
          }
          acc = fmaf(cy, part, acc);
        }
      } else
      for (int y = ylo; y <= yhi; ++y) {
        int y0, y1; float wy;
        bil_src(y, Hs, H, y0, y1, wy);
        const float cy = (y0 == qy ? 1.f - wy : 0.f) + (y1 == qy ? wy : 0.f);
        if (cy == 0.f) continue;
        // fixed batches of 8 window columns with unconditional (clamped) loads: 8 loads in flight per thread instead
        // of one load per dependent branch (a x8 up-sample gathers 16 x 16 values per output, latency-bound otherwise)
        const float* grow = g + (((long)b * H + y) * W) * ldG + goff + c;
        float part = 0.f;
        for (int xb = xlo; xb <= xhi; xb += 8) {
          float v[8], wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int x = min(xb + u, xhi);
            int x0, x1; float wx;
            bil_src(x, Ws, W, x0, x1, wx);
            const float cx = (x0 == qx ? 1.f - wx : 0.f) + (x1 == qx ? wx : 0.f);
            wv[u] = xb + u <= xhi ? cx : 0.f;
            v[u] = grow[(long)x * ldG];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) part = fmaf(wv[u], v[u], part);
        }
        acc = fmaf(cy, part, acc);
      }
      const long i = q * C + c;
      du[i] = acc;
      a1[j] += acc;
      if (ylow != nullptr) a2[j] = fmaf(acc, ylow[i], a2[j]);
    }
  }
  if (stats == nullptr) return;
  // the R row groups of the block meet through plain stores ([R][2*C]) + one pass (LDS float atomics with R lanes per address
  // cost ~1.5 us at the end of every launch: act_bwd_vec_kernel)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = c0 + 256 * j;
    if (active && c < C) { sacc[(size_t)r * 2 * C + c] = a1[j]; sacc[(size_t)r * 2 * C + C + c] = a2[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int rr = 0; rr < R; ++rr) t += sacc[(size_t)rr * 2 * C + i];
    const size_t cp = (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * C;
    hrf_atomic_add(&stats[cp + i], (double)t);
  }
}

// out[pix][off + c] = bilinear_up(x)[pix][c]  (align_corners=False; identity copy when Hs == H): the HRFPN
// concat (hrfpn.py:80-84) written straight into the channel slice of the concatenated row buffer
__global__ __launch_bounds__(256) void bilinear_up_into_kernel(const float* x, int Hs, int Ws, int C, float* out, int ldOut,
                                                               int off, int B, int H, int W) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    int y0, y1, x0, x1; float wy, wx;
    bil_src(yy, Hs, H, y0, y1, wy);
    bil_src(xx, Ws, W, x0, x1, wx);
    const float* base = x + (long)b * Hs * Ws * C + c;
    const float v00 = base[((long)y0 * Ws + x0) * C], v01 = base[((long)y0 * Ws + x1) * C];
    const float v10 = base[((long)y1 * Ws + x0) * C], v11 = base[((long)y1 * Ws + x1) * C];
    const float top = (1.f - wx) * v00 + wx * v01, bot = (1.f - wx) * v10 + wx * v11;   // ATen association
    out[pix * ldOut + off + c] = (1.f - wy) * top + wy * bot;
  }
}

// Device-side input pipeline: Normalize (+ BGR->RGB) -> horizontal flip -> zero pad to the padded grid ->
// sensor drop -> channels-last packing, one pass (transforms.py:706-753,440-466,649-664,487-514; formating.py:212-227).
// in: [B][H0][W0][C] float32 or uint8 (HWC as the decoders deliver it); out: [B][Hp][Wp][C] float32.
// out(b, y, x, c) = drop[b] ? 0 : (y < H0 && x < W0 ? (in(b, y, flip[b] ? W0-1-x : x, to_rgb ? C-1-c : c) - mean[c]) * stdinv[c] : 0)
// with a separate float32 subtract and multiply (cv2.subtract / cv2.multiply; contraction is off for this library).
__global__ __launch_bounds__(256) void pack_input_kernel(const void* in, int is_u8, int B, int H0, int W0, int C,
                                                         const float* mean, const float* stdinv, int to_rgb,
                                                         const unsigned char* flip, const unsigned char* drop,
                                                         float* out, int Hp, int Wp) {
  const long total = (long)B * Hp * Wp * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int x = (int)(pix % Wp), y = (int)((pix / Wp) % Hp), b = (int)(pix / ((long)Wp * Hp));
    float v = 0.f;
    const bool dropped = drop != nullptr && drop[b] != 0;
    if (!dropped && y < H0 && x < W0) {
      const int xs = (flip != nullptr && flip[b] != 0) ? W0 - 1 - x : x;
      const int cs = to_rgb ? C - 1 - c : c;
      const long si = (((long)b * H0 + y) * W0 + xs) * C + cs;
      const float raw = is_u8 ? (float)static_cast<const unsigned char*>(in)[si] : static_cast<const float*>(in)[si];
      const float d = raw - mean[c];
      v = d * stdinv[c];
    }
    out[i] = v;
  }
}

// dst[row][c] (+)= src[row][off + c]: a channel slice of wider rows (adjoint of the identity branch of the HRFPN concat)
__global__ __launch_bounds__(256) void slice_cols_kernel(const float* src, int ld, int off, long rows, int C, float* dst,
                                                         int accumulate) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const float v = src[row * ld + off + c];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

// cols[(b, yo, xo)][ci * 9 + tap] = x(b, ci, yo * stride - 1 + tap / 3, xo * stride - 1 + tap % 3) (zero outside): the 3x3 / pad-1 patches
// of a FEW-channel input as rows (the column order is the OIHW weight's memory order, so the 3x3 convolution is the row GEMM
// cols . w.view(Cout, 9 Cin)^T and its weight gradient the 1x1 form dY^T . cols)
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* x, long sB, long sY, long sX, long sC, int B, int H, int W,
                                                        int Cin, int stride, int Ho, int Wo, float* cols, int ld) {
  const int K = Cin * 9;
  const long total = (long)B * Ho * Wo * K;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / K;
    const int k = (int)(i - row * K);
    const int ci = k / 9, tap = k - 9 * ci;
    const int xo = (int)(row % Wo);
    const long r2 = row / Wo;
    const int yo = (int)(r2 % Ho), b = (int)(r2 / Ho);
    const int yi = yo * stride - 1 + tap / 3, xi = xo * stride - 1 + tap % 3;
    const bool ok = (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W;
    const float v = x[ok ? b * sB + yi * sY + xi * sX + ci * sC : 0];
    cols[row * ld + k] = ok ? v : 0.f;
  }
}

// avg_pool2d(kernel = stride = k) on NHWC rows and its adjoint (hrfpn.py:90-91)
__global__ __launch_bounds__(256) void avg_pool_kernel(const float* x, int B, int H, int W, int C, int k, float* out) {
  const int Ho = H / k, Wo = W / k;
  const long total = (long)B * Ho * Wo * C;
  const float inv = 1.0f / (float)(k * k);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long q = i / C;
    const int xo = (int)(q % Wo), yo = (int)((q / Wo) % Ho), b = (int)(q / ((long)Wo * Ho));
    float s = 0.f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) s += x[(((long)b * H + yo * k + dy) * W + xo * k + dx) * C + c];
    out[i] = s * inv;
  }
}

__global__ __launch_bounds__(256) void avg_pool_bwd_kernel(const float* g, int B, int H, int W, int C, int k, float* dx,
                                                           int accumulate) {
  const int Ho = H / k, Wo = W / k;
  const long total = (long)B * H * W * C;
  const float inv = 1.0f / (float)(k * k);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    const int yo = yy / k, xo = xx / k;
    float v = 0.f;
    if (yo < Ho && xo < Wo) v = g[(((long)b * Ho + yo) * Wo + xo) * C + c] * inv;   // rows/cols cut off by floor(H/k)
    dx[i] = accumulate ? dx[i] + v : v;
  }
}

// dst[map[i]] += sum_k scratch[k*copy_stride + i]: folds the replicated parameter-gradient accumulators
// (LayerNorm gamma/beta, depthwise weights/bias) into the flat gradient arena, once per step.
__global__ __launch_bounds__(256) void fold_copies_kernel(const float* scratch, long copy_stride, const int* map,
                                                          float* dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < HRF_STAT_COPIES; ++k) s += scratch[(long)k * copy_stride + i];
    dst[map[i]] += s;
  }
}

// ------------------------------------------------------------------------------- optimizer
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, const float* wd_mask,
                                                    long n, float lr, float b1, float b2, float eps, float wd,
                                                    const float* state, float gscale) {
  const float bc1 = state[0], bc2 = state[1];
  if (lr < 0.f) lr = state[3];                                   // learning rate kept on the device (hipGraph replays follow a schedule)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float wm = wd_mask ? wd_mask[i] : 1.f;
    if (wm < 0.f) continue;                                      // parameter without a gradient (torch skips `grad is None`)
    const float gr = g[i] * gscale;
    float pi = p[i];
    pi -= lr * wd * wm * pi;                                     // decoupled weight decay (torch.optim.AdamW)
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

// device-resident step counter so the optimizer stays correct under hipGraph replay
__global__ void adamw_tick_kernel(float* state, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float t = state[2] + 1.f;
    state[2] = t;
    state[0] = 1.f - powf(b1, t);
    state[1] = 1.f - powf(b2, t);
  }
}

static int g_pw_knob[4] = {0, 0, 0, 0};      // tuning aids (hrf_debug_knob keys 16..19)

inline int ew_grid(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int hrf_bn_finalize(const double* stats, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, double count, float eps, float momentum, int update_running,
                               float* scale, float* shift, float* mean_out, float* invstd_out, int C, void* stream) {
  HRF_LAUNCH(bn_finalize_kernel, dim3(hrf_cdiv(C, 64)), dim3(64), 0, stream, stats, gamma, beta, running_mean,
             running_var, count, eps, momentum, update_running, scale, shift, mean_out, invstd_out, C);
  return hrf_check_launch();
}

extern "C" int hrf_bn_bwd_finalize(const double* gstats, const double* gstats_local, const float* gamma,
                                   const float* mean, const float* invstd,
                                   double count, int train, float* dgamma, float* dbeta, float* cA, float* cB,
                                   float* cC, int C, void* stream) {
  HRF_LAUNCH(bn_bwd_finalize_kernel, dim3(hrf_cdiv(C, 64)), dim3(64), 0, stream, gstats, gstats_local, gamma, mean, invstd, count,
             train, dgamma, dbeta, cA, cB, cC, C);
  return hrf_check_launch();
}

extern "C" int hrf_ln_stats(const float* x, int rows, int C, float eps, float* rowstat, void* stream) {
  HRF_GROUP_CALL();
  if (rows <= 0) return HRF_OK;
  HRF_LAUNCH_G(ln_stats_kernel, dim3(hrf_cdiv(rows, 16)), dim3(256), 0, stream, (LnStatsArgs{x, rows, C, eps, rowstat}));
  return hrf_check_launch();
}

extern "C" int hrf_ln_bwd(const float* da, const float* x, const float* rowstat, const float* gamma, int rows, int C,
                          float* dx, int accumulate, float* dgamma, float* dbeta, long copy_stride, void* stream) {
  HRF_GROUP_CALL();
  if (rows <= 0) return HRF_OK;
  if (C > 640) return HRF_ERR_ARG;
  const int lpr = C > 80 ? 64 : 16, nch = hrf_cdiv(C, lpr);
  const int nrb = hrf_cdiv(rows, 256 / lpr);
  // passes per block: ~640 blocks at most - every block ends with 2*C global atomics, and at 1 920 blocks those were the
  // kernel (156 channels x 7 680 rows: 12.9 us at 480 blocks, 18.2 us at 1 920)
  int grid = hrf_cdiv(nrb, g_pw_knob[0] > 0 ? g_pw_knob[0] : (nrb + 639) / 640);
  if (grid > 2048) grid = 2048;
#define HRF_LNB(N_, L_) HRF_LAUNCH_G((ln_bwd_kernel<N_, L_>), dim3(grid), dim3(256), 0, stream, \
                                     (LnBwdArgs{da, x, rowstat, gamma, rows, C, dx, accumulate, dgamma, dbeta, copy_stride}))
  if (lpr == 16) { if (nch <= 2) { HRF_LNB(2, 16); } else if (nch <= 3) { HRF_LNB(3, 16); } else { HRF_LNB(5, 16); } }
  else if (nch <= 3) { HRF_LNB(3, 64); } else if (nch <= 5) { HRF_LNB(5, 64); } else { HRF_LNB(10, 64); }
  return hrf_check_launch();
}

static long gn_chunk(long rows_per_sample, int B) {
  // ~1024 blocks per launch, at least 16 pixels per block
  long chunks = 1024 / (B > 0 ? B : 1);
  if (chunks < 1) chunks = 1;
  long chunk = (rows_per_sample + chunks - 1) / chunks;
  return chunk < 16 ? 16 : chunk;
}
extern "C" int hrf_gn_moments(const float* v, const float* w, int B, long rows_per_sample, int C, double* out, void* stream) {
  if (B <= 0 || rows_per_sample <= 0 || C <= 0) return HRF_OK;
  if (v == nullptr || out == nullptr || C > 4096) return HRF_ERR_ARG;
  const long chunk = gn_chunk(rows_per_sample, B);
  HRF_LAUNCH(gn_moments_kernel, dim3(hrf_cdiv(rows_per_sample, chunk), B), dim3(256), (unsigned)(2 * C * sizeof(float)), stream,
             v, w, rows_per_sample, C, out, chunk);
  return hrf_check_launch();
}
extern "C" int hrf_gn_apply(const float* raw, const double* mom, const float* gamma, const float* beta, float eps, int B,
                            long rows_per_sample, int C, int G, float* y, float* stat, void* stream) {
  if (B <= 0 || rows_per_sample <= 0 || C <= 0) return HRF_OK;
  if (G <= 0 || C % G != 0 || raw == nullptr || mom == nullptr || y == nullptr || stat == nullptr || G > 4096) return HRF_ERR_ARG;
  const long chunk = gn_chunk(rows_per_sample, B);
  HRF_LAUNCH(gn_apply_kernel, dim3(hrf_cdiv(rows_per_sample, chunk), B), dim3(256), (unsigned)(2 * G * sizeof(float)), stream,
             raw, mom, gamma, beta, eps, rows_per_sample, C, G, y, stat, chunk);
  return hrf_check_launch();
}
extern "C" int hrf_gn_bwd(const float* du, const float* raw, const float* stat, const double* gmom, const float* gamma, int B,
                          long rows_per_sample, int C, int G, float* draw, float* dgamma, float* dbeta, void* stream) {
  if (B <= 0 || rows_per_sample <= 0 || C <= 0) return HRF_OK;
  if (G <= 0 || C % G != 0 || du == nullptr || raw == nullptr || stat == nullptr || gmom == nullptr || draw == nullptr || G > 4096)
    return HRF_ERR_ARG;
  const long chunk = gn_chunk(rows_per_sample, B);
  HRF_LAUNCH(gn_bwd_kernel, dim3(hrf_cdiv(rows_per_sample, chunk), B), dim3(256), (unsigned)(4 * G * sizeof(float)), stream,
             du, raw, stat, gmom, gamma, B, rows_per_sample, C, G, draw, dgamma, dbeta, chunk);
  return hrf_check_launch();
}

extern "C" int hrf_affine_act_res(const float* y1, const float* sc1, const float* sh1, const float* y2,
                                  const float* sc2, const float* sh2, const float* res, const float* rowscale,
                                  int rows_per_sample, int act, int act_first, float* out, long rows, int C,
                                  float* ln_rowstat, float ln_eps, const hrf_bn_fin_t* fin1, const hrf_bn_fin_t* fin2,
                                  void* stream) {
  HRF_GROUP_CALL();
  const long total = rows * C;
  if (total <= 0) return HRF_OK;
  if ((fin1 != nullptr && (fin1->C != C || C > HRF_FIN_MAXC || fin1->stats == nullptr)) ||
      (fin2 != nullptr && (fin2->C != C || C > HRF_FIN_MAXC || fin2->stats == nullptr || y2 == nullptr))) return HRF_ERR_ARG;
  const hrf_bn_fin_t f1 = fin1 != nullptr ? *fin1 : hrf_bn_fin_t{}, f2 = fin2 != nullptr ? *fin2 : hrf_bn_fin_t{};
  if (ln_rowstat != nullptr && act_first && y2 == nullptr && C <= 640) {
    const int nch = hrf_cdiv(C, 16);
    const dim3 grid(hrf_cdiv(rows, 16));
#define HRF_FT(N_) HRF_LAUNCH_G(ffn_tail_kernel<N_>, grid, dim3(256), 0, stream, \
                                (FfnTailArgs{y1, sc1, sh1, res, rowscale, rows_per_sample, act, out, (int)rows, C, ln_eps, ln_rowstat, f1}))
    if (nch <= 2) { HRF_FT(2); } else if (nch <= 3) { HRF_FT(3); } else if (nch <= 5) { HRF_FT(5); }
    else if (nch <= 10) { HRF_FT(10); } else if (nch <= 20) { HRF_FT(20); } else { HRF_FT(40); }
    return hrf_check_launch();
  }
  const int vec4 = g_pw_knob[2] == 1 ? 0 : (C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 0));   // channels per thread (hrf_debug_knob 18 = 1: dword paths)
  HRF_LAUNCH_G(affine_act_res_kernel, dim3(ew_grid(vec4 ? total / vec4 : total)), dim3(256), 0, stream,
               (AffineActResArgs{y1, sc1, sh1, y2, sc2, sh2, res, rowscale, rows_per_sample, act, act_first, out, total, C, f1, f2, vec4}));
  if (ln_rowstat != nullptr) { if (hrf_check_launch() != HRF_OK) return HRF_ERR_LAUNCH; return hrf_ln_stats(out, (int)rows, C, ln_eps, ln_rowstat, stream); }
  return hrf_check_launch();
}

extern "C" int hrf_scale_add(const float* y, const float* mask, float mscale, const float* rowscale,
                             int rows_per_sample, const float* res, const float* res2, float* out, long rows, int C,
                             void* stream) {
  HRF_GROUP_CALL();
  const long total = rows * C;
  if (total <= 0) return HRF_OK;
  HRF_LAUNCH_G(scale_add_kernel, dim3(ew_grid(total)), dim3(256), 0, stream,
               (ScaleAddArgs{y, mask, mscale, rowscale, rows_per_sample, res, res2, out, total, C}));
  return hrf_check_launch();
}

extern "C" int hrf_act_bwd(const float* dout, const float* out, const float* y1, const float* sc, const float* sh,
                           const float* rowscale, int rows_per_sample, int mode, float* g, const float* y2,
                           const float* y3, double* st1, double* st2, double* st3, long rows, int C, void* stream) {
  HRF_GROUP_CALL();
  if (rows * C <= 0) return HRF_OK;
  if (C > 768) return HRF_ERR_ARG;
  const int passes = g_pw_knob[1] > 0 ? g_pw_knob[1] : 4;                          // passes per block
  const int vw = g_pw_knob[2] == 1 ? 0 : (C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 0));   // (hrf_debug_knob 18 = 1: dword kernel)
  if (vw != 0 && C >= 4 * vw && C / vw <= 256) {
    int grid = hrf_cdiv(hrf_cdiv(rows, 256 / (C / vw)), passes);
    if (grid > 2048) grid = 2048;
    const ActBwdArgs a{dout, out, y1, sc, sh, rowscale, rows_per_sample, mode, g, y2, y3, st1, st2, st3, rows, C};
    const unsigned smem = (unsigned)((256 / (C / vw) + 1) * 4 * C * sizeof(float));     // [R + 1][4*C]: <= 17 KB
    if (vw == 4) { HRF_LAUNCH_G(act_bwd_vec_kernel<4>, dim3(grid), dim3(256), smem, stream, a); }
    else { HRF_LAUNCH_G(act_bwd_vec_kernel<2>, dim3(grid), dim3(256), smem, stream, a); }
    return hrf_check_launch();
  }
  const int R = C <= 256 ? 256 / C : 1;
  int grid = hrf_cdiv(hrf_cdiv(rows, R), passes);
  if (grid > 2048) grid = 2048;
  HRF_LAUNCH_G(act_bwd_kernel, dim3(grid), dim3(256), (unsigned)(4 * C * sizeof(float)), stream,
               (ActBwdArgs{dout, out, y1, sc, sh, rowscale, rows_per_sample, mode, g, y2, y3, st1, st2, st3, rows, C}));
  return hrf_check_launch();
}

extern "C" int hrf_fuse_sum(int type0, const float* p0, const float* sc0, const float* sh0, int Hs0, int Ws0,
                            int type1, const float* p1, const float* sc1, const float* sh1, int Hs1, int Ws1,
                            int type2, const float* p2, const float* sc2, const float* sh2, int Hs2, int Ws2,
                            int type3, const float* p3, const float* sc3, const float* sh3, int Hs3, int Ws3,
                            float* out, int B, int H, int W, int C, const hrf_bn_fin_t* fins, void* stream) {
  HRF_GROUP_CALL();
  FuseArgs a;
  for (int k = 0; k < 4; ++k) {
    a.fin[k] = hrf_bn_fin_t{};
    if (fins != nullptr && fins[k].stats != nullptr) {
      if (fins[k].C != C || C > HRF_FIN_MAXC / 2) return HRF_ERR_ARG;
      a.fin[k] = fins[k];
    }
  }
  a.t[0] = FuseTerm{type0, p0, sc0, sh0, Hs0, Ws0};
  a.t[1] = FuseTerm{type1, p1, sc1, sh1, Hs1, Ws1};
  a.t[2] = FuseTerm{type2, p2, sc2, sh2, Hs2, Ws2};
  a.t[3] = FuseTerm{type3, p3, sc3, sh3, Hs3, Ws3};
  a.out = out; a.B = B; a.H = H; a.W = W; a.C = C;
  for (int k = 0; k < 4; ++k)
    if (a.t[k].type == 4 && (a.t[k].Hs <= 0 || a.t[k].Ws <= 0 || H % a.t[k].Hs || W % a.t[k].Ws)) return HRF_ERR_ARG;
  const long total = (long)B * H * W * C;
  if (total <= 0) return HRF_OK;
  HRF_LAUNCH_G(fuse_sum_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, a);
  return hrf_check_launch();
}

extern "C" int hrf_bilinear_up_bwd(const float* g, int ldG, int goff, int B, int H, int W, int C, const float* ylow,
                                   int Hs, int Ws, float* du, double* stats, void* stream) {
  HRF_GROUP_CALL();
  const long npix = (long)B * Hs * Ws;
  if (npix * C <= 0) return HRF_OK;
  if (C > 768) return HRF_ERR_ARG;
  const int R = C <= 256 ? 256 / C : 1;
  int grid = hrf_cdiv(npix, R);
  if (grid > 1024) grid = 1024;
  HRF_LAUNCH_G(bilinear_up_bwd_kernel, dim3(grid), dim3(256), (unsigned)((size_t)R * 2 * C * sizeof(float)), stream,
               (BilUpBwdArgs{g, ldG, goff, B, H, W, C, ylow, Hs, Ws, du, stats}));
  return hrf_check_launch();
}

extern "C" int hrf_nearest_up_bwd(const float* g, int ldG, int goff, int B, int H, int W, int C, const float* ylow,
                                  int Hs, int Ws, float* du, double* stats, void* stream) {
  const long npix = (long)B * Hs * Ws;
  if (npix * C <= 0) return HRF_OK;
  if (C > 768 || Hs <= 0 || Ws <= 0 || H % Hs || W % Ws) return HRF_ERR_ARG;
  const int R = C <= 256 ? 256 / C : 1;
  int grid = hrf_cdiv(npix, R);
  if (grid > 1024) grid = 1024;
  HRF_LAUNCH(nearest_up_bwd_kernel, dim3(grid), dim3(256), (size_t)R * 2 * C * sizeof(float), stream, g, ldG, goff,
             B, H, W, C, ylow, Hs, Ws, du, stats);
  return hrf_check_launch();
}

extern "C" int hrf_bilinear_up_into(const float* x, int Hs, int Ws, int C, float* out, int ldOut, int off, int B, int H,
                                    int W, void* stream) {
  const long total = (long)B * H * W * C;
  if (total <= 0) return HRF_OK;
  HRF_LAUNCH(bilinear_up_into_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, x, Hs, Ws, C, out, ldOut, off, B, H, W);
  return hrf_check_launch();
}

extern "C" int hrf_pack_input(const void* in, int is_u8, int B, int H0, int W0, int C, const float* mean,
                              const float* stdinv, int to_rgb, const unsigned char* flip, const unsigned char* drop,
                              float* out, int Hp, int Wp, void* stream) {
  if (C < 1 || Hp < H0 || Wp < W0) return HRF_ERR_ARG;
  const long total = (long)B * Hp * Wp * C;
  if (total <= 0) return HRF_OK;
  HRF_LAUNCH(pack_input_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, in, is_u8, B, H0, W0, C, mean, stdinv, to_rgb,
             flip, drop, out, Hp, Wp);
  return hrf_check_launch();
}

extern "C" int hrf_im2col3x3(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin, int stride,
                             float* cols, int ld, void* stream) {
  if (Cin < 1 || (stride != 1 && stride != 2) || ld < 9 * Cin) return HRF_ERR_ARG;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const long total = (long)B * Ho * Wo * Cin * 9;
  if (total <= 0) return HRF_OK;
  long g = (total + 255) / 256;
  HRF_LAUNCH(im2col3x3_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, stream, x, (long)sB, (long)sY, (long)sX, (long)sC,
             B, H, W, Cin, stride, Ho, Wo, cols, ld);
  return hrf_check_launch();
}

extern "C" int hrf_slice_cols(const float* src, int ld, int off, long rows, int C, float* dst, int accumulate, void* stream) {
  const long total = rows * C;
  if (total <= 0) return HRF_OK;
  HRF_LAUNCH(slice_cols_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, src, ld, off, rows, C, dst, accumulate);
  return hrf_check_launch();
}

extern "C" int hrf_avg_pool(const float* x, int B, int H, int W, int C, int k, float* out, void* stream) {
  if (k < 1 || H / k < 1 || W / k < 1) return HRF_ERR_ARG;
  const long total = (long)B * (H / k) * (W / k) * C;
  HRF_LAUNCH(avg_pool_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, x, B, H, W, C, k, out);
  return hrf_check_launch();
}

extern "C" int hrf_avg_pool_bwd(const float* g, int B, int H, int W, int C, int k, float* dx, int accumulate, void* stream) {
  if (k < 1 || H / k < 1 || W / k < 1) return HRF_ERR_ARG;
  const long total = (long)B * H * W * C;
  HRF_LAUNCH(avg_pool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, g, B, H, W, C, k, dx, accumulate);
  return hrf_check_launch();
}

// (reached through hrf_debug_knob only: not exported)
extern "C" __attribute__((visibility("hidden"))) int hrf_pw_knob(int key, int value) { if (key < 0 || key >= 4) return HRF_ERR_ARG; g_pw_knob[key] = value; return HRF_OK; }

extern "C" int hrf_fold_copies(const float* scratch, long copy_stride, const int* map, float* dst, long n,
                               void* stream) {
  if (n <= 0) return HRF_OK;
  HRF_LAUNCH(fold_copies_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, scratch, copy_stride, map, dst, n);
  return hrf_check_launch();
}

extern "C" int hrf_adamw_tick(float* state, float beta1, float beta2, void* stream) {
  HRF_LAUNCH(adamw_tick_kernel, dim3(1), dim3(64), 0, stream, state, beta1, beta2);
  return hrf_check_launch();
}

extern "C" int hrf_adamw(float* p, const float* g, float* m, float* v, const float* wd_mask, long n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, const float* state,
                         float grad_scale, void* stream) {
  if (n <= 0) return HRF_OK;
  HRF_LAUNCH(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, p, g, m, v, wd_mask, n, lr, beta1, beta2, eps,
             weight_decay, state, grad_scale);
  return hrf_check_launch();
}

extern "C" int hrf_bn_pack(const double* const* stats, const int* C, int n, const double* rows, double* packed, void* stream) {
  if (n <= 0) return HRF_OK;
  int off = 0, tail = 0;
  for (int k = 0; k < n; ++k) tail += 2 * C[k];                 // the counts sit behind the sums
  for (int b = 0; b < n; b += PK_MAX) {
    BnPackArgs a{};
    const int m = n - b < PK_MAX ? n - b : PK_MAX;
    int cmax = 0;
    for (int k = 0; k < m; ++k) {
      a.src[k] = stats[b + k]; a.C[k] = C[b + k]; a.off[k] = off; off += 2 * C[b + k]; if (C[b + k] > cmax) cmax = C[b + k];
      a.roff[k] = rows != nullptr ? tail + b + k : -1;
      a.rows[k] = rows != nullptr ? rows[b + k] : 0.0;
    }
    HRF_LAUNCH(bn_pack_kernel, dim3(hrf_cdiv(2 * cmax, 256), m), dim3(256), 0, stream, a, packed);
  }
  return hrf_check_launch();
}

extern "C" int hrf_bn_finalize_packed(const hrf_bn_fin_t* fins, int n, const double* packed, void* stream) {
  if (n <= 0) return HRF_OK;
  int off = 0;
  for (int b = 0; b < n; b += PK_MAX) {
    BnFinPackArgs a{};
    const int m = n - b < PK_MAX ? n - b : PK_MAX;
    int cmax = 0;
    for (int k = 0; k < m; ++k) { a.f[k] = fins[b + k]; a.off[k] = off; off += 2 * fins[b + k].C; if (fins[b + k].C > cmax) cmax = fins[b + k].C; }
    HRF_LAUNCH(bn_finalize_packed_kernel, dim3(hrf_cdiv(cmax, 256), m), dim3(256), 0, stream, a, packed);
  }
  return hrf_check_launch();
}

extern "C" int hrf_bn_bwd_finalize_packed(const hrf_bn_bfin_t* bfins, int n, const double* packed, const double* packed_local,
                                          void* stream) {
  if (n <= 0) return HRF_OK;
  int off = 0;
  for (int b = 0; b < n; b += PK_MAX) {
    BnBFinPackArgs a{};
    const int m = n - b < PK_MAX ? n - b : PK_MAX;
    int cmax = 0;
    for (int k = 0; k < m; ++k) { a.f[k] = bfins[b + k]; a.off[k] = off; off += 2 * bfins[b + k].C; if (bfins[b + k].C > cmax) cmax = bfins[b + k].C; }
    HRF_LAUNCH(bn_bwd_finalize_packed_kernel, dim3(hrf_cdiv(cmax, 256), m), dim3(256), 0, stream, a, packed, packed_local);
  }
  return hrf_check_launch();
}

// GPU timestamp (the constant-rate 100 MHz counter behind wall_clock64()) written by one thread when the stream reaches this
// point - works inside a replayed hipGraph, where HIP events cannot be timed: per-stage durations of the real captured step
__global__ void stamp_kernel(long long* dst) {
#ifdef HRF_EMUL
  *dst = 0;
#else
  *dst = (long long)wall_clock64();
#endif
}

extern "C" int hrf_stamp(long long* dst, void* stream) {
  if (dst == nullptr) return HRF_ERR_ARG;
  HRF_LAUNCH(stamp_kernel, dim3(1), dim3(1), 0, stream, dst);
  return hrf_check_launch();
}

// critical-lane probe: one workgroup that does nothing for `ticks` of the 100 MHz wall clock (see hrf_debug_spin in the header)
__global__ void spin_kernel(long long ticks) {
#ifndef HRF_EMUL
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
#else
  (void)ticks;
#endif
}

extern "C" int hrf_debug_spin(long ticks, void* stream) {
  if (ticks < 0 || ticks > 100000000L) return HRF_ERR_ARG;          // at most 1 s
  HRF_LAUNCH(spin_kernel, dim3(1), dim3(1), 0, stream, (long long)ticks);
  return hrf_check_launch();
}

extern "C" int hrf_memset(void* ptr, int value, long bytes, void* stream) {
  if (bytes <= 0) return HRF_OK;
#ifdef HRF_EMUL
  memset(ptr, value, (size_t)bytes);
  return HRF_OK;
#else
  return hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream) == hipSuccess ? HRF_OK : HRF_ERR_LAUNCH;
#endif
}
