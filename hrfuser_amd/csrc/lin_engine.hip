// LDS-free fp32 MFMA "row GEMM" kernels for stride-1 1x1 convolutions / Linear layers (gfx950).
//
//   lin_fwd       Y[m][n]  = sum_k tf(X)[m][k] * W[n][k]  + bias + res + res2     (+ BN moments)
//   lin_bwd_data  dX[m][n] = sum_k bnbwd(dY)[m][k] * W[k][n]   (+= | * act'(.) and BN moments)
//
// ~750 of the ~1200 GEMM launches of a HRFuser-T training step are of this kind, with M = 480..30720
// pixels and K, N = 18..576: tiny, latency-bound problems.  Both operands of v_mfma_f32_16x16x4_f32
// want "16 rows x 4 consecutive k" per lane group, and both X rows and W rows are K-contiguous, so
// a lane fetches its fragment values for FOUR MFMAs with one 16-byte global load (the k order inside
// a 16-deep slab is permuted identically for A and B, which a dot product does not care about):
// no LDS staging, no barrier, no transposition, one dependent memory round trip per 32-deep K batch.
// The accumulator tile is D[channel][pixel], so a lane owns 4 consecutive output channels of one
// pixel: bias / residual rows are loaded straight INTO the accumulators and the result leaves as
// one 16-byte store.  BatchNorm / LayerNorm / activation of the producer are applied to the X
// fragment in registers ("transform on load").
//
// Reference ops replaced: nn.Linear / 1x1 nn.Conv2d call sites of hrformer.py:233-237,281-333,
// utils/transformer.py:932-1018, resnet.py:263-302, hrnet.py:417-455 and their autograd backward.
#include "hrf_common.h"
#include "hrf_lin.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int NTM = 9;    // 16-channel output tiles per wave (<= 144 channels per channel group)
constexpr int SB = 2;     // 16-deep K slabs whose loads are issued before the first use

// Elements base..base+3 of a row of which only the first `nvalid` exist (nvalid <= 0: none, >= 4: all);
// missing elements read as 0.  Never touches memory outside the row: a partial group is fetched as
// the 4 elements ENDING at the row end and rotated into place (rows have >= 4 elements).
__device__ __forceinline__ hrf_f4 ld4_guard(const float* p, long base, int nvalid) {
  const int sh = (nvalid >= 4 || nvalid <= 0) ? 0 : 4 - nvalid;
  const hrf_f4 v = hrf_ld4(p + (nvalid > 0 ? base - sh : 0));
  hrf_f4 r;
  r[0] = nvalid > 0 ? (sh == 0 ? v[0] : (sh == 1 ? v[1] : (sh == 2 ? v[2] : v[3]))) : 0.f;
  r[1] = nvalid > 1 ? (sh == 0 ? v[1] : (sh == 1 ? v[2] : v[3])) : 0.f;
  r[2] = nvalid > 2 ? (sh == 0 ? v[2] : v[3]) : 0.f;
  r[3] = nvalid > 3 ? v[3] : 0.f;
  return r;
}

__device__ __forceinline__ void st4_guard(float* p, hrf_f4 v, int nvalid) {
  if (nvalid >= 4) {
    hrf_st4(p, v);
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (r < nvalid) p[r] = v[r];
  }
}

// per-channel (sum, sum*w) of one accumulator tile into the block's LDS moments
__device__ __forceinline__ void tile_moments(float* sStat, int t, int q, bool pixv, int nval, hrf_f4 v, hrf_f4 w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (pixv && r < nval) {
      hrf_atomic_add(&sStat[16 * t + 4 * q + r], v[r]);
      hrf_atomic_add(&sStat[16 * NTM + 16 * t + 4 * q + r], v[r] * w[r]);
    }
  }
}

__device__ __forceinline__ void flush_moments(const float* sStat, double* stats, int nt, int t0, int N) {
  double* st = stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * N;
  for (int i = threadIdx.x; i < 16 * nt; i += 256) {
    const int ch = 16 * t0 + i;
    if (ch < N) {
      hrf_atomic_add(&st[ch], (double)sStat[i]);
      hrf_atomic_add(&st[N + ch], (double)sStat[16 * NTM + i]);
    }
  }
}

// --------------------------------------------------------------------------------- forward
template <int TF>
__global__ __launch_bounds__(256) void lin_fwd_kernel(LinFwdArgs a, int ntw) {
  __shared__ float sStat[2 * 16 * NTM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int T = (a.N + 15) >> 4;
  const int t0 = blockIdx.y * ntw;
  const int nt = min(ntw, T - t0);
  const int pix = blockIdx.x * 64 + wave * 16 + j;
  const bool pixv = pix < a.M;
  const long pc = pixv ? pix : a.M - 1;
  if (a.stats != nullptr)
    for (int i = tid; i < 2 * 16 * NTM; i += 256) sStat[i] = 0.f;

  // accumulators start as bias + residual rows (D = A*B + C): no separate epilogue loads
  hrf_f4 acc[NTM];
#pragma unroll
  for (int t = 0; t < NTM; ++t) {
    acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
    if (t < nt) {
      const int chb = 16 * (t0 + t) + 4 * q, nval = a.N - chb;
      if (a.bias != nullptr) acc[t] = ld4_guard(a.bias, chb, nval);
      if (a.res != nullptr) {
        const hrf_f4 rv = ld4_guard(a.res, pc * a.ldR + chb, nval);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += rv[r];
      }
      if (a.res2 != nullptr) {
        const hrf_f4 rv = ld4_guard(a.res2, pc * a.ldR + chb, nval);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += rv[r];
      }
    }
  }
  float mean = 0.f, rstd = 1.f;
  if (TF == HRF_TF_LN) { mean = a.tf_rowstat[2 * pc]; rstd = a.tf_rowstat[2 * pc + 1]; }

  const int nslab = (a.K + 15) >> 4;
  const long xrow = pc * a.ldX;
  for (int kb = 0; kb < nslab; kb += SB) {
    hrf_f4 xa[SB], sc[SB], sh[SB], wv[SB][NTM];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int kbase = 16 * (kb + s) + 4 * q, kval = a.K - kbase;
      xa[s] = ld4_guard(a.x, xrow + kbase, kval);
      if (TF != HRF_TF_NONE) { sc[s] = ld4_guard(a.tf_scale, kbase, kval); sh[s] = ld4_guard(a.tf_shift, kbase, kval); }
#pragma unroll
      for (int t = 0; t < NTM; ++t) {
        if (t < nt) {
          const int n = 16 * (t0 + t) + j;
          wv[s][t] = ld4_guard(a.w, (long)n * a.K + kbase, n < a.N ? kval : 0);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      float xt[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = xa[s][r];
        if (TF == HRF_TF_LN) v = fmaf((v - mean) * rstd, sc[s][r], sh[s][r]);
        else if (TF != HRF_TF_NONE) v = fmaf(v, sc[s][r], sh[s][r]);
        xt[r] = TF == HRF_TF_AFFINE_RELU ? fmaxf(v, 0.f) : (TF == HRF_TF_AFFINE_GELU ? hrf_gelu(v) : v);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * (kb + s) + r < a.K) {           // uniform: MFMA r of this slab holds at least one valid k
#pragma unroll
          for (int t = 0; t < NTM; ++t)
            if (t < nt) acc[t] = hrf_mfma16(wv[s][t][r], xt[r], acc[t]);
        }
      }
    }
  }

  if (a.stats != nullptr) __syncthreads();
#pragma unroll
  for (int t = 0; t < NTM; ++t) {
    if (t < nt) {
      const int chb = 16 * (t0 + t) + 4 * q, nval = a.N - chb;
      if (pixv) st4_guard(a.y + (long)pix * a.ldY + a.yoff + chb, acc[t], nval);
      if (a.stats != nullptr) tile_moments(sStat, t, q, pixv, nval, acc[t], acc[t]);
    }
  }
  if (a.stats != nullptr) {
    __syncthreads();
    flush_moments(sStat, a.stats, nt, t0, a.N);
  }
}

// --------------------------------------------------------------------------------- backward data
template <bool BNB>
__global__ __launch_bounds__(256) void lin_bwd_data_kernel(LinBwdDataArgs a, int ntw) {
  __shared__ float sStat[2 * 16 * NTM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int T = (a.N + 15) >> 4;
  const int t0 = blockIdx.y * ntw;
  const int nt = min(ntw, T - t0);
  const int pix = blockIdx.x * 64 + wave * 16 + j;
  const bool pixv = pix < a.M;
  const long pc = pixv ? pix : a.M - 1;
  const bool want_stats = a.epi == 1 && a.stats != nullptr;
  if (want_stats)
    for (int i = tid; i < 2 * 16 * NTM; i += 256) sStat[i] = 0.f;

  hrf_f4 acc[NTM], xr[NTM];
#pragma unroll
  for (int t = 0; t < NTM; ++t) {
    acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
    xr[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
    if (t < nt) {
      const int chb = 16 * (t0 + t) + 4 * q, nval = a.N - chb;
      if (a.epi == 1) xr[t] = ld4_guard(a.xraw, pc * a.ldXr + chb, nval);
      else if (a.accumulate) acc[t] = ld4_guard(a.dx, pc * a.ldDx + chb, nval);
    }
  }

  const int nslab = (a.K + 15) >> 4;
  const long drow = pc * a.ldD + a.doff;
  for (int kb = 0; kb < nslab; kb += SB) {
    hrf_f4 dv[SB], yv[SB], ca[SB], cb[SB], cc[SB], wv[SB][NTM];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int kbase = 16 * (kb + s) + 4 * q, kval = a.K - kbase;
      dv[s] = ld4_guard(a.dy, drow + kbase, kval);
      if (BNB) {
        yv[s] = ld4_guard(a.yraw, drow + kbase, kval);
        ca[s] = ld4_guard(a.cA, kbase, kval); cb[s] = ld4_guard(a.cB, kbase, kval); cc[s] = ld4_guard(a.cC, kbase, kval);
      }
#pragma unroll
      for (int t = 0; t < NTM; ++t) {
        if (t < nt) {
          const int ci = 16 * (t0 + t) + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) {              // W[co][ci]: the contraction index is the ROW here
            const bool ok = r < kval && ci < a.N;
            const float wr = a.w[ok ? (long)(kbase + r) * a.N + ci : 0];
            wv[s][t][r] = ok ? wr : 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      float d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r] = BNB ? fmaf(ca[s][r], dv[s][r], fmaf(cb[s][r], yv[s][r], cc[s][r])) : dv[s][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * (kb + s) + r < a.K) {
#pragma unroll
          for (int t = 0; t < NTM; ++t)
            if (t < nt) acc[t] = hrf_mfma16(wv[s][t][r], d[r], acc[t]);
        }
      }
    }
  }

  if (want_stats) __syncthreads();
#pragma unroll
  for (int t = 0; t < NTM; ++t) {
    if (t < nt) {
      const int chb = 16 * (t0 + t) + 4 * q, nval = a.N - chb;
      hrf_f4 v = acc[t];
      if (a.epi == 1) {
        const hrf_f4 sc = ld4_guard(a.tf_scale, chb, nval), sh = ld4_guard(a.tf_shift, chb, nval);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= hrf_act_grad(a.act, fmaf(xr[t][r], sc[r], sh[r]));
        if (want_stats) tile_moments(sStat, t, q, pixv, nval, v, xr[t]);
      }
      if (pixv) st4_guard(a.dx + (long)pix * a.ldDx + chb, v, nval);
    }
  }
  if (want_stats) {
    __syncthreads();
    flush_moments(sStat, a.stats, nt, t0, a.N);
  }
}

// tiles per wave: as many as fit (X fragment reuse) while keeping >= ~768 waves in flight
inline int pick_ntw(int M, int T) {
  const long ptiles = (M + 15) / 16;
  int ntw = T < NTM ? T : NTM;
  while (ntw > 1 && ptiles * ((T + ntw - 1) / ntw) < 768) --ntw;
  const int groups = (T + ntw - 1) / ntw;
  return (T + groups - 1) / groups;
}

}  // namespace

#define HRF_LF_LAUNCH(TF_) HRF_LAUNCH((lin_fwd_kernel<TF_>), grid, dim3(256), 0, stream, a, ntw)

int hrf_lin_fwd_launch(const LinFwdArgs& a, void* stream) {
  if (a.K < 4 || a.N < 4 || a.M <= 0) return -1;
  const int T = (a.N + 15) / 16;
  const int ntw = pick_ntw(a.M, T);
  const dim3 grid(hrf_cdiv(a.M, 64), hrf_cdiv(T, ntw));
  switch (a.tf_mode) {
    case HRF_TF_NONE: HRF_LF_LAUNCH(HRF_TF_NONE); break;
    case HRF_TF_AFFINE: HRF_LF_LAUNCH(HRF_TF_AFFINE); break;
    case HRF_TF_AFFINE_RELU: HRF_LF_LAUNCH(HRF_TF_AFFINE_RELU); break;
    case HRF_TF_AFFINE_GELU: HRF_LF_LAUNCH(HRF_TF_AFFINE_GELU); break;
    case HRF_TF_LN: HRF_LF_LAUNCH(HRF_TF_LN); break;
    default: return -1;
  }
  return hrf_check_launch();
}

int hrf_lin_bwd_data_launch(const LinBwdDataArgs& a, void* stream) {
  if (a.K < 4 || a.N < 4 || a.M <= 0) return -1;
  const int T = (a.N + 15) / 16;
  const int ntw = pick_ntw(a.M, T);
  const dim3 grid(hrf_cdiv(a.M, 64), hrf_cdiv(T, ntw));
  if (a.cA != nullptr) { HRF_LAUNCH((lin_bwd_data_kernel<true>), grid, dim3(256), 0, stream, a, ntw); }
  else { HRF_LAUNCH((lin_bwd_data_kernel<false>), grid, dim3(256), 0, stream, a, ntw); }
  return hrf_check_launch();
}
