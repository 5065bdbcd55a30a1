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
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

#ifndef HRF_LIN_BWD_EPI_MAX_NT
#define HRF_LIN_BWD_EPI_MAX_NT 5
#endif
constexpr int LIN_BWD_EPI_MAX_NT = HRF_LIN_BWD_EPI_MAX_NT;
// forward: 5 tiles per wave at most (HRFuser-B: 56.3 -> 55.5 ms; 9 tiles starve the launch of waves)
#ifndef HRF_LIN_FWD_MAX_NT
#define HRF_LIN_FWD_MAX_NT 5
#endif
constexpr int LIN_FWD_MAX_NT = HRF_LIN_FWD_MAX_NT;
// SB = 16-deep K slabs whose loads are issued before the first use (one dependent round trip per batch).  Same-box A/B of the
// captured steps: HRFuser-T (narrow tiles, K <= 64 where the tiles are wide) 2 slabs 13.27 ms, 3 slabs 13.06, 4 slabs 13.35
// (registers); HRFuser-B (5 channel tiles per wave, K = 78 ... 1 248) 2 slabs 46.4 ms, 3 slabs 46.8, 4 slabs 46.9.  Hence 3,
// except for >= 5 tiles per wave with a long contraction (chosen at launch: LIN_SB_WIDE_K).
constexpr int LIN_SB = 3, LIN_SB_WIDE = 2, LIN_SB_WIDE_K = 64;

// Out-of-range fragment groups are read from this zero block instead of being masked after the
// load: `cond ? loaded : 0` makes the compiler sink the load into an exec-masked branch with its own
// s_waitcnt (one dependent round trip per load); a select between two ADDRESSES keeps every load
// unconditional and back-to-back.
__device__ float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};

// 4 consecutive elements p[off..off+3]; V4: one 16-byte load (all-or-nothing validity: row lengths are
// multiples of 4), otherwise four dword loads with per-element validity (nvalid of them exist).
template <bool V4>
__device__ __forceinline__ hrf_f4 ld_group(const float* p, long off, int nvalid) {
  if (V4) return hrf_ld4(nvalid > 0 ? p + off : g_zero4);
  hrf_f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = *(e < nvalid ? p + off + e : g_zero4);
  return r;
}

// uniform (per slab / per tile) choice between the 16-byte and the dword path: `full` is the same for
// every lane of the block, so this is a scalar branch; a 16-channel tile or 16-deep slab that lies
// completely inside the row uses one 16-byte access per lane even when the row length is not a
// multiple of 4 (18, 54, 78 ... channels: only the last, partial tile / slab takes the dword path)
__device__ __forceinline__ hrf_f4 ld_sel(bool full, const float* p, long off, int nvalid) {
  if (full) return ld_group<true>(p, off, nvalid);
  return ld_group<false>(p, off, nvalid);
}

template <bool V4>
__device__ __forceinline__ void st_group(float* p, hrf_f4 v, int nvalid) {
  if (V4) {
    if (nvalid > 0) hrf_st4(p, v);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < nvalid) p[e] = v[e];
  }
}

// per-channel (sum v, sum v*w) of the NT accumulator tiles of one wave -> sStat[wave][2][NT*16]
template <int NT>
__device__ __forceinline__ void wave_moments(float* sStat, int wave, int j, int q, bool pixv, int n0w, int N,
                                             const hrf_f4* v, const hrf_f4* w) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int nval = N - (n0w + 16 * t + 4 * q);
    float s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m = (pixv && r < nval) ? 1.f : 0.f;
      s1[r] = hrf_row16_sum(v[t][r] * m);
      s2[r] = hrf_row16_sum(v[t][r] * w[t][r] * m);
    }
    if (j == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sStat[(wave * 2 + 0) * (NT * 16) + 16 * t + 4 * q + r] = s1[r];
        sStat[(wave * 2 + 1) * (NT * 16) + 16 * t + 4 * q + r] = s2[r];
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void flush_moments(const float* sStat, double* stats, int n0w, int N) {
  double* st = stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * N;
  for (int i = threadIdx.x; i < 2 * NT * 16; i += 256) {
    const int which = i / (NT * 16), cidx = i - which * (NT * 16);
    const int ch = n0w + cidx;
    if (ch < N) {
      const float s = sStat[(0 * 2 + which) * (NT * 16) + cidx] + sStat[(1 * 2 + which) * (NT * 16) + cidx] +
                      sStat[(2 * 2 + which) * (NT * 16) + cidx] + sStat[(3 * 2 + which) * (NT * 16) + cidx];
      hrf_atomic_add(&st[which * N + ch], (double)s);
    }
  }
}

// --------------------------------------------------------------------------------- forward
// NT = 16-channel output tiles per wave (compile time: the unrolled code carries no guards)
// SK ("split K"): few rows, long contraction (the 24x40 / 12x20 branches: K = 288 ... 2 496 at 480 ... 1 920 rows) - the four
// waves of a block share ONE 16-pixel tile and take a quarter of the K slabs each (K / 64 dependent load round trips per wave
// instead of K / 16 / SB ... and four times the waves in flight); partial tiles meet in LDS, wave 0 runs the epilogue.
template <int NT, int TF, int SB, bool SK, bool ONE = false>
__global__ __launch_bounds__(256) void lin_fwd_kernel(HrfGroup<LinFwdArgs> grp) {
  const LinFwdArgs& a = grp.sel();
  __shared__ float sStat[4 * 2 * NT * 16];
  __shared__ __attribute__((aligned(16))) float sRed[SK ? 3 * NT * 256 : 4];
  __shared__ __attribute__((aligned(16))) float sFin[(TF >= HRF_TF_AFFINE && TF <= HRF_TF_AFFINE_GELU) ? 2 * HRF_FIN_MAXC : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n0w = blockIdx.y * (NT * 16);
  const int pix = SK ? blockIdx.x * 16 + j : blockIdx.x * 64 + wave * 16 + j;
  const bool pixin = pix < a.M;
  const bool pixv = pixin && (!SK || wave == 0);           // the wave that owns the tile's epilogue
  const long pc = pixin ? pix : a.M - 1;
  // BatchNorm of the input finalised on load (hrf_bn_fin_t): scale / shift come from LDS instead of memory
  const float* scp = a.tf_scale;
  const float* shp = a.tf_shift;
  if (TF >= HRF_TF_AFFINE && TF <= HRF_TF_AFFINE_GELU) {
    if (a.fin.stats != nullptr) {
      hrf_bn_fin_onload(a.fin, sFin, sFin + HRF_FIN_MAXC, tid, 256, blockIdx.x == 0 && blockIdx.y == 0);
      __syncthreads();
      scp = sFin; shp = sFin + HRF_FIN_MAXC;
    }
  }

  // accumulators start as bias + residual rows (D = A*B + C): no separate epilogue loads
  hrf_f4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int chb = n0w + 16 * t + 4 * q, nval = a.N - chb;
    const bool nfull = n0w + 16 * (t + 1) <= a.N;
    const bool own = !SK || wave == 0;                      // (uniform per wave)
    acc[t] = ld_sel(nfull, a.bias, chb, (own && a.bias != nullptr) ? nval : 0);
    const hrf_f4 r1 = ld_sel(nfull, a.res, pc * a.ldR + chb, (own && a.res != nullptr) ? nval : 0);
    const hrf_f4 r2 = ld_sel(nfull, a.res2, pc * a.ldR + chb, (own && a.res2 != nullptr) ? nval : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] += r1[r] + r2[r];
  }
  float mean = 0.f, rstd = 1.f;
  if (TF == HRF_TF_LN) { mean = a.tf_rowstat[2 * pc]; rstd = a.tf_rowstat[2 * pc + 1]; }

  const int nslab = (a.K + 15) >> 4;
  const int kper = SK ? (nslab + 3) >> 2 : nslab;
  const int k0 = SK ? wave * kper : 0, k1 = SK ? min(nslab, k0 + kper) : nslab;
  const long xrow = pc * a.ldX;
#pragma unroll 1
  for (int kb = k0; kb < (ONE ? k0 + 1 : k1); kb += SB) {        // ONE: the whole contraction is one batch (K <= 16 * SB): straight-line code
    hrf_f4 xa[SB], sc[SB], sh[SB], wv[SB][NT];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int kbase = 16 * (kb + s) + 4 * q, kval = (!SK || kb + s < k1) ? a.K - kbase : 0;   // (slabs of the next wave: masked)
      const bool kfull = 16 * (kb + s + 1) <= a.K;
      xa[s] = ld_sel(kfull, a.x, xrow + kbase, kval);
      if (TF != HRF_TF_NONE) { sc[s] = ld_sel(kfull, scp, kbase, kval); sh[s] = ld_sel(kfull, shp, kbase, kval); }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = n0w + 16 * t + j;
        wv[s][t] = ld_sel(kfull, a.w, (long)n * a.K + kbase, n < a.N ? kval : 0);
      }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = xa[s][r];
        if (TF == HRF_TF_LN) v = fmaf((v - mean) * rstd, sc[s][r], sh[s][r]);
        else if (TF != HRF_TF_NONE) v = fmaf(v, sc[s][r], sh[s][r]);
        v = TF == HRF_TF_AFFINE_RELU ? fmaxf(v, 0.f) : (TF == HRF_TF_AFFINE_GELU ? hrf_gelu(v) : v);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = hrf_mfma16(wv[s][t][r], v, acc[t]);   // k beyond K: W == 0
      }
    }
  }
  if (SK) {                                                // partial tiles of waves 1..3 -> wave 0
    if (wave != 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) hrf_st4(sRed + (((wave - 1) * NT + t) * 64 + lane) * 4, acc[t]);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const hrf_f4 v = hrf_ld4(sRed + ((w * NT + t) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] += v[r];
        }
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int chb = n0w + 16 * t + 4 * q;
    if (n0w + 16 * (t + 1) <= a.N) st_group<true>(a.y + (long)pc * a.ldY + a.yoff + chb, acc[t], pixv ? a.N - chb : 0);
    else st_group<false>(a.y + (long)pc * a.ldY + a.yoff + chb, acc[t], pixv ? a.N - chb : 0);
  }
  if (a.ln_out != nullptr && gridDim.y == 1) {
    // LayerNorm row statistics of the output row (all N channels of pixel j live in this wave: lanes
    // j, j+16, j+32, j+48 hold 4*NT channels each) - two-pass like ln_stats_kernel
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sm += (16 * t + 4 * q + r < a.N) ? acc[t][r] : 0.f;
    sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
    const float mu = sm / (float)a.N;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float dd = (16 * t + 4 * q + r < a.N) ? acc[t][r] - mu : 0.f; sq = fmaf(dd, dd, sq); }
    sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
    if (q == 0 && pixv) { a.ln_out[2 * (long)pix] = mu; a.ln_out[2 * (long)pix + 1] = 1.0f / sqrtf(sq / (float)a.N + a.ln_eps); }
  }
  if (a.stats != nullptr) {
    wave_moments<NT>(sStat, wave, j, q, pixv, n0w, a.N, acc, acc);
    __syncthreads();
    flush_moments<NT>(sStat, a.stats, n0w, a.N);
  }
}

// --------------------------------------------------------------------------------- backward data
template <int NT, bool BNB, int SB, bool SK, bool ONE = false>
__global__ __launch_bounds__(256) void lin_bwd_data_kernel(HrfGroup<LinBwdDataArgs> grp) {
  const LinBwdDataArgs& a = grp.sel();
  __shared__ float sStat[4 * 2 * NT * 16];
  __shared__ __attribute__((aligned(16))) float sRed[SK ? 3 * NT * 256 : 4];
  __shared__ __attribute__((aligned(16))) float sFin[BNB ? 3 * HRF_FIN_MAXC : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int n0w = blockIdx.y * (NT * 16);
  const int pix = SK ? blockIdx.x * 16 + j : blockIdx.x * 64 + wave * 16 + j;
  const bool pixin = pix < a.M;
  const bool pixv = pixin && (!SK || wave == 0);           // (SK: see lin_fwd_kernel)
  const long pc = pixin ? pix : a.M - 1;
  // BatchNorm-backward coefficients of dY derived on load (hrf_bn_bfin_t)
  const float* cAp = a.cA;
  const float* cBp = a.cB;
  const float* cCp = a.cC;
  if (BNB) {
    if (a.bfin.gstats != nullptr) {
      hrf_bn_bfin_onload(a.bfin, sFin, sFin + HRF_FIN_MAXC, sFin + 2 * HRF_FIN_MAXC, tid, 256, blockIdx.x == 0 && blockIdx.y == 0);
      __syncthreads();
      cAp = sFin; cBp = sFin + HRF_FIN_MAXC; cCp = sFin + 2 * HRF_FIN_MAXC;
    }
  }

  hrf_f4 acc[NT], xr[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int chb = n0w + 16 * t + 4 * q, nval = a.N - chb;
    const bool nfull = n0w + 16 * (t + 1) <= a.N;
    xr[t] = ld_sel(nfull, a.xraw, pc * a.ldXr + chb, a.epi == 1 ? nval : 0);
    acc[t] = ld_sel(nfull, a.dx, pc * a.ldDx + chb, (a.epi != 1 && a.accumulate && (!SK || wave == 0)) ? nval : 0);
  }

  const int nslab = (a.K + 15) >> 4;
  const int kper = SK ? (nslab + 3) >> 2 : nslab;
  const int k0 = SK ? wave * kper : 0, k1 = SK ? min(nslab, k0 + kper) : nslab;
  const long drow = pc * a.ldD + a.doff;
#pragma unroll 1
  for (int kb = k0; kb < (ONE ? k0 + 1 : k1); kb += SB) {        // ONE: the whole contraction is one batch (K <= 16 * SB): straight-line code
    hrf_f4 dv[SB], yv[SB], ca[SB], cb[SB], cc[SB], wv[SB][NT];
#pragma unroll
    for (int s = 0; s < SB; ++s) {
      const int kbase = 16 * (kb + s) + 4 * q, kval = (!SK || kb + s < k1) ? a.K - kbase : 0;
      const bool kfull = 16 * (kb + s + 1) <= a.K;
      dv[s] = ld_sel(kfull, a.dy, drow + kbase, kval);
      if (BNB) {
        yv[s] = ld_sel(kfull, a.yraw, drow + kbase, kval);
        ca[s] = ld_sel(kfull, cAp, kbase, kval); cb[s] = ld_sel(kfull, cBp, kbase, kval); cc[s] = ld_sel(kfull, cCp, kbase, kval);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ci = n0w + 16 * t + j;
#pragma unroll
        for (int r = 0; r < 4; ++r)                  // W[co][ci]: the contraction index is the ROW here
          wv[s][t][r] = *((r < kval && ci < a.N) ? a.w + (long)(kbase + r) * a.N + ci : g_zero4);
      }
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = BNB ? fmaf(ca[s][r], dv[s][r], fmaf(cb[s][r], yv[s][r], cc[s][r])) : dv[s][r];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = hrf_mfma16(wv[s][t][r], d, acc[t]);
      }
    }
  }
  if (SK) {                                                // partial tiles of waves 1..3 -> wave 0
    if (wave != 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) hrf_st4(sRed + (((wave - 1) * NT + t) * 64 + lane) * 4, acc[t]);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const hrf_f4 v = hrf_ld4(sRed + ((w * NT + t) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] += v[r];
        }
    }
  }

  if (a.epi == 1) {
    hrf_f4 esc[NT], esh[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int chb = n0w + 16 * t + 4 * q, nval = a.N - chb;
      const bool nfull = n0w + 16 * (t + 1) <= a.N;
      esc[t] = ld_sel(nfull, a.tf_scale, chb, nval); esh[t] = ld_sel(nfull, a.tf_shift, chb, nval);
    }
    // one uniform branch per activation kind around the 4 NT elements (hrf_with_act: per element it was a branch each)
    hrf_with_act(a.act, [&](auto kind) HRF_KIND_INLINE {
      constexpr int ACT = decltype(kind)::value;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] *= hrf_act_grad(ACT, fmaf(xr[t][r], esc[t][r], esh[t][r]));
    });
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int chb = n0w + 16 * t + 4 * q;
    if (n0w + 16 * (t + 1) <= a.N) st_group<true>(a.dx + (long)pc * a.ldDx + chb, acc[t], pixv ? a.N - chb : 0);
    else st_group<false>(a.dx + (long)pc * a.ldDx + chb, acc[t], pixv ? a.N - chb : 0);
  }
  if (a.epi == 1 && a.stats != nullptr) {
    wave_moments<NT>(sStat, wave, j, q, pixv, n0w, a.N, acc, xr);
    __syncthreads();
    flush_moments<NT>(sStat, a.stats, n0w, a.N);
  }
}

// tiles per wave: as many as fit (X fragment reuse, <= 9) while keeping >= ~768 waves in flight;
// rounded up to an instantiated size (masked tiles cost MFMA issue slots only)
inline int pick_ntw(int M, int T) {
  const long ptiles = (M + 15) / 16;
#ifndef HRF_LIN_WAVES
#define HRF_LIN_WAVES 768
#endif
  long groups = (HRF_LIN_WAVES + ptiles - 1) / ptiles;     // channel groups wanted for parallelism
  if (groups > T) groups = T;
  const long gmin = (T + 8) / 9;
  if (groups < gmin) groups = gmin;
  const int need = (int)((T + groups - 1) / groups);
  return need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 3 ? 3 : (need <= 5 ? 5 : 9)));
}

}  // namespace

// split K over the four waves of a block (lin_*_kernel<..., SK = true>): few rows and a long contraction
#ifndef HRF_LIN_SK_MAX_M
#define HRF_LIN_SK_MAX_M 4096
#endif
#ifndef HRF_LIN_SK_MIN_SLABS
#define HRF_LIN_SK_MIN_SLABS 8
#endif
constexpr int LIN_SK_MAX_M = HRF_LIN_SK_MAX_M, LIN_SK_MIN_SLABS = HRF_LIN_SK_MIN_SLABS;
#ifndef HRF_LIN_LN_ROW_MAX_M
#define HRF_LIN_LN_ROW_MAX_M 8192
#endif
constexpr int LIN_LN_ROW_MAX_M = HRF_LIN_LN_ROW_MAX_M;
inline bool lin_use_sk(int M, int K) { return M <= LIN_SK_MAX_M && ((K + 15) >> 4) >= LIN_SK_MIN_SLABS; }
// tiles per wave and the split decision of a forward launch (hrf_lin_fwd_emits_ln must agree with the launch)
inline void lin_fwd_plan(const LinFwdArgs& a, int& ntw, bool& sk) {
  const int T = (a.N + 15) / 16;
  sk = lin_use_sk(a.M, a.K);
  ntw = pick_ntw(sk ? 4 * a.M : a.M, T);                  // (a 16-pixel tile per BLOCK: four times the blocks for the same rows)
  if (ntw > LIN_FWD_MAX_NT) ntw = LIN_FWD_MAX_NT;
  // the caller wants the LayerNorm statistics of the output rows (the per-op transformer blocks: out_proj -> norm2): a wave that holds
  // the WHOLE row (<= 144 channels) emits them from its accumulators and the hrf_ln_stats launch behind this one disappears - worth
  // more than the parallelism of several channel groups when the rows are few (72 / 144 channels at 24x40 / 12x20)
  if (a.ln_out != nullptr && T <= 9 && a.M <= LIN_LN_ROW_MAX_M) ntw = T <= 1 ? 1 : (T <= 2 ? 2 : (T <= 3 ? 3 : (T <= 5 ? 5 : 9)));
}

#define HRF_LF_LAUNCH(NT_, TF_, SB_, SK_) HRF_LAUNCH_G((lin_fwd_kernel<NT_, TF_, SB_, SK_>), grid, dim3(256), 0, stream, a)
// the whole contraction in ONE batch of 1 / 2 / 3 slabs (K <= 48): straight-line variants with fewer live registers
#define HRF_LF_ONE(NT_, TF_)                                                                                                   \
  { if (a.K <= 16) { HRF_LAUNCH_G((lin_fwd_kernel<NT_, TF_, 1, false, true>), grid, dim3(256), 0, stream, a); }                  \
    else if (a.K <= 32) { HRF_LAUNCH_G((lin_fwd_kernel<NT_, TF_, 2, false, true>), grid, dim3(256), 0, stream, a); }             \
    else { HRF_LAUNCH_G((lin_fwd_kernel<NT_, TF_, 3, false, true>), grid, dim3(256), 0, stream, a); } }
#define HRF_LF_WIDE(NT_, TF_, SK_) { if (a.K <= LIN_SB_WIDE_K) { HRF_LF_LAUNCH(NT_, TF_, LIN_SB, SK_); } else { HRF_LF_LAUNCH(NT_, TF_, LIN_SB_WIDE, SK_); } }
#define HRF_LF_NT(TF_, SK_)                              \
  if (!SK_ && a.K <= 48) {                               \
    switch (ntw) {                                       \
      case 1: HRF_LF_ONE(1, TF_) break;                  \
      case 2: HRF_LF_ONE(2, TF_) break;                  \
      case 3: HRF_LF_ONE(3, TF_) break;                  \
      case 5: HRF_LF_ONE(5, TF_) break;                  \
      default: HRF_LF_WIDE(9, TF_, SK_) break;           \
    }                                                    \
  } else                                                 \
  switch (ntw) {                                         \
    case 1: HRF_LF_LAUNCH(1, TF_, LIN_SB, SK_); break;   \
    case 2: HRF_LF_LAUNCH(2, TF_, LIN_SB, SK_); break;   \
    case 3: HRF_LF_LAUNCH(3, TF_, LIN_SB, SK_); break;   \
    case 5: HRF_LF_WIDE(5, TF_, SK_) break;              \
    default: HRF_LF_WIDE(9, TF_, SK_) break;             \
  }
#define HRF_LF_V4(TF_) { if (sk) { HRF_LF_NT(TF_, true) } else { HRF_LF_NT(TF_, false) } }

bool hrf_lin_fwd_emits_ln(const LinFwdArgs& a) {
  if (a.K < 4 || a.N < 4 || a.M <= 0) return false;
  int ntw; bool sk;
  lin_fwd_plan(a, ntw, sk);
  return hrf_cdiv((a.N + 15) / 16, ntw) == 1;
}

int hrf_lin_fwd_launch(const LinFwdArgs& a, void* stream) {
  if (a.K < 4 || a.N < 4 || a.M <= 0) return -1;
  const int T = (a.N + 15) / 16;
  int ntw; bool sk;
  lin_fwd_plan(a, ntw, sk);
  const dim3 grid(hrf_cdiv(a.M, sk ? 16 : 64), hrf_cdiv(T, ntw));
  switch (a.tf_mode) {
    case HRF_TF_NONE: HRF_LF_V4(HRF_TF_NONE) break;
    case HRF_TF_AFFINE: HRF_LF_V4(HRF_TF_AFFINE) break;
    case HRF_TF_AFFINE_RELU: HRF_LF_V4(HRF_TF_AFFINE_RELU) break;
    case HRF_TF_AFFINE_GELU: HRF_LF_V4(HRF_TF_AFFINE_GELU) break;
    case HRF_TF_LN: HRF_LF_V4(HRF_TF_LN) break;
    default: return -1;
  }
  return hrf_check_launch();
}

#define HRF_LB_LAUNCH(NT_, BNB_, SB_, SK_) HRF_LAUNCH_G((lin_bwd_data_kernel<NT_, BNB_, SB_, SK_>), grid, dim3(256), 0, stream, a)
#define HRF_LB_ONE(NT_, BNB_)                                                                                                   \
  { if (a.K <= 16) { HRF_LAUNCH_G((lin_bwd_data_kernel<NT_, BNB_, 1, false, true>), grid, dim3(256), 0, stream, a); }             \
    else if (a.K <= 32) { HRF_LAUNCH_G((lin_bwd_data_kernel<NT_, BNB_, 2, false, true>), grid, dim3(256), 0, stream, a); }        \
    else { HRF_LAUNCH_G((lin_bwd_data_kernel<NT_, BNB_, 3, false, true>), grid, dim3(256), 0, stream, a); } }
#define HRF_LB_WIDE(NT_, BNB_, SK_) { if (a.K <= LIN_SB_WIDE_K) { HRF_LB_LAUNCH(NT_, BNB_, LIN_SB, SK_); } else { HRF_LB_LAUNCH(NT_, BNB_, LIN_SB_WIDE, SK_); } }
#define HRF_LB_NT(BNB_, SK_)                             \
  if (!SK_ && a.K <= 48) {                               \
    switch (ntw) {                                       \
      case 1: HRF_LB_ONE(1, BNB_) break;                 \
      case 2: HRF_LB_ONE(2, BNB_) break;                 \
      case 3: HRF_LB_ONE(3, BNB_) break;                 \
      case 5: HRF_LB_ONE(5, BNB_) break;                 \
      default: HRF_LB_ONE(5, BNB_) break;                \
    }                                                    \
  } else                                                 \
  switch (ntw) {                                         \
    case 1: HRF_LB_LAUNCH(1, BNB_, LIN_SB, SK_); break;  \
    case 2: HRF_LB_LAUNCH(2, BNB_, LIN_SB, SK_); break;  \
    case 3: HRF_LB_LAUNCH(3, BNB_, LIN_SB, SK_); break;  \
    default: HRF_LB_WIDE(5, BNB_, SK_) break;            \
  }
#define HRF_LB_V4(BNB_) { if (sk) { HRF_LB_NT(BNB_, true) } else { HRF_LB_NT(BNB_, false) } }

int hrf_lin_bwd_data_launch(const LinBwdDataArgs& a, void* stream) {
  if (a.K < 4 || a.N < 4 || a.M <= 0) return -1;
  const int T = (a.N + 15) / 16;
  // (split K: only where the output is narrow - with >= 10 channel tiles a launch has blocks enough, and HRFuser-B's 1 248 -> 312 /
  // 2 496 -> 624 data gradients lost 0.4 ms per step to it)
  const bool sk = T <= 9 && lin_use_sk(a.M, a.K);
  int ntw = pick_ntw(sk ? 4 * a.M : a.M, T);
  // the epilogue variants keep their raw-input tile and the activation temporaries next to the accumulators: 9 tiles per
  // wave do not fit the register file (HRFuser-B, 312 output channels: 150 us at 10 TFLOP/s); more channel groups instead
  // (round 5: the cap holds for every variant - the 9-tile instantiations held 256 VGPRs + 71 ... 186 AGPRs, ONE wave per SIMD)
  if (ntw > LIN_BWD_EPI_MAX_NT) ntw = LIN_BWD_EPI_MAX_NT;
  const dim3 grid(hrf_cdiv(a.M, sk ? 16 : 64), hrf_cdiv(T, ntw));
  if (a.cA != nullptr) { HRF_LB_V4(true) } else { HRF_LB_V4(false) }
  return hrf_check_launch();
}
