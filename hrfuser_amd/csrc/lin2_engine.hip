// LDS-tiled fp32 MFMA row GEMM with transform-on-load and fused epilogues, for the WIDE stride-1 1x1 convolutions / Linear
// layers (gfx950): min(K, N) >= 64 - every CrossFFN, qkv / out_proj and fuse 1x1 of HRFuser-B (78 ... 2496 channels,
// configs/hrfuser/cascade_rcnn_hrfuser_b_1x_nus_r640_l_r_fusion.py:6-41; hrformer.py:267-295) and the 64 <-> 256 Bottleneck
// convolutions of every stem (resnet.py:263-302).
//
//   lin2_fwd       Y[m][n]  = sum_k tf(X)[m][k] * W[n][k]  + bias + res + res2      (+ BN moments, + LayerNorm row statistics)
//   lin2_bwd_data  dX[m][n] = sum_k bnbwd(dY)[m][k] * W[k][n]    (+= | * act'(.) and BN moments)
//
// Same contract as lin_engine.hip (hrf_lin.h), different data path.  The register-only kernels there re-fetch the weight
// fragments of every 16-pixel tile from L2: at 312 output channels that is 190 MB of weight traffic for 10 MB of
// activations, which is what held HRFuser-B's GEMMs at 0.16 of the MFMA peak.  Here a 512-thread block owns 128 rows x up
// to 256 output channels; both operand tiles of a 16-deep K step pass through a 2-slot LDS ring (global loads three steps
// ahead, the fragments of step s+1 fetched in the middle of step s's MFMAs - the schedule of rowgemm_kernel in
// conv3w_engine.hip), the contraction index is permuted inside a step (MFMA m takes k = 4q + m) so that every operand fetch is
// one ds_read_b128, and every weight element is staged ONCE per 128 rows.  What the neck's rowgemm lacks and the backbone
// needs is done where the data passes anyway:
//   * X is transformed while it is staged (BatchNorm affine finalised on load + ReLU / GELU, or LayerNorm), dY gets the
//     BatchNorm backward (cA*dy + cB*y + cC, coefficients derived on load) the same way - the tables live in LDS;
//   * weights are read in the reference layouts ((out, in) rows; the backward stages W[k][n] transposed), ragged K / N
//     are guarded at the loads (no packed copies, no padding);
//   * bias / residual rows, activation' of the producer, per-channel (sum, sum*.) moments and LayerNorm row statistics are
//     applied to the accumulators (D[channel][pixel]: a lane owns 4 consecutive channels of one pixel).
#include <type_traits>
#include "hrf_common.h"
#include "hrf_lin.h"
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int GK = 16, GLP = GK + 4, GROWS = 128, NTHR = 512;
constexpr int TBL = HRF_FIN_MAXC;          // widest contraction whose per-k tables fit in LDS

#ifdef HRF_EMUL
#define L2_SCHED_FENCE() ((void)0)
#define L2_WAIT_LDS() ((void)0)
#else
#define L2_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define L2_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)      // lgkmcnt(0) only
#endif
// the pipeline stages are lambdas over ~100 live registers: an outlined call would pass them through memory
#define L2_INLINE __attribute__((always_inline))

__device__ float g_zero4l[4] = {0.f, 0.f, 0.f, 0.f};
static int g_l2_knob[4] = {0, 0, 0, 0};    // 0: 1 = use this engine whenever the shape is supported, 2 = never (tests / A-B); 1: forced WN

__device__ __forceinline__ hrf_f4 l2_ld4(const float* p) {
#ifdef HRF_EMUL
  return hrf_ld4(p);
#else
  return *reinterpret_cast<const hrf_f4*>(p);      // ds_read_b128 (16-byte aligned by construction)
#endif
}
__device__ __forceinline__ void l2_st4(float* p, hrf_f4 v) {
#ifdef HRF_EMUL
  hrf_st4(p, v);
#else
  *reinterpret_cast<hrf_f4*>(p) = v;
#endif
}
// 4 consecutive floats p[0..3] of which the first `nvalid` exist (the others read as 0); unaligned 16-byte load when all
// four exist, selected ADDRESSES otherwise (never a load under a branch: hrf_lin / DESIGN 2)
__device__ __forceinline__ hrf_f4 gl_ld4(const float* p, int nvalid) {
  if (nvalid >= 4) return hrf_ld4(p);
  hrf_f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = *(e < nvalid ? p + e : g_zero4l);
  return r;
}

// row[idx .. idx+3] of a row of `len` >= 4 floats, elements at or beyond `len` read as 0 (idx may lie beyond the row): ONE
// unconditional 16-byte load from a clamped index, then register selects - a load under a (per-lane) condition gets its own
// dependent round trip, which is what a ragged K / N would cost in every step of the pipeline otherwise
__device__ __forceinline__ hrf_f4 ld4_ragged(const float* row, int idx, int len) {
  const int ic = idx > len - 4 ? len - 4 : idx;
  const int d = idx - ic;                                // 0: aligned with the request; 1..3: shifted; >= 4: nothing valid
  const hrf_f4 v = hrf_ld4(row + ic);
  hrf_f4 o;
  o[0] = d == 0 ? v[0] : (d == 1 ? v[1] : (d == 2 ? v[2] : (d == 3 ? v[3] : 0.f)));
  o[1] = d == 0 ? v[1] : (d == 1 ? v[2] : (d == 2 ? v[3] : 0.f));
  o[2] = d == 0 ? v[2] : (d == 1 ? v[3] : 0.f);
  o[3] = d == 0 ? v[3] : 0.f;
  return o;
}

// per-channel (sum v, sum v*w) over the pixels of a wave's tiles -> atomics into the block's replicated copy.
// acc[rr][tt][r] belongs to channel cb + tt*16 + 4q + r and pixel lane i of row tile rr.
template <int WN>
__device__ __forceinline__ void block_moments(float* sRed, double* stats, int N, int n0, int chg, int rg, int lane,
                                              const hrf_f4 (*v)[4], const hrf_f4 (*w)[4], const bool* rowv) {
  // sRed: [8 / WN row groups][2][NB] floats
  constexpr int NB = WN * 64, RGN = 8 / WN;
  const int i = lane & 15, q = lane >> 4;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) {
      const float m = rowv[rr] ? 1.f : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] += v[rr][tt][r] * m; s2[r] += v[rr][tt][r] * w[rr][tt][r] * m; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = hrf_row16_sum(s1[r]); s2[r] = hrf_row16_sum(s2[r]); }
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = chg * 64 + tt * 16 + 4 * q + r;
        sRed[(rg * 2 + 0) * NB + c] = s1[r];
        sRed[(rg * 2 + 1) * NB + c] = s2[r];
      }
    }
  }
  __syncthreads();
  double* st = stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * N;
  for (int e = threadIdx.x; e < 2 * NB; e += NTHR) {
    const int which = e / NB, c = e - which * NB;
    if (n0 + c < N) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < RGN; ++g) s += sRed[(g * 2 + which) * NB + c];
      hrf_atomic_add(&st[which * N + n0 + c], (double)s);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
// WN = 64-channel groups per block (1, 2 or 4).  8 waves: wave -> (channel group = wave % WN, row group = wave / WN);
// each wave owns WN consecutive 16-row tiles x 64 channels.
template <int WN, int TF>
__global__ __launch_bounds__(NTHR) void lin2_fwd_kernel(HrfGroup<LinFwdArgs> grp) {
  const LinFwdArgs& a = grp.sel();
  constexpr int NB = WN * 64, SLOT = (GROWS + NB) * GLP;
  constexpr int NWV = (NB * 4 + NTHR - 1) / NTHR;      // weight float4 per thread and step
  constexpr bool TBLS = TF != HRF_TF_NONE;
  HRF_DYN_SMEM(float, smem);                            // [2][SLOT] ring | [2][TBL] per-k tables | reduction scratch
  float* sTab = smem + 2 * SLOT;                        // scale | shift (BatchNorm affine or LayerNorm gamma / beta)
  float* sRed = sTab + (TBLS ? 2 * TBL : 0);            // [8 / WN][2][NB] moments, or [GROWS][WN] row sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: tile predicates become scalar branches
  const int i = lane & 15, q = lane >> 4;
  const int chg = wave % WN, rg = wave / WN;
  const long m0 = (long)blockIdx.x * GROWS;
  const int n0 = blockIdx.y * NB;

  // ---- per-k transform tables in LDS (finalised on load when the producer's BatchNorm is handed over as moments)
  if (TBLS) {
    if (TF != HRF_TF_LN && a.fin.stats != nullptr) {
      hrf_bn_fin_onload(a.fin, sTab, sTab + TBL, tid, NTHR, blockIdx.x == 0 && blockIdx.y == 0);
    } else {
      for (int k = tid; k < a.K; k += NTHR) { sTab[k] = a.tf_scale[k]; sTab[TBL + k] = a.tf_shift[k]; }
    }
    for (int k = a.K + tid; k < ((a.K + 15) & ~15); k += NTHR) { sTab[k] = 0.f; sTab[TBL + k] = 0.f; }
  }

  hrf_f4 acc[WN][4];
#pragma unroll
  for (int rr = 0; rr < WN; ++rr)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[rr][tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  // staging: x tile = 128 rows x 4 float4 (one per thread); weight tile = NB rows x 4 float4
  const long xr = m0 + (tid >> 2) < a.M ? m0 + (tid >> 2) : a.M - 1;          // tail rows re-read the last row, never stored
  const int xdst = (tid >> 2) * GLP + 4 * (tid & 3);
  float mean = 0.f, rstd = 1.f;
  if (TF == HRF_TF_LN) { mean = a.tf_rowstat[2 * xr]; rstd = a.tf_rowstat[2 * xr + 1]; }
  const float* xrow = a.x + xr * a.ldX;
  const float* wsrc[NWV]; int wdst[NWV]; float wmask[NWV];
#pragma unroll
  for (int e = 0; e < NWV; ++e) {
    const int f = tid + e * NTHR, n = f >> 2;
    const bool on = f < NB * 4 && n0 + n < a.N;
    wmask[e] = on ? 1.f : 0.f;                           // rows beyond N: a valid row is read and multiplied by 0
    wsrc[e] = a.w + (long)(on ? n0 + n : 0) * a.K;
    wdst[e] = f < NB * 4 ? (GROWS + n) * GLP + 4 * (f & 3) : -1;
  }
  hrf_f4 xpre, wpre[NWV];
  int kpre = 0;
  auto load_tile = [&](int s) L2_INLINE {
    const int kb = s * GK + 4 * (tid & 3);
    kpre = kb;
    if ((s + 1) * GK <= a.K) {                           // (uniform) whole step inside the rows: plain 16-byte loads
      xpre = hrf_ld4(xrow + kb);
#pragma unroll
      for (int e = 0; e < NWV; ++e) wpre[e] = hrf_ld4(wsrc[e] + kb);
    } else {
      xpre = ld4_ragged(xrow, kb, a.K);
#pragma unroll
      for (int e = 0; e < NWV; ++e) wpre[e] = ld4_ragged(wsrc[e], kb, a.K);
    }
#pragma unroll
    for (int e = 0; e < NWV; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) wpre[e][r] *= wmask[e];
  };
  auto store_tile = [&](int slot) L2_INLINE {
    float* d = smem + slot * SLOT;
    hrf_f4 v = xpre;
    if (TBLS) {
      const hrf_f4 sc = l2_ld4(sTab + kpre), sh = l2_ld4(sTab + TBL + kpre);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = TF == HRF_TF_LN ? fmaf((v[r] - mean) * rstd, sc[r], sh[r]) : fmaf(v[r], sc[r], sh[r]);
        v[r] = TF == HRF_TF_AFFINE_RELU ? fmaxf(u, 0.f) : (TF == HRF_TF_AFFINE_GELU ? hrf_gelu(u) : u);
      }
    }
    l2_st4(d + xdst, v);
#pragma unroll
    for (int e = 0; e < NWV; ++e)
      if (wdst[e] >= 0) l2_st4(d + wdst[e], wpre[e]);
  };
  hrf_f4 fa[2][WN], fb[2][4];
  const float* abase = smem + (rg * 16 * WN + i) * GLP + 4 * q;
  const float* bbase = smem + (GROWS + chg * 64 + i) * GLP + 4 * q;
  auto read_frags = [&](int slot, int set) L2_INLINE {
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) fa[set][rr] = l2_ld4(abase + slot * SLOT + rr * 16 * GLP);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) fb[set][tt] = l2_ld4(bbase + slot * SLOT + tt * 16 * GLP);
  };
  bool tile_on[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) tile_on[tt] = n0 + chg * 64 + tt * 16 < a.N;
  const bool all_on = tile_on[3];                       // (tiles switch off from the top: the common case is all four)
  // ALL: every channel tile of this wave is inside N - the whole K loop is compiled twice, so that the common case carries
  // no per-tile predicate (and no accumulator copies at the joins of a predicated version)
  auto mma = [&](auto ALL, int set, int half) L2_INLINE {
    if (decltype(ALL)::value) {
#pragma unroll
      for (int m = 2 * half; m < 2 * half + 2; ++m)
#pragma unroll
        for (int rr = 0; rr < WN; ++rr)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) acc[rr][tt] = hrf_mfma16(fb[set][tt][m], fa[set][rr][m], acc[rr][tt]);
    } else {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        if (!tile_on[tt]) break;
#pragma unroll
        for (int m = 2 * half; m < 2 * half + 2; ++m)
#pragma unroll
          for (int rr = 0; rr < WN; ++rr) acc[rr][tt] = hrf_mfma16(fb[set][tt][m], fa[set][rr][m], acc[rr][tt]);
      }
    }
  };

  const int S = (a.K + GK - 1) / GK;
  if (TBLS) __syncthreads();                            // tables complete before the first transform
  load_tile(0);
  store_tile(0);
  if (S > 1) { load_tile(1); store_tile(1); }
  if (S > 2) load_tile(2);
  __syncthreads();
  read_frags(0, 0);
  L2_WAIT_LDS();
  __syncthreads();                                      // slot 0 is rewritten in step 0: every wave holds its fragments first
  auto step = [&](auto ALL, int s, int set) L2_INLINE {
    L2_SCHED_FENCE();
    mma(ALL, set, 0);
    L2_SCHED_FENCE();
    if (s + 2 < S) store_tile(s & 1);
    if (s + 3 < S) load_tile(s + 3);
    read_frags((s + 1) & 1, set ^ 1);          // complete since the previous barrier (stale but valid after the last step)
    L2_SCHED_FENCE();
    mma(ALL, set, 1);
    L2_SCHED_FENCE();
    __syncthreads();
  };
  auto kloop = [&](auto ALL) L2_INLINE {
    for (int s = 0; s < S; s += 2) {
      step(ALL, s, 0);
      if (s + 1 < S) step(ALL, s + 1, 1);
    }
  };
  if (all_on) kloop(std::true_type{}); else kloop(std::false_type{});

  // ---- epilogue: acc[rr][tt][r] = y(row m0 + rg*16*WN + rr*16 + i, channel n0 + chg*64 + tt*16 + 4q + r)
  bool rowv[WN];
#pragma unroll
  for (int rr = 0; rr < WN; ++rr) rowv[rr] = m0 + rg * 16 * WN + rr * 16 + i < a.M;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int ch = n0 + chg * 64 + tt * 16 + 4 * q;
    const int nval = tile_on[tt] ? a.N - ch : 0;
    const hrf_f4 bv = gl_ld4(a.bias != nullptr ? a.bias + ch : g_zero4l, a.bias != nullptr ? nval : 0);
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) {
      const long m = m0 + rg * 16 * WN + rr * 16 + i;
      const long mc = rowv[rr] ? m : a.M - 1;
      hrf_f4 v = acc[rr][tt];
      const hrf_f4 r1 = gl_ld4(a.res != nullptr ? a.res + mc * a.ldR + ch : g_zero4l, a.res != nullptr ? nval : 0);
      const hrf_f4 r2 = gl_ld4(a.res2 != nullptr ? a.res2 + mc * a.ldR + ch : g_zero4l, a.res2 != nullptr ? nval : 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = r < nval ? v[r] + bv[r] + r1[r] + r2[r] : 0.f;
      acc[rr][tt] = v;
      if (rowv[rr] && nval > 0) {
        float* o = a.y + m * a.ldY + a.yoff + ch;
        if (nval >= 4) hrf_st4(o, v);
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nval) o[r] = v[r];
        }
      }
    }
  }
  if (a.ln_out != nullptr && gridDim.y == 1) {
    // LayerNorm (mean, rstd) of the output rows, two passes like ln_stats_kernel: a row's channels sit in the WN waves of
    // its row group (4 tiles x 4 registers x 4 lane groups each); partial sums meet in sRed[row][chg]
    float mu[WN];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int rr = 0; rr < WN; ++rr) {
        float s = 0.f;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool cv = chg * 64 + tt * 16 + 4 * q + r < a.N;
            const float d = pass == 0 ? acc[rr][tt][r] : acc[rr][tt][r] - mu[rr];
            s += cv ? (pass == 0 ? d : d * d) : 0.f;
          }
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        if (q == 0) sRed[(rg * 16 * WN + rr * 16 + i) * WN + chg] = s;
      }
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < WN; ++rr) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < WN; ++c) t += sRed[(rg * 16 * WN + rr * 16 + i) * WN + c];
        if (pass == 0) mu[rr] = t / (float)a.N;
        else if (chg == 0 && q == 0 && rowv[rr]) {
          const long m = m0 + rg * 16 * WN + rr * 16 + i;
          a.ln_out[2 * m] = mu[rr];
          a.ln_out[2 * m + 1] = 1.0f / sqrtf(t / (float)a.N + a.ln_eps);
        }
      }
      __syncthreads();
    }
  }
  if (a.stats != nullptr) block_moments<WN>(sRed, a.stats, a.N, n0, chg, rg, lane, acc, acc, rowv);
}

// ------------------------------------------------------------------------------------------------------ backward data
// dX[m][n] = sum_k d(m, k) * W[k][n],  d = BNB ? cA[k]*dy + cB[k]*yraw + cC[k] : dy.  The weight tile [n][k-step] is staged
// TRANSPOSED from the (out = k, in = n) rows of W.  epi: * act'(sc[n]*xraw + sh[n]) and the (sum du, sum du*xraw) moments.
template <int WN, bool BNB>
__global__ __launch_bounds__(NTHR) void lin2_bwd_data_kernel(HrfGroup<LinBwdDataArgs> grp) {
  const LinBwdDataArgs& a = grp.sel();
  constexpr int NB = WN * 64, SLOT = (GROWS + NB) * GLP;
  constexpr int NWV = (NB * 4 + NTHR - 1) / NTHR;      // weight float4 (along n) per thread and step
  HRF_DYN_SMEM(float, smem);
  float* sTab = smem + 2 * SLOT;                        // cA | cB | cC
  float* sRed = sTab + (BNB ? 3 * TBL : 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: tile predicates become scalar branches
  const int i = lane & 15, q = lane >> 4;
  const int chg = wave % WN, rg = wave / WN;
  const long m0 = (long)blockIdx.x * GROWS;
  const int n0 = blockIdx.y * NB;

  if (BNB) {
    if (a.bfin.gstats != nullptr) {
      hrf_bn_bfin_onload(a.bfin, sTab, sTab + TBL, sTab + 2 * TBL, tid, NTHR, blockIdx.x == 0 && blockIdx.y == 0);
    } else {
      for (int k = tid; k < a.K; k += NTHR) { sTab[k] = a.cA[k]; sTab[TBL + k] = a.cB[k]; sTab[2 * TBL + k] = a.cC[k]; }
    }
    for (int k = a.K + tid; k < ((a.K + 15) & ~15); k += NTHR) { sTab[k] = 0.f; sTab[TBL + k] = 0.f; sTab[2 * TBL + k] = 0.f; }
  }

  hrf_f4 acc[WN][4];
#pragma unroll
  for (int rr = 0; rr < WN; ++rr)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[rr][tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  const long xr = m0 + (tid >> 2) < a.M ? m0 + (tid >> 2) : a.M - 1;
  const float* drow = a.dy + xr * a.ldD + a.doff;
  const float* yrow = BNB ? a.yraw + xr * a.ldD + a.doff : nullptr;
  const int xdst = (tid >> 2) * GLP + 4 * (tid & 3);
  // weight staging: thread -> (kk = f & 15, n4 = f >> 4): a float4 of W[k][n .. n+3], written to LDS rows n .. n+3 at column kk
  int wkk[NWV], wn[NWV]; bool won[NWV];
#pragma unroll
  for (int e = 0; e < NWV; ++e) {
    const int f = tid + e * NTHR;
    wkk[e] = f & 15; wn[e] = 4 * (f >> 4);
    won[e] = f < NB * 4;
  }
  hrf_f4 dpre, ypre, wpre[NWV];
  int kpre = 0;
  auto load_tile = [&](int s) L2_INLINE {
    const int kb = s * GK + 4 * (tid & 3);
    kpre = kb;
    if ((s + 1) * GK <= a.K) {                           // (uniform)
      dpre = hrf_ld4(drow + kb);
      if (BNB) ypre = hrf_ld4(yrow + kb);
    } else {
      dpre = ld4_ragged(drow, kb, a.K);
      if (BNB) ypre = ld4_ragged(yrow, kb, a.K);
    }
#pragma unroll
    for (int e = 0; e < NWV; ++e) {
      const int k = s * GK + wkk[e];
      const float km = (won[e] && k < a.K) ? 1.f : 0.f;  // rows beyond K: row K-1 is read and multiplied by 0
      wpre[e] = ld4_ragged(a.w + (long)(k < a.K ? k : a.K - 1) * a.N, n0 + wn[e], a.N);
#pragma unroll
      for (int r = 0; r < 4; ++r) wpre[e][r] *= km;
    }
  };
  auto store_tile = [&](int slot) L2_INLINE {
    float* d = smem + slot * SLOT;
    hrf_f4 v = dpre;
    if (BNB) {
      const hrf_f4 ca = l2_ld4(sTab + kpre), cb = l2_ld4(sTab + TBL + kpre), cc = l2_ld4(sTab + 2 * TBL + kpre);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf(ca[r], v[r], fmaf(cb[r], ypre[r], cc[r]));
    }
    l2_st4(d + xdst, v);
#pragma unroll
    for (int e = 0; e < NWV; ++e)
      if (won[e]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[(GROWS + wn[e] + r) * GLP + wkk[e]] = wpre[e][r];
      }
  };
  hrf_f4 fa[2][WN], fb[2][4];
  const float* abase = smem + (rg * 16 * WN + i) * GLP + 4 * q;
  const float* bbase = smem + (GROWS + chg * 64 + i) * GLP + 4 * q;
  auto read_frags = [&](int slot, int set) L2_INLINE {
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) fa[set][rr] = l2_ld4(abase + slot * SLOT + rr * 16 * GLP);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) fb[set][tt] = l2_ld4(bbase + slot * SLOT + tt * 16 * GLP);
  };
  bool tile_on[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) tile_on[tt] = n0 + chg * 64 + tt * 16 < a.N;
  const bool all_on = tile_on[3];                       // (tiles switch off from the top: the common case is all four)
  // ALL: every channel tile of this wave is inside N - the whole K loop is compiled twice, so that the common case carries
  // no per-tile predicate (and no accumulator copies at the joins of a predicated version)
  auto mma = [&](auto ALL, int set, int half) L2_INLINE {
    if (decltype(ALL)::value) {
#pragma unroll
      for (int m = 2 * half; m < 2 * half + 2; ++m)
#pragma unroll
        for (int rr = 0; rr < WN; ++rr)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) acc[rr][tt] = hrf_mfma16(fb[set][tt][m], fa[set][rr][m], acc[rr][tt]);
    } else {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        if (!tile_on[tt]) break;
#pragma unroll
        for (int m = 2 * half; m < 2 * half + 2; ++m)
#pragma unroll
          for (int rr = 0; rr < WN; ++rr) acc[rr][tt] = hrf_mfma16(fb[set][tt][m], fa[set][rr][m], acc[rr][tt]);
      }
    }
  };

  const int S = (a.K + GK - 1) / GK;
  if (BNB) __syncthreads();
  load_tile(0);
  store_tile(0);
  if (S > 1) { load_tile(1); store_tile(1); }
  if (S > 2) load_tile(2);
  __syncthreads();
  read_frags(0, 0);
  L2_WAIT_LDS();
  __syncthreads();                                      // slot 0 is rewritten in step 0: every wave holds its fragments first
  auto step = [&](auto ALL, int s, int set) L2_INLINE {
    L2_SCHED_FENCE();
    mma(ALL, set, 0);
    L2_SCHED_FENCE();
    if (s + 2 < S) store_tile(s & 1);
    if (s + 3 < S) load_tile(s + 3);
    read_frags((s + 1) & 1, set ^ 1);          // complete since the previous barrier (stale but valid after the last step)
    L2_SCHED_FENCE();
    mma(ALL, set, 1);
    L2_SCHED_FENCE();
    __syncthreads();
  };
  auto kloop = [&](auto ALL) L2_INLINE {
    for (int s = 0; s < S; s += 2) {
      step(ALL, s, 0);
      if (s + 1 < S) step(ALL, s + 1, 1);
    }
  };
  if (all_on) kloop(std::true_type{}); else kloop(std::false_type{});

  bool rowv[WN];
#pragma unroll
  for (int rr = 0; rr < WN; ++rr) rowv[rr] = m0 + rg * 16 * WN + rr * 16 + i < a.M;
  hrf_f4 xr4[WN][4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int ch = n0 + chg * 64 + tt * 16 + 4 * q;
    const int nval = tile_on[tt] ? a.N - ch : 0;
    hrf_f4 sc = hrf_f4{0.f, 0.f, 0.f, 0.f}, sh = sc;
    if (a.epi == 1) { sc = gl_ld4(a.tf_scale + ch, nval); sh = gl_ld4(a.tf_shift + ch, nval); }
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) {
      const long m = m0 + rg * 16 * WN + rr * 16 + i;
      const long mc = rowv[rr] ? m : a.M - 1;
      hrf_f4 v = acc[rr][tt];
      xr4[rr][tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};
      if (a.epi == 1) {
        const hrf_f4 xv = gl_ld4(a.xraw + mc * a.ldXr + ch, nval);
        xr4[rr][tt] = xv;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= hrf_act_grad(a.act, fmaf(xv[r], sc[r], sh[r]));
      } else if (a.accumulate) {
        const hrf_f4 p = gl_ld4(a.dx + mc * a.ldDx + ch, nval);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = r < nval ? v[r] : 0.f;
      acc[rr][tt] = v;
      if (rowv[rr] && nval > 0) {
        float* o = a.dx + m * a.ldDx + ch;
        if (nval >= 4) hrf_st4(o, v);
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nval) o[r] = v[r];
        }
      }
    }
  }
  if (a.epi == 1 && a.stats != nullptr) block_moments<WN>(sRed, a.stats, a.N, n0, chg, rg, lane, acc, xr4, rowv);
}

inline int pick_wn(long M, int N) {
  if (g_l2_knob[1] == 1 || g_l2_knob[1] == 2 || g_l2_knob[1] == 4) return g_l2_knob[1];      // tests: force a block width
  const long mt = (M + GROWS - 1) / GROWS;
  if (N > 128 && mt * ((N + 255) / 256) >= 96) return 4;      // enough blocks to occupy the chip at one block per CU
  if (N > 64) return 2;
  return 1;
}
inline size_t smem_bytes(int wn, int tables) {
  const int NB = wn * 64;
  const int red = (8 / wn) * 2 * NB > GROWS * wn ? (8 / wn) * 2 * NB : GROWS * wn;
  return ((size_t)2 * (GROWS + NB) * GLP + (size_t)tables * TBL + red) * sizeof(float);
}
inline bool wide_enough(long M, int K, int N) {
  if (g_l2_knob[0] == 2) return false;
  if (K < 16 || N < 16 || M <= 0) return false;
  if (g_l2_knob[0] == 1) return true;
  // a block owns 128 rows: fewer than ~200 blocks cannot fill 256 CUs and the register-only kernels (64 rows per block, one
  // wave per 16 rows) win (measured on HRFuser-B's 24x40 and 12x20 branches: 31.9 -> 90.6 us for 312 -> 936 at 1920 rows)
  const long blocks = ((M + GROWS - 1) / GROWS) * ((N + 255) / 256);
  return (K < N ? K : N) >= 64 && blocks >= 200;
}

template <class KERN>
int set_smem(KERN kern, size_t smem) {
#ifndef HRF_EMUL
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return HRF_ERR_LAUNCH;
#endif
  return HRF_OK;
}

template <int WN, int TF>
int launch_fwd(const LinFwdArgs& a, void* stream) {
  const size_t smem = smem_bytes(WN, TF != HRF_TF_NONE ? 2 : 0);
  static bool once = false;
  if (!once) { if (set_smem(&lin2_fwd_kernel<WN, TF>, smem) != HRF_OK) return HRF_ERR_LAUNCH; once = true; }
  const dim3 grid(hrf_cdiv(a.M, GROWS), hrf_cdiv(a.N, WN * 64));
  return HRF_LAUNCH_G((lin2_fwd_kernel<WN, TF>), grid, dim3(NTHR), (unsigned)smem, stream, a);
}
template <int WN>
int launch_fwd_tf(const LinFwdArgs& a, void* stream) {
  switch (a.tf_mode) {
    case HRF_TF_NONE: return launch_fwd<WN, HRF_TF_NONE>(a, stream);
    case HRF_TF_AFFINE: return launch_fwd<WN, HRF_TF_AFFINE>(a, stream);
    case HRF_TF_AFFINE_RELU: return launch_fwd<WN, HRF_TF_AFFINE_RELU>(a, stream);
    case HRF_TF_AFFINE_GELU: return launch_fwd<WN, HRF_TF_AFFINE_GELU>(a, stream);
    case HRF_TF_LN: return launch_fwd<WN, HRF_TF_LN>(a, stream);
    default: return -1;
  }
}
template <int WN, bool BNB>
int launch_bwd(const LinBwdDataArgs& a, void* stream) {
  const size_t smem = smem_bytes(WN, BNB ? 3 : 0);
  static bool once = false;
  if (!once) { if (set_smem(&lin2_bwd_data_kernel<WN, BNB>, smem) != HRF_OK) return HRF_ERR_LAUNCH; once = true; }
  const dim3 grid(hrf_cdiv(a.M, GROWS), hrf_cdiv(a.N, WN * 64));
  return HRF_LAUNCH_G((lin2_bwd_data_kernel<WN, BNB>), grid, dim3(NTHR), (unsigned)smem, stream, a);
}

}  // namespace

// -1: shape not served by this engine (the caller falls back to lin_engine.hip)
int hrf_lin2_fwd_launch(const LinFwdArgs& a, void* stream) {
  if (!wide_enough(a.M, a.K, a.N)) return -1;
  if (a.tf_mode != HRF_TF_NONE && a.K > TBL) return -1;
  const int wn = pick_wn(a.M, a.N);
  return wn == 4 ? launch_fwd_tf<4>(a, stream) : (wn == 2 ? launch_fwd_tf<2>(a, stream) : launch_fwd_tf<1>(a, stream));
}
bool hrf_lin2_fwd_emits_ln(const LinFwdArgs& a) {
  if (!wide_enough(a.M, a.K, a.N) || (a.tf_mode != HRF_TF_NONE && a.K > TBL)) return false;
  return a.N <= pick_wn(a.M, a.N) * 64;
}
int hrf_lin2_bwd_data_launch(const LinBwdDataArgs& a, void* stream) {
  if (!wide_enough(a.M, a.K, a.N)) return -1;
  if (a.cA != nullptr && a.K > TBL) return -1;
  const int wn = pick_wn(a.M, a.N);
  if (a.cA != nullptr) return wn == 4 ? launch_bwd<4, true>(a, stream) : (wn == 2 ? launch_bwd<2, true>(a, stream) : launch_bwd<1, true>(a, stream));
  return wn == 4 ? launch_bwd<4, false>(a, stream) : (wn == 2 ? launch_bwd<2, false>(a, stream) : launch_bwd<1, false>(a, stream));
}
extern "C" int hrf_lin2_knob(int key, int value) {
  if (key < 0 || key >= 4) return HRF_ERR_ARG;
  g_l2_knob[key] = value;
  return HRF_OK;
}
