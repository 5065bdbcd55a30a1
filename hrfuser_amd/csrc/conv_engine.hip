// C-ABI entry points of every dense convolution / Linear of the HRFuser backbone (forward, backward
// data, backward weight) for gfx950, NHWC activations, OIHW weights, and two of the engines behind them:
//
//   hrf_conv_fwd / hrf_conv_bwd_data dispatch, by shape, to
//     lin_engine.hip    stride-1 1x1 on channel-contiguous rows: LDS-free row-GEMM kernels
//     conv3_engine.hip  3x3 with >= 32 input channels: halo-staged engine (fwd, bwd s1, bwd s2 by parity)
//     this file         everything else (NCHW stem input, strided 3x3 forward, narrow 3x3):
//                       conv_fwd_kernel / conv_bwd_data_kernel, LDS implicit GEMM, 64x64 tiles
//   hrf_conv_bwd_weight dispatches to
//     wgrad_dense_kernel (this file)  1x1 and NHWC 3x3: pixel-major fragments straight from global
//     conv_bwd_wgt_kernel (this file) the remaining strided/NCHW cases
//
// All contractions run on v_mfma_f32_16x16x4_f32 (exact fp32: parity with the fp32 reference is a
// hard requirement - SURVEY.md 7 "hard parts": bf16/fp16 inputs fail the 1e-3 gate).
// BatchNorm / LayerNorm / activation are never materialised: the loaders read the producer's RAW
// output and apply the per-channel affine / activation on the fragment ("transform on load"); the
// epilogues emit the per-channel sums the next BatchNorm needs into HRF_STAT_COPIES replicated accumulators
// (same-address global atomics serialise at ~25 ns each on MI355X, see include/hrfuser_hip.h).
//
// Shape of the problem (HRFuser-T, 2 images/GPU): M = 480..30720 pixels, K = 18..2304, N = 18..576,
// ~1200 launches per training step, each far too small to fill 256 CUs: the kernels are built for
// LATENCY (all global loads of a batch issued unconditionally and back-to-back, compile-time tile
// counts instead of guards, small code: the instruction cache is cold at every launch).
//
// Reference ops replaced: every nn.Conv2d(k=1|3, groups=1) + nn.Linear reached from
// mmdet/models/backbones/{hrfuser_hrformer_based,hrformer,hrnet,resnet}.py.
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "hrf_common.h"
#include "hrf_group.h"
#include "hrf_lin.h"
#include "../../include/hrfuser_hip.h"

#include "hrf_wgrad.h"

namespace {

constexpr int BM = 64;      // output rows (pixels) per block: one 16-row MFMA tile per wave
constexpr int BK = 64;      // K elements staged per step (16 MFMA k-substeps)
constexpr int LDK = 65;     // LDS pitch of the [row][k] tiles: bank = (row + k) % 32
constexpr float SENT = -1.0e30f;   // staged pre-activation of a zero-padded tap: relu/gelu(SENT) == 0

static int g_knob[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ float act_at_read(int tf, float u) {
  return tf == HRF_TF_AFFINE_RELU ? fmaxf(u, 0.f) : (tf == HRF_TF_AFFINE_GELU ? hrf_gelu(u) : u);
}

struct ConvFwdArgs {
  const float* x; const float* w; const float* bias;
  float* y; int ldY; int yoff;
  const float* res; int ldR; const float* res2;
  const float* tf_scale; const float* tf_shift; const float* tf_rowstat;
  double* stats;
  hrf_bn_fin_t fin;                            // fin.stats != null: BatchNorm of x finalised on load
  int B, H, W, Cin, Ho, Wo, Cout, stride, pad;
  int sB, sY, sX, sC;
  int M, K;
  int ksplit; float* part;                     // ksplit > 1: slice blockIdx.y / column blocks takes 1 / ksplit of the K steps; slice 0 stores its partial
                                               // tile (+ bias / residuals) to y, slice k >= 1 to part[(k - 1) * M * Cout ...] (dense [M][Cout]); no moments
};

// --------------------------------------------------------------------------------- forward
template <int NT, int KH, int TF>
__global__ __launch_bounds__(256) void conv_fwd_kernel(HrfGroup<ConvFwdArgs> grp) {
  const ConvFwdArgs& a = grp.sel();
  constexpr int BN = NT * 16, RP = BM / 4, RQ = BN / 4;
  __shared__ float As[BM * LDK];
  __shared__ float Bs[BN * LDK];
  __shared__ float sStat[4 * 2 * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = tid & 63, r0 = tid >> 6;
  const int nyb = (a.Cout + BN - 1) / BN, ksl = (int)blockIdx.y / nyb;           // (split over K: blockIdx.y = slice * nyb + column block;
  const int m0 = blockIdx.x * BM, n0 = ((int)blockIdx.y - ksl * nyb) * BN;        //  blockIdx.z is the problem index of HrfGroup)
  __shared__ float sFin[(TF >= HRF_TF_AFFINE && TF <= HRF_TF_AFFINE_GELU) ? 2 * HRF_FIN_MAXC : 4];
  const float* scp = a.tf_scale;
  const float* shp = a.tf_shift;
  if (TF >= HRF_TF_AFFINE && TF <= HRF_TF_AFFINE_GELU) {
    if (a.fin.stats != nullptr) {
      hrf_bn_fin_onload(a.fin, sFin, sFin + HRF_FIN_MAXC, tid, 256, blockIdx.x == 0 && blockIdx.y == 0);
      __syncthreads();
      scp = sFin; shp = sFin + HRF_FIN_MAXC;
    }
  }

  // per-thread row bookkeeping for its 16 staged rows (r0 + 4p)
  int rb[RP];
  unsigned ryx[KH == 3 ? RP : 1];
  float rmean[TF == HRF_TF_LN ? RP : 1], rrstd[TF == HRF_TF_LN ? RP : 1];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    const int m = m0 + r0 + 4 * p;
    const bool mv = m < a.M;
    const int mc = mv ? m : 0;
    const int b = mc / HoWo, rem = mc - b * HoWo;
    const int yo = rem / a.Wo, xo = rem - yo * a.Wo;
    const int ry = yo * a.stride - a.pad, rx = xo * a.stride - a.pad;
    if (KH == 3) {
      rb[p] = b * a.sB + ry * a.sY + rx * a.sX;
      ryx[p] = mv ? ((unsigned)(ry + 1) << 16) | (unsigned)(rx + 1) : 0xFFFF0000u;
    } else {
      rb[p] = mv ? b * a.sB + ry * a.sY + rx * a.sX : -1;
    }
    if (TF == HRF_TF_LN) { rmean[p] = a.tf_rowstat[2 * mc]; rrstd[p] = a.tf_rowstat[2 * mc + 1]; }
  }

  float areg[RP], breg[RQ];
  auto load_tile = [&](int k0) {
    const int k = k0 + kl;
    const bool kv = k < a.K;
    int dy = 0, dx = 0, ci = kv ? k : 0;
    if (KH == 3) { const int tap = ci / a.Cin; ci -= tap * a.Cin; dy = tap / 3; dx = tap - 3 * dy; }
    const int koff = dy * a.sY + dx * a.sX + ci * a.sC;
    float sc = 1.f, sh = 0.f;
    if (TF != HRF_TF_NONE) { sc = scp[ci]; sh = shp[ci]; }
    float raw[RP];
    bool okv[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      bool ok;
      if (KH == 3) {
        const int yy = (int)(ryx[p] >> 16) - 1 + dy, xx = (int)(ryx[p] & 0xFFFFu) - 1 + dx;
        ok = kv && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      } else {
        ok = kv && rb[p] >= 0;
      }
      okv[p] = ok;
      raw[p] = a.x[ok ? rb[p] + koff : 0];
    }
    const int wk = (KH == 3) ? ci * 9 + dy * 3 + dx : ci;
    const int wrow = (KH == 3) ? a.Cin * 9 : a.Cin;
    float wraw[RQ];
    bool wok[RQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int n = n0 + r0 + 4 * q;
      wok[q] = kv && n < a.Cout;
      wraw[q] = a.w[wok[q] ? n * wrow + wk : 0];
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float v = raw[p];
      if (TF == HRF_TF_LN) v = fmaf((v - rmean[p]) * rrstd[p], sc, sh);
      else if (TF != HRF_TF_NONE) v = fmaf(v, sc, sh);
      const float zero = (TF == HRF_TF_AFFINE_RELU || TF == HRF_TF_AFFINE_GELU) ? SENT : 0.f;
      areg[p] = okv[p] ? v : zero;
    }
#pragma unroll
    for (int q = 0; q < RQ; ++q) breg[q] = wok[q] ? wraw[q] : 0.f;
  };

  hrf_f4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  // split over K (a.ksplit > 1, slice ksl): the 2 304-deep stride-2 transition convolutions have 120 row blocks for 256 CUs and 36
  // serial 64-deep steps each; four slices of 9 steps fill the chip; their partial tiles are added in a FIXED order by
  // splitk_reduce_kernel (no atomics: the forward stays bit-reproducible)
  const int nstep = (a.K + BK - 1) / BK, sper = (nstep + a.ksplit - 1) / a.ksplit;
  const int klo = ksl * sper * BK, khi = min(a.K, klo + sper * BK);
  load_tile(klo);
  for (int k0 = klo; k0 < khi; k0 += BK) {
#pragma unroll
    for (int p = 0; p < RP; ++p) As[(r0 + 4 * p) * LDK + kl] = areg[p];
#pragma unroll
    for (int q = 0; q < RQ; ++q) Bs[(r0 + 4 * q) * LDK + kl] = breg[q];
    __syncthreads();
    if (k0 + BK < khi) load_tile(k0 + BK);
    const int ksub = min(16, (khi - k0 + 3) >> 2);
#pragma unroll 4
    for (int kk = 0; kk < ksub; ++kk) {
      const int kc = kk * 4 + (lane >> 4);
      const float af = act_at_read(TF, As[(wave * 16 + (lane & 15)) * LDK + kc]);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = hrf_mfma16(af, Bs[(j * 16 + (lane & 15)) * LDK + kc], acc[j]);
    }
    __syncthreads();
  }

  // epilogue: bias, residual(s), store, per-channel (sum, sumsq) for the following BatchNorm
  const int col = lane & 15;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + col;
    const bool nv = n < a.Cout;
    const bool first = ksl == 0;                        // (split over K: bias / residuals join the first slice)
    const float bv = (a.bias != nullptr && first) ? a.bias[nv ? n : 0] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + (lane >> 4) * 4 + r;
      const bool ok = nv && m < a.M;
      float v = acc[j][r] + bv;
      const int ro = ok ? m * a.ldR + n : 0;
      if (a.res != nullptr && first) v += a.res[ro];
      if (a.res2 != nullptr && first) v += a.res2[ro];
      if (ok) {
        if (first) a.y[m * a.ldY + a.yoff + n] = v;
        else a.part[(size_t)(ksl - 1) * a.M * a.Cout + (size_t)m * a.Cout + n] = v;
        s1 += v; s2 = fmaf(v, v, s2);
      }
    }
    if (a.stats != nullptr) {
      s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
      if (lane < 16) { sStat[wave * 2 * BN + j * 16 + lane] = s1; sStat[wave * 2 * BN + BN + j * 16 + lane] = s2; }
    }
  }
  if (a.stats != nullptr) {
    __syncthreads();
    if (tid < 2 * BN) {
      const int c = tid < BN ? tid : tid - BN;
      if (n0 + c < a.Cout) {
        const float s = sStat[tid] + sStat[2 * BN + tid] + sStat[4 * BN + tid] + sStat[6 * BN + tid];
        hrf_atomic_add(&a.stats[(size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * a.Cout + (tid < BN ? 0 : a.Cout) + n0 + c], (double)s);
      }
    }
  }
}

// --------------------------------------------------------------------------------- backward data
struct ConvBwdDataArgs {
  const float* dy; int ldD; int doff;          // incoming grad wrt conv output (B,Ho,Wo,ldD)
  const float* yraw;                           // raw conv output (same indexing) for bn-bwd
  const float* cA; const float* cB; const float* cC;   // bn-bwd coefficients per Cout (nullable)
  const float* w;
  float* dx; int sB, sY, sX, sC; int accumulate;       // epi 0: dX (generic strides)
  int epi;                                     // 0 store/accumulate, 1 act-backward + stats
  const float* xraw; int ldXr;                 // epi 1: raw producer output (NHWC, ld)
  const float* tf_scale; const float* tf_shift; int act;
  double* stats;                               // epi 1: (sum du, sum du*xraw) per Cin
  hrf_bn_bfin_t bfin;                          // bfin.gstats != null: cA/cB/cC derived on load
  int B, H, W, Cin, Ho, Wo, Cout, stride, pad;
  int M, K;                                    // M = B*H*W, K = KH*KH*Cout
};

template <int NT, int KH, bool BNB>
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(HrfGroup<ConvBwdDataArgs> grp) {
  const ConvBwdDataArgs& a = grp.sel();
  constexpr int BN = NT * 16, RP = BM / 4, RQ = BN / 4;
  __shared__ float As[BM * LDK];
  __shared__ float Bs[BN * LDK];
  __shared__ float sStat[4 * 2 * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kl = tid & 63, r0 = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  __shared__ float sFin[BNB ? 3 * HRF_FIN_MAXC : 4];
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

  int rbase[RP];
  unsigned ryx[RP];
  const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int p = 0; p < RP; ++p) {
    const int m = m0 + r0 + 4 * p;
    const bool mv = m < a.M;
    const int mc = mv ? m : 0;
    const int b = mc / HW, rem = mc - b * HW;
    const int yi = rem / a.W, xi = rem - yi * a.W;
    rbase[p] = b * HoWo;
    ryx[p] = mv ? ((unsigned)(yi + a.pad) << 16) | (unsigned)(xi + a.pad) : 0xFFFFFFFFu;
  }
  float areg[RP], breg[RQ];
  auto load_tile = [&](int k0) {
    const int k = k0 + kl;
    const bool kv = k < a.K;
    int dyy = 0, dxx = 0, co = kv ? k : 0;
    if (KH == 3) { const int tap = co / a.Cout; co -= tap * a.Cout; dyy = tap / 3; dxx = tap - 3 * dyy; }
    float ca = 1.f, cb = 0.f, cc = 0.f;
    if (BNB) { ca = cAp[co]; cb = cBp[co]; cc = cCp[co]; }
    float dv[RP], yv[RP];
    bool okv[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      const int ty = (int)(ryx[p] >> 16) - dyy, tx = (int)(ryx[p] & 0xFFFFu) - dxx;
      bool ok = kv && ryx[p] != 0xFFFFFFFFu && ty >= 0 && tx >= 0;
      int yo = ty, xo = tx;
      if (a.stride == 2) { ok = ok && ((ty | tx) & 1) == 0; yo = ty >> 1; xo = tx >> 1; }
      ok = ok && yo < a.Ho && xo < a.Wo;
      const int idx = ok ? (rbase[p] + yo * a.Wo + xo) * a.ldD + a.doff + co : 0;
      okv[p] = ok;
      dv[p] = a.dy[idx];
      yv[p] = BNB ? a.yraw[idx] : 0.f;
    }
    float wraw[RQ];
    bool wok[RQ];
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int n = n0 + r0 + 4 * q;   // n = ci
      wok[q] = kv && n < a.Cin;
      const int wi = (KH == 3) ? (co * a.Cin + n) * 9 + dyy * 3 + dxx : co * a.Cin + n;
      wraw[q] = a.w[wok[q] ? wi : 0];
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) {
      float v = dv[p];
      if (BNB) v = fmaf(ca, v, fmaf(cb, yv[p], cc));
      areg[p] = okv[p] ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < RQ; ++q) breg[q] = wok[q] ? wraw[q] : 0.f;
  };

  hrf_f4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  for (int k0 = 0; k0 < a.K; k0 += BK) {
#pragma unroll
    for (int p = 0; p < RP; ++p) As[(r0 + 4 * p) * LDK + kl] = areg[p];
#pragma unroll
    for (int q = 0; q < RQ; ++q) Bs[(r0 + 4 * q) * LDK + kl] = breg[q];
    __syncthreads();
    if (k0 + BK < a.K) load_tile(k0 + BK);
    const int ksub = min(16, (a.K - k0 + 3) >> 2);
#pragma unroll 4
    for (int kk = 0; kk < ksub; ++kk) {
      const int kc = kk * 4 + (lane >> 4);
      const float af = As[(wave * 16 + (lane & 15)) * LDK + kc];
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = hrf_mfma16(af, Bs[(j * 16 + (lane & 15)) * LDK + kc], acc[j]);
    }
    __syncthreads();
  }

  const int col = lane & 15;
  int orow[4];
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + wave * 16 + (lane >> 4) * 4 + r;
    rowok[r] = m < a.M;
    const int mc = rowok[r] ? m : 0;
    const int b = mc / HW, rem = mc - b * HW;
    const int yi = rem / a.W, xi = rem - yi * a.W;
    orow[r] = b * a.sB + yi * a.sY + xi * a.sX;
  }
  // (one uniform branch per activation kind around the whole epilogue: hrf_with_act)
  hrf_with_act(a.epi == 1 ? a.act : HRF_ACT_NONE, [&](auto kind) HRF_KIND_INLINE {
    constexpr int ACT = decltype(kind)::value;
  #pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 16 + col;
      const bool nv = n < a.Cin;
      float sc = 1.f, sh = 0.f;
      if (a.epi == 1) { sc = a.tf_scale[nv ? n : 0]; sh = a.tf_shift[nv ? n : 0]; }
      float s1 = 0.f, s2 = 0.f;
  #pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 16 + (lane >> 4) * 4 + r;
        const bool ok = nv && rowok[r];
        const int o = ok ? orow[r] + n * a.sC : 0;
        float v = acc[j][r];
        if (a.epi == 1) {
          const float xr = a.xraw[ok ? m * a.ldXr + n : 0];
          v *= hrf_act_grad(ACT, fmaf(xr, sc, sh));
          if (ok) { s1 += v; s2 = fmaf(v, xr, s2); a.dx[o] = v; }
        } else {
          const float prev = a.accumulate ? a.dx[o] : 0.f;
          if (ok) a.dx[o] = prev + v;
        }
      }
      if (a.epi == 1 && a.stats != nullptr) {
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        if (lane < 16) { sStat[wave * 2 * BN + j * 16 + lane] = s1; sStat[wave * 2 * BN + BN + j * 16 + lane] = s2; }
      }
    }
  });
  if (a.epi == 1 && a.stats != nullptr) {
    __syncthreads();
    if (tid < 2 * BN) {
      const int c = tid < BN ? tid : tid - BN;
      if (n0 + c < a.Cin) {
        const float s = sStat[tid] + sStat[2 * BN + tid] + sStat[4 * BN + tid] + sStat[6 * BN + tid];
        hrf_atomic_add(&a.stats[(size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * a.Cin + (tid < BN ? 0 : a.Cin) + n0 + c], (double)s);
      }
    }
  }
}

// --------------------------------------------------------------------------------- backward weight
struct ConvBwdWgtArgs {
  const float* dy; int ldD; int doff; const float* yraw;
  const float* cA; const float* cB; const float* cC;
  const float* x; int sB, sY, sX, sC;
  int tf_mode; const float* tf_scale; const float* tf_shift; const float* tf_rowstat;
  float* dw; float* dbias;
  int B, H, W, Cin, Ho, Wo, Cout, stride, pad, KH;
  int Mpix, Np;                  // Mpix = B*Ho*Wo (reduction), Np = KH*KH*Cin
  int chunk;                     // pixels per split (multiple of WK)
  int dbg_plain;                 // tuning aid: plain stores instead of atomics (WRONG results)
};

constexpr int WK = 64;     // pixels (reduction elements) staged per step: 4 k-substeps per wave
constexpr int WLD = 65;    // LDS pitch of the [row][pixel] tiles

// TFA: activation applied to the X operand when its MFMA fragment is read (0 none, 2 relu, 3 gelu)
template <bool DENSE1, bool BNB, int TFA>
__global__ __launch_bounds__(256) void conv_bwd_wgt_kernel(ConvBwdWgtArgs a) {
  // tile: 64 (co) x 64 (n' = tap*Cin+ci); reduction over pixels.  Every wave owns 16 of the 64
  // pixels of a step (4 MFMA k-substeps) and a full 4x4 grid of 16x16 accumulators; the four
  // partial tiles are summed through LDS (plain ld/st rounds, no atomics) and each block issues
  // ONE fp32 atomic per output element - the split count is kept small by the launcher.
  __shared__ float As[64 * WLD];
  __shared__ float Bs[64 * WLD];
  __shared__ float sBias[4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int pbeg = blockIdx.z * a.chunk, pend = min(a.Mpix, pbeg + a.chunk);
  const int mtiles = min(4, (a.Cout - m0 + 15) / 16), ntiles = min(4, (a.Np - n0 + 15) / 16);

  const int co = m0 + lane;
  const bool cov = co < a.Cout;
  float ca = 1.f, cb = 0.f, cc = 0.f;
  if (BNB) { const int cs = cov ? co : 0; ca = a.cA[cs]; cb = a.cB[cs]; cc = a.cC[cs]; }
  const int np = n0 + lane;
  const bool npv = np < a.Np;
  int tap = 0, ci = npv ? np : 0;
  if (a.KH == 3) { tap = ci / a.Cin; ci -= tap * a.Cin; }
  const int dyy = tap / 3, dxx = tap - 3 * dyy;
  float sc = 1.f, sh = 0.f;
  if (a.tf_mode != HRF_TF_NONE) { sc = a.tf_scale[ci]; sh = a.tf_shift[ci]; }
  const bool ln = a.tf_mode == HRF_TF_LN;
  const int HoWo = a.Ho * a.Wo;
  float bias_part = 0.f;

  // Output tiles (16x16) are dealt round-robin to the four waves (tile t -> wave t % 4), so every
  // wave reduces its own tiles over ALL 64 pixels of a step: no cross-wave reduction, the epilogue
  // is one fp32 atomic per owned element straight from the accumulator registers.
  const int ntl = mtiles * ntiles;
  int ti[4], tj[4];
  bool tv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = wave + 4 * q;
    tv[q] = t < ntl;
    ti[q] = tv[q] ? t / ntiles : 0;
    tj[q] = tv[q] ? t - ti[q] * ntiles : 0;
  }
  hrf_f4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  float areg[16], breg[16];
  auto load_tile = [&](int p0) {
    int pix = p0 + wave;                       // this thread's pixel slots: p0 + wave + 4*s
    int b = 0, yo = 0, xo = 0;
    if (!DENSE1) { b = pix / HoWo; const int rem = pix - b * HoWo; yo = rem / a.Wo; xo = rem - yo * a.Wo; }
    // all loads of a step are issued unconditionally (clamped addresses) before any use
    float ad[16], ay[16], bx[16], bm[16], br[16];
    bool aok[16], bok[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bool pv = pix < pend;
      aok[s] = pv && cov;
      const int idx = aok[s] ? pix * a.ldD + a.doff + co : 0;
      ad[s] = a.dy[idx];
      ay[s] = BNB ? a.yraw[idx] : 0.f;
      int xo_ = 0, row = 0;
      bool ok = pv && npv;
      if (DENSE1) {
        row = pix;
        xo_ = pix * a.sX + ci * a.sC;
      } else {
        const int yi = yo * a.stride - a.pad + dyy, xi = xo * a.stride - a.pad + dxx;
        ok = ok && (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W;
        row = (b * a.H + yi) * a.W + xi;
        xo_ = b * a.sB + yi * a.sY + xi * a.sX + ci * a.sC;
      }
      bok[s] = ok;
      bx[s] = a.x[ok ? xo_ : 0];
      bm[s] = 0.f; br[s] = 0.f;
      if (ln) { bm[s] = a.tf_rowstat[ok ? 2 * row : 0]; br[s] = a.tf_rowstat[ok ? 2 * row + 1 : 0]; }
      pix += 4;
      if (!DENSE1) { xo += 4; while (xo >= a.Wo) { xo -= a.Wo; if (++yo == a.Ho) { yo = 0; ++b; } } }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float av = ad[s];
      if (BNB) av = fmaf(ca, av, fmaf(cb, ay[s], cc));
      float bv = bx[s];
      if (ln) bv = fmaf((bv - bm[s]) * br[s], sc, sh);
      else bv = fmaf(bv, sc, sh);               // (sc, sh) = (1, 0) when no transform
      areg[s] = aok[s] ? av : 0.f;
      breg[s] = bok[s] ? bv : (TFA != 0 ? SENT : 0.f);
    }
  };

  if (pbeg < pend) load_tile(pbeg);
  for (int p0 = pbeg; p0 < pend; p0 += WK) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      As[lane * WLD + wave + 4 * s] = areg[s];
      Bs[lane * WLD + wave + 4 * s] = breg[s];
      bias_part += areg[s];
    }
    __syncthreads();
    if (p0 + WK < pend) load_tile(p0 + WK);
#pragma unroll 2
    for (int kk = 0; kk < 16; ++kk) {
      const int kq = kk * 4 + (lane >> 4);
      const float bf0 = act_at_read(TFA, Bs[(tj[0] * 16 + (lane & 15)) * WLD + kq]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (tv[q]) {
          const float af = As[(ti[q] * 16 + (lane & 15)) * WLD + kq];
          const float bf = (q == 0 || tj[q] == tj[0]) ? bf0 : act_at_read(TFA, Bs[(tj[q] * 16 + (lane & 15)) * WLD + kq]);
          acc[q] = hrf_mfma16(af, bf, acc[q]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (tv[q]) {
      const int nn = n0 + tj[q] * 16 + (lane & 15);
      int t9 = 0, c9 = nn;
      if (a.KH == 3) { t9 = nn / a.Cin; c9 = nn - t9 * a.Cin; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cco = m0 + ti[q] * 16 + (lane >> 4) * 4 + r;
        if (cco < a.Cout && nn < a.Np) {
          const int o = a.KH == 3 ? (cco * a.Cin + c9) * 9 + t9 : cco * a.Cin + nn;
          if (a.dbg_plain) a.dw[o] = acc[q][r]; else hrf_atomic_add(&a.dw[o], acc[q][r]);
        }
      }
    }
  }
  sBias[wave * 64 + lane] = bias_part;
  __syncthreads();
  if (a.dbias != nullptr && blockIdx.y == 0 && tid < 64 && m0 + tid < a.Cout)
    hrf_atomic_add(&a.dbias[m0 + tid], sBias[tid] + sBias[64 + tid] + sBias[128 + tid] + sBias[192 + tid]);
}

// ----------------------------------------------------------- backward weight, dense 1x1 (no LDS staging)
// dW[co][ci] += sum_pix bnbwd(dY)[pix][co] * tf(X)[pix][ci] for stride-1 1x1 convs / Linears on
// channel-contiguous rows (90 % of the weight-gradient launches).  Both operands are "pixel-major",
// which is exactly the v_mfma_f32_16x16x4_f32 fragment layout (A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15] with k = pixel): fragments are loaded STRAIGHT from global memory, one
// dword per lane, no LDS staging, no barrier in the reduction loop.  Every wave streams its own
// pixels with WU k-steps (4 pixels each) of loads in flight and keeps an MT x NT grid of 16x16
// accumulators (compile-time: no guards in the unrolled code - executed code size and scalar
// branches, not FLOPs or bytes, set the duration of these ~5 us kernels: the instruction cache
// is cold at every launch).  The 8 waves of a block are merged through LDS with plain ld/st rounds
// (LDS float atomics run at ~1 lane/clk: measured 45 us for a 64x64 tile) and the block issues ONE
// coalesced fp32 atomic per output element; the split count is capped at 128 because same-address
// global atomics serialise at ~25 ns each.
constexpr int WUMAX = 8;   // k-steps (of 4 pixels) whose loads are issued before the first use
constexpr int WNW = 8;     // waves per block (512 threads)

// ACT: activation applied to the X operand (0 none, 1 ReLU, 2 GELU)
// TAP: 3x3 convolution (pad 1): the N index is n' = ci*9 + tap (the OIHW memory order, so the block's
// output atomics stay coalesced); every lane owns its own (ci, tap) and gathers the X operand from
// the correspondingly shifted input pixel (exact zero outside the image).
// TAPM 2 ("tap-blocked", NT = 9): tile j of the wave IS tap j and its 16 lanes are 16 CONSECUTIVE input channels of the
// shifted pixel - a 64-byte run per pixel instead of 16 scattered dwords (the n' = ci*9+tap order makes every lane of a
// fragment load its own (ci, tap): 64 cache-line lookups per wave instruction, which is what the 3x3 problems spent
// their time on), every dY fragment is reused by the 9 taps, and a block's output is 144 CONTIGUOUS floats per output
// channel (16 ci x 9 taps in OIHW order), written by a transposing final pass.
// phase stamps of one workgroup (wave 0) for tools/time_wgrad_phases.py: compiled in only with -DHRF_WG_TIMING
#if defined(HRF_WG_TIMING) && !defined(HRF_EMUL)
__device__ long long g_wg_t[16];
#define WG_T(k) do { if (blockIdx.x == (gridDim.x > 17 ? 17u : 0u) && threadIdx.x == 0) g_wg_t[k] = wall_clock64(); } while (0)
extern "C" int hrf_wgrad_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_t), sizeof(long long) * 16) == hipSuccess ? HRF_OK : HRF_ERR_LAUNCH;
}
#else
#define WG_T(k)
#endif
template <int MT, int NT, bool BNB, int ACT, int TAPM>
__global__ __launch_bounds__(64 * WNW) void wgrad_dense_kernel(WgradGroup grp_args) {
  WG_T(0);
  constexpr bool TAP = TAPM != 0, TAPB = TAPM == 2;
  static_assert(!TAPB || NT == 9, "tap-blocked: one tile per tap");
  int prob = 0;
  while (prob + 1 < grp_args.nprob && (int)blockIdx.x >= grp_args.bstart[prob + 1]) ++prob;   // (wave-uniform, <= 15 steps)
  const WgradDenseArgs& a = grp_args.p[prob];
  const int bid = (int)blockIdx.x - grp_args.bstart[prob];
  constexpr int WU = MT * NT >= 25 ? 2 : (MT * NT > 16 ? WUMAX / 2 : WUMAX);   // keep the many-tile variants inside 256 VGPRs
  // BatchNorm-backward coefficients of the many-tile variants live in LDS (12 loop-invariant registers at MT = 4 that the
  // allocator spilled: a scratch reload is a vector-memory load whose wait retires every outstanding load of the pixel loop)
  constexpr bool COEF_LDS = BNB && MT * NT >= 25;
  __shared__ float sCoef[COEF_LDS ? 3 * MT * 16 : 1];
  constexpr int NREG = MT * NT > 25 ? 1 : 2;                 // merge regions in LDS (one: 7 rounds instead of 3)
  __shared__ __attribute__((aligned(16))) float sAcc[NREG * MT * NT * 256];   // merge region(s), fragment order
  __shared__ float sBias[WNW * MT * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, kq = lane >> 4;
  // XCD-aware block mapping: workgroups are dealt round-robin to the 8 XCDs (private 4 MB L2 each).
  // All (co group, n group) blocks that reduce the SAME pixel chunk are placed on one XCD and run
  // back-to-back there, so the chunk's dY / X rows are fetched from the fabric once, not once per
  // block (measured on the 3x3 64->64 weight gradient: 206 MB of fabric reads for 24 MB of data).
  const int G = a.gx * a.gy;
  const int xcd = bid & 7, slot = bid >> 3;
  const int grp = slot % G, bz = (slot / G) * 8 + xcd;
  if (bz >= a.sp) return;                                   // padding blocks (sp rounded up to 8)
  const int bx = grp % a.gx, by = grp / a.gx;
  const int m0 = bx * (16 * MT), n0 = TAPB ? by * 16 : by * (16 * NT);     // TAPB: n0 = first input channel of the block
  const int Np = TAP ? a.Cin * 9 : a.Cin;
  const int pbeg = bz * a.chunk, pend = min(a.Mpix, pbeg + a.chunk);
  const bool ln = a.tf_rowstat != nullptr;

  // (tap-blocked: every tile of a lane is the SAME input channel n0 + c at another tap - one validity flag, one affine and
  // one channel offset per lane instead of nine: the nine copies were what spilled the <4, 9, ...> variants, and a scratch
  // reload retires every outstanding load of the pixel loop first)
  constexpr int NB = TAPB ? 1 : NT;
  int aoff[MT], boff[NB], tdy[NT], tdx[NT];
  bool aval[MT], bval[NB];
  float ca[MT], cb[MT], cc[MT], sc[NB], sh[NB], bsum[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int co = m0 + 16 * i + c;
    aval[i] = co < a.Cout;
    aoff[i] = a.doff + (aval[i] ? co : 0);
    ca[i] = 1.f; cb[i] = 0.f; cc[i] = 0.f; bsum[i] = 0.f;
    if (BNB && !COEF_LDS) { const int cs = aval[i] ? co : 0; ca[i] = a.cA[cs]; cb[i] = a.cB[cs]; cc[i] = a.cC[cs]; }
  }
  if (COEF_LDS) {
    for (int e = tid; e < MT * 16; e += 64 * WNW) {
      const int cs = m0 + e < a.Cout ? m0 + e : 0;
      sCoef[e] = a.cA[cs]; sCoef[MT * 16 + e] = a.cB[cs]; sCoef[2 * MT * 16 + e] = a.cC[cs];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    tdy[j] = 0; tdx[j] = 0;
    if (TAPB) { tdy[j] = j / 3 - 1; tdx[j] = j - (j / 3) * 3 - 1; }              // (compile-time after unrolling)
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int np = TAPB ? n0 + c : n0 + 16 * j + c;
    bval[j] = TAPB ? np < a.Cin : np < Np;
    int ci = bval[j] ? np : 0;
    if (TAP && !TAPB) { const int tp = ci % 9; ci /= 9; tdy[j] = tp / 3 - 1; tdx[j] = tp - (tp / 3) * 3 - 1; }
    boff[j] = (TAP && !TAPB) ? (tdy[j] * a.W + tdx[j]) * a.ldX + ci : ci;        // ci*9+tap order: + offset of the shifted pixel
    sc[j] = 1.f; sh[j] = 0.f;
    if (a.tf_scale != nullptr) { sc[j] = a.tf_scale[ci]; sh[j] = a.tf_shift[ci]; }
  }
  const int xrow = TAPB ? a.W * a.ldX : 0;                                       // tap-blocked: one image row of X, in floats
  hrf_f4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  WG_T(1);
  // wave w owns k-steps w, w+WNW, ... of the block's chunk: neighbouring waves read neighbouring rows
#pragma unroll 1
  for (int p0 = pbeg + 4 * wave; p0 < pend; p0 += 4 * WNW * WU) {
    float ar[WU][MT], yr[WU][MT], br[WU][NT], rm[WU], rr[WU];
    bool pv[WU];
    unsigned bmask[WU];
    // TAP: (image, row, column) of this lane's first pixel, advanced by 4*WNW pixels per k-step
    int bi = 0, yo = 0, xo = 0;
    if (TAP) {
      const int pf = min(p0 + kq, a.Mpix - 1), HoWo = a.Ho * a.Wo;
      bi = pf / HoWo;
      const int rem = pf - bi * HoWo;
      yo = rem / a.Wo; xo = rem - yo * a.Wo;
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int pix = p0 + 4 * WNW * u + kq;
      pv[u] = pix < pend;
      const int pc = pv[u] ? pix : pbeg;
      const unsigned arow = (unsigned)pc * (unsigned)a.ldD;          // 32-bit offsets: tensors are < 2^31 elements
      const unsigned brow = (unsigned)pc * (unsigned)a.ldX;
      const int yb = yo * a.stride, xb = xo * a.stride;
      const int tbase = ((bi * a.H + yb) * a.W + xb) * a.ldX;        // TAP: centre input pixel of this output pixel
      bmask[u] = 0u;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ar[u][i] = a.dy[arow + (unsigned)aoff[i]];
        yr[u][i] = BNB ? a.yraw[arow + (unsigned)aoff[i]] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (TAPB) {
          const bool inb = pv[u] && bval[0] && (unsigned)(yb + tdy[j]) < (unsigned)a.H && (unsigned)(xb + tdx[j]) < (unsigned)a.W;
          bmask[u] |= inb ? (1u << j) : 0u;
          br[u][j] = a.x[inb ? (unsigned)(tbase + boff[0] + tdy[j] * xrow + tdx[j] * a.ldX) : 0u];
        } else if (TAP) {
          const bool inb = pv[u] && bval[j] && (unsigned)(yb + tdy[j]) < (unsigned)a.H && (unsigned)(xb + tdx[j]) < (unsigned)a.W;
          bmask[u] |= inb ? (1u << j) : 0u;
          br[u][j] = a.x[inb ? (unsigned)(tbase + boff[j]) : 0u];
        } else {
          br[u][j] = a.x[brow + (unsigned)boff[j]];
        }
      }
      if (TAP) {                               // next k-step of this lane: 4*WNW pixels further
        xo += 4 * WNW;
        while (xo >= a.Wo) { xo -= a.Wo; if (++yo == a.Ho) { yo = 0; ++bi; } }
      }
      rm[u] = 0.f; rr[u] = 1.f;
      if (ln) { rm[u] = a.tf_rowstat[2 * pc]; rr[u] = a.tf_rowstat[2 * pc + 1]; }
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float v = ar[u][i];
        if (COEF_LDS) v = fmaf(sCoef[16 * i + c], v, fmaf(sCoef[MT * 16 + 16 * i + c], yr[u][i], sCoef[2 * MT * 16 + 16 * i + c]));
        else if (BNB) v = fmaf(ca[i], v, fmaf(cb[i], yr[u][i], cc[i]));
        av[i] = (pv[u] && aval[i]) ? v : 0.f;
        bsum[i] += av[i];
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float w = fmaf((br[u][j] - rm[u]) * rr[u], sc[TAPB ? 0 : j], sh[TAPB ? 0 : j]);    // (mean, rstd, sc, sh) = (0, 1, 1, 0) when unused
        bv[j] = ACT == 1 ? fmaxf(w, 0.f) : (ACT == 2 ? hrf_gelu(w) : w);
        if (TAP) bv[j] = (bmask[u] >> j) & 1u ? bv[j] : 0.f;      // zero padding applies AFTER the activation
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = hrf_mfma16(av[i], bv[j], acc[i][j]);
    }
  }

  // merge the 8 waves: region = wave >> 2, four rounds (waves r and r+4 work in parallel).  The tile lives in LDS in
  // FRAGMENT order (one 16-byte word per lane and 16x16 tile: conflict-free b128 accesses) and every round loads all
  // of its words BEFORE it stores any: a ld/add/st per element is a chain of dependent LDS round trips (the compiler
  // must assume that a store aliases the next load) - that chain, not the reduction, was the duration of this kernel
  // (~20 us for every problem size; tools/bench_wgrad.py).
  WG_T(2);
  constexpr int WPR = WNW / NREG;                            // waves per merge region
  hrf_f4* S = reinterpret_cast<hrf_f4*>(sAcc) + (wave / WPR) * (MT * NT * 64);
  if (wave % WPR == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) S[(i * NT + j) * 64 + lane] = acc[i][j];
  }
  __syncthreads();
#pragma unroll 1
  for (int round = 1; round < WPR; ++round) {
    if (wave % WPR == round) {
      // (in batches of at most MB rows of tiles: the accumulators are live beside `old` - 2 x 144 registers at MT x NT = 36)
      constexpr int MB = MT * NT > 25 ? 2 : MT;
#pragma unroll
      for (int i0 = 0; i0 < MT; i0 += MB) {
        hrf_f4 old[MB][NT];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) if (i0 + i < MT) old[i][j] = S[((i0 + i) * NT + j) * 64 + lane];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (i0 + i >= MT) continue;
            hrf_f4 v = old[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += acc[i0 + i][j][r];
            S[((i0 + i) * NT + j) * 64 + lane] = v;
          }
      }
    }
    __syncthreads();
  }
  WG_T(3);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float bsm = bsum[i];
    bsm += __shfl_xor(bsm, 16); bsm += __shfl_xor(bsm, 32);
    if (lane < 16) sBias[wave * (MT * 16) + 16 * i + lane] = bsm;
  }
  if (TAPB) {
    // transposing pass: per output channel the block owns 144 contiguous floats (16 ci x 9 taps, OIHW)
    for (int e = tid; e < MT * 16 * 144; e += 64 * WNW) {
      const int row = e / 144, o = e - row * 144;
      const int cil = o / 9, tap = o - cil * 9;
      const int ti = row >> 4, rr = row & 15;
      const int w = ((ti * NT + tap) * 64 + (rr >> 2) * 16 + cil) * 4 + (rr & 3);
      float v = sAcc[w];
      if (NREG == 2) v += sAcc[MT * NT * 256 + w];
      if (m0 + row < a.Cout && n0 + cil < a.Cin) hrf_atomic_add(&a.dw[(long)(m0 + row) * Np + (n0 + cil) * 9 + tap], v);
    }
  } else {
    const hrf_f4* S0 = reinterpret_cast<const hrf_f4*>(sAcc);
    for (int e = tid; e < MT * NT * 64; e += 64 * WNW) {
      const int tile = e >> 6, l = e & 63;
      const int ti = tile / NT, tj = tile - ti * NT;
      hrf_f4 v0 = S0[e];
      if (NREG == 2) { const hrf_f4 v1 = S0[MT * NT * 64 + e]; v0[0] += v1[0]; v0[1] += v1[1]; v0[2] += v1[2]; v0[3] += v1[3]; }
      const int col = n0 + 16 * tj + (l & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * ti + 4 * (l >> 4) + r;
        if (row < a.Cout && col < Np) hrf_atomic_add(&a.dw[(long)row * Np + col], v0[r]);
      }
    }
  }
  WG_T(4);
  __syncthreads();
  WG_T(5);
  if (a.dbias != nullptr && by == 0 && tid < MT * 16 && m0 + tid < a.Cout) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WNW; ++w) t += sBias[w * (MT * 16) + tid];
    hrf_atomic_add(&a.dbias[m0 + tid], t);
  }
}

// y[row][c] += part[0][row][c] + part[1][row][c] + ... in a fixed order (dense [M][C] output of a split-over-K forward), and the
// BatchNorm moments (sum, sum of squares) of the result: a thread keeps ONE channel (C <= 256: 256 / C rows per pass), its
// partial sums meet in LDS in a fixed order, one fp64 atomic per channel and block - like every other producer of moments (the
// generic hrf_gn_moments sums through LDS float atomics: the eager and the captured step then differed by 1.5e-5 in the gradients)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(float* y, const float* part, int M, int C, int nparts, double* stats) {
  HRF_DYN_SMEM(float, sacc);                               // [R][2*C]
  const int R = 256 / C, r = threadIdx.x / C, c = threadIdx.x - r * C;
  const size_t n = (size_t)M * C;
  float s1 = 0.f, s2 = 0.f;
  if (r < R) {
    for (int row = blockIdx.x * R + r; row < M; row += gridDim.x * R) {
      const size_t i = (size_t)row * C + c;
      float v = y[i];
      for (int k = 0; k < nparts; ++k) v += part[(size_t)k * n + i];
      y[i] = v;
      s1 += v; s2 = fmaf(v, v, s2);
    }
    sacc[(size_t)r * 2 * C + c] = s1; sacc[(size_t)r * 2 * C + C + c] = s2;
  }
  if (stats == nullptr) return;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    double t = 0.0;
    for (int rr = 0; rr < R; ++rr) t += (double)sacc[(size_t)rr * 2 * C + i];
    hrf_atomic_add(&stats[(size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * C + i], t);
  }
}

inline int pick_nt(int C) {
  const int T = (C + 15) / 16;
  int best = 2, cost = 1 << 30;
  for (int nt = 2; nt <= 4; ++nt) {
    const int c = ((T + nt - 1) / nt) * nt;
    if (c <= cost) { cost = c; best = nt; }
  }
  return best;
}

}  // namespace

#define HRF_CF_LAUNCH(NT_, KH_, TF_) \
  HRF_LAUNCH_G((conv_fwd_kernel<NT_, KH_, TF_>), dim3(hrf_cdiv(a.M, BM), hrf_cdiv(Cout, NT_ * 16) * a.ksplit), dim3(256), 0, stream, a)
#define HRF_CF_NT(KH_, TF_)                          \
  switch (nt) {                                      \
    case 2: HRF_CF_LAUNCH(2, KH_, TF_); break;       \
    case 3: HRF_CF_LAUNCH(3, KH_, TF_); break;       \
    default: HRF_CF_LAUNCH(4, KH_, TF_); break;      \
  }
#define HRF_CF_TF(KH_)                                            \
  switch (tf_mode) {                                              \
    case HRF_TF_NONE: HRF_CF_NT(KH_, HRF_TF_NONE) break;          \
    case HRF_TF_AFFINE: HRF_CF_NT(KH_, HRF_TF_AFFINE) break;      \
    case HRF_TF_AFFINE_RELU: HRF_CF_NT(KH_, HRF_TF_AFFINE_RELU) break; \
    default: HRF_CF_NT(KH_, HRF_TF_AFFINE_GELU) break;            \
  }

// the policy of the split over K (shared by hrf_conv_fwd_split_scratch and the launch): slices, or 1
static int conv_fwd_ksplit(int B, int H, int W, int Cin, int KH, int stride, int Cout, int ldY, int yoff, bool dense_nhwc) {
  static const bool off = std::getenv("HRF_NO_KSPLIT") != nullptr;               // (A/B switch, like hrf_debug_knob(9, 1))
  if (KH != 3 || g_knob[9] == 1 || off || ldY != Cout || yoff != 0 || Cout > 256) return 1;
  if (stride == 1 && Cin >= 32 && dense_nhwc && g_knob[6] == 0) return 1;          // (the 3x3 halo engine takes these)
  const int pad = 1, Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KH) / stride + 1;
  const long M = (long)B * Ho * Wo;
  const int nt = pick_nt(Cout);
  if (M <= 0 || KH * KH * Cin < 1024 || hrf_cdiv(M, BM) * hrf_cdiv(Cout, nt * 16) > 128) return 1;
  return 4;
}

static int conv_fwd_impl(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                         const float* w, const float* bias, int KH, int stride, int Cout,
                         float* y, int ldY, int yoff, const float* res, const float* res2, int ldR,
                         int tf_mode, const float* tf_scale, const float* tf_shift,
                         const float* tf_rowstat, double* stats, const hrf_bn_fin_t* tf_fin, float* ln_rowstat,
                         float ln_eps, float* scratch, void* stream) {
  HRF_GROUP_CALL();
  if ((KH != 1 && KH != 3) || (stride != 1 && stride != 2)) return HRF_ERR_ARG;
  if (tf_mode < 0 || tf_mode > 4) return HRF_ERR_ARG;
  if (ln_rowstat != nullptr && (ldY != Cout || yoff != 0)) return HRF_ERR_ARG;   // row statistics of a full output row
  if (tf_mode == HRF_TF_LN && (KH != 1 || stride != 1)) return HRF_ERR_ARG;
  if (tf_fin != nullptr && (tf_mode < HRF_TF_AFFINE || tf_mode > HRF_TF_AFFINE_GELU || tf_fin->C != Cin || Cin > HRF_FIN_MAXC ||
                            tf_fin->stats == nullptr)) return HRF_ERR_ARG;
  ConvFwdArgs a;
  const int pad = KH / 2;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.ldY = ldY; a.yoff = yoff; a.res = res; a.res2 = res2; a.ldR = ldR;
  a.tf_scale = tf_scale; a.tf_shift = tf_shift; a.tf_rowstat = tf_rowstat;
  a.fin = hrf_bn_fin_t{};
  if (tf_fin != nullptr) a.fin = *tf_fin;
  a.stats = stats; a.B = B; a.H = H; a.W = W; a.Cin = Cin;
  a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KH) / stride + 1;
  a.Cout = Cout; a.stride = stride; a.pad = pad; a.sB = sB; a.sY = sY; a.sX = sX; a.sC = sC;
  a.M = B * a.Ho * a.Wo; a.K = KH * KH * Cin; a.ksplit = 1; a.part = nullptr;
  if (a.M <= 0) return HRF_OK;
  if (KH == 1 && stride == 1 && sC == 1 && sY == W * sX && sB == H * sY && g_knob[4] == 0) {
    // channel-contiguous rows: LDS-free row-GEMM kernel (lin_engine.hip)
    LinFwdArgs l;
    l.x = x; l.ldX = sX; l.w = w; l.bias = bias; l.y = y; l.ldY = ldY; l.yoff = yoff;
    l.res = res; l.res2 = res2; l.ldR = ldR; l.tf_mode = tf_mode; l.tf_scale = tf_scale; l.tf_shift = tf_shift;
    l.tf_rowstat = tf_rowstat; l.stats = stats; l.M = a.M; l.K = Cin; l.N = Cout; l.ln_out = ln_rowstat; l.ln_eps = ln_eps;
    l.fin = a.fin;
    const int rc2 = hrf_lin2_fwd_launch(l, stream);          // wide problems: LDS-tiled engine (lin2_engine.hip)
    if (rc2 == HRF_OK && l.ln_out != nullptr && !hrf_lin2_fwd_emits_ln(l)) return hrf_ln_stats(y, a.M, Cout, ln_eps, ln_rowstat, stream);
    if (rc2 >= 0) return rc2;
    const int rc = hrf_lin_fwd_launch(l, stream);
    if (rc == HRF_OK && l.ln_out != nullptr && !hrf_lin_fwd_emits_ln(l)) return hrf_ln_stats(y, a.M, Cout, ln_eps, ln_rowstat, stream);
    if (rc >= 0) return rc;
  }
  if (KH == 3 && stride == 1 && Cin >= 32 && sC == 1 && sY == W * sX && sB == H * sY && g_knob[6] == 0) {
    Conv3Args c{};
    c.in = x; c.ldIn = sX; c.t0 = tf_scale; c.t1 = tf_shift; c.tf_mode = tf_mode; c.w = w; c.wCin = Cin; c.bias = bias;
    c.out = y; c.ldOut = ldY; c.ooff = yoff; c.res = res; c.res2 = res2; c.ldR = ldR; c.stats = stats;
    c.B = B; c.H = H; c.W = W; c.Cin = Cin; c.Cout = Cout;
    c.fin = a.fin;
    const int rc3 = hrf_conv3_fwd_launch(c, stream);
    if (rc3 == HRF_OK && ln_rowstat != nullptr) return hrf_ln_stats(y, a.M, Cout, ln_eps, ln_rowstat, stream);
    return rc3;
  }
  const int nt = pick_nt(Cout);
  // deep contraction, few row blocks (the 256 -> 36 stride-2 transition: K = 2 304, 120 blocks): split over K, the moments of the
  // summed output by a pass of their own (hrf_debug_knob(9, 1): off)
  // (HRFuser-T: 120 row blocks, 115.6 -> ~60 us with the two extra launches; STF's 234 blocks in two slices: 118 -> 111 us, not taken.
  // Not while a merged multi-problem launch is being collected: that launch is issued LATER, at hrf_group_end - the passes behind
  // it here would run first)
  const int ks = (scratch != nullptr && !hrf_grp_collecting())
                     ? conv_fwd_ksplit(B, H, W, Cin, KH, stride, Cout, ldY, yoff, sC == 1 && sY == W * sX && sB == H * sY) : 1;
  const bool ksp = ks > 1;
  if (ksp) { a.ksplit = ks; a.part = scratch; a.stats = nullptr; }
  if (KH == 1) {
    if (tf_mode == HRF_TF_LN) { HRF_CF_NT(1, HRF_TF_LN) } else { HRF_CF_TF(1) }
  } else {
    HRF_CF_TF(3)
  }
  if (ksp) {
    if (hrf_check_launch() != HRF_OK) return HRF_ERR_LAUNCH;
    const int R = 256 / Cout;
    HRF_LAUNCH(splitk_reduce_kernel, dim3((unsigned)std::min<long>(1024, hrf_cdiv(a.M, R))), dim3(256), (unsigned)((size_t)R * 2 * Cout * sizeof(float)),
               stream, y, (const float*)scratch, a.M, Cout, ks - 1, stats);
  }
  if (ln_rowstat != nullptr) { if (hrf_check_launch() != HRF_OK) return HRF_ERR_LAUNCH; return hrf_ln_stats(y, a.M, Cout, ln_eps, ln_rowstat, stream); }
  return hrf_check_launch();
}

extern "C" int hrf_conv_fwd(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                            const float* w, const float* bias, int KH, int stride, int Cout,
                            float* y, int ldY, int yoff, const float* res, const float* res2, int ldR,
                            int tf_mode, const float* tf_scale, const float* tf_shift,
                            const float* tf_rowstat, double* stats, const hrf_bn_fin_t* tf_fin, float* ln_rowstat,
                            float ln_eps, void* stream) {
  return conv_fwd_impl(x, sB, sY, sX, sC, B, H, W, Cin, w, bias, KH, stride, Cout, y, ldY, yoff, res, res2, ldR, tf_mode, tf_scale,
                       tf_shift, tf_rowstat, stats, tf_fin, ln_rowstat, ln_eps, nullptr, stream);
}

extern "C" long hrf_conv_fwd_split_scratch(int sB, int sY, int sX, int sC, int B, int H, int W, int Cin, int KH, int stride, int Cout,
                                           int ldY, int yoff) {
  const int ks = conv_fwd_ksplit(B, H, W, Cin, KH, stride, Cout, ldY, yoff, sC == 1 && sY == W * sX && sB == H * sY);
  if (ks <= 1) return 0;
  const int Ho = (H + 2 - KH) / stride + 1, Wo = (W + 2 - KH) / stride + 1;
  return (long)(ks - 1) * B * Ho * Wo * Cout;
}

extern "C" int hrf_conv_fwd_split(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                                  const float* w, const float* bias, int KH, int stride, int Cout,
                                  float* y, int ldY, int yoff, const float* res, const float* res2, int ldR,
                                  int tf_mode, const float* tf_scale, const float* tf_shift,
                                  const float* tf_rowstat, double* stats, const hrf_bn_fin_t* tf_fin, float* ln_rowstat,
                                  float ln_eps, float* scratch, void* stream) {
  return conv_fwd_impl(x, sB, sY, sX, sC, B, H, W, Cin, w, bias, KH, stride, Cout, y, ldY, yoff, res, res2, ldR, tf_mode, tf_scale,
                       tf_shift, tf_rowstat, stats, tf_fin, ln_rowstat, ln_eps, scratch, stream);
}

#define HRF_BD_LAUNCH(NT_, KH_, BNB_) \
  HRF_LAUNCH_G((conv_bwd_data_kernel<NT_, KH_, BNB_>), dim3(hrf_cdiv(a.M, BM), hrf_cdiv(Cin, NT_ * 16)), dim3(256), 0, stream, a)
#define HRF_BD_NT(KH_, BNB_)                         \
  switch (nt) {                                      \
    case 2: HRF_BD_LAUNCH(2, KH_, BNB_); break;      \
    case 3: HRF_BD_LAUNCH(3, KH_, BNB_); break;      \
    default: HRF_BD_LAUNCH(4, KH_, BNB_); break;     \
  }

extern "C" int hrf_conv_bwd_data(const float* dy, int ldD, int doff, const float* yraw,
                                 const float* cA, const float* cB, const float* cC, const hrf_bn_bfin_t* bfin,
                                 const float* w, int KH, int stride, int Cout,
                                 int B, int H, int W, int Cin,
                                 float* dx, int sB, int sY, int sX, int sC, int accumulate,
                                 int epi, const float* xraw, int ldXr, const float* tf_scale,
                                 const float* tf_shift, int act, double* stats, void* stream) {
  HRF_GROUP_CALL();
  if ((KH != 1 && KH != 3) || (stride != 1 && stride != 2)) return HRF_ERR_ARG;
  if (bfin != nullptr && (cA == nullptr || bfin->C != Cout || Cout > HRF_FIN_MAXC || bfin->gstats == nullptr)) return HRF_ERR_ARG;
  ConvBwdDataArgs a;
  const int pad = KH / 2;
  a.bfin = hrf_bn_bfin_t{};
  if (bfin != nullptr) a.bfin = *bfin;
  a.dy = dy; a.ldD = ldD; a.doff = doff; a.yraw = yraw; a.cA = cA; a.cB = cB; a.cC = cC; a.w = w;
  a.dx = dx; a.sB = sB; a.sY = sY; a.sX = sX; a.sC = sC; a.accumulate = accumulate; a.epi = epi;
  a.xraw = xraw; a.ldXr = ldXr; a.tf_scale = tf_scale; a.tf_shift = tf_shift; a.act = act; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin;
  a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KH) / stride + 1;
  a.Cout = Cout; a.stride = stride; a.pad = pad;
  a.M = B * H * W; a.K = KH * KH * Cout;
  if (a.M <= 0) return HRF_OK;
  if (KH == 1 && stride == 1 && sC == 1 && sY == W * sX && sB == H * sY && g_knob[4] == 0) {
    LinBwdDataArgs l;
    l.dy = dy; l.ldD = ldD; l.doff = doff; l.yraw = yraw; l.cA = cA; l.cB = cB; l.cC = cC; l.w = w;
    l.dx = dx; l.ldDx = sX; l.accumulate = accumulate; l.epi = epi; l.xraw = xraw; l.ldXr = ldXr;
    l.tf_scale = tf_scale; l.tf_shift = tf_shift; l.act = act; l.stats = stats; l.bfin = a.bfin;
    l.M = a.M; l.K = Cout; l.N = Cin;
    const int rc2 = hrf_lin2_bwd_data_launch(l, stream);
    if (rc2 >= 0) return rc2;
    const int rc = hrf_lin_bwd_data_launch(l, stream);
    if (rc >= 0) return rc;
  }
  if (KH == 3 && stride == 1 && Cout >= (g_knob[5] > 0 ? g_knob[5] : 16) && sC == 1 && sY == W * sX && sB == H * sY && g_knob[6] == 0) {
    Conv3Args c{};
    c.in = dy + doff; c.ldIn = ldD; c.in2 = cA != nullptr ? yraw + doff : nullptr; c.t0 = cA; c.t1 = cB; c.t2 = cC;
    c.w = w; c.wCin = Cin; c.out = dx; c.ldOut = sX; c.accumulate = accumulate; c.epi = epi; c.xraw = xraw; c.ldXr = ldXr;
    c.esc = tf_scale; c.esh = tf_shift; c.act = act; c.stats = stats; c.bfin = a.bfin;
    c.B = B; c.H = H; c.W = W; c.Cin = Cout; c.Cout = Cin;
    return hrf_conv3_bwd_data_launch(c, stream);
  }
  if (KH == 3 && stride == 2 && Cout >= (g_knob[5] > 0 ? g_knob[5] : 16) && sC == 1 && sY == W * sX && sB == H * sY && g_knob[6] == 0) {
    Conv3Args c{};
    c.in = dy + doff; c.ldIn = ldD; c.in2 = cA != nullptr ? yraw + doff : nullptr; c.t0 = cA; c.t1 = cB; c.t2 = cC;
    c.w = w; c.wCin = Cin; c.out = dx; c.ldOut = sX; c.accumulate = accumulate; c.epi = epi; c.xraw = xraw; c.ldXr = ldXr;
    c.esc = tf_scale; c.esh = tf_shift; c.act = act; c.stats = stats; c.bfin = a.bfin;
    c.B = B; c.H = H; c.W = W; c.Cin = Cout; c.Cout = Cin; c.Hs = a.Ho; c.Ws = a.Wo;
    return hrf_conv3s2_bwd_data_launch(c, stream);
  }
  const int nt = pick_nt(Cin);
  if (KH == 1) {
    if (cA != nullptr) { HRF_BD_NT(1, true) } else { HRF_BD_NT(1, false) }
  } else {
    if (cA != nullptr) { HRF_BD_NT(3, true) } else { HRF_BD_NT(3, false) }
  }
  return hrf_check_launch();
}

// ---- the packed-weight front-end engine (conv3x_engine.hip)
extern "C" int hrf_conv3x_supported(int Cin, int Cout, int KH, int stride, int dir) {
  if (KH != 3 || Cin <= 0 || Cout <= 0) return 0;
  if (dir == 0) return (stride == 1 || stride == 2) && Cout > 32 ? 1 : 0;
  if (stride == 2) return Cin > 32 && Cout <= 64 ? 1 : 0;       // (the parity-class walk keeps ONE halo slab: <= 64 channels of dY)
  return stride == 1 && Cin > 32 ? 1 : 0;
}

extern "C" int hrf_conv_fwd_packed(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                                   const float* w, const float* bias, int KH, int stride, int Cout,
                                   float* y, int ldY, int yoff, const float* res, const float* res2, int ldR,
                                   int tf_mode, const float* tf_scale, const float* tf_shift,
                                   const float* tf_rowstat, double* stats, const hrf_bn_fin_t* tf_fin, float* ln_rowstat,
                                   float ln_eps, const float* wp, void* stream) {
  HRF_GROUP_CALL();
  (void)w; (void)tf_rowstat;
  if (wp == nullptr || !hrf_conv3x_supported(Cin, Cout, KH, stride, 0)) return HRF_ERR_ARG;
  if (!(sC == 1 && sY == W * sX && sB == H * sY) || tf_mode < 0 || tf_mode > HRF_TF_AFFINE_GELU) return HRF_ERR_ARG;
  if (ln_rowstat != nullptr && (ldY != Cout || yoff != 0)) return HRF_ERR_ARG;
  if (tf_fin != nullptr && (tf_mode < HRF_TF_AFFINE || tf_fin->C != Cin || Cin > 256 || tf_fin->stats == nullptr)) return HRF_ERR_ARG;
  C3xArgs c{};
  c.in = x; c.ldIn = sX; c.t0 = tf_scale; c.t1 = tf_shift; c.tf_mode = tf_mode; c.wp = wp;
  c.Np = (Cout + 63) & ~63; c.Kp = (Cin + 31) & ~31; c.bias = bias;
  c.out = y; c.ldOut = ldY; c.ooff = yoff; c.res = res; c.res2 = res2; c.ldR = ldR; c.stats = stats;
  c.B = B; c.H = H; c.W = W; c.Cin = Cin; c.Cout = Cout;
  if (tf_fin != nullptr) c.fin = *tf_fin;
  int rc;
  if (stride == 2) {                                             // the source grid is the input, tiles walk the output grid
    c.Hs = H; c.Ws = W; c.H = (H + 2 - KH) / 2 + 1; c.W = (W + 2 - KH) / 2 + 1;
    rc = hrf_conv3xs2_fwd_launch(c, stream);
  } else {
    rc = hrf_conv3x_fwd_launch(c, stream);
  }
  if (rc == HRF_OK && ln_rowstat != nullptr) return hrf_ln_stats(y, (long)B * c.H * c.W, Cout, ln_eps, ln_rowstat, stream);
  return rc;
}

extern "C" int hrf_conv_bwd_data_packed(const float* dy, int ldD, int doff, const float* yraw,
                                        const float* cA, const float* cB, const float* cC, const hrf_bn_bfin_t* bfin,
                                        const float* w, int KH, int stride, int Cout,
                                        int B, int H, int W, int Cin,
                                        float* dx, int sB, int sY, int sX, int sC, int accumulate,
                                        int epi, const float* xraw, int ldXr, const float* tf_scale,
                                        const float* tf_shift, int act, double* stats, const float* wp, void* stream) {
  HRF_GROUP_CALL();
  (void)w;
  if (wp == nullptr || !hrf_conv3x_supported(Cin, Cout, KH, stride, 1)) return HRF_ERR_ARG;
  if (!(sC == 1 && sY == W * sX && sB == H * sY)) return HRF_ERR_ARG;
  if (bfin != nullptr && (cA == nullptr || bfin->C != Cout || Cout > 256 || bfin->gstats == nullptr)) return HRF_ERR_ARG;
  C3xArgs c{};
  c.in = dy + doff; c.ldIn = ldD; c.in2 = cA != nullptr ? yraw + doff : nullptr; c.t0 = cA; c.t1 = cB; c.t2 = cC;
  c.wp = wp; c.Np = (Cin + 63) & ~63; c.Kp = (Cout + 31) & ~31;
  c.out = dx; c.ldOut = sX; c.accumulate = accumulate; c.epi = epi; c.xraw = xraw; c.ldXr = ldXr;
  c.esc = tf_scale; c.esh = tf_shift; c.act = act; c.stats = stats;
  if (bfin != nullptr) c.bfin = *bfin;
  c.B = B; c.H = H; c.W = W; c.Cin = Cout; c.Cout = Cin;
  if (stride == 1) return hrf_conv3x_bwd_data_launch(c, stream);
  c.Hs = (H + 2 - KH) / stride + 1; c.Ws = (W + 2 - KH) / stride + 1;
  return hrf_conv3xs2_bwd_data_launch(c, stream);
}

extern "C" __attribute__((visibility("hidden"))) int hrf_pw_knob(int key, int value);
extern "C" __attribute__((visibility("hidden"))) int hrf_conv3w_knob(int key, int value);
extern "C" __attribute__((visibility("hidden"))) int hrf_lin2_knob(int key, int value);
extern "C" __attribute__((visibility("hidden"))) int hrf_w3x_knob(int key, int value);
extern "C" __attribute__((visibility("hidden"))) int hrf_dw_knob(int key, int value);
extern "C" int hrf_debug_knob(int key, int value) {
  if (key >= 40 && key < 44) return hrf_dw_knob(key - 40, value);      // dwconv.hip: 40 = float4-lane kernels (1 off, 2 / 3 tile height), 41 = grid threshold
  if (key >= 32 && key < 36) return hrf_w3x_knob(key - 32, value);     // wgrad3x_engine.hip: 32 = blocks per problem, 33 = smallest problem (output pixels)
  if (key >= 28 && key < 32) return hrf_lin2_knob(key - 28, value);    // lin2_engine.hip: 28 = 1 force / 2 disable the LDS-tiled row GEMM
  if (key >= 16 && key < 20) return hrf_pw_knob(key - 16, value);      // pointwise.hip tuning aids
  if (key >= 24 && key < 28) return hrf_conv3w_knob(key - 24, value);  // conv3w_engine.hip tuning aids
  if (key < 0 || key >= 12) return HRF_ERR_ARG;      // (8: the LDS-tiled weight gradient, 1 = off, 2 = every 1x1 problem)
  g_knob[key] = value;
  return HRF_OK;
}

struct WgPending { int key; int blocks; WgradDenseArgs d; };
static bool g_wg_collect = false;
static std::vector<WgPending> g_wg_pending;
static int wgrad_dense_launch(int key, const WgradGroup& d, int total_blocks, void* stream);

#define HRF_BW_LAUNCH(D1_, BNB_, TFA_) \
  HRF_LAUNCH((conv_bwd_wgt_kernel<D1_, BNB_, TFA_>), grid, dim3(256), 0, stream, a)
#define HRF_BW_TFA(D1_, BNB_)                                       \
  if (tfa == 2) { HRF_BW_LAUNCH(D1_, BNB_, 2); }                    \
  else if (tfa == 3) { HRF_BW_LAUNCH(D1_, BNB_, 3); }               \
  else { HRF_BW_LAUNCH(D1_, BNB_, 0); }

extern "C" int hrf_conv_bwd_weight(const float* dy, int ldD, int doff, const float* yraw,
                                   const float* cA, const float* cB, const float* cC,
                                   const float* x, int sB, int sY, int sX, int sC,
                                   int B, int H, int W, int Cin, int KH, int stride, int Cout,
                                   int tf_mode, const float* tf_scale, const float* tf_shift,
                                   const float* tf_rowstat, float* dw, float* dbias, void* stream) {
  if ((KH != 1 && KH != 3) || (stride != 1 && stride != 2)) return HRF_ERR_ARG;
  ConvBwdWgtArgs a;
  const int pad = KH / 2;
  a.dy = dy; a.ldD = ldD; a.doff = doff; a.yraw = yraw; a.cA = cA; a.cB = cB; a.cC = cC;
  a.x = x; a.sB = sB; a.sY = sY; a.sX = sX; a.sC = sC;
  a.tf_mode = tf_mode; a.tf_scale = tf_scale; a.tf_shift = tf_shift; a.tf_rowstat = tf_rowstat;
  a.dw = dw; a.dbias = dbias; a.B = B; a.H = H; a.W = W; a.Cin = Cin;
  a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KH) / stride + 1;
  a.Cout = Cout; a.stride = stride; a.pad = pad; a.KH = KH;
  a.Mpix = B * a.Ho * a.Wo; a.Np = KH * KH * Cin;
  if (a.Mpix <= 0) return HRF_OK;
  const int gx = hrf_cdiv(Cout, 64), gy = hrf_cdiv(a.Np, 64);
  // split the pixel reduction so that ~256-512 blocks exist, but never below 4 steps (256 pixels)
  // per block: the per-block epilogue (LDS reduce + one atomic per output) must stay amortised.
  int splits = hrf_cdiv(a.Mpix, 256);
  int cap = hrf_cdiv(512, gx * gy);
  if (cap > 64) cap = 64;                                  // atomic fan-in per output element
  if (g_knob[0] > 0) cap = g_knob[0];
  a.dbg_plain = g_knob[1];
  if (splits > cap) splits = cap;
  if (splits < 1) splits = 1;
  a.chunk = hrf_cdiv(hrf_cdiv(a.Mpix, splits), WK) * WK;
  splits = hrf_cdiv(a.Mpix, a.chunk);
  const bool dense1 = KH == 1 && stride == 1 && sC == 1 && sY == W * sX && sB == H * sY;
  const int tfa = tf_mode == HRF_TF_AFFINE_RELU ? 2 : (tf_mode == HRF_TF_AFFINE_GELU ? 3 : 0);
  const bool tap3 = KH == 3 && sC == 1 && sY == W * sX && sB == H * sY && tf_mode != HRF_TF_LN;
  if ((dense1 || tap3) && g_knob[2] == 0) {
    WgradDenseArgs d;
    d.dy = dy; d.ldD = ldD; d.doff = doff; d.yraw = yraw; d.cA = cA; d.cB = cB; d.cC = cC;
    d.x = x; d.ldX = sX; d.tf_scale = tf_mode != HRF_TF_NONE ? tf_scale : nullptr; d.tf_shift = tf_shift;
    d.tf_rowstat = tf_mode == HRF_TF_LN ? tf_rowstat : nullptr;
    d.dw = dw; d.dbias = dbias; d.Cout = Cout; d.Cin = Cin; d.Mpix = a.Mpix;
    d.H = H; d.W = W; d.Ho = a.Ho; d.Wo = a.Wo; d.stride = stride;
    const int act = tf_mode == HRF_TF_AFFINE_RELU ? 1 : (tf_mode == HRF_TF_AFFINE_GELU ? 2 : 0);
    if (dense1 && g_knob[1] == 0) {
      // wide 1x1 problems: the LDS-tiled kernel (wgrad_tiled.hip)
      int tkey = 0, tblocks = 0;
      if (hrf_wgrad_tiled_plan(d, cA != nullptr, act, g_wg_collect, g_knob[8], tkey, tblocks)) {
        if (g_wg_collect) { g_wg_pending.push_back(WgPending{tkey, tblocks, d}); return HRF_OK; }
        WgradGroup one;
        one.nprob = 1; one.bstart[0] = 0; one.bstart[1] = tblocks; one.p[0] = d;
        return wgrad_dense_launch(tkey, one, tblocks, stream);
      }
    }
    // 16x16 tiles per wave in each dimension: the count in {2..5} with the least padding (+ one unit per
    // extra group, which re-reads the other operand); the 3x3 path keeps {2, 4} (N' = 9*Cin is large)
    auto pick_tiles = [](int C, bool wide) {
      const int T = hrf_cdiv(C, 16);
      if (!wide) return T <= 2 ? 2 : 4;
      int best = 2, cost = 1 << 30;
      for (int t = 2; t <= 5; ++t) {
        const int g = hrf_cdiv(T, t), c = g * t + g;
        if (c <= cost) { cost = c; best = t; }
      }
      return best;
    };
    // 3x3 with >= 8 input channels: tap-blocked variant (one tile per tap, 16 consecutive input channels per block)
    const bool tapb = tap3 && Cin >= 64 && g_knob[1] != 2;      // (debug knob 1 = 2: the ci*9+tap variant)
    const int mt = tapb ? (Cout <= 32 ? 2 : (Cout <= 48 ? 3 : 4)) : pick_tiles(Cout, !tap3);
    const int nt = tapb ? 9 : pick_tiles(tap3 ? Cin * 9 : Cin, !tap3);
    d.gyc = tapb ? hrf_cdiv(Cin, 16) : hrf_cdiv(tap3 ? Cin * 9 : Cin, 16 * nt);
    int sp = hrf_cdiv(a.Mpix, 4 * WNW * WUMAX);
    // atomic fan-in per output element (128 x 25 ns = 3 us tail); the 3x3 path already has 9x the blocks
    const int gxy0 = hrf_cdiv(Cout, 16 * mt) * d.gyc;
    const int cap_tapb = gxy0 >= 64 ? 8 : (512 / gxy0 > 128 ? 128 : 512 / gxy0);       // ~512 blocks
    // The problems of a step are issued GROUPED (8 ... 16 per launch), so a problem does not have to fill the chip on its own: with
    // 16 pixel splits instead of 128 every block streams 8x the pixels per 8-wave merge + atomic pass (same-box A/B of the captured
    // steps, cap 128 / 64 / 32 / 16 / 8 / 4: HRFuser-T 12.66 / 12.63 / 12.56 / 12.49 / 12.46 / 13.6 ms, HRFuser-B 46.3 / 45.9 / 45.6 /
    // 45.5 / 46.1 ms)
    // (a problem launched on its own - outside hrf_wgrad_group_begin / _end - keeps the wide split: it has the chip to itself)
    const int cap_split = g_wg_collect ? 16 : (tap3 ? 32 : 128);
    const int cap2 = g_knob[3] > 0 ? g_knob[3] : (tapb ? (cap_tapb < cap_split ? cap_tapb : cap_split) : cap_split);
    if (sp > cap2) sp = cap2;
    if (sp < 1) sp = 1;
    const int gxy = hrf_cdiv(Cout, 16 * mt) * d.gyc;
    if (sp > 8) {                                            // whole pixel chunks per XCD (32 CUs): one round if possible
      int c = hrf_cdiv(sp, 8);
      if (gxy * c > 32 && gxy <= 32) c = 32 / gxy;
      sp = 8 * c;
    }
    d.chunk = hrf_cdiv(hrf_cdiv(a.Mpix, sp), 4 * WNW) * 4 * WNW;
    sp = hrf_cdiv(a.Mpix, d.chunk);
    d.gx = hrf_cdiv(Cout, 16 * mt); d.gy = d.gyc; d.sp = sp;
    const int nblocks = d.gx * d.gy * hrf_cdiv(sp, 8) * 8;
    const int key = ((((mt * 8 + (tapb ? 7 : nt)) * 2 + (cA != nullptr ? 1 : 0)) * 4 + act) * 2) + (tap3 ? 1 : 0);   // nt code 7 = tap-blocked (NT 9)
    if (g_wg_collect) {                                      // queued: launched by hrf_wgrad_group_end
      g_wg_pending.push_back(WgPending{key, nblocks, d});
      return HRF_OK;
    }
    WgradGroup one;
    one.nprob = 1; one.bstart[0] = 0; one.bstart[1] = nblocks; one.p[0] = d;
    return wgrad_dense_launch(key, one, nblocks, stream);
  }
  const dim3 grid(gx, gy, splits);
  if (dense1) {
    if (cA != nullptr) { HRF_BW_TFA(true, true) } else { HRF_BW_TFA(true, false) }
  } else {
    if (cA != nullptr) { HRF_BW_TFA(false, true) } else { HRF_BW_TFA(false, false) }
  }
  return hrf_check_launch();
}

// ---- the LDS-staged 3x3 weight gradient (wgrad3x_engine.hip): per-split slabs in caller-owned scratch + a fold launch
long hrf_wgrad3x_scratch(int B, int H, int W, int Cin, int KH, int stride, int Cout, int has_bias, int tf_mode, bool dense_nhwc);
int hrf_wgrad3x_launch(const float* dy, int ldD, const float* yraw, const float* cA, const float* cB, const float* cC,
                       const float* x, int ldX, int B, int H, int W, int Cin, int stride, int Cout,
                       int tf_mode, const float* tf_scale, const float* tf_shift, float* dw, float* scratch, void* stream);

extern "C" long hrf_conv_bwd_weight_scratch(int sB, int sY, int sX, int sC, int B, int H, int W, int Cin, int KH, int stride, int Cout,
                                            int tf_mode, int has_bias) {
  return hrf_wgrad3x_scratch(B, H, W, Cin, KH, stride, Cout, has_bias, tf_mode, sC == 1 && sY == W * sX && sB == H * sY);
}

extern "C" int hrf_conv_bwd_weight_s(const float* dy, int ldD, int doff, const float* yraw,
                                     const float* cA, const float* cB, const float* cC,
                                     const float* x, int sB, int sY, int sX, int sC,
                                     int B, int H, int W, int Cin, int KH, int stride, int Cout,
                                     int tf_mode, const float* tf_scale, const float* tf_shift,
                                     const float* tf_rowstat, float* dw, float* dbias, float* scratch, void* stream) {
  if (scratch == nullptr || hrf_conv_bwd_weight_scratch(sB, sY, sX, sC, B, H, W, Cin, KH, stride, Cout, tf_mode, dbias != nullptr) <= 0)
    return hrf_conv_bwd_weight(dy, ldD, doff, yraw, cA, cB, cC, x, sB, sY, sX, sC, B, H, W, Cin, KH, stride, Cout, tf_mode, tf_scale,
                               tf_shift, tf_rowstat, dw, dbias, stream);
  return hrf_wgrad3x_launch(dy + doff, ldD, cA != nullptr ? yraw + doff : nullptr, cA, cB, cC, x, sX, B, H, W, Cin, stride, Cout, tf_mode,
                            tf_scale, tf_shift, dw, scratch, stream);
}

// one launch of kernel variant `key` = (mt, nt, BatchNorm-backward, activation, 3x3) over a group of problems
static int wgrad_dense_launch(int key, const WgradGroup& d, int total_blocks, void* stream) {
    if (key >= 4096) return hrf_wgrad_tiled_launch(key, d, total_blocks, stream);   // wgrad_tiled.hip
    const bool tap3 = key & 1;
    const int act = (key >> 1) & 3;
    const bool bnb = (key >> 3) & 1;
    const int nt = (key >> 4) & 7, mt = key >> 7;
    const dim3 g2(total_blocks);
#define HRF_WD_LAUNCH(MT_, NT_, BNB_, ACT_, TAP_) \
    HRF_LAUNCH((wgrad_dense_kernel<MT_, NT_, BNB_, ACT_, TAP_>), g2, dim3(64 * WNW), 0, stream, d)
#define HRF_WD_ACT(MT_, NT_, BNB_, TAP_)                           \
    if (act == 1) { HRF_WD_LAUNCH(MT_, NT_, BNB_, 1, TAP_); }      \
    else if (act == 2) { HRF_WD_LAUNCH(MT_, NT_, BNB_, 2, TAP_); } \
    else { HRF_WD_LAUNCH(MT_, NT_, BNB_, 0, TAP_); }
#define HRF_WD_BNB(MT_, NT_, TAP_) \
    if (bnb) { HRF_WD_ACT(MT_, NT_, true, TAP_) } else { HRF_WD_ACT(MT_, NT_, false, TAP_) }
#define HRF_WD_NT(MT_)                                    \
    switch (nt) {                                         \
      case 2: HRF_WD_BNB(MT_, 2, 0) break;                \
      case 3: HRF_WD_BNB(MT_, 3, 0) break;                \
      case 4: HRF_WD_BNB(MT_, 4, 0) break;                \
      default: HRF_WD_BNB(MT_, 5, 0) break;               \
    }
    if (tap3 && nt == 7) {
      if (mt == 2) { HRF_WD_BNB(2, 9, 2) }
      else if (mt == 3) { HRF_WD_BNB(3, 9, 2) }
      else { HRF_WD_BNB(4, 9, 2) }
    } else if (tap3) {
      if (mt == 2 && nt == 2) { HRF_WD_BNB(2, 2, 1) }
      else if (mt == 2) { HRF_WD_BNB(2, 4, 1) }
      else if (nt == 2) { HRF_WD_BNB(4, 2, 1) }
      else { HRF_WD_BNB(4, 4, 1) }
    } else {
      switch (mt) {
        case 2: HRF_WD_NT(2) break;
        case 3: HRF_WD_NT(3) break;
        case 4: HRF_WD_NT(4) break;
        default: HRF_WD_NT(5) break;
      }
    }
    return hrf_check_launch();
}

// ---- grouped weight-gradient launches (see WgradGroup): between begin and end every hrf_conv_bwd_weight call that
// maps to the pixel-major kernel is queued; end issues them, WGMAX problems of one kernel variant per launch.
bool hrf_wgrad_collecting() { return g_wg_collect; }
int hrf_dw_wgt_flush(void* stream);     // dwconv.hip: the depthwise weight gradients queued while collecting

extern "C" int hrf_wgrad_group_begin(void) {
  g_wg_pending.clear();
  g_wg_collect = true;
  return HRF_OK;
}

// in-situ timing of the grouped launches (hrf_debug_knob 7 = 1): HIP events on the launching stream around every
// launch of an eager step; hrf_wgrad_group_report aggregates them per kernel variant (bench.py's roofline leg)
#ifndef HRF_EMUL
struct WgTimed { int key, nprob; double bytes, flops; int Cin, Cout, H, W, stride, KH; hipEvent_t e0, e1; };
static std::vector<WgTimed> g_wg_timed;
#endif

extern "C" long hrf_wgrad_group_report(double* out, long cap_rows) {
  // rows of 12 doubles: key, launches, problems, total_us, algorithmic bytes, flops, then the dims of the
  // heaviest problem of the variant (Cin, Cout, H, W, stride, KH); returns the number of rows; clears the log
#ifdef HRF_EMUL
  (void)out; (void)cap_rows;
  return 0;
#else
  std::vector<std::vector<double>> rows;
  for (auto& t : g_wg_timed) {
    float ms = 0.f;
    hipEventSynchronize(t.e1);
    hipEventElapsedTime(&ms, t.e0, t.e1);
    hipEventDestroy(t.e0); hipEventDestroy(t.e1);
    std::vector<double>* r = nullptr;
    for (auto& q : rows) if ((int)q[0] == t.key) r = &q;
    if (r == nullptr) { rows.push_back(std::vector<double>(13, 0.0)); r = &rows.back(); (*r)[0] = t.key; }
    (*r)[1] += 1; (*r)[2] += t.nprob; (*r)[3] += ms * 1e3; (*r)[4] += t.bytes; (*r)[5] += t.flops;
    if (t.bytes / t.nprob > (*r)[12]) {
      (*r)[12] = t.bytes / t.nprob;
      (*r)[6] = t.Cin; (*r)[7] = t.Cout; (*r)[8] = t.H; (*r)[9] = t.W; (*r)[10] = t.stride; (*r)[11] = t.KH;
    }
  }
  g_wg_timed.clear();
  long n = 0;
  for (auto& q : rows) {
    if (n >= cap_rows) break;
    for (int k = 0; k < 12; ++k) out[n * 12 + k] = q[k];
    ++n;
  }
  return n;
#endif
}

extern "C" int hrf_wgrad_group_end(void* stream) {
  g_wg_collect = false;
  std::stable_sort(g_wg_pending.begin(), g_wg_pending.end(),
                   [](const WgPending& x, const WgPending& y) { return x.key < y.key; });
  int rc = HRF_OK;
  size_t i = 0;
  while (i < g_wg_pending.size()) {
    WgradGroup g;
    g.nprob = 0; g.bstart[0] = 0;
    const int key = g_wg_pending[i].key;
    while (i < g_wg_pending.size() && g_wg_pending[i].key == key && g.nprob < WGMAX) {
      g.p[g.nprob] = g_wg_pending[i].d;
      g.bstart[g.nprob + 1] = g.bstart[g.nprob] + g_wg_pending[i].blocks;
      ++g.nprob; ++i;
    }
    for (int k = g.nprob + 1; k <= WGMAX; ++k) g.bstart[k] = g.bstart[g.nprob];
#ifndef HRF_EMUL
    WgTimed tm;
    if (g_knob[7]) {
      tm.key = key; tm.nprob = g.nprob; tm.bytes = 0; tm.flops = 0;
      double best = -1;
      for (int k = 0; k < g.nprob; ++k) {
        const WgradDenseArgs& q = g.p[k];
        const bool tiled = key >= 4096, tap = !tiled && (key & 1), bnb = tiled ? (key >> 2) & 1 : (key >> 3) & 1;
        const double np = (tap ? 9.0 : 1.0) * q.Cin;
        const double inpix = tap ? (double)q.Mpix / ((double)q.Ho * q.Wo) * q.H * q.W : (double)q.Mpix;
        const double b = 4.0 * (inpix * q.Cin + (double)q.Mpix * q.Cout * (bnb ? 2 : 1) + (double)q.Cout * np);
        tm.bytes += b; tm.flops += 2.0 * q.Mpix * q.Cout * np;
        if (b > best) { best = b; tm.Cin = q.Cin; tm.Cout = q.Cout; tm.H = q.H; tm.W = q.W; tm.stride = q.stride; tm.KH = tap ? 3 : 1; }
      }
      hipEventCreate(&tm.e0); hipEventCreate(&tm.e1);
      hipEventRecord(tm.e0, (hipStream_t)stream);
    }
#endif
    const int r = wgrad_dense_launch(key, g, g.bstart[g.nprob], stream);
#ifndef HRF_EMUL
    if (g_knob[7]) { hipEventRecord(tm.e1, (hipStream_t)stream); g_wg_timed.push_back(tm); }
#endif
    if (r != HRF_OK) rc = r;
  }
  g_wg_pending.clear();
  const int rd = hrf_dw_wgt_flush(stream);
  return rc != HRF_OK ? rc : rd;
}
