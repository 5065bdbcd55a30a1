// 3x3 / stride-1 / pad-1 convolution, forward and backward-data, fp32 MFMA (gfx950).
//
// These are the FLOP carriers of the backbone (stem / layer1 Bottleneck conv2 / transition1:
// 2.3-4.5 GFLOP per launch, resnet.py:263-302, hrnet.py:417-455).  The generic implicit-GEMM engine
// re-reads every input pixel once per tap (9x) from L2 and pays one dependent global->LDS round
// trip per 64-deep K step; here a block stages the (8+2)x(16+2) input halo of a 64-channel slab
// ONCE (BatchNorm/activation applied once per element, not once per tap), walks all 9 taps x 64
// channels out of LDS, and only the weight tile (64 x 64 floats per K step, contiguous rows of the
// OIHW tensor because k = ci*9 + tap is its memory order) streams through a register prefetch.
// One block = 8x16 output pixels x up to 64 output channels, 8 waves; wave w owns output row w as one
// 16-pixel MFMA row tile (2 waves per SIMD: the LDS/VALU latency of one hides under the other's MFMAs).
#include "hrf_common.h"
#include "hrf_lin.h"
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, NPIX = IH * IW;   // 180 halo pixels
constexpr int CS = 64;       // channel slab
constexpr int CP = 66;       // halo pitch (floats per pixel): bank = (2*i + q) % 32
constexpr int BP = 66;       // weight-tile pitch: bank = (2*j + q) % 32 (ds_read_b32 is served in 32-lane halves: q in {0, 1} or {2, 3}; 68 was 2-way)
constexpr int NWV = 8;        // waves per block: one output row (16-pixel MFMA row tile) each, 2 waves per SIMD
constexpr int NHL = (NPIX + NWV - 1) / NWV;   // halo pixels per thread (NWV pixel groups x 64 channels)

__device__ float g_zero1[4] = {0.f, 0.f, 0.f, 0.f};

// MODE 0: forward (in = x, transform-on-load, bias/res epilogue, (sum, sumsq) moments)
// MODE 1: backward data (in = dY [+ BN-backward], flipped/transposed weights, += | act' epilogue)
// MODE 2: backward data of a STRIDE-2 convolution, one input-parity class (Y%2, X%2) per block: the
//         16 pixels of an MFMA row tile share their parity, so the set of contributing taps is uniform
//         (1, 2, 2 or 4 of the 9) and no masked/zero work is issued (the generic kernel ran all 9).
// KH = 1 (not instantiated): the same engine on wide 1x1 convolutions measured no better than the
//         register-only row-GEMM kernel (64<->256 channels at 30720 pixels: 30.6 vs 30.3 us).
template <int NT, int MODE, int KH>
__global__ __launch_bounds__(64 * NWV) void conv3_kernel(HrfGroup<Conv3Args> grp) {
  const Conv3Args& a = grp.sel();
  constexpr int IWm = KH == 1 ? TW : (MODE == 2 ? TW + 1 : IW);     // staged source tile width / pixel count
  constexpr int NPm = KH == 1 ? TH * TW : (MODE == 2 ? (TH + 1) * (TW + 1) : NPIX);
  constexpr int ORG = (KH == 1 || MODE == 2) ? 0 : -1;              // source tile origin relative to (y0, x0)
  constexpr int KK = KH * KH;
  __shared__ float sIn[NPIX * CP];
  __shared__ float sB[NT * 16 * BP];
  __shared__ float sStat[NWV * 2 * NT * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY; t /= a.tilesY;
  const int cls = MODE == 2 ? (t & 3) : 0, b = MODE == 2 ? (t >> 2) : t;
  const int cpy = cls >> 1, cpx = cls & 1;                          // MODE 2: parity class of the output pixels
  const int Hs = MODE == 2 ? a.Hs : a.H, Ws = MODE == 2 ? a.Ws : a.W;   // source (staged) grid
  const int y0 = ty * TH, x0 = tx * TW;
  const int n0 = blockIdx.y * (NT * 16);
  // BatchNorm of the input finalised on load (forward) / BatchNorm-backward coefficients derived on load (backward)
  __shared__ __attribute__((aligned(16))) float sFin[3 * HRF_FIN_MAXC];
  const float* t0p = a.t0;
  const float* t1p = a.t1;
  const float* t2p = a.t2;
  if (MODE == 0) {
    if (a.fin.stats != nullptr) {
      hrf_bn_fin_onload(a.fin, sFin, sFin + HRF_FIN_MAXC, tid, 64 * NWV, blockIdx.x == 0 && blockIdx.y == 0);
      __syncthreads();
      t0p = sFin; t1p = sFin + HRF_FIN_MAXC;
    }
  } else if (a.bfin.gstats != nullptr) {
    hrf_bn_bfin_onload(a.bfin, sFin, sFin + HRF_FIN_MAXC, sFin + 2 * HRF_FIN_MAXC, tid, 64 * NWV, blockIdx.x == 0 && blockIdx.y == 0);
    __syncthreads();
    t0p = sFin; t1p = sFin + HRF_FIN_MAXC; t2p = sFin + 2 * HRF_FIN_MAXC;
  }

  hrf_f4 acc[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) acc[tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  const int hc = tid & 63, hg = tid >> 6;          // halo staging: channel, pixel group
  const int bk = tid & 63, bc = tid >> 6;          // weight staging: k within step, row group
  const int aBase = (wave * IWm + i) * CP;         // staged address of (output row `wave`, pixel i, tap (0,0))

  for (int c0 = 0; c0 < a.Cin; c0 += CS) {
    // K order inside a slab is TAP-major (k = tap*64 + ci): the tap of a 64-deep step is uniform, so the
    // halo address of an A fragment is (uniform tap offset) + 4*kk + q - no per-lane index walk at all.
    const int csz = min(CS, a.Cin - c0);
    const int nsx = MODE == 2 ? 1 + cpx : KH, nsteps = MODE == 2 ? (1 + cpy) * nsx : KK;
    // ---- stage the halo slab (all loads first, transform afterwards)
    float hv[NHL], hv2[NHL];
    const bool cv = hc < csz;
    float p0 = 1.f, p1 = 0.f, p2 = 0.f;
    if (MODE == 0) { if (a.tf_mode != HRF_TF_NONE) { p0 = t0p[cv ? c0 + hc : 0]; p1 = t1p[cv ? c0 + hc : 0]; } }
    else if (t0p != nullptr) { p0 = t0p[cv ? c0 + hc : 0]; p1 = t1p[cv ? c0 + hc : 0]; p2 = t2p[cv ? c0 + hc : 0]; }
#pragma unroll
    for (int e = 0; e < NHL; ++e) {
      const int pix = e * NWV + hg;
      const int py = pix / IWm, px = pix - py * IWm;
      const int gy = y0 + ORG + py, gx = x0 + ORG + px;
      const bool ok = cv && pix < NPm && (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
      const long off = ((long)(b * Hs + gy) * Ws + gx) * a.ldIn + c0 + hc;
      hv[e] = *(ok ? a.in + off : g_zero1);
      hv2[e] = (MODE != 0 && a.in2 != nullptr) ? *(ok ? a.in2 + off : g_zero1) : 0.f;
    }
    // ---- first weight tile of the slab
    float bw[64 / NWV];
    auto load_b = [&](int step) {
      int tp = step;                                // tap of this step, bk = channel within the slab
      if (MODE == 2) {
        const int iy = step / nsx, ix = step - iy * nsx;
        tp = (cpy ? 2 * iy : 1) * 3 + (cpx ? 2 * ix : 1);
      }
#pragma unroll
      for (int e = 0; e < 64 / NWV; ++e) {
        const int j = bc + NWV * e;                 // output channel within the block column
        const bool ok = bk < csz && j < NT * 16 && n0 + j < a.Cout;
        long off;
        if (MODE == 0) off = ((long)(n0 + j) * a.wCin + c0 + bk) * KK + tp;
        else off = ((long)(c0 + bk) * a.wCin + n0 + j) * KK + (MODE == 1 ? KK - 1 - tp : tp);   // W[co = in ch][ci = out ch][.]
        bw[e] = *(ok ? a.w + off : g_zero1);
      }
    };
    load_b(0);
    __syncthreads();                                 // previous slab's readers are done
    // raw values (BN-backward combine in MODE 1) go to LDS first; the forward transform-on-load runs
    // afterwards as a ROLLED in-place pass (one copy of the BN/GELU code instead of 45 unrolled ones)
#pragma unroll
    for (int e = 0; e < NHL; ++e) {
      const int pix = e * NWV + hg;
      float v = hv[e];
      if (MODE != 0 && t0p != nullptr) {
        const int py = pix / IWm, px = pix - py * IWm;
        const int gy = y0 + ORG + py, gx = x0 + ORG + px;
        const bool ok = cv && (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
        v = ok ? fmaf(p0, v, fmaf(p1, hv2[e], p2)) : 0.f;
      }
      if (pix < NPm) sIn[pix * CP + hc] = v;
    }
    if (MODE == 0 && a.tf_mode != HRF_TF_NONE) {
#pragma unroll 1
      for (int e = 0; e < NHL; ++e) {
        const int pix = e * NWV + hg;
        const int py = pix / IWm, px = pix - py * IWm;
        const int gy = y0 + ORG + py, gx = x0 + ORG + px;
        const bool ok = cv && pix < NPm && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const int li = min(pix, NPm - 1) * CP + hc;
        const float v = hrf_tf_affine(a.tf_mode, sIn[li], p0, p1);      // own element: no barrier needed
        if (pix < NPm) sIn[li] = ok ? v : 0.f;
      }
    }
    const int nkk = (csz + 3) >> 2;
    for (int step = 0; step < nsteps; ++step) {
#pragma unroll
      for (int e = 0; e < 64 / NWV; ++e)
        if (bc + NWV * e < NT * 16) sB[(bc + NWV * e) * BP + bk] = bw[e];
      __syncthreads();
      if (step + 1 < nsteps) load_b(step + 1);
      int dy = step / KH, dx = step - KH * dy;                       // source offset of this tap
      if (MODE == 2) {
        const int iy = step / nsx, ix = step - iy * nsx;
        dy = cpy ? 1 - iy : 0; dx = cpx ? 1 - ix : 0;                // tap dy = 2*iy reads source row y' + 1 - iy
      }
      const float* ap = sIn + aBase + (dy * IWm + dx) * CP + q;
      const float* bp = sB + i * BP + q;
#pragma unroll 4
      for (int kk = 0; kk < nkk; ++kk) {
        const float a0 = ap[kk * 4];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = hrf_mfma16(a0, bp[tt * 16 * BP + kk * 4], acc[tt]);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: D[row = pixel 4q + r][col = channel i]
  const int xq = x0 + 4 * q;
  float s1[NT], s2[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) { s1[tt] = 0.f; s2[tt] = 0.f; }
  {
    const int y = MODE == 2 ? 2 * (y0 + wave) + cpy : y0 + wave;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int ch = n0 + tt * 16 + i;
      const bool chv = ch < a.Cout;
      const int chc = chv ? ch : 0;
      float bv = 0.f, esc = 1.f, esh = 0.f;
      if (MODE == 0) { if (a.bias != nullptr) bv = a.bias[chc]; }
      else if (a.epi == 1) { esc = a.esc[chc]; esh = a.esh[chc]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = MODE == 2 ? 2 * (xq + r) + cpx : xq + r;
        const bool ok = chv && y < a.H && x < a.W;
        const long prow = (long)(b * a.H + y) * a.W + x;
        float v = acc[tt][r];
        if (MODE == 0) {
          v += bv;
          if (a.res != nullptr) v += *(ok ? a.res + prow * a.ldR + ch : g_zero1);
          if (a.res2 != nullptr) v += *(ok ? a.res2 + prow * a.ldR + ch : g_zero1);
          if (ok) { a.out[prow * a.ldOut + a.ooff + ch] = v; s1[tt] += v; s2[tt] = fmaf(v, v, s2[tt]); }
        } else if (a.epi == 1) {
          const float xr = *(ok ? a.xraw + prow * a.ldXr + ch : g_zero1);
          v *= hrf_act_grad(a.act, fmaf(xr, esc, esh));
          if (ok) { a.out[prow * a.ldOut + ch] = v; s1[tt] += v; s2[tt] = fmaf(v, xr, s2[tt]); }
        } else {
          const float prev = a.accumulate ? *(ok ? a.out + prow * a.ldOut + ch : g_zero1) : 0.f;
          if (ok) a.out[prow * a.ldOut + ch] = prev + v;
        }
      }
    }
  }
  if (a.stats != nullptr && (MODE == 0 || a.epi == 1)) {   // (uniform condition)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      float u1 = s1[tt], u2 = s2[tt];
      u1 += __shfl_xor(u1, 16); u1 += __shfl_xor(u1, 32);
      u2 += __shfl_xor(u2, 16); u2 += __shfl_xor(u2, 32);
      if (lane < 16) { sStat[(wave * 2 + 0) * (NT * 16) + tt * 16 + lane] = u1; sStat[(wave * 2 + 1) * (NT * 16) + tt * 16 + lane] = u2; }
    }
    __syncthreads();
    double* st = a.stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * a.Cout;
    for (int e = tid; e < 2 * NT * 16; e += 64 * NWV) {
      const int which = e / (NT * 16), cidx = e - which * (NT * 16);
      const int ch = n0 + cidx;
      if (ch < a.Cout) {
        float sm = 0.f;
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) sm += sStat[(wv * 2 + which) * (NT * 16) + cidx];
        hrf_atomic_add(&st[which * a.Cout + ch], (double)sm);
      }
    }
  }
}

inline int conv3_nt(int C) { return C <= 32 ? 2 : 4; }

}  // namespace

template <int MODE, int KH>
static int conv3_launch(Conv3Args a, void* stream) {
  if (a.B <= 0 || a.H <= 0 || a.W <= 0) return HRF_OK;
  a.tilesX = hrf_cdiv(MODE == 2 ? (a.W + 1) / 2 : a.W, TW); a.tilesY = hrf_cdiv(MODE == 2 ? (a.H + 1) / 2 : a.H, TH);
  const int nt = conv3_nt(a.Cout);
  const dim3 grid(a.tilesX * a.tilesY * a.B * (MODE == 2 ? 4 : 1), hrf_cdiv(a.Cout, nt * 16));
  if (nt == 2) { HRF_LAUNCH_G((conv3_kernel<2, MODE, KH>), grid, dim3(64 * NWV), 0, stream, a); }
  else { HRF_LAUNCH_G((conv3_kernel<4, MODE, KH>), grid, dim3(64 * NWV), 0, stream, a); }
  return hrf_check_launch();
}

int hrf_conv3_fwd_launch(const Conv3Args& a, void* stream) { return conv3_launch<0, 3>(a, stream); }
int hrf_conv3_bwd_data_launch(const Conv3Args& a, void* stream) { return conv3_launch<1, 3>(a, stream); }
int hrf_conv3s2_bwd_data_launch(const Conv3Args& a, void* stream) { return conv3_launch<2, 3>(a, stream); }

