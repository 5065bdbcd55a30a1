// LDS-tiled fp32 MFMA weight gradient for WIDE stride-1 1x1 convolutions / Linears on channel-contiguous rows (gfx950).
//
//   dW[co][ci] += sum_pix bnbwd(dY)[pix][co] * tf(X)[pix][ci]          (+ dbias[co] += sum_pix bnbwd(dY)[pix][co])
//
// The pixel-major kernel of conv_engine.hip loads every MFMA fragment straight from memory, one dword per lane, and lets the
// 8 waves of a block split the PIXELS: no reuse between waves, MT + NT (+ MT) vector-memory instructions per MT * NT MFMAs, a
// load -> wait -> compute loop per wave and an 8-wave merge through LDS at the end.  That is the right shape for the narrow
// problems of HRFuser-T (18 ... 72 channels, 10 output tiles), and 10 - 25 % of the MFMA peak on the wide ones (HRFuser-B:
// 78 ... 2 496 channels; the 64 <-> 256 Bottleneck convolutions).  Here a block owns a (16 MT) x 128 tile of the output and
// streams its pixel chunk through LDS:
//   * both operands arrive as 16-byte row segments (coalesced), the BatchNorm-backward affine of dY and the producer's
//     affine / LayerNorm / activation of X are applied ONCE per element on the way into LDS;
//   * chunk i + 1 is in flight (registers) while chunk i is multiplied: double-buffered LDS, one barrier per 32 pixels;
//   * wave w multiplies the M-side panel (MT tiles, shared by all waves) with ITS 16 columns of the N-side panel: MT + 1
//     conflict-free ds_read_b32 per MT MFMAs, no merge - a wave's accumulators are final for the block's pixel chunk;
//   * either operand can be the M side (SWAP): the 128-wide side takes the dimension that pads least (Cin = 312, Cout = 78:
//     M = Cout = 80, N = Cin = 3 x 128);  the transposed tiles of SWAP pass through LDS once so that the atomics of a wave
//     stay 64-byte runs (the atomic units work cache line by cache line).
// Pixel splits, grouped launches and the XCD-aware block order are those of the pixel-major kernel (hrf_wgrad.h).
//
// Reference op replaced: the weight / bias gradient of every nn.Linear / 1x1 nn.Conv2d of the path (autograd of hrformer.py:
// 233-237, 281-333, utils/transformer.py:932-1018, resnet.py:263-302).
#include <algorithm>
#include <cstdlib>
#include "hrf_common.h"
#include "hrf_wgrad.h"

namespace {

constexpr int TK = 32;             // pixels per staged chunk
constexpr int TNW = 8;             // waves per block = 16-channel column tiles of the block
constexpr int TN = 16 * TNW;       // 128
constexpr int TNP = TN + 16;       // pitch: rows k, k+1, k+2, k+3 of a fragment start 16 banks apart
constexpr int NTHR = 64 * TNW;

__device__ float g_wt_zero[4] = {0.f, 0.f, 0.f, 0.f};

// 4 consecutive channels p[0..3] of which nv exist (p = a zero block for slots that do not exist: every load unconditional)
__device__ __forceinline__ hrf_f4 ld_grp(const float* p, int nv) {
  if (nv >= 4) return hrf_ld4(p);
  hrf_f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float v = p[e < nv ? e : 0]; r[e] = e < nv ? v : 0.f; }
  return r;
}

template <int MT, bool SWAP, bool BNB, int ACT>
__global__ __launch_bounds__(NTHR, 2) void wgrad_tiled_kernel(WgradGroup grp) {
  constexpr int TM = 16 * MT, TMP = 16 * (MT | 1);             // (pitch = odd multiple of 16: see TNP)
  constexpr int DW = SWAP ? TN : TM, DP = SWAP ? TNP : TMP;    // dY' panel: channels, LDS pitch
  constexpr int XW = SWAP ? TM : TN, XP = SWAP ? TMP : TNP;    // X' panel
  constexpr int GD = DW / 4, GX = XW / 4;                      // 16-byte groups per panel row
  constexpr int ND = (TK * GD + NTHR - 1) / NTHR, NX = (TK * GX + NTHR - 1) / NTHR;   // groups per thread and chunk
  __shared__ __attribute__((aligned(16))) float sD[2 * TK * DP];
  __shared__ __attribute__((aligned(16))) float sX[2 * TK * XP];
  __shared__ __attribute__((aligned(16))) float sCo[BNB ? 3 * DW : 4];   // cA | cB | cC of the block's dY channels
  __shared__ __attribute__((aligned(16))) float sAf[2 * XW];             // scale | shift of the block's X channels
  int prob = 0;
  while (prob + 1 < grp.nprob && (int)blockIdx.x >= grp.bstart[prob + 1]) ++prob;   // (wave-uniform, <= 15 steps)
  const WgradDenseArgs& a = grp.p[prob];
  const int bid = (int)blockIdx.x - grp.bstart[prob];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, kq = lane >> 4;
  // XCD-aware block order (see wgrad_dense_kernel): the (m, n) blocks of one pixel chunk sit on one XCD, back to back
  const int G = a.gx * a.gy;
  const int xcd = bid & 7, slot = bid >> 3;
  const int grpi = slot % G, bz = (slot / G) * 8 + xcd;
  if (bz >= a.sp) return;                                      // padding blocks (sp rounded up to 8)
  const int bx = grpi % a.gx, by = grpi / a.gx;
  const int m0 = bx * TM, n0 = by * TN;
  const int Nd = SWAP ? a.Cout : a.Cin;
  const int co0 = SWAP ? n0 : m0, ci0 = SWAP ? m0 : n0;
  const int pbeg = bz * a.chunk, pend = min(a.Mpix, pbeg + a.chunk);
  const bool ln = a.tf_rowstat != nullptr;

  for (int e = tid; e < DW; e += NTHR) {
    const int ch = co0 + e;
    if (BNB) {
      const bool v = ch < a.Cout;
      const int cs = v ? ch : 0;
      const float ca = a.cA[cs], cb = a.cB[cs], cc = a.cC[cs];
      sCo[e] = v ? ca : 0.f; sCo[DW + e] = v ? cb : 0.f; sCo[2 * DW + e] = v ? cc : 0.f;
    }
  }
  for (int e = tid; e < XW; e += NTHR) {
    const int ch = ci0 + e;
    const bool v = ch < a.Cin;
    float sc = 1.f, sh = 0.f;
    if (a.tf_scale != nullptr) { sc = a.tf_scale[v ? ch : 0]; sh = a.tf_shift[v ? ch : 0]; }
    sAf[e] = v ? sc : 0.f; sAf[XW + e] = v ? sh : 0.f;
  }

  // this thread's groups of a chunk (the same panel positions for every chunk): row (pixel of the chunk), first channel
  int rowD[ND], gD[ND], nvD[ND], rowX[NX], gX[NX], nvX[NX];
#pragma unroll
  for (int s = 0; s < ND; ++s) {
    const int e = tid + NTHR * s;
    rowD[s] = e / GD; gD[s] = 4 * (e - rowD[s] * GD);
    const int left = a.Cout - (co0 + gD[s]);
    nvD[s] = rowD[s] < TK ? (left < 0 ? 0 : (left > 4 ? 4 : left)) : 0;
  }
#pragma unroll
  for (int s = 0; s < NX; ++s) {
    const int e = tid + NTHR * s;
    rowX[s] = e / GX; gX[s] = 4 * (e - rowX[s] * GX);
    const int left = a.Cin - (ci0 + gX[s]);
    nvX[s] = rowX[s] < TK ? (left < 0 ? 0 : (left > 4 ? 4 : left)) : 0;
  }
  hrf_f4 rd[ND], ry[ND], rx[NX];
  float rm[NX], rr[NX];
  bool okD[ND], okX[NX];
  auto fetch = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < ND; ++s) {
      const int pix = p0 + rowD[s];
      okD[s] = nvD[s] > 0 && pix < pend;
      const unsigned off = (unsigned)pix * (unsigned)a.ldD + (unsigned)(a.doff + co0 + gD[s]);   // 32-bit offsets: tensors are < 2^31 elements
      rd[s] = ld_grp(okD[s] ? a.dy + off : g_wt_zero, okD[s] ? nvD[s] : 4);
      if (BNB) ry[s] = ld_grp(okD[s] ? a.yraw + off : g_wt_zero, okD[s] ? nvD[s] : 4);
    }
#pragma unroll
    for (int s = 0; s < NX; ++s) {
      const int pix = p0 + rowX[s];
      okX[s] = nvX[s] > 0 && pix < pend;
      const unsigned off = (unsigned)pix * (unsigned)a.ldX + (unsigned)(ci0 + gX[s]);
      rx[s] = ld_grp(okX[s] ? a.x + off : g_wt_zero, okX[s] ? nvX[s] : 4);
      rm[s] = 0.f; rr[s] = 1.f;
      if (ln) { const int pc = okX[s] ? pix : pbeg; rm[s] = a.tf_rowstat[2 * pc]; rr[s] = a.tf_rowstat[2 * pc + 1]; }
    }
  };
  // transform on the way into LDS: dY' = cA dY + cB Y + cC, X' = act(((X - mean) rstd) scale + shift); exact zeros for
  // pixels beyond the chunk and channels beyond the tensor (zero padding applies AFTER the transforms)
  auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < ND; ++s) {
      if (rowD[s] >= TK) continue;
      hrf_f4 v = rd[s];
      if (BNB) {
        const hrf_f4 ca = *reinterpret_cast<const hrf_f4*>(sCo + gD[s]), cb = *reinterpret_cast<const hrf_f4*>(sCo + DW + gD[s]),
                     cc = *reinterpret_cast<const hrf_f4*>(sCo + 2 * DW + gD[s]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(ca[r], v[r], fmaf(cb[r], ry[s][r], cc[r]));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (okD[s] && r < nvD[s]) ? v[r] : 0.f;
      *reinterpret_cast<hrf_f4*>(sD + buf * (TK * DP) + rowD[s] * DP + gD[s]) = v;
    }
#pragma unroll
    for (int s = 0; s < NX; ++s) {
      if (rowX[s] >= TK) continue;
      const hrf_f4 sc = *reinterpret_cast<const hrf_f4*>(sAf + gX[s]), sh = *reinterpret_cast<const hrf_f4*>(sAf + XW + gX[s]);
      hrf_f4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float w = fmaf((rx[s][r] - rm[s]) * rr[s], sc[r], sh[r]);            // (mean, rstd, sc, sh) = (0, 1, 1, 0) when unused
        const float u = ACT == 1 ? fmaxf(w, 0.f) : (ACT == 2 ? hrf_gelu(w) : w);
        v[r] = (okX[s] && r < nvX[s]) ? u : 0.f;
      }
      *reinterpret_cast<hrf_f4*>(sX + buf * (TK * XP) + rowX[s] * XP + gX[s]) = v;
    }
  };

  hrf_f4 acc[MT];
  float bs[MT], bq = 0.f;
#pragma unroll
  for (int t = 0; t < MT; ++t) { acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; bs[t] = 0.f; }
  const bool wact = n0 + 16 * wave < Nd;                       // (wave-uniform: a column tile beyond the tensor does no MFMAs)
  const int nch = (pend - pbeg + TK - 1) / TK;
  __syncthreads();                                             // coefficient tables
  fetch(pbeg);
  stage(0);
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch < nch; ++ch) {
    const bool more = ch + 1 < nch;
    if (more) fetch(pbeg + (ch + 1) * TK);                     // in flight while this chunk is multiplied
    if (wact) {
      const float* P = (SWAP ? sX : sD) + (ch & 1) * (TK * TMP);   // M side (pitch TMP)
      const float* Q = (SWAP ? sD : sX) + (ch & 1) * (TK * TNP);   // N side (pitch TNP)
#pragma unroll
      for (int ks = 0; ks < TK / 4; ++ks) {
        const int row = 4 * ks + kq;
        const float b = Q[row * TNP + 16 * wave + c];
        float av[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) av[t] = P[row * TMP + 16 * t + c];
#pragma unroll
        for (int t = 0; t < MT; ++t) { acc[t] = hrf_mfma16(av[t], b, acc[t]); bs[t] += av[t]; }
        bq += b;
      }
    }
    if (more) stage((ch + 1) & 1);
    __syncthreads();
  }

  const int Np = a.Cin;
  if (!SWAP) {
    if (wact) {
      const int col = n0 + 16 * wave + c;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + 16 * t + 4 * kq + r;
          if (row < a.Cout && col < a.Cin) hrf_atomic_add(&a.dw[(long)row * Np + col], acc[t][r]);
        }
    }
    if (a.dbias != nullptr && by == 0 && wave == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        float b = bs[t];
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (kq == 0 && m0 + 16 * t + c < a.Cout) hrf_atomic_add(&a.dbias[m0 + 16 * t + c], b);
      }
    }
  } else {
    // D[i = input channel][j = output channel]: transposed through a private 16 x 17 LDS tile per wave, so that a wave's
    // atomic instruction covers 64-byte runs of dW[co][ci .. ci + 15] (the panels are dead: barrier at the end of the loop)
    float* sT = sX + wave * (16 * 17);
    if (wact) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sT[(4 * kq + r) * 17 + c] = acc[t][r];
        HRF_WAVE_SYNC();
        const int ci = m0 + 16 * t + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = n0 + 16 * wave + 4 * kq + r;
          const float v = sT[c * 17 + 4 * kq + r];
          if (co < a.Cout && ci < a.Cin) hrf_atomic_add(&a.dw[(long)co * Np + ci], v);
        }
        HRF_WAVE_SYNC();
      }
      if (a.dbias != nullptr && bx == 0) {
        float b = bq;
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (kq == 0 && n0 + 16 * wave + c < a.Cout) hrf_atomic_add(&a.dbias[n0 + 16 * wave + c], b);
      }
    }
  }
}

inline int tiles_for(int C) {                                  // 16-row tiles per block on the M side: least padding, 2 .. 5
  const int T = hrf_cdiv(C, 16);
  int best = 2, cost = 1 << 30;
  for (int t = 2; t <= 5; ++t) {
    const int c = hrf_cdiv(T, t) * t;
    if (c <= cost) { cost = c; best = t; }
  }
  return best;
}

}  // namespace

// key of a tiled variant: 4096 + (((mt * 2 + swap) * 2 + bnb) * 4 + act)
bool hrf_wgrad_tiled_plan(WgradDenseArgs& d, bool bnb, int act, bool queued, int knob, int& key, int& nblocks) {
  // HRF_WGRAD_TILED=0 / hrf_debug_knob(8, 1): off; hrf_debug_knob(8, 2): every stride-1 1x1 problem (the tests' small shapes)
  static const int env = [] { const char* e = std::getenv("HRF_WGRAD_TILED"); return e != nullptr ? std::atoi(e) : 1; }();
  const bool force = knob == 2;
  if (knob == 1 || (env == 0 && !force)) return false;
  // Few pixels, wide channels: there the pixel-major kernel has too few pixels per 8-wave block to amortise its merge (12 grouped
  // problems, MI355X: 2 496 -> 624 at 12x20 9.9 -> 18.8 TFLOP/s, 624 -> 156 at 48x80 45.5 -> 57.5, 288 -> 72 at 24x40 21.8 -> 23.8).  On
  // the 96x160 maps it streams 30 720 pixels per problem at 55 - 62 TFLOP/s and this kernel, whose loads have one chunk (40 MFMAs
  // per wave) to land, reaches 32 - 48: those stay where they were (tools/bench_wgrad_tiled.py).
  if (!force && (d.Mpix < 8 * TK || d.Mpix > 8192 || d.Cin < 32 || d.Cout < 32)) return false;
  auto eff = [](int Cm, int Cn, int& mt) {
    mt = tiles_for(Cm);
    return (double)Cm / (hrf_cdiv(Cm, 16 * mt) * 16 * mt) * (double)Cn / (hrf_cdiv(Cn, TN) * TN);
  };
  int mt0, mt1;
  const double e0 = eff(d.Cout, d.Cin, mt0), e1 = eff(d.Cin, d.Cout, mt1);
  const bool swap = e1 > e0;
  const int mt = swap ? mt1 : mt0;
  if (!force && (swap ? e1 : e0) < 0.6) return false;
  d.gx = hrf_cdiv(swap ? d.Cin : d.Cout, 16 * mt);
  d.gy = hrf_cdiv(swap ? d.Cout : d.Cin, TN);
  const int G = d.gx * d.gy;
  int sp = queued ? 16 : std::min(64, hrf_cdiv(512, G));
  sp = std::min(sp, d.Mpix / ((force ? 1 : 4) * TK));          // >= 4 chunks per block
  if (sp < 1) sp = 1;
  if (sp > 8) sp = hrf_cdiv(sp, 8) * 8;
  d.chunk = hrf_cdiv(hrf_cdiv(d.Mpix, sp), TK) * TK;
  sp = hrf_cdiv(d.Mpix, d.chunk);
  d.sp = sp;
  nblocks = G * hrf_cdiv(sp, 8) * 8;
  key = 4096 + (((mt * 2 + (swap ? 1 : 0)) * 2 + (bnb ? 1 : 0)) * 4 + act);
  return true;
}

int hrf_wgrad_tiled_launch(int key, const WgradGroup& g, int total_blocks, void* stream) {
  const int k = key - 4096;
  const int act = k & 3, mt = k >> 4;
  const bool bnb = (k >> 2) & 1, swap = (k >> 3) & 1;
  const dim3 grid(total_blocks);
#define HRF_WT_LAUNCH(MT_, SW_, BNB_, ACT_) HRF_LAUNCH((wgrad_tiled_kernel<MT_, SW_, BNB_, ACT_>), grid, dim3(NTHR), 0, stream, g)
#define HRF_WT_ACT(MT_, SW_, BNB_)                              \
  if (act == 1) { HRF_WT_LAUNCH(MT_, SW_, BNB_, 1); }           \
  else if (act == 2) { HRF_WT_LAUNCH(MT_, SW_, BNB_, 2); }      \
  else { HRF_WT_LAUNCH(MT_, SW_, BNB_, 0); }
#define HRF_WT_BNB(MT_, SW_) if (bnb) { HRF_WT_ACT(MT_, SW_, true) } else { HRF_WT_ACT(MT_, SW_, false) }
#define HRF_WT_SW(MT_) if (swap) { HRF_WT_BNB(MT_, true) } else { HRF_WT_BNB(MT_, false) }
  switch (mt) {
    case 2: HRF_WT_SW(2) break;
    case 3: HRF_WT_SW(3) break;
    case 4: HRF_WT_SW(4) break;
    default: HRF_WT_SW(5) break;
  }
  return hrf_check_launch();
}
