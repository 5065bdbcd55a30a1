// Depthwise 3x3 convolution (pad 1, stride 1|2) for NHWC fp32 on gfx950 - HBM/LDS-bound stencil.
//
// The CrossFFN depthwise conv (hrformer.py:271-277: 4C channels, bias, stride 1) and the HRModule
// fuse-down depthwise convs (hrformer.py:532-541: stride 2, no bias) are ~95 launches per forward
// of the reference.  Channels are innermost, so a wave reads 32 consecutive channels of one pixel
// (128 B) and each LDS bank holds exactly one channel: the 3x3 window walks the LDS tile with zero
// bank conflicts.  The producer BatchNorm+GELU/ReLU is applied ONCE per element while staging the
// halo tile ("transform on load"), never per tap.
#include <algorithm>
#include <type_traits>
#include <vector>
#include "hrf_common.h"
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int CB = 32;    // channels per block (one LDS bank each)
constexpr int TW = 16;

template <int S> struct DwTile {
  static constexpr int TH = S == 1 ? 8 : 4;              // output tile rows
  static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
};

struct DwFwdArgs {
  const float* x; const float* w; const float* bias; float* y; double* stats; hrf_bn_fin_t fin;
  int tf_mode; const float* tf_scale; const float* tf_shift;
  int B, H, W, C, Ho, Wo, tilesX, tilesY;
  int nc4;                                               // dw4 kernels: float4 lanes (channels / 4) of a block's channel slab
};

template <int S>
__global__ __launch_bounds__(256) void dw_fwd_kernel(HrfGroup<DwFwdArgs> grp) {
  const DwFwdArgs& a = grp.sel();
  constexpr int TH = DwTile<S>::TH, IH = DwTile<S>::IH, IW = DwTile<S>::IW;
  __shared__ float sIn[IH * IW * CB];
  __shared__ float sStat[8 * 2 * CB];                    // per row-group partial moments (no LDS atomics)
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY; const int b = t / a.tilesY;
  const int c0 = blockIdx.y * CB;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  // the filter taps, the bias and the whole halo tile are requested FIRST; the BatchNorm of the input is finalised (this
  // block's CB channels only, hrf_bn_fin_t) while they are in flight - in front of the loads it was a memory round trip +
  // fp64 arithmetic + a barrier of its own: 1.4 of the kernel's 13.3 us at 2x96x160x72
  __shared__ float sFin[2 * CB];
  const float* scp = a.tf_scale;
  const float* shp = a.tf_shift;
  float wr[9];
  {
    const int cgw = c0 + (tid & 31);
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = a.w[(cgw < a.C ? cgw : c0) * 9 + k];
  }
  const float bv0 = a.bias ? a.bias[c0 + (tid & 31) < a.C ? c0 + (tid & 31) : c0] : 0.f;
  {
    const int c = tid & 31, cg = c0 + c;
    const bool cv = cg < a.C;
    // unconditional clamped loads (no load under a per-element branch), value selected afterwards
    constexpr int NIT = (IH * IW + 7) / 8;
    float raw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pix = it * 8 + (tid >> 5);
      const int iy = pix / IW, ix = pix - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool ok = cv && pix < IH * IW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      raw[it] = a.x[ok ? (((long)b * a.H + gy) * a.W + gx) * a.C + cg : 0];
    }
    if (a.fin.stats != nullptr) {
      hrf_bn_fin_onload(a.fin, sFin, sFin + CB, tid, 256, blockIdx.x == 0, c0, CB);
      __syncthreads();
      scp = sFin - c0; shp = sFin + CB - c0;
    }
    float sc = 1.f, sh = 0.f;
    if (a.tf_mode != HRF_TF_NONE) { sc = scp[cv ? cg : c0]; sh = shp[cv ? cg : c0]; }
    hrf_with_tf(a.tf_mode, [&](auto kind) HRF_KIND_INLINE {
      constexpr int MODE = decltype(kind)::value;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pix = it * 8 + (tid >> 5);
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool ok = cv && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        float v = raw[it];
        if (MODE != HRF_TF_NONE) v = hrf_tf_affine(MODE, v, sc, sh);
        if (pix < IH * IW) sIn[pix * CB + c] = ok ? v : 0.f;
      }
    });
  }
  __syncthreads();
  const int c = tid & 31, cg = c0 + c, rg = tid >> 5;
  const bool cv = cg < a.C;
  const float bv = cv ? bv0 : 0.f;
  // S=1: thread = one output row of 16; S=2: 8 row-groups over 4 rows -> half rows of 8
  const int row = S == 1 ? rg : (rg >> 1);
  const int xbeg = S == 1 ? 0 : (rg & 1) * 8, xcnt = S == 1 ? 16 : 8;
  const int oy = oy0 + row;
  float s1 = 0.f, s2 = 0.f;
  if (S == 1) {
    // stride 1: the thread's three halo rows (18 columns) are read ONCE into registers - 54 LDS reads for 16 outputs instead of
    // 144 - and the row is straight-line code (the column loop used to stop at the image edge with a `break`: rolled, one
    // dependent LDS round trip per output); outputs beyond the edge are computed and not stored
    float win[3][IW];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int x = 0; x < IW; ++x) win[dy][x] = sIn[((row + dy) * IW + x) * CB + c];
    const bool rowv = cv && oy < a.Ho;
    float* yrow = a.y + (((long)b * a.Ho + (rowv ? oy : 0)) * a.Wo) * a.C + cg;
#pragma unroll
    for (int q = 0; q < TW; ++q) {
      float acc = bv;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) acc = fmaf(win[dy][q + dx], wr[dy * 3 + dx], acc);
      const bool ok = rowv && ox0 + q < a.Wo;
      if (ok) yrow[(long)(ox0 + q) * a.C] = acc;
      s1 += ok ? acc : 0.f; s2 = ok ? fmaf(acc, acc, s2) : s2;
    }
  } else if (cv && oy < a.Ho) {
    for (int q = 0; q < xcnt; ++q) {
      const int oxl = xbeg + q, ox = ox0 + oxl;
      if (ox >= a.Wo) break;
      float acc = bv;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          acc = fmaf(sIn[((row * S + dy) * IW + oxl * S + dx) * CB + c], wr[dy * 3 + dx], acc);
      a.y[(((long)b * a.Ho + oy) * a.Wo + ox) * a.C + cg] = acc;
      s1 += acc; s2 = fmaf(acc, acc, s2);
    }
  }
  if (a.stats) {
    sStat[rg * 2 * CB + c] = s1;
    sStat[rg * 2 * CB + CB + c] = s2;
    __syncthreads();
    if (tid < 2 * CB && c0 + (tid & (CB - 1)) < a.C) {
      float tot = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) tot += sStat[g * 2 * CB + tid];
      double* st = a.stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * a.C;
      hrf_atomic_add(&st[(tid < CB ? 0 : a.C) + c0 + (tid & (CB - 1))], (double)tot);
    }
  }
}

// ------------------------------------------------------------------------------- float4-lane kernels (stride 1, C % 4 == 0)
// The kernels above give a lane ONE channel: a wave's global access is 2 pixels x 128 B, 23 dword loads + 16 dword stores per
// thread, 54 ds_read_b32 per 16 outputs, and at C = 72 the third 32-channel block runs 8 of its 32 lanes.  Here a lane owns
// FOUR consecutive channels (one float4) of one tile column: thread = (c4 < nc4, column < 16), nc4 = the largest divisor of
// C / 4 up to 18 (72 channels per slab: HRFuser-T's 72 / 144 / 288 / 576 have no ragged slab), so
//   * a wave's global access is contiguous ((column, c4) -> 16 B each: 1 KB per instruction at nc4 = 16+), the halo is 12 (7)
//     dwordx4 loads per thread, the outputs 8 (4) dwordx4 stores;
//   * the staged tile is [halo pixel][nc4 float4]: the same (column, c4) -> 16 B map makes every ds_read_b128 / ds_write_b128
//     of a wave one contiguous run (no bank conflict);
//   * the thread walks DOWN its column: each halo row is read once (3 float4) and feeds the (up to) three output rows it
//     touches - 3 (TH + 2) LDS reads for TH float4 outputs.
constexpr int D4_MAXL = 18;                               // float4 lanes of a slab (72 channels)

__host__ __device__ inline int dw4_lanes(int C) {
  if (C & 3) return 0;
  const int n = C >> 2;
  for (int d = D4_MAXL; d >= 8; --d)
    if (n % d == 0) return d;
  return 0;
}

template <int TH>
__global__ __launch_bounds__(16 * D4_MAXL) void dw4_fwd_kernel(HrfGroup<DwFwdArgs> grp) {
  const DwFwdArgs& a = grp.sel();
  constexpr int IH = TH + 2, IW = TW + 2, NPX = IH * IW, NIT = (NPX + 15) / 16;
  __shared__ __attribute__((aligned(16))) float sIn[NPX * D4_MAXL * 4];
  __shared__ __attribute__((aligned(16))) float sRed[16 * 2 * D4_MAXL * 4];
  __shared__ __attribute__((aligned(16))) float sFin[2 * D4_MAXL * 4];
  const int tid = threadIdx.x, nc4 = a.nc4, nt = 16 * nc4, CS = 4 * nc4;
  const int col = tid / nc4, c4 = tid - col * nc4;
  int t = blockIdx.x;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY; const int b = t / a.tilesY;
  const int c0 = blockIdx.y * CS, cg = c0 + 4 * c4;
  const int oy0 = ty * TH, ox0 = tx * TW;
  // every global operand is requested first (taps, bias, halo); the BatchNorm of the input is finalised while they fly
  hrf_f4 wreg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wreg[k] = hrf_ld4(a.w + (long)cg * 9 + 4 * k);
  hrf_f4 bv = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bv = hrf_ld4(a.bias + cg);
  hrf_f4 raw[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = col + 16 * it;                         // (the thread's float4 lane is the same for every element it stages)
    const int iy = p / IW, ix = p - iy * IW;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    const bool ok = p < NPX && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    raw[it] = hrf_ld4(a.x + (ok ? (((long)b * a.H + gy) * a.W + gx) * a.C : 0) + cg);
  }
  hrf_f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (a.fin.stats != nullptr) {
    hrf_bn_fin_onload(a.fin, sFin, sFin + 4 * D4_MAXL, tid, nt, blockIdx.x == 0, c0, CS);
    __syncthreads();
    sc = *reinterpret_cast<const hrf_f4*>(sFin + 4 * c4); sh = *reinterpret_cast<const hrf_f4*>(sFin + 4 * D4_MAXL + 4 * c4);
  } else if (a.tf_mode != HRF_TF_NONE) {
    sc = hrf_ld4(a.tf_scale + cg); sh = hrf_ld4(a.tf_shift + cg);
  }
  // ONE uniform branch per transform kind around the whole staging loop: with the kind tested per element the loop body was
  // 12 x 4 separately branched GELU chains (rcp -> 5 dependent FMAs -> exp2), none overlapping the next
  auto stage = [&](auto kind) HRF_KIND_INLINE {
    constexpr int MODE = decltype(kind)::value;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = col + 16 * it;
      const int iy = p / IW, ix = p - iy * IW;
      const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
      const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      hrf_f4 v = raw[it];
      if (MODE != HRF_TF_NONE) {
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = hrf_tf_affine(MODE, v[m], sc[m], sh[m]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = ok ? v[m] : 0.f;  // zero padding of the TRANSFORMED tensor
      if (p < NPX) *reinterpret_cast<hrf_f4*>(sIn + ((long)p * nc4 + c4) * 4) = v;
    }
  };
  hrf_with_tf(a.tf_mode, stage);
  __syncthreads();
  hrf_f4 acc[TH];
#pragma unroll
  for (int o = 0; o < TH; ++o) acc[o] = bv;
  // taps of channel cg + m: the 36 floats w[cg .. cg + 3][9] in memory order, element m * 9 + k
#define DW4_W(m, k) wreg[((m) * 9 + (k)) >> 2][((m) * 9 + (k)) & 3]
#pragma unroll
  for (int h = 0; h < IH; ++h) {
    const float* rp = sIn + ((long)(h * IW + col) * nc4 + c4) * 4;
    hrf_f4 in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) in[j] = *reinterpret_cast<const hrf_f4*>(rp + j * CS);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int o = h - ky;
      if (o >= 0 && o < TH) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[o][m] = fmaf(in[kx][m], DW4_W(m, ky * 3 + kx), acc[o][m]);
      }
    }
  }
#undef DW4_W
  hrf_f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const int ox = ox0 + col;
#pragma unroll
  for (int o = 0; o < TH; ++o) {
    const int oy = oy0 + o;
    const bool ok = oy < a.Ho && ox < a.Wo;
    if (ok) hrf_st4(a.y + (((long)b * a.Ho + oy) * a.Wo + ox) * a.C + cg, acc[o]);
#pragma unroll
    for (int m = 0; m < 4; ++m) { s1[m] += ok ? acc[o][m] : 0.f; s2[m] = ok ? fmaf(acc[o][m], acc[o][m], s2[m]) : s2[m]; }
  }
  if (a.stats) {
    // moments: the 16 columns of a channel meet in LDS, one atomic per channel and moment and block
    *reinterpret_cast<hrf_f4*>(sRed + ((col * 2 + 0) * nc4 + c4) * 4) = s1;
    *reinterpret_cast<hrf_f4*>(sRed + ((col * 2 + 1) * nc4 + c4) * 4) = s2;
    __syncthreads();
    if (tid < 2 * CS) {
      const int which = tid / CS, ch = tid - which * CS;
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) tot += sRed[(q * 2 + which) * CS + ch];
      double* st = a.stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * a.C;
      hrf_atomic_add(&st[(which ? a.C : 0) + c0 + ch], (double)tot);
    }
  }
}

// Tail of the weight-gradient kernels: sAcc = [8 row groups][10][CB] partial sums (taps 0..8, bias 9) of this block's CB
// channels.  dw is [C][9]: the block's slice is ONE contiguous run of 9 * CB floats, and walked in MEMORY order a wave's
// atomic instruction touches 2 cache lines; walked tap-major (lanes = channels, 36 B apart) it touched 18, and the atomic
// units work line by line: the tail was 9.4 of dw_bwd_data_kernel<1, true>'s 25.3 us at 2x96x160x72.
__device__ __forceinline__ void dw_wgt_tail(const float* sAcc, float* dw, float* dbias, long cp, int c0, int C, int tid) {
  const int nc = C - c0 < CB ? C - c0 : CB;
  for (int i = tid; i < 9 * nc; i += 256) {
    const int cl = i / 9, k = i - 9 * cl;
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) tot += sAcc[(g * 10 + k) * CB + cl];
    hrf_atomic_add(&dw[cp + (long)c0 * 9 + i], tot);
  }
  const int cl = tid - (256 - CB);                        // the bias sums: the last CB threads (they sit out the second pass above)
  if (dbias != nullptr && cl >= 0 && cl < nc) {
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) tot += sAcc[(g * 10 + 9) * CB + cl];
    hrf_atomic_add(&dbias[cp + c0 + cl], tot);
  }
}

// ------------------------------------------------------------------------------- backward data
struct DwBwdDataArgs {
  const float* dy; const float* yraw; const float* cA; const float* cB; const float* cC;
  const float* w;
  float* dx; int accumulate;
  int epi; const float* xraw; const float* tf_scale; const float* tf_shift; int act; double* stats;
  hrf_bn_bfin_t bfin;
  float* dw; float* dbias; long copy_stride;               // WG: weight / bias gradient accumulators (replicated copies)
  int B, H, W, C, Ho, Wo, tilesX, tilesY;
};

// WG (stride 1, epi 1): the weight gradient of the SAME convolution from the same pass.  dW[ky][kx] = sum_p x[p] *
// dy'[p - (ky-1, kx-1)] reads exactly the staged dy' element that dx[p] multiplies with w[ky][kx], and x[p] = act(u) is a
// by-product of the epilogue's act'(u): nine more FMAs per element instead of a second kernel that re-reads dY, Y and X.
template <int S, bool WG>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(HrfGroup<DwBwdDataArgs> grp) {
  const DwBwdDataArgs& a = grp.sel();
  // tile over INPUT pixels 8 x 16; staged dY region: S=1 (10 x 18, origin -1), S=2 (5 x 9, origin y0/2)
  constexpr int TH = 8, RH = S == 1 ? 10 : 5, RW = S == 1 ? 18 : 9;
  __shared__ float sD[RH * RW * CB];
  __shared__ float sStat[8 * 2 * CB];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY; const int b = t / a.tilesY;
  const int c0 = blockIdx.y * CB;
  const int y0 = ty * TH, x0 = tx * TW;
  const int ry0 = S == 1 ? y0 - 1 : y0 / 2, rx0 = S == 1 ? x0 - 1 : x0 / 2;
  const int c = tid & 31, cg = c0 + c;
  const bool cv = cg < a.C;
  // everything the kernel needs from memory is requested with the first batch of loads: the filter taps, the producer's
  // affine, this thread's row of raw producer outputs / previous dx values and the staged dY region; the BatchNorm-backward
  // coefficients (this block's CB channels only, hrf_bn_bfin_t) are derived while those are in flight
  __shared__ float sFin[3 * CB];
  const float* cAp = a.cA;
  const float* cBp = a.cB;
  const float* cCp = a.cC;
  const int r = tid >> 5;                                 // input row within tile (0..7)
  float wr[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wr[k] = a.w[(cv ? cg : c0) * 9 + k];
  float sc = 1.f, sh = 0.f;
  if (a.epi == 1) { sc = a.tf_scale[cv ? cg : c0]; sh = a.tf_shift[cv ? cg : c0]; }
  const int yi = y0 + r;
  float pre[TW];                                          // epi 1: raw producer output; else previous dx
#pragma unroll
  for (int q = 0; q < TW; ++q) {
    const bool ok = cv && yi < a.H && x0 + q < a.W;
    const long o = ok ? (((long)b * a.H + yi) * a.W + x0 + q) * a.C + cg : 0;
    pre[q] = a.epi == 1 ? a.xraw[o] : (a.accumulate ? a.dx[o] : 0.f);
  }
  {
    const bool bnb = a.cA != nullptr;
    constexpr int NIT = (RH * RW + 7) / 8;
    float rd[NIT], ry[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pix = it * 8 + (tid >> 5);
      const int ly = pix / RW, lx = pix - ly * RW;
      const int oy = ry0 + ly, ox = rx0 + lx;
      const bool ok = cv && pix < RH * RW && (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
      const long idx = ok ? (((long)b * a.Ho + oy) * a.Wo + ox) * a.C + cg : 0;
      rd[it] = a.dy[idx];
      ry[it] = bnb ? a.yraw[idx] : 0.f;
    }
    if (a.bfin.gstats != nullptr) {
      hrf_bn_bfin_onload(a.bfin, sFin, sFin + CB, sFin + 2 * CB, tid, 256, blockIdx.x == 0, c0, CB);
      __syncthreads();
      cAp = sFin - c0; cBp = sFin + CB - c0; cCp = sFin + 2 * CB - c0;
    }
    float ca = 1.f, cb = 0.f, cc = 0.f;
    if (bnb) { const int cs = cv ? cg : c0; ca = cAp[cs]; cb = cBp[cs]; cc = cCp[cs]; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pix = it * 8 + (tid >> 5);
      const int ly = pix / RW, lx = pix - ly * RW;
      const int oy = ry0 + ly, ox = rx0 + lx;
      const bool ok = cv && (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
      float v = rd[it];
      if (bnb) v = fmaf(ca, v, fmaf(cb, ry[it], cc));
      if (pix < RH * RW) sD[pix * CB + c] = ok ? v : 0.f;
    }
  }
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  float wacc[WG ? 10 : 1];
#pragma unroll
  for (int k = 0; k < (WG ? 10 : 1); ++k) wacc[k] = 0.f;
  // stride 1: the thread's three staged rows (18 columns) are read ONCE into registers: 54 LDS reads for 16 outputs instead of 144
  float win[S == 1 ? 3 : 1][S == 1 ? RW : 1];
  if (S == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int x = 0; x < RW; ++x) win[k][x] = sD[((r + k) * RW + x) * CB + c];
  }
  // act'(u) (WG: and x = act(u), from ONE evaluation) of the thread's 16 pixels, one uniform branch per kind around the loop
  float xvq[WG ? TW : 1], gvq[TW];
  if (a.epi == 1) {
    hrf_with_act(a.act, [&](auto kind) HRF_KIND_INLINE {
      constexpr int ACT = decltype(kind)::value;
#pragma unroll
      for (int q = 0; q < TW; ++q) {
        const float u = fmaf(pre[q], sc, sh);
        if (WG) hrf_act_both(ACT, u, xvq[q], gvq[q]);
        else gvq[q] = hrf_act_grad(ACT, u);
      }
    });
  }
#pragma unroll
  for (int q = 0; q < TW; ++q) {
    const int xi = x0 + q;
    const bool ok = cv && yi < a.H && xi < a.W;
    float acc = 0.f;
    // WG: every staged dy' element is read once for both products
    float xv = 0.f;
    if (WG) xv = ok ? xvq[q] : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (S == 1) {
          const float d = win[2 - dy][q + 2 - dx];
          acc = fmaf(d, wr[dy * 3 + dx], acc);
          if (WG) {
            wacc[dy * 3 + dx] = fmaf(d, xv, wacc[dy * 3 + dx]);
            if (dy == 1 && dx == 1) wacc[9] += ok ? d : 0.f;
          }
        } else {
          const int ty2 = r + 1 - dy, tx2 = q + 1 - dx;   // relative to (y0, x0), both even
          if (ty2 >= 0 && tx2 >= 0 && ((ty2 | tx2) & 1) == 0)
            acc = fmaf(sD[((ty2 >> 1) * RW + (tx2 >> 1)) * CB + c], wr[dy * 3 + dx], acc);
        }
      }
    const long o = (((long)b * a.H + yi) * a.W + xi) * a.C + cg;
    if (a.epi == 1) {
      const float xr = pre[q];
      acc *= gvq[q];
      if (ok) { s1 += acc; s2 = fmaf(acc, xr, s2); a.dx[o] = acc; }
    } else {
      if (ok) a.dx[o] = pre[q] + acc;
    }
  }
  if (a.epi == 1 && a.stats) {
    sStat[r * 2 * CB + c] = s1;
    sStat[r * 2 * CB + CB + c] = s2;
    __syncthreads();
    if (tid < 2 * CB && c0 + (tid & (CB - 1)) < a.C) {
      float tot = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) tot += sStat[g * 2 * CB + tid];
      double* st = a.stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * a.C;
      hrf_atomic_add(&st[(tid < CB ? 0 : a.C) + c0 + (tid & (CB - 1))], (double)tot);
    }
  }
  if (WG) {
    __syncthreads();                                      // every read of the staged dy' tile is done: reuse it
    float* sAcc = sD;                                     // [8 rows][10][CB] (10 KB of the 23 KB tile)
#pragma unroll
    for (int k = 0; k < 10; ++k) sAcc[(r * 10 + k) * CB + c] = wacc[k];
    __syncthreads();
    dw_wgt_tail(sAcc, a.dw, a.dbias, (long)(blockIdx.x % HRF_STAT_COPIES) * a.copy_stride, c0, a.C, tid);
  }
}

// ------------------------------------------------------------------------------- backward weight
struct DwBwdWgtArgs {
  const float* dy; const float* yraw; const float* cA; const float* cB; const float* cC;
  const float* x; int tf_mode; const float* tf_scale; const float* tf_shift;
  float* dw; float* dbias; long copy_stride;
  int B, H, W, C, Ho, Wo, tilesX, tilesY;
};

template <int S>
__device__ __forceinline__ void dw_bwd_wgt_body(const DwBwdWgtArgs& a) {
  constexpr int TH = DwTile<S>::TH, IH = DwTile<S>::IH, IW = DwTile<S>::IW;
  __shared__ float sIn[IH * IW * CB];
  __shared__ float sAcc[8 * 10 * CB];                    // per row-group partials (plain stores, no LDS atomics)
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY; const int b = t / a.tilesY;
  const int c0 = blockIdx.y * CB;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int c = tid & 31, cg = c0 + c;
  const bool cv = cg < a.C;
  const int rg = tid >> 5;
  const int row = S == 1 ? rg : (rg >> 1);
  constexpr int XC = S == 1 ? 16 : 8;
  const int xbeg = S == 1 ? 0 : (rg & 1) * 8;
  const int oy = oy0 + row;
  const bool bnb = a.cA != nullptr;
  float gq[XC], yq[XC];
#pragma unroll
  for (int q = 0; q < XC; ++q) {                          // this thread's dY row: issued before the halo loads
    const int ox = ox0 + xbeg + q;
    const bool ok = cv && oy < a.Ho && ox < a.Wo;
    const long idx = ok ? (((long)b * a.Ho + oy) * a.Wo + ox) * a.C + cg : 0;
    gq[q] = a.dy[idx];
    yq[q] = bnb ? a.yraw[idx] : 0.f;
  }
  {
    float sc = 1.f, sh = 0.f;
    if (a.tf_mode != HRF_TF_NONE) { sc = a.tf_scale[cv ? cg : 0]; sh = a.tf_shift[cv ? cg : 0]; }
    // unconditional clamped loads (no load under a per-element branch), value selected afterwards
    constexpr int NIT = (IH * IW + 7) / 8;
    float raw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pix = it * 8 + (tid >> 5);
      const int iy = pix / IW, ix = pix - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool ok = cv && pix < IH * IW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      raw[it] = a.x[ok ? (((long)b * a.H + gy) * a.W + gx) * a.C + cg : 0];
    }
    hrf_with_tf(a.tf_mode, [&](auto kind) HRF_KIND_INLINE {
      constexpr int MODE = decltype(kind)::value;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pix = it * 8 + (tid >> 5);
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool ok = cv && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        float v = raw[it];
        if (MODE != HRF_TF_NONE) v = hrf_tf_affine(MODE, v, sc, sh);
        if (pix < IH * IW) sIn[pix * CB + c] = ok ? v : 0.f;
      }
    });
  }
  __syncthreads();
  float ca = 1.f, cb = 0.f, cc = 0.f;
  if (bnb && cv) { ca = a.cA[cg]; cb = a.cB[cg]; cc = a.cC[cg]; }
  float acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.f;
#pragma unroll
  for (int q = 0; q < XC; ++q) {
    const int oxl = xbeg + q, ox = ox0 + oxl;
    const bool ok = cv && oy < a.Ho && ox < a.Wo;
    float g = gq[q];
    if (bnb) g = fmaf(ca, g, fmaf(cb, yq[q], cc));
    g = ok ? g : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        acc[dy * 3 + dx] = fmaf(g, sIn[((row * S + dy) * IW + oxl * S + dx) * CB + c], acc[dy * 3 + dx]);
    acc[9] += g;
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) sAcc[(rg * 10 + k) * CB + c] = acc[k];
  __syncthreads();
  dw_wgt_tail(sAcc, a.dw, a.dbias, (long)(blockIdx.x % HRF_STAT_COPIES) * a.copy_stride, c0, a.C, tid);
}

template <int S>
__global__ __launch_bounds__(256) void dw_bwd_wgt_kernel(DwBwdWgtArgs a) { dw_bwd_wgt_body<S>(a); }

// Weight gradients are leaves of the backward graph: between hrf_wgrad_group_begin and _end the depthwise ones are queued like
// the dense ones and issued DWG problems per launch (blockIdx.z = problem; blocks beyond a problem's own grid leave at once):
// HRFuser-T has 33 of these 10 us launches per step, 8 per lane of the weight-gradient phase.
constexpr int DWG = 12;
struct DwWgtGroup { DwBwdWgtArgs p[DWG]; };
template <int S>
__global__ __launch_bounds__(256) void dw_bwd_wgt_grp_kernel(DwWgtGroup g) {
  const DwBwdWgtArgs& a = g.p[blockIdx.z];
  if ((int)blockIdx.x >= a.tilesX * a.tilesY * a.B || (int)blockIdx.y * CB >= a.C) return;   // (uniform)
  dw_bwd_wgt_body<S>(a);
}

struct DwWgtPending { int stride; DwBwdWgtArgs a; };
std::vector<DwWgtPending> g_dw_pending;

}  // namespace

bool hrf_wgrad_collecting();            // conv_engine.hip: inside hrf_wgrad_group_begin / _end

// called by hrf_wgrad_group_end: issue the queued depthwise weight gradients
int hrf_dw_wgt_flush(void* stream) {
  int rc = HRF_OK;
  for (int stride = 1; stride <= 2; ++stride) {
    size_t i = 0;
    while (true) {
      DwWgtGroup g;
      int n = 0, gx = 0, gy = 0;
      for (; i < g_dw_pending.size() && n < DWG; ++i) {
        if (g_dw_pending[i].stride != stride) continue;
        const DwBwdWgtArgs& a = g_dw_pending[i].a;
        g.p[n++] = a;
        gx = std::max(gx, a.tilesX * a.tilesY * a.B); gy = std::max(gy, hrf_cdiv(a.C, CB));
      }
      if (n == 0) break;
      for (int k = n; k < DWG; ++k) g.p[k] = g.p[0];
      const dim3 grid(gx, gy, n);
      if (stride == 1) { HRF_LAUNCH(dw_bwd_wgt_grp_kernel<1>, grid, dim3(256), 0, stream, g); }
      else { HRF_LAUNCH(dw_bwd_wgt_grp_kernel<2>, grid, dim3(256), 0, stream, g); }
      if (hrf_check_launch() != HRF_OK) rc = HRF_ERR_LAUNCH;
    }
  }
  g_dw_pending.clear();
  return rc;
}

// tuning aids (hrf_debug_knob 40 / 41): [0] float4-lane forward: 0 auto, 1 off (default), 2 always 8-row tiles, 3 always 4-row tiles;
// [1] smallest 8-row-tile grid that keeps 8-row tiles
// The float4-lane forward is NOT the default (round 6, same-box A/B of two library builds, three interleaved pairs): isolated it is
// 5 % (72 channels) ... 20 % (144 - 576) faster, in the captured step 10.71 -> 10.77 ms - its 288-thread blocks (61 KB of LDS, 169
// registers) hold more of a CU than the 256-thread / 25 KB ones beside the kernels of the other lanes.  hrf_debug_knob(40, 0 / 2 / 3)
// selects it (tests, tools/bench_dw.py); -DHRF_DW4_DEFAULT=0 builds a library that uses it.
#ifndef HRF_DW4_DEFAULT
#define HRF_DW4_DEFAULT 1
#endif
static int g_dw4[4] = {HRF_DW4_DEFAULT, 200, 0, 0};
extern "C" __attribute__((visibility("hidden"))) int hrf_dw_knob(int key, int value) {
  if (key < 0 || key >= 4) return HRF_ERR_ARG;
  g_dw4[key] = value;
  return HRF_OK;
}

extern "C" int hrf_dwconv_fwd(const float* x, int B, int H, int W, int C, const float* w, const float* bias,
                              int stride, int tf_mode, const float* tf_scale, const float* tf_shift, float* y,
                              double* stats, const hrf_bn_fin_t* tf_fin, void* stream) {
  HRF_GROUP_CALL();
  if (stride != 1 && stride != 2) return HRF_ERR_ARG;
  if (tf_fin != nullptr && (tf_mode < HRF_TF_AFFINE || tf_mode > HRF_TF_AFFINE_GELU || tf_fin->C != C || tf_fin->stats == nullptr)) return HRF_ERR_ARG;
  DwFwdArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats; a.tf_mode = tf_mode; a.tf_scale = tf_scale;
  a.fin = hrf_bn_fin_t{};
  if (tf_fin != nullptr) a.fin = *tf_fin;
  a.tf_shift = tf_shift; a.B = B; a.H = H; a.W = W; a.C = C;
  a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
  const int th = stride == 1 ? 8 : 4;
  a.tilesX = hrf_cdiv(a.Wo, TW); a.tilesY = hrf_cdiv(a.Ho, th);
  if ((long)B * a.Ho * a.Wo <= 0) return HRF_OK;
  a.nc4 = (stride == 1 && g_dw4[0] != 1) ? dw4_lanes(C) : 0;
  if (a.nc4 > 0) {
    // float4-lane kernel: 8-row tiles when they still give every CU ~two blocks, 4-row tiles otherwise
    const int slabs = C / (4 * a.nc4);
    const bool th8 = g_dw4[0] == 2 || (g_dw4[0] != 3 && (long)a.tilesX * a.tilesY * B * slabs >= g_dw4[1]);
    if (!th8) a.tilesY = hrf_cdiv(a.Ho, 4);
    dim3 grid4(a.tilesX * a.tilesY * B, slabs);
    if (th8) { HRF_LAUNCH_G(dw4_fwd_kernel<8>, grid4, dim3(16 * a.nc4), 0, stream, a); }
    else { HRF_LAUNCH_G(dw4_fwd_kernel<4>, grid4, dim3(16 * a.nc4), 0, stream, a); }
    return hrf_check_launch();
  }
  dim3 grid(a.tilesX * a.tilesY * B, hrf_cdiv(C, CB));
  if (stride == 1) { HRF_LAUNCH_G(dw_fwd_kernel<1>, grid, dim3(256), 0, stream, a); }
  else { HRF_LAUNCH_G(dw_fwd_kernel<2>, grid, dim3(256), 0, stream, a); }
  return hrf_check_launch();
}

static int dw_bwd_data_launch(const float* dy, const float* yraw, const float* cA, const float* cB,
                              const float* cC, const hrf_bn_bfin_t* bfin, const float* w, int stride, int B, int H, int W, int C,
                              float* dx, int accumulate, int epi, const float* xraw, const float* tf_scale,
                              const float* tf_shift, int act, double* stats, float* dw, float* dbias, long copy_stride,
                              void* stream) {
  if (stride != 1 && stride != 2) return HRF_ERR_ARG;
  if (dw != nullptr && (stride != 1 || epi != 1 || xraw == nullptr)) return HRF_ERR_ARG;
  if (bfin != nullptr && (cA == nullptr || bfin->C != C || bfin->gstats == nullptr)) return HRF_ERR_ARG;
  DwBwdDataArgs a;
  a.bfin = hrf_bn_bfin_t{};
  if (bfin != nullptr) a.bfin = *bfin;
  a.dy = dy; a.yraw = yraw; a.cA = cA; a.cB = cB; a.cC = cC; a.w = w; a.dx = dx; a.accumulate = accumulate;
  a.epi = epi; a.xraw = xraw; a.tf_scale = tf_scale; a.tf_shift = tf_shift; a.act = act; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
  a.tilesX = hrf_cdiv(W, TW); a.tilesY = hrf_cdiv(H, 8);
  if ((long)B * H * W <= 0) return HRF_OK;
  dim3 grid(a.tilesX * a.tilesY * B, hrf_cdiv(C, CB));
  a.dw = dw; a.dbias = dbias; a.copy_stride = copy_stride;
  if (dw != nullptr) { HRF_LAUNCH_G((dw_bwd_data_kernel<1, true>), grid, dim3(256), 0, stream, a); }
  else if (stride == 1) { HRF_LAUNCH_G((dw_bwd_data_kernel<1, false>), grid, dim3(256), 0, stream, a); }
  else { HRF_LAUNCH_G((dw_bwd_data_kernel<2, false>), grid, dim3(256), 0, stream, a); }
  return hrf_check_launch();
}

extern "C" int hrf_dwconv_bwd_data(const float* dy, const float* yraw, const float* cA, const float* cB,
                                   const float* cC, const hrf_bn_bfin_t* bfin, const float* w, int stride, int B, int H, int W, int C,
                                   float* dx, int accumulate, int epi, const float* xraw, const float* tf_scale,
                                   const float* tf_shift, int act, double* stats, void* stream) {
  HRF_GROUP_CALL();
  return dw_bwd_data_launch(dy, yraw, cA, cB, cC, bfin, w, stride, B, H, W, C, dx, accumulate, epi, xraw, tf_scale, tf_shift,
                            act, stats, nullptr, nullptr, 0, stream);
}

extern "C" int hrf_dwconv_bwd_data_weight(const float* dy, const float* yraw, const float* cA, const float* cB,
                                          const float* cC, const hrf_bn_bfin_t* bfin, const float* w, int B, int H, int W, int C,
                                          float* dx, const float* xraw, const float* tf_scale, const float* tf_shift, int act,
                                          double* stats, float* dw, float* dbias, long copy_stride, void* stream) {
  HRF_GROUP_CALL();
  if (dw == nullptr) return HRF_ERR_ARG;
  return dw_bwd_data_launch(dy, yraw, cA, cB, cC, bfin, w, 1, B, H, W, C, dx, 0, 1, xraw, tf_scale, tf_shift, act, stats, dw,
                            dbias, copy_stride, stream);
}

extern "C" int hrf_dwconv_bwd_weight(const float* dy, const float* yraw, const float* cA, const float* cB,
                                     const float* cC, const float* x, int B, int H, int W, int C, int stride,
                                     int tf_mode, const float* tf_scale, const float* tf_shift, float* dw,
                                     float* dbias, long copy_stride, void* stream) {
  if (stride != 1 && stride != 2) return HRF_ERR_ARG;
  DwBwdWgtArgs a;
  a.dy = dy; a.yraw = yraw; a.cA = cA; a.cB = cB; a.cC = cC; a.x = x; a.tf_mode = tf_mode;
  a.tf_scale = tf_scale; a.tf_shift = tf_shift; a.dw = dw; a.dbias = dbias; a.copy_stride = copy_stride;
  a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
  const int th = stride == 1 ? 8 : 4;
  a.tilesX = hrf_cdiv(a.Wo, TW); a.tilesY = hrf_cdiv(a.Ho, th);
  if ((long)B * a.Ho * a.Wo <= 0) return HRF_OK;
  if (hrf_wgrad_collecting()) { g_dw_pending.push_back(DwWgtPending{stride, a}); return HRF_OK; }
  dim3 grid(a.tilesX * a.tilesY * B, hrf_cdiv(C, CB));
  if (stride == 1) { HRF_LAUNCH(dw_bwd_wgt_kernel<1>, grid, dim3(256), 0, stream, a); }
  else { HRF_LAUNCH(dw_bwd_wgt_kernel<2>, grid, dim3(256), 0, stream, a); }
  return hrf_check_launch();
}
