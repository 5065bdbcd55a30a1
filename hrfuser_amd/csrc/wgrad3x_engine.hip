// Weight gradient of the front-end 3x3 convolutions (stride 1 | 2, pad 1, >= 32 input channels), fp32 MFMA 32x32x2 (gfx950):
//   dW[co][ci][tap] += sum_pix dY'[pix][co] * X'[pix * stride + tap][ci],  dY' = cA du + cB yraw + cC (BatchNorm backward on load),
//   X' = act(sc x + sh) (the producer's BatchNorm + activation on load, zero padding AFTER it).
// The stems' Bottleneck conv2 / stem conv2 (64 -> 64) and transition1 (256 -> 18 | 36) carry 27 of the 70 GFLOP of a step's weight
// gradients (resnet.py:263-302, hrnet.py:341-358,430-459).  The pixel-major kernel of conv_engine.hip (rounds 2-5) feeds
// v_mfma_f32_16x16x4 straight from global memory with one DWORD per lane and fragment (13 loads per 36 MFMAs, every dY row
// fetched once per 16-channel group), merges its 8 waves through LDS in 7 rounds and ends in 9 216 fp32 atomics per block:
// 33-44 TFLOP/s, and its merge + atomic tail is ~25 of its 68 us.  Here
//   * a block owns a 32 x 32 (co, ci) tile of ALL NINE taps and a share of the pixel tiles; its three waves take three taps each
//     (3 accumulator tiles of 32 x 32 = 48 registers): no merge between waves, no atomics - the block's sums leave as plain
//     128-byte-run stores into a per-split slab part[split][tap][co][ci], folded by wgrad3x_fold_kernel in a second, small
//     launch of the same C call;
//   * both operands are staged ONCE per pixel tile (16-byte coalesced loads, transform applied once per element) into
//     double-buffered LDS tiles [pixel][32 channels]: K = pixels, so a fragment is 32 consecutive floats of one pixel row -
//     conflict-free ds_read_b32 for any tap and either stride (the tap only moves the pixel);
//   * v_mfma_f32_32x32x2_f32: 4 LDS reads per 3 MFMAs (one dY fragment shared by the wave's three taps);
//   * 43-50 KB of LDS, 144 registers: three blocks per CU.
// Selected by hrf_conv_bwd_weight_s (conv_engine.hip) for problems of >= 16 384 output pixels; the OIHW layout of dw is unchanged.
#include <cstdlib>
#include "hrf_common.h"
#include "hrf_wgrad.h"
#include "../../include/hrfuser_hip.h"

namespace {

#ifndef W3_VARIANT
#define W3_VARIANT 0      // timing experiments only (tools): 1 no MFMA, 2 no fold launch, 3 no staging after the first tile
#endif
constexpr int GNT = 192;                  // threads per block: three waves, three taps each

#ifdef HRF_EMUL
struct w3_f16 {
  float d[16];
  float& operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
};
inline w3_f16 w3_mfma32(float a, float b, w3_f16 c) {       // v_mfma_f32_32x32x2_f32 (see conv3x_engine.hip)
  char* buf = static_cast<char*>(hrf_emul::wave_buf());
  const int lane = hrf_emul::cur_lane;
  float ab[2] = {a, b};
  std::memcpy(buf + 16 * lane, ab, 8);
  hrf_emul::sync_wave();
  auto A = [&](int i, int k) { float v; std::memcpy(&v, buf + 16 * (k * 32 + i), 4); return v; };
  auto B = [&](int k, int j) { float v; std::memcpy(&v, buf + 16 * (k * 32 + j) + 4, 4); return v; };
  w3_f16 d = c;
  const int col = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(A(row, k), B(k, col), acc);
    d[r] = acc;
  }
  hrf_emul::sync_wave();
  return d;
}
#define W3_INLINE
#define W3_FENCE() ((void)0)
#else
typedef float w3_f16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ w3_f16 w3_mfma32(float a, float b, w3_f16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
#define W3_INLINE __attribute__((always_inline))
#define W3_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

__device__ float g_zero4g[4] = {0.f, 0.f, 0.f, 0.f};

struct W3xArgs {
  const float* dy; int ldD; const float* yraw;       // dY rows (column offset added by the launcher); raw conv output or null
  const float* cA; const float* cB; const float* cC;
  const float* x; int ldX;
  const float* sc; const float* sh; int act;         // transform of x (sc == null: none); act 0 none | 1 ReLU | 2 GELU
  float* part;                                       // [splits][9][Cout][Cin]
  int B, H, W, Ho, Wo, Cin, Cout;
  int tilesX, tilesY, ntiles, splits, cig;           // cig = 32-channel groups of the input
};

// STRIDE 1: 4 x 16 output pixels per tile, halo 6 x 18; STRIDE 2: 2 x 16 output pixels, source patch 5 x 33
// RAGY: the channel count of dY is not a multiple of 4 (transition1's 18): element loads for its rows
template <int STRIDE, bool RAGY>
__global__ __launch_bounds__(GNT, 2) void wgrad3x_kernel(W3xArgs a) {
  constexpr int TR = STRIDE == 1 ? 4 : 2, TP = TR * 16;                     // output rows / pixels per tile
  constexpr int HR = STRIDE == 1 ? TR + 2 : 2 * TR + 1, HC = STRIDE == 1 ? 18 : 33, HP = HR * HC;   // staged source patch
  constexpr int NYE = (TP * 8 + GNT - 1) / GNT, NXE = (HP * 8 + GNT - 1) / GNT;   // float4 per thread and tile
  __shared__ __attribute__((aligned(16))) float sY[2][TP * 32];
  __shared__ __attribute__((aligned(16))) float sX[2][HP * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int split = blockIdx.x;
  const int cog = blockIdx.y / a.cig, cig = blockIdx.y - cog * a.cig;
  const int co0 = cog * 32, ci0 = cig * 32;
  const int c4 = tid & 7;                                                    // the thread's four channels of every staged row

  float ka[4], kb[4], kc[4], ks[4], kh[4];
  const bool bnb = a.cA != nullptr, tfx = a.sc != nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = min(co0 + 4 * c4 + r, a.Cout - 1), ci = min(ci0 + 4 * c4 + r, a.Cin - 1);
    ka[r] = bnb ? a.cA[co] : 1.f; kb[r] = bnb ? a.cB[co] : 0.f; kc[r] = bnb ? a.cC[co] : 0.f;
    ks[r] = tfx ? a.sc[ci] : 1.f; kh[r] = tfx ? a.sh[ci] : 0.f;
  }
  hrf_f4 ry[NYE], rz[NYE], rx[NXE];
  bool oky[NYE], okx[NXE];
  // every load is UNCONDITIONAL from a clamped in-tensor address (padding pixels / channel groups past the tensor re-read row 0
  // and are zeroed when the tile is stored): `cond ? *p : 0` and `a && b ? p : q` become exec-masked branches with a wait of
  // their own per load - the tile's 11 loads then cost 11 dependent round trips (first form of this kernel: 52 us instead of 30)
  const bool cov = co0 + 4 * c4 < a.Cout, civ = ci0 + 4 * c4 < a.Cin;
  const int coff = cov ? co0 + 4 * c4 : 0, cioff = civ ? ci0 + 4 * c4 : 0;
  const float* yz = bnb ? a.yraw : a.dy;                     // (no BatchNorm backward: kb = 0, any finite value does)
  auto load_tile = [&](int tile) W3_INLINE {
    int t = tile;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY; const int b = t / a.tilesY;
    const int y0 = ty * TR, x0 = tx * 16;
#pragma unroll
    for (int e = 0; e < NYE; ++e) {
      const int f = tid + e * GNT;
      const int p = min(f >> 3, TP - 1);
      const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
      oky[e] = (gy < a.Ho) & (gx < a.Wo);
      const long row = oky[e] ? ((long)(b * a.Ho + gy) * a.Wo + gx) * a.ldD : 0;
      if (!RAGY) {
        ry[e] = hrf_ld4(a.dy + row + coff);
        rz[e] = hrf_ld4(yz + row + coff);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = min(coff + r, a.Cout - 1);
          ry[e][r] = a.dy[row + c];
          rz[e][r] = yz[row + c];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < NXE; ++e) {
      const int f = tid + e * GNT;
      const int p = min(f >> 3, HP - 1);
      const int py = p / HC, px = p - py * HC;
      const int gy = y0 * STRIDE - 1 + py, gx = x0 * STRIDE - 1 + px;
      okx[e] = ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
      const long row = okx[e] ? ((long)(b * a.H + gy) * a.W + gx) * a.ldX : 0;
      rx[e] = hrf_ld4(a.x + row + cioff);
    }
  };
  auto store_tile = [&](int buf) W3_INLINE {
#pragma unroll
    for (int e = 0; e < NYE; ++e) {
      const int f = tid + e * GNT;
      hrf_f4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float u = fmaf(ka[r], ry[e][r], fmaf(kb[r], rz[e][r], kc[r]));
        v[r] = (oky[e] & (co0 + 4 * c4 + r < a.Cout)) ? u : 0.f;
      }
      if (f < TP * 8) hrf_st4(&sY[buf][(f >> 3) * 32 + 4 * c4], v);
    }
    if (a.act == HRF_ACT_GELU) {                              // (uniform; no front-end convolution takes a GELU input)
#pragma unroll
      for (int e = 0; e < NXE; ++e) {
        const int f = tid + e * GNT;
        hrf_f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (okx[e] & (ci0 + 4 * c4 + r < a.Cin)) ? hrf_gelu(fmaf(rx[e][r], ks[r], kh[r])) : 0.f;
        if (f < HP * 8) hrf_st4(&sX[buf][(f >> 3) * 32 + 4 * c4], v);
      }
      return;
    }
    const float lo = a.act == HRF_ACT_RELU ? 0.f : -3.0e38f;
#pragma unroll
    for (int e = 0; e < NXE; ++e) {
      const int f = tid + e * GNT;
      hrf_f4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float u = fmaxf(fmaf(rx[e][r], ks[r], kh[r]), lo);
        v[r] = (okx[e] & (ci0 + 4 * c4 + r < a.Cin)) ? u : 0.f;      // zero padding applies AFTER the transform
      }
      if (f < HP * 8) hrf_st4(&sX[buf][(f >> 3) * 32 + 4 * c4], v);
    }
  };

  w3_f16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // the wave's three taps: tap = 3 wave + t -> (dy, dx) = (wave, t); source pixel of output pixel (r, c): stride 1: halo
  // (r + dy, c + dx), stride 2: patch (2 r + dy, 2 c + dx)
  int tile = split, buf = 0;
  if (tile < a.ntiles) { load_tile(tile); store_tile(0); }
  __syncthreads();
  for (; tile < a.ntiles; tile += a.splits) {
    const int nxt = tile + a.splits;
    if (W3_VARIANT != 3 && nxt < a.ntiles) load_tile(nxt);
    const float* ay = &sY[buf][h * 32 + j];
    const float* bx = &sX[buf][((STRIDE * 0 + wave) * HC + STRIDE * h) * 32 + j];
    // k-steps in batches of 8 (pixels 2 s + h: row s >> 3, column 2 (s & 7) + h): the 32 fragments of batch b+1 are fetched in
    // front of the 24 MFMAs of batch b (two register sets, order pinned by scheduling fences).  Left to the compiler, every
    // k-step waited for its own four ds_read_b32 in front of its three MFMAs: a round trip of ~130 cycles per 192 cycles of
    // matrix work with one to three waves per SIMD to cover it - 40 us where the matrix pipe needs 18.
    float fa[2][8], fb[2][8][3];
    auto fetch = [&](int bt, int set) W3_INLINE {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = 8 * bt + u;
        fa[set][u] = ay[s * 64];
        const int off = (STRIDE * (s >> 3) * HC + STRIDE * 2 * (s & 7)) * 32;
#pragma unroll
        for (int t = 0; t < 3; ++t) fb[set][u][t] = bx[off + t * 32];
      }
    };
    auto mma = [&](int set) W3_INLINE {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#if W3_VARIANT == 1
          acc[t][0] += fa[set][u] * fb[set][u][t];                // timing experiment: no MFMA
#else
          acc[t] = w3_mfma32(fa[set][u], fb[set][u][t], acc[t]);
#endif
        }
    };
    constexpr int NBT = TP / 16;                             // batches per tile (4 at stride 1, 2 at stride 2)
    fetch(0, 0);
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      W3_FENCE();
      if (bt + 1 < NBT) fetch(bt + 1, (bt + 1) & 1);
      W3_FENCE();
      mma(bt & 1);
      W3_FENCE();
    }
    if (W3_VARIANT != 3 && nxt < a.ntiles) store_tile(buf ^ 1);
    __syncthreads();
    if (W3_VARIANT != 3) buf ^= 1;
  }

  // ---- the block's sums: acc[t][r] = dW[co0 + (r & 3) + 8 (r >> 2) + 4 h][ci0 + j][tap 3 wave + t], 128-byte runs over ci
  const int ci = ci0 + j;
  if (ci < a.Cin) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float* pb = a.part + ((long)(split * 9 + 3 * wave + t) * a.Cout) * a.Cin + ci;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < a.Cout) pb[(long)co * a.Cin] = acc[t][r];
      }
    }
  }
}

// dw[co][ci][t] += sum_s part[s][t][co][ci]: a block owns 64 consecutive (co, ci) elements of one tap; its four waves take every
// fourth split (8 loads in flight per thread) and meet in LDS.  (wgrad3w_fold_kernel gives ONE thread all the splits of an element:
// fine for the neck's <= 16 splits, 30 us of dependent round trips for 128.)
__global__ __launch_bounds__(256) void wgrad3x_fold_kernel(const float* part, int splits, int Cout, int Cin, float* dw) {
  __shared__ float sRed[3][64];
  const long n = (long)Cout * Cin, total = 9 * n;
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long f = (long)blockIdx.x * 64 + l;                        // index into [t][co][ci]
  float acc = 0.f;
  if (f < total) {
    const float* p = part + f;
    int sp = g;
    for (; sp + 28 < splits; sp += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long)(sp + 4 * u) * total];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sp < splits; sp += 4) acc += p[(long)sp * total];
  }
  if (g > 0) sRed[g - 1][l] = acc;
  __syncthreads();
  if (g == 0 && f < total) {
    const int t = (int)(f / n);
    const long e = f - (long)t * n;
    dw[e * 9 + t] += (acc + sRed[0][l]) + (sRed[1][l] + sRed[2][l]);
  }
}

int g_w3x_blocks = 0, g_w3x_minpix = 16384;     // (7 680 output pixels, 256 -> 36 stride 2: 44 us here, 37 us pixel-major)

// pixel splits of a problem: ~512 blocks of three waves (1.5 waves per SIMD), at least two pixel tiles per block
int w3x_splits(int ntiles, int groups) {
  const int want = g_w3x_blocks > 0 ? g_w3x_blocks : 512;
  int s = want / (groups < 1 ? 1 : groups);
  if (s > ntiles / 2) s = ntiles / 2;
  return s < 1 ? 1 : s;
}

}  // namespace

// -> floats of scratch hrf_conv_bwd_weight_s wants for this problem; 0: the problem stays with hrf_conv_bwd_weight
long hrf_wgrad3x_scratch(int B, int H, int W, int Cin, int KH, int stride, int Cout, int has_bias, int tf_mode, bool dense_nhwc) {
  static const bool off = std::getenv("HRF_NO_WGRAD3X") != nullptr;
  if (off || KH != 3 || (stride != 1 && stride != 2) || !dense_nhwc || has_bias || Cin < 32 || (Cin & 3) != 0 || Cout < 1) return 0;
  if (tf_mode < HRF_TF_NONE || tf_mode > HRF_TF_AFFINE_GELU) return 0;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  if ((long)B * Ho * Wo < g_w3x_minpix) return 0;
  const int tr = stride == 1 ? 4 : 2;
  const int ntiles = B * hrf_cdiv(Ho, tr) * hrf_cdiv(Wo, 16);
  const int groups = hrf_cdiv(Cout, 32) * hrf_cdiv(Cin, 32);
  return (long)w3x_splits(ntiles, groups) * 9 * Cout * Cin;
}

int hrf_wgrad3x_launch(const float* dy, int ldD, const float* yraw, const float* cA, const float* cB, const float* cC,
                       const float* x, int ldX, int B, int H, int W, int Cin, int stride, int Cout,
                       int tf_mode, const float* tf_scale, const float* tf_shift, float* dw, float* scratch, void* stream) {
  W3xArgs a;
  a.dy = dy; a.ldD = ldD; a.yraw = yraw; a.cA = cA; a.cB = cB; a.cC = cC; a.x = x; a.ldX = ldX;
  a.sc = tf_mode != HRF_TF_NONE ? tf_scale : nullptr; a.sh = tf_shift;
  a.act = tf_mode == HRF_TF_AFFINE_RELU ? HRF_ACT_RELU : (tf_mode == HRF_TF_AFFINE_GELU ? HRF_ACT_GELU : HRF_ACT_NONE);
  a.part = scratch;
  a.B = B; a.H = H; a.W = W; a.Ho = (H + 2 - 3) / stride + 1; a.Wo = (W + 2 - 3) / stride + 1; a.Cin = Cin; a.Cout = Cout;
  const int tr = stride == 1 ? 4 : 2;
  a.tilesX = hrf_cdiv(a.Wo, 16); a.tilesY = hrf_cdiv(a.Ho, tr); a.ntiles = B * a.tilesX * a.tilesY;
  a.cig = hrf_cdiv(Cin, 32);
  const int groups = hrf_cdiv(Cout, 32) * a.cig;
  a.splits = w3x_splits(a.ntiles, groups);
  const dim3 grid(a.splits, groups);
  const bool ragy = (Cout & 3) != 0;
  if (stride == 1) {
    if (ragy) { HRF_LAUNCH((wgrad3x_kernel<1, true>), grid, dim3(GNT), 0, stream, a); }
    else { HRF_LAUNCH((wgrad3x_kernel<1, false>), grid, dim3(GNT), 0, stream, a); }
  } else if (ragy) { HRF_LAUNCH((wgrad3x_kernel<2, true>), grid, dim3(GNT), 0, stream, a); }
  else { HRF_LAUNCH((wgrad3x_kernel<2, false>), grid, dim3(GNT), 0, stream, a); }
  if (hrf_check_launch() != HRF_OK) return HRF_ERR_LAUNCH;
#if W3_VARIANT != 2
  HRF_LAUNCH(wgrad3x_fold_kernel, dim3(hrf_cdiv(9L * Cout * Cin, 64)), dim3(256), 0, stream, (const float*)scratch, a.splits, Cout, Cin, dw);
#endif
  return hrf_check_launch();
}

extern "C" __attribute__((visibility("hidden"))) int hrf_w3x_knob(int key, int value) {
  if (key == 0) { g_w3x_blocks = value; return HRF_OK; }
  if (key == 1) { g_w3x_minpix = value > 0 ? value : 16384; return HRF_OK; }      // (tests: small problems on the CPU emulator)
  return HRF_ERR_ARG;
}
