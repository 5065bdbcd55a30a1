// 3x3 / pad-1 convolution of the backbone's front end on TAP-MAJOR PACKED weights: forward, backward-data of the stride-1
// convolution and backward-data of the stride-2 convolution, fp32 MFMA 32x32x2 (gfx950).
//
// The stems, the Bottleneck conv2 of layer1 and transition1 (hrnet.py:341-358,417-459, resnet.py:263-302,
// hrfuser_hrformer_based.py:380-396) carry half of the network's FLOPs in launches of 2.3 GFLOP over 30 720 pixels:
// 120 pixels x 64 channels per CU, so what a launch costs is its fixed part.  conv3_engine.hip (round 1-5) gathered OIHW
// weights with a 36-byte stride, passed a 64 x 64 weight tile through LDS between TWO barriers per tap, fed
// v_mfma_f32_16x16x4 from ds_read_b32 (1.25 LDS reads per MFMA) and ran one 8-wave block per CU: 0.30 MFMA-busy.  Here
//   * the weights are re-packed once per step as wp[tap][n][k] (hrf_conv3x_pack, one launch for every front-end
//     convolution, both directions): a K step of a block is one contiguous 128-byte run per output channel and
//     backward-data is the SAME kernel on the transposed pack;
//   * v_mfma_f32_32x32x2_f32: D[pixel][channel] tiles of 32 x 32 with ONE accumulator chain per wave (64-cycle issue =
//     64-cycle dependent latency) and half the operand fetches per FLOP; the contraction index is permuted inside a group of
//     eight (lane half h of MFMA m takes k = 8g + 4h + m) so that EVERY operand fetch is a ds_read_b128: 8 LDS reads per 16 MFMAs;
//   * a block is 4 waves (one per SIMD) on a 4 x 16 pixel tile x 64 channels, 57 KB of LDS: TWO blocks per CU, 480 blocks
//     for 2 x 96 x 160 - one block's prologue / epilogue runs beside the other's K loop;
//   * the weight tiles (one tap x 32 input channels x 64 output channels = 8 KB) go through a 3-slot LDS ring with ONE
//     barrier per step, global loads three steps ahead; the operand fragments of step s+1 are fetched in front of the
//     MFMAs of step s;
//   * the halo row pitch is 0 mod 64 banks and the pixel pitch 68 floats; weight rows are stored at row slot
//     ((n & 15) << 1 | n >> 4) with the 16-byte chunk index XOR (n & 7): both fragment fetches are bank-conflict free for the
//     16-lane groups ds_read_b128 is served in;
//   * stride-2 backward-data: ONE block owns a tile of the SOURCE grid and walks the four output-parity classes over the same
//     staged halo (1 + 2 + 2 + 4 = 9 taps, as many steps as a stride-1 tile) with four epilogues - the round-5 form launched
//     one block per class (uneven: 1 ... 4 taps) and staged the halo four times.
// Contract as conv3_engine.hip (BatchNorm finalised on load / BatchNorm-backward combined on load, raw output + moments,
// bias / residual / act' epilogues); selected by the caller through hrf_conv_fwd_packed / hrf_conv_bwd_data_packed.
#include "hrf_common.h"
#include "hrf_lin.h"
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int XW = 16;                    // tile width in pixels
constexpr int XPP = 68;                   // halo pixel pitch (floats): 64 channels + 4 (granule index of pixel x = 17 x: distinct mod 16)
constexpr int XRP = 18 * XPP + 56;        // halo row pitch = 1280 floats = 0 mod 64 banks
constexpr int XNT = 256;                  // threads per block
constexpr int XFC = 256;                  // channels of an on-load BatchNorm the kernel can finalise

__device__ float g_zero4x[4] = {0.f, 0.f, 0.f, 0.f};

// timing experiments (tools/c3x_variants.sh): parts of the kernel compiled out, results WRONG - never defined in the product build
#ifndef X_VARIANT
#define X_VARIANT 0
#endif

#ifdef HRF_EMUL
#define X_SCHED_FENCE() ((void)0)
#define X_WAIT_LDS() ((void)0)
#define X_INLINE
struct hrf_f16 {
  float d[16];
  float& operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
};
// v_mfma_f32_32x32x2_f32 semantics: lane l supplies A[l & 31][l >> 5] and B[l >> 5][l & 31]; result register r of lane l =
// D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]; exact fp32 fmaf chain in k order.
inline hrf_f16 hrf_mfma32(float a, float b, hrf_f16 c) {
  char* buf = static_cast<char*>(hrf_emul::wave_buf());
  const int lane = hrf_emul::cur_lane;
  float ab[2] = {a, b};
  std::memcpy(buf + 16 * lane, ab, 8);
  hrf_emul::sync_wave();
  auto A = [&](int i, int k) { float v; std::memcpy(&v, buf + 16 * (k * 32 + i), 4); return v; };
  auto B = [&](int k, int j) { float v; std::memcpy(&v, buf + 16 * (k * 32 + j) + 4, 4); return v; };
  hrf_f16 d = c;
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
inline hrf_f4 xlds_ld4(const float* p) { return hrf_ld4(p); }
inline void xlds_st4(float* p, hrf_f4 v) { hrf_st4(p, v); }
#else
#define X_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)      // lgkmcnt(0) only
#define X_INLINE __attribute__((always_inline))
typedef float hrf_f16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ hrf_f16 hrf_mfma32(float a, float b, hrf_f16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ hrf_f4 xlds_ld4(const float* p) { return *reinterpret_cast<const hrf_f4*>(p); }   // ds_read_b128
__device__ __forceinline__ void xlds_st4(float* p, hrf_f4 v) { *reinterpret_cast<hrf_f4*>(p) = v; }
#endif

// the (halo slab, parity class, tap, 32-channel sub-slab) sequence of a block: walked twice, by the weight loader (three
// steps ahead) and by the consumer; every member is wave-uniform
template <int MODE>
struct XStep {
  static constexpr int HSW = MODE == 3 ? 32 : 64;            // channels of a staged halo slab
  int col, hs, cls, ti, sub, nsub, cin, nhs, ncol;
  __device__ __forceinline__ int nsub_of(int h) const { return (min(HSW, cin - HSW * h) + 31) >> 5; }
  __device__ __forceinline__ void init(int Cin, int ncols) {
    cin = Cin; nhs = (Cin + HSW - 1) / HSW; ncol = ncols; col = 0; hs = 0; cls = 0; ti = 0; sub = 0; nsub = nsub_of(0);
  }
  __device__ __forceinline__ bool done() const { return col >= ncol; }
  __device__ __forceinline__ int ntaps() const { return MODE == 2 ? (1 + (cls >> 1)) * (1 + (cls & 1)) : 9; }
  // last step of a GROUP = the steps that share one halo slab and one accumulator tile: a halo slab of a column group
  // (MODE 0 / 1), a parity class of a column group (MODE 2)
  __device__ __forceinline__ bool last_of_group() const { return sub + 1 >= nsub && ti + 1 >= ntaps(); }
  __device__ __forceinline__ bool last_group_of_col() const { return MODE == 2 ? cls == 3 : hs + 1 >= nhs; }
  // weight tap slice, source offset (rows, pixels) of the halo relative to the output pixel's halo position
  __device__ __forceinline__ void decode(int& wtap, int& dy, int& dx) const {
    if (MODE == 2) {
      const int cpy = cls >> 1, cpx = cls & 1, nsx = 1 + cpx;
      const int iy = ti >= nsx ? 1 : 0, ix = ti - iy * nsx;
      wtap = (cpy ? 2 * iy : 1) * 3 + (cpx ? 2 * ix : 1);
      dy = cpy ? 1 - iy : 0; dx = cpx ? 1 - ix : 0;         // tap ky = 2 iy reads source row y' + 1 - iy
    } else if (MODE == 3) {
      // stride-2 forward: the staged patch lives in four PARITY PLANES (row / column parity of the source pixel 2 r + ky,
      // 2 c + kx relative to the patch origin): plane (ky & 1, kx & 1) starts at row {0, 5, 10, 14}, tap (ky, kx) reads its
      // pixel (r + (ky >> 1), c + (kx >> 1)) - `dy` carries the plane's first row, so the fragment address keeps one form
      const int ky = ti >= 6 ? 2 : (ti >= 3 ? 1 : 0), kx = ti - 3 * ky;
      const int pl = (ky & 1) * 2 + (kx & 1);
      dy = (pl == 0 ? 0 : (pl == 1 ? 5 : (pl == 2 ? 10 : 14))) + (ky >> 1); dx = kx >> 1;
      wtap = ti;
    } else {
      dy = ti >= 6 ? 2 : (ti >= 3 ? 1 : 0); dx = ti - 3 * dy;
      wtap = MODE == 1 ? 8 - ti : ti;
    }
  }
  __device__ __forceinline__ void next() {
    if (++sub >= nsub) {
      sub = 0;
      if (++ti >= ntaps()) {
        ti = 0;
        if (MODE == 2) { if (++cls >= 4) { cls = 0; ++col; } }
        else if (++hs >= nhs) { hs = 0; ++col; }
        nsub = nsub_of(hs);
      }
    }
  }
};

// NCG = 32-channel groups per block (2: 4 x 16 pixels x 64 channels; 1: 8 x 16 pixels x 32 channels)
template <int MODE, int NCG>
__global__ __launch_bounds__(XNT, 2) void conv3x_kernel(HrfGroup<C3xArgs> grp) {
  const C3xArgs& a = grp.sel();
  constexpr bool FWD = MODE == 0 || MODE == 3;                 // forward contract (transform on load, bias / residual, moments of y)
  constexpr int RP = 4 / NCG, TH = 2 * RP;                       // row pairs (= waves per channel group), tile height
  // staged source patch: rows x columns, origin relative to the tile origin; MODE 3: the 9 x 33 patch of the stride-2 forward
  constexpr int SH = MODE == 2 ? TH + 1 : (MODE == 3 ? 2 * TH + 1 : TH + 2), SW = MODE == 2 ? XW + 1 : (MODE == 3 ? 2 * XW + 1 : XW + 2);
  constexpr int ORG = MODE == 2 ? 0 : -1, SM = MODE == 3 ? 2 : 1;
  constexpr int HSW = XStep<MODE>::HSW, C4N = HSW / 4;           // channels / float4 per staged pixel
  constexpr int PPX = MODE == 3 ? 36 : XPP;                      // pixel pitch: 9 (17) granules - odd, so 16 pixels hit 16 bank groups
  constexpr int RPX = MODE == 3 ? 640 : XRP;                     // row pitch = 0 mod 64 banks
  constexpr int LROWS = MODE == 3 ? 18 : SH;                     // LDS rows (MODE 3: 5 + 5 + 4 + 4 rows of the four parity planes)
  constexpr int NPX = SH * SW;                                   // staged source pixels
  constexpr int NB = NCG * 32;                                   // output channels per block
  constexpr int WT = NB * 32;                                    // floats of a weight tile
  constexpr int NHE = (NPX * C4N + XNT - 1) / XNT;               // halo float4 per thread
  __shared__ __attribute__((aligned(16))) float sIn[LROWS * RPX];
  __shared__ __attribute__((aligned(16))) float sW[3 * WT];
  __shared__ __attribute__((aligned(16))) float sFin[3 * XFC];
  __shared__ float sStat[RP * 2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int rp = wave % RP, cg = wave / RP;
  // XCD-aware tile order: consecutive block ids go to different XCDs, so XCD x walks the contiguous tile range [x per, (x+1) per)
  int t = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if (t >= a.ntiles) return;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY; const int b = t / a.tilesY;
  const int y0 = ty * TH, x0 = tx * XW;                          // tile origin (MODE 2: on the source grid)
  // column groups (64 output channels each): a block whose input fits ONE halo slab walks all of them over the staged halo
  // (the transitions' backward: 18 / 36 -> 256 channels); otherwise one group per blockIdx.y
  const int ncolb = a.cols_per_block;
  const int ncol0 = blockIdx.y * ncolb;
  const bool writer = blockIdx.x == 0 && blockIdx.y == 0;

  // ---- staging maps (the thread's four channels are the same for every halo element it stages)
  const int c4 = tid & (C4N - 1);
  const float* hsrc[NHE]; int hdst[NHE]; bool hin[NHE];
#pragma unroll
  for (int e = 0; e < NHE; ++e) {
    const int f = tid + e * XNT;
    const int pix = min(f / C4N, NPX - 1);
    const int py = pix / SW, px = pix - py * SW;
    const int gy = y0 * SM + ORG + py, gx = x0 * SM + ORG + px;
    hin[e] = f < NPX * C4N && (unsigned)gy < (unsigned)a.Hs && (unsigned)gx < (unsigned)a.Ws;
    hsrc[e] = a.in + ((long)(b * a.Hs + gy) * a.Ws + gx) * a.ldIn;
    int lrow = py, lcol = px;
    if (MODE == 3) {                                             // parity planes (see XStep::decode)
      const int pl = (py & 1) * 2 + (px & 1);
      lrow = (pl == 0 ? 0 : (pl == 1 ? 5 : (pl == 2 ? 10 : 14))) + (py >> 1); lcol = px >> 1;
    }
    hdst[e] = f < NPX * C4N ? lrow * RPX + lcol * PPX + 4 * c4 : -1;
  }
  hrf_f4 hv[NHE], hv2[NHE];
  const bool two = !FWD && a.in2 != nullptr;
  const long d2 = two ? a.in2 - a.in : 0;
  auto load_halo = [&](int hs) X_INLINE {
    const int c = hs * HSW + 4 * c4;
    if ((a.Cin & 3) == 0) {
      const bool cv = c < a.Cin;
#pragma unroll
      for (int e = 0; e < NHE; ++e) {
        const float* p = (hin[e] && cv) ? hsrc[e] + c : g_zero4x;
        hv[e] = hrf_ld4(p);
        if (!FWD) hv2[e] = hrf_ld4(two && hin[e] && cv ? p + d2 : g_zero4x);
      }
    } else {                                                     // ragged channel count (18, 36 + 2 ...): element loads
#pragma unroll
      for (int e = 0; e < NHE; ++e) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = hin[e] && c + r < a.Cin;
          const float* p = ok ? hsrc[e] + c + r : g_zero4x;
          hv[e][r] = *p;
          if (!FWD) hv2[e][r] = *(two && ok ? p + d2 : g_zero4x);
        }
      }
    }
  };
  const float* t0p = sFin;                                       // per-channel transform coefficients: always through LDS
  const float* t1p = sFin + XFC;
  const float* t2p = sFin + 2 * XFC;
  auto store_halo = [&](int hs) X_INLINE {
    const int c = hs * HSW + 4 * c4;
    float p0[4], p1[4], p2[4];
    const bool tf = FWD ? a.tf_mode != HRF_TF_NONE : a.t0 != nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cc = min(c + r, a.Cin - 1);
      p0[r] = tf ? t0p[cc] : 1.f; p1[r] = tf ? t1p[cc] : 0.f; p2[r] = (tf && !FWD) ? t2p[cc] : 0.f;
    }
    if (FWD && a.tf_mode == HRF_TF_AFFINE_GELU) {
#pragma unroll
      for (int e = 0; e < NHE; ++e) {
        hrf_f4 v = hv[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (hin[e] && c + r < a.Cin) ? hrf_gelu(fmaf(v[r], p0[r], p1[r])) : 0.f;
        if (hdst[e] >= 0) xlds_st4(sIn + hdst[e], v);
      }
      return;
    }
    const bool relu = FWD && a.tf_mode == HRF_TF_AFFINE_RELU;
#pragma unroll
    for (int e = 0; e < NHE; ++e) {
      hrf_f4 v = hv[e];
      if (tf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float u = FWD ? fmaf(v[r], p0[r], p1[r]) : fmaf(p0[r], v[r], fmaf(p1[r], hv2[e][r], p2[r]));
          if (relu) u = fmaxf(u, 0.f);
          v[r] = (hin[e] && c + r < a.Cin) ? u : 0.f;            // zero padding applies AFTER the transform
        }
      }
      if (hdst[e] >= 0) xlds_st4(sIn + hdst[e], v);
    }
  };

  // weight tile of a step: thread -> rows (tid >> 3) + 32 e, 16-byte chunk tid & 7
  const int wn = tid >> 3, wc = tid & 7;
  const long wtap_stride = (long)a.Np * a.Kp;
  const float* wrow = a.wp + (long)(ncol0 * NB + wn) * a.Kp + 4 * wc;
  const int wdst = ((((wn & 15) << 1) | (wn >> 4)) << 5) + 4 * (wc ^ (wn & 7));
  hrf_f4 wreg[NCG];
  auto load_w = [&](const XStep<MODE>& s) X_INLINE {
    int wtap, dy, dx;
    s.decode(wtap, dy, dx);
    const float* p = wrow + wtap * wtap_stride + (long)s.col * NB * a.Kp + s.hs * HSW + s.sub * 32;
#pragma unroll
    for (int e = 0; e < NCG; ++e) wreg[e] = hrf_ld4(p + (long)e * 32 * a.Kp);
  };
  auto store_w = [&](int slot) X_INLINE {
#pragma unroll
    for (int e = 0; e < NCG; ++e) xlds_st4(sW + slot * WT + e * 1024 + wdst, wreg[e]);
  };

  // ---- prologue: global loads first, the BatchNorm finalisation while they fly
  XStep<MODE> cs, ls;
  cs.init(a.Cin, min(ncolb, (a.Cout + NB - 1) / NB - ncol0));
  ls = cs;
  load_halo(0);
  load_w(ls); ls.next();
  if (FWD) {
    if (a.fin.stats != nullptr) { if (X_VARIANT != 6) hrf_bn_fin_onload(a.fin, sFin, sFin + XFC, tid, XNT, writer); }
    else if (a.tf_mode != HRF_TF_NONE)
      for (int c = tid; c < a.Cin; c += XNT) { sFin[c] = a.t0[c]; sFin[XFC + c] = a.t1[c]; }
  } else if (a.bfin.gstats != nullptr) {
    hrf_bn_bfin_onload(a.bfin, sFin, sFin + XFC, sFin + 2 * XFC, tid, XNT, writer);
  } else if (a.t0 != nullptr) {
    for (int c = tid; c < a.Cin; c += XNT) { sFin[c] = a.t0[c]; sFin[XFC + c] = a.t1[c]; sFin[2 * XFC + c] = a.t2[c]; }
  }
  store_w(0);
  bool have = false;
  if (!ls.done()) { load_w(ls); ls.next(); have = true; }
  __syncthreads();                                               // sFin complete
  store_halo(0);
  if (have) { store_w(1); have = false; }
  if (!ls.done()) { load_w(ls); ls.next(); have = true; }       // tile 2 stays in registers until the top of step 0
  __syncthreads();

#if !defined(HRF_EMUL) && (X_VARIANT == 8)
  // the two blocks of a CU start in lock-step; the block in the odd wave slot of each SIMD gets the higher priority
  if (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1) __builtin_amdgcn_s_setprio(2);
#endif
  hrf_f16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float s1 = 0.f, s2 = 0.f;                                      // moments of channel n0 + 32 cg + j over this lane's pixels

  // fragment addressing: A = halo (pixel i = lane & 31 of the wave's 2 x 16 pixels, k half h), B = weight row j of group cg
  const float* abase = sIn + (2 * rp + (j >> 4)) * RPX + (j & 15) * PPX + 4 * h;
  const float* bbase = sW + cg * 1024 + ((((j & 15) << 1) | (j >> 4)) << 5);
  int bsw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bsw[g] = 4 * ((2 * g + h) ^ (j & 7));
  hrf_f4 fa[2][4], fb[2][4];
  auto read_frags = [&](const XStep<MODE>& s, int slot, int set) X_INLINE {
    int wtap, dy, dx;
    s.decode(wtap, dy, dx);
    const float* ap = abase + dy * RPX + dx * PPX + s.sub * 32;
    const float* bp = bbase + slot * WT;
#pragma unroll
    for (int g = 0; g < 4; ++g) { fa[set][g] = xlds_ld4(ap + 8 * g); fb[set][g] = xlds_ld4(bp + bsw[g]); }
  };
  auto mma = [&](int set) X_INLINE {
    if (X_VARIANT == 1) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = hrf_mfma32(fa[set][g][m], fb[set][g][m], acc);
  };

  // ---- epilogue of one accumulator tile: acc[r] = out(pixel p = (r & 3) + 8 ((r >> 2) & 1) + 4 h of row r >> 3, channel j of
  // column group `col`).  (Fetching its per-element operand - residual / raw producer output / accumulated gradient - in front
  // of the K loop measured SLOWER: vmcnt retires in order, so the first weight tile of the ring waited for those loads too:
  // 64 -> 64 backward 32.8 -> 35.5 us, stride-2 backward 40.6 -> 53.5.)
  auto epilogue = [&](int cls, int col) X_INLINE {
    if (X_VARIANT == 3) { if (acc[0] == 12345.f) a.out[0] = acc[1] + acc[15]; return; }
    const int ch = (ncol0 + col) * NB + cg * 32 + j;
    const bool chv = ch < a.Cout;
    const int chc = chv ? ch : 0;
    const int cpy = cls >> 1, cpx = cls & 1;
    float bv = 0.f, esc = 1.f, esh = 0.f;
    if (FWD) { if (a.bias != nullptr) bv = a.bias[chc]; }
    else if (a.epi == 1) { esc = a.esc[chc]; esh = a.esh[chc]; }
    long prow[16]; bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int yy = y0 + 2 * rp + (r >> 3), xx = x0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;
      const int y = MODE == 2 ? 2 * yy + cpy : yy, x = MODE == 2 ? 2 * xx + cpx : xx;
      ok[r] = chv && y < a.H && x < a.W;
      prow[r] = (long)(b * a.H + y) * a.W + x;
    }
    // (one uniform branch per epilogue KIND around whole loops - the same code with the kind tested per element inside one loop
    // measured 10 us slower on the stride-2 backward, whose blocks run four epilogues)
    if (FWD) {
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rv[r] = a.res != nullptr ? *(ok[r] ? a.res + prow[r] * a.ldR + ch : g_zero4x) : 0.f;
        if (a.res2 != nullptr) rv[r] += *(ok[r] ? a.res2 + prow[r] * a.ldR + ch : g_zero4x);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[r] + bv + rv[r];
        if (ok[r]) { a.out[prow[r] * a.ldOut + a.ooff + ch] = v; s1 += v; s2 = fmaf(v, v, s2); }
      }
    } else if (a.epi == 1) {
      float xr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xr[r] = *(ok[r] ? a.xraw + prow[r] * a.ldXr + ch : g_zero4x);
      if (a.act == HRF_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[r] * hrf_gelu_grad(fmaf(xr[r], esc, esh));
          if (ok[r]) { a.out[prow[r] * a.ldOut + ch] = v; s1 += v; s2 = fmaf(v, xr[r], s2); }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float u = fmaf(xr[r], esc, esh);
          const float v = a.act == HRF_ACT_RELU ? (u > 0.f ? acc[r] : 0.f) : acc[r];
          if (ok[r]) { a.out[prow[r] * a.ldOut + ch] = v; s1 += v; s2 = fmaf(v, xr[r], s2); }
        }
      }
    } else {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[r] = a.accumulate ? *(ok[r] ? a.out + prow[r] * a.ldOut + ch : g_zero4x) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (ok[r]) a.out[prow[r] * a.ldOut + ch] = pv[r] + acc[r];
    }
  };
  // moments of a finished column group: lanes -> waves (LDS) -> one atomic per channel and block
  auto flush_stats = [&](int col) X_INLINE {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) { sStat[(rp * 2 + 0) * NB + cg * 32 + j] = s1; sStat[(rp * 2 + 1) * NB + cg * 32 + j] = s2; }
    __syncthreads();
    if (tid < 2 * NB) {
      const int which = tid / NB, cidx = tid - which * NB;
      const int c = (ncol0 + col) * NB + cidx;
      if (c < a.Cout) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < RP; ++w) sm += sStat[(w * 2 + which) * NB + cidx];
        double* st = a.stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * a.Cout;
        hrf_atomic_add(&st[which * a.Cout + c], (double)sm);
      }
    }
    s1 = 0.f; s2 = 0.f;
    __syncthreads();                                             // sStat is reused by the next column group
  };

  // ---- K loop: one barrier per step; the fragments of step s+1 are fetched in front of the MFMAs of step s.
  // Two loop levels: the inner one holds nothing but the steady state (weight ring, fragment fetch, 16 MFMAs, barrier) so that
  // the waitcnt pass sees no halo load / epilogue state at its header; group boundaries (a new halo slab, or the next parity
  // class of MODE 2) are handled between two runs of it.  (Explicit LDS wait in front of it: with nothing pending at the loop
  // header on either path the pass does not drain the prefetched fragments of step s+1 in front of the MFMAs of step s; the
  // prefetch itself is unconditional for the same reason - behind the last step of a group it fetches stale but addressable
  // LDS, replaced at the boundary.)
  int slot = 0;
  auto step = [&](int set) X_INLINE {
    if (X_VARIANT == 7) { cs.next(); mma(set); return; }
    const int s2slot = slot >= 1 ? slot - 1 : 2;                 // (slot + 2) % 3
    if (X_VARIANT != 4) {
      if (have) { store_w(s2slot); have = false; }
      if (!ls.done()) { load_w(ls); ls.next(); have = true; }
    }
    cs.next();
    const int nslot = slot == 2 ? 0 : slot + 1;
    X_SCHED_FENCE();
    read_frags(cs, nslot, set ^ 1);
    X_SCHED_FENCE();
    mma(set);
    X_SCHED_FENCE();
    slot = nslot;
    if (X_VARIANT != 5) __syncthreads();
  };
  const bool want_stats = a.stats != nullptr && (FWD || a.epi == 1);      // (uniform)
  for (; X_VARIANT != 2;) {
    read_frags(cs, slot, 0);
    X_WAIT_LDS();
    const int cls = cs.cls, col = cs.col;
    const bool col_end = cs.last_group_of_col();
    for (;;) {
      bool ge = cs.last_of_group();
      step(0);
      if (ge) break;
      ge = cs.last_of_group();
      step(1);
      if (ge) break;
    }
    if (MODE == 2 || col_end) {
      epilogue(cls, col);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (col_end && want_stats) flush_stats(col);
    }
    if (cs.done()) break;
    if (MODE != 2 && cs.nhs > 1) {
      load_halo(cs.hs);                                          // every wave is behind the barrier of the slab's last step
      store_halo(cs.hs);
      __syncthreads();
    }
  }
  if (X_VARIANT == 2) { epilogue(0, 0); if (want_stats) flush_stats(0); }
}

struct PackJobs {
  hrf_conv3x_pack_job_t job[16];
  long first[17];                              // first output element of job i in the concatenated index space
};

// wp[tap][n][k] (n < Np = N rounded up to 64, k < Kp = K rounded up to 32, zero outside the tensor): dir 0: w[n][k][tap]; dir 1: w[k][n][tap]
__global__ __launch_bounds__(256) void conv3x_pack_kernel(PackJobs pj, int njobs) {
  const long total = pj.first[njobs];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    int ji = 0;
    while (ji + 1 < njobs && e >= pj.first[ji + 1]) ++ji;
    const hrf_conv3x_pack_job_t& jb = pj.job[ji];
    const int N = jb.dir ? jb.Cin : jb.Cout, K = jb.dir ? jb.Cout : jb.Cin;
    const int Np = (N + 63) & ~63, Kp = (K + 31) & ~31;
    const long o = e - pj.first[ji];
    const int k = (int)(o % Kp);
    const long r = o / Kp;
    const int n = (int)(r % Np), tap = (int)(r / Np);
    float v = 0.f;
    if (n < N && k < K) v = jb.dir ? jb.w[((long)k * jb.Cin + n) * 9 + tap] : jb.w[((long)n * jb.Cin + k) * 9 + tap];
    jb.wp[o] = v;
  }
}

template <int MODE>
int conv3x_launch(C3xArgs a, void* stream) {
  if (a.B <= 0 || a.H <= 0 || a.W <= 0) return HRF_OK;
  if (MODE == 0 || MODE == 1) { a.Hs = a.H; a.Ws = a.W; }
  // tiles: of the output grid, except MODE 2 (tiles of the SOURCE grid, four output parity classes each)
  a.tilesX = hrf_cdiv(MODE == 2 ? a.Ws : a.W, XW); a.tilesY = hrf_cdiv(MODE == 2 ? a.Hs : a.H, 4);
  a.ntiles = a.tilesX * a.tilesY * a.B;
  a.per_xcd = hrf_cdiv(a.ntiles, 8);
  const int ncols = hrf_cdiv(a.Cout, 64);
  // one halo slab and enough tiles to fill the chip twice over: the block walks every column group over its staged halo
  // (2 x 16 x 24 pixels, 36 -> 256: 16 blocks x 4 groups 115 us, 64 blocks 20 us)
  a.cols_per_block = (MODE != 3 && a.Cin <= 64 && a.ntiles >= 448) ? ncols : 1;
  const dim3 grid(a.per_xcd * 8, hrf_cdiv(ncols, a.cols_per_block));
  HRF_LAUNCH_G((conv3x_kernel<MODE, 2>), grid, dim3(XNT), 0, stream, a);
  return hrf_check_launch();
}

}  // namespace

int hrf_conv3x_fwd_launch(const C3xArgs& a, void* stream) { return conv3x_launch<0>(a, stream); }
int hrf_conv3x_bwd_data_launch(const C3xArgs& a, void* stream) { return conv3x_launch<1>(a, stream); }
int hrf_conv3xs2_bwd_data_launch(const C3xArgs& a, void* stream) { return conv3x_launch<2>(a, stream); }
int hrf_conv3xs2_fwd_launch(const C3xArgs& a, void* stream) { return conv3x_launch<3>(a, stream); }

extern "C" long hrf_conv3x_pack_size(int Cout, int Cin, int dir) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const int N = dir ? Cin : Cout, K = dir ? Cout : Cin;
  return 9L * ((N + 63) & ~63) * ((K + 31) & ~31);      // rows in blocks of 64 output channels, zero rows / columns past the tensor
}

extern "C" int hrf_conv3x_pack(const hrf_conv3x_pack_job_t* jobs, int n, void* stream) {
  if (n < 0 || (n > 0 && jobs == nullptr)) return HRF_ERR_ARG;
  for (int i0 = 0; i0 < n; i0 += 16) {
    PackJobs pj;
    const int m = n - i0 < 16 ? n - i0 : 16;
    pj.first[0] = 0;
    for (int i = 0; i < m; ++i) {
      const hrf_conv3x_pack_job_t& jb = jobs[i0 + i];
      if (jb.w == nullptr || jb.wp == nullptr || jb.Cout <= 0 || jb.Cin <= 0) return HRF_ERR_ARG;
      pj.job[i] = jb;
      pj.first[i + 1] = pj.first[i] + hrf_conv3x_pack_size(jb.Cout, jb.Cin, jb.dir);
    }
    for (int i = m; i < 16; ++i) { pj.job[i] = pj.job[0]; pj.first[i + 1] = pj.first[m]; }
    const long total = pj.first[m];
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    HRF_LAUNCH(conv3x_pack_kernel, dim3(grid), dim3(256), 0, stream, pj, m);
  }
  return hrf_check_launch();
}
