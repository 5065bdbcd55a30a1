// Wide 3x3 / stride-1 / pad-1 convolution on TAP-MAJOR PACKED weights, fp32 MFMA (gfx950).
//
// The HRFPN output convolutions (256 -> 256 at 96x160 and its average-pooled pyramid, hrfpn.py:60-70,92-100) are the
// only MFMA-bound launches of the detector's front half: 36 GFLOP in one launch.  conv3_engine.hip was shaped for the
// backbone's 64-channel convolutions (one 64-channel slab, 64 output channels per block); at 256 channels it re-stages
// the input halo once per 64-output-channel block column, gathers OIHW weights with a 36-byte stride and issues 1.25 LDS
// reads per MFMA.  This engine instead
//   * takes the weights re-packed once per step as wp[tap][n][k] (hrf_conv3_pack): every K step of a block is ONE
//     contiguous 128-byte run per output channel, and backward-data is the SAME kernel on the flipped/transposed pack;
//   * lets one block own an 8x16 pixel tile x up to 256 output channels: the halo of a 32-channel slab is staged once
//     for all of them (wave = 4 pixel rows x 64 channels = 16 MFMA tiles, 64 accumulator registers);
//   * permutes the contraction index inside a 16-channel chunk (MFMA m of 4 takes k = 4q + m) so that a lane's four
//     K values are contiguous in LDS: every operand fetch is a ds_read_b128 - 8 LDS reads per 64 MFMAs;
//   * keeps a 3-slot ring of weight tiles in LDS (137 KB of the 160 KB with the halo): global loads run two steps
//     ahead and the operand fragments of the next chunk are fetched while the current one is on the matrix cores.
// Output fragments are produced transposed (D[row = channel][col = pixel]) so that a lane holds four consecutive
// channels of one pixel: the epilogue (bias, += for gradient accumulation) is one 16-byte load/store per tile.
#include "hrf_common.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, NPIX = IH * IW;   // 180 halo pixels
constexpr int KS = 32;        // channels per K slab
constexpr int LP = KS + 4;    // LDS pitch of the halo (floats): 16-byte aligned rows
constexpr int WP = KS;        // weight tile: no padding, the 16-byte chunk index is XOR-swizzled with (row & 7) instead
constexpr int NTHR = 512;

__device__ float g_zero4w[4] = {0.f, 0.f, 0.f, 0.f};

struct C3wArgs {
  const float* x; int ldX;            // [B*H*W][ldX], K valid channels per row
  const float* wp;                    // [9][N][K]
  const float* bias;                  // [N] or null
  float* y; int ldY; int accumulate;  // [B*H*W][ldY], N channels
  int B, H, W, K, N;
  int tilesX, tilesY;
};

#ifdef HRF_EMUL
#define HRF_SCHED_FENCE() ((void)0)
#define HRF_WAIT_LDS() ((void)0)
#else
#define HRF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define HRF_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)      // lgkmcnt(0) only
#endif

__device__ __forceinline__ hrf_f4 lds_ld4(const float* p) {
#ifdef HRF_EMUL
  return hrf_ld4(p);
#else
  return *reinterpret_cast<const hrf_f4*>(p);      // ds_read_b128 (16-byte aligned by construction)
#endif
}
__device__ __forceinline__ void lds_st4(float* p, hrf_f4 v) {
#ifdef HRF_EMUL
  hrf_st4(p, v);
#else
  *reinterpret_cast<hrf_f4*>(p) = v;
#endif
}

// WN = 64-channel groups per block (1, 2 or 4).  8 waves: wave -> (channel group = wave % WN, row group = wave / WN);
// each wave owns WN consecutive pixel rows x 64 channels.
//
// Schedule of a step s (one tap x 32 input channels = two 16-channel chunks), steady state:
//   top      weights of step s+2 (global loads issued one step ago) -> LDS ring slot (s+2)%3; global loads of step s+3
//   chunk 0  fragments are ALREADY in registers (fetched during step s-1); half of its 16*WN MFMAs, the chunk 1
//            fragment reads, the other half
//   chunk 1  half of its MFMAs, the chunk 0 fragment reads of step s+1 (ring slot (s+1)%3 is complete since the
//            previous barrier), the other half
//   barrier
// so no LDS read is waited on with an idle matrix core except at the 32-channel slab boundaries (every 9th step),
// where the single halo buffer is replaced.  Measured on MI355X, 256->256 at 2x96x160: weights double-buffered with
// fragment reads at the head of each chunk 334 us; MFMAs alone 288 us; everything but the MFMAs 121 us.
template <int WN>
__global__ __launch_bounds__(NTHR) void conv3w_kernel(C3wArgs a) {
  constexpr int NB = WN * 64;                 // output channels per block
  constexpr int NHV = (NPIX * (KS / 4) + NTHR - 1) / NTHR;     // halo float4 per thread (3)
  HRF_DYN_SMEM(float, smem);
  float* sIn = smem;                          // [NPIX * LP]
  float* sB = smem + NPIX * LP;               // [3][NB * WP], chunk c of row n stored at chunk c ^ (n & 7)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int chg = wave % WN, rg = wave / WN;
  int t = blockIdx.x;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY; const int b = t / a.tilesY;
  const int y0 = ty * TH, x0 = tx * TW;
  const int n0 = blockIdx.y * NB;
  const bool wave_on = n0 + chg * 64 < a.N;

  hrf_f4 acc[WN][4];
#pragma unroll
  for (int rr = 0; rr < WN; ++rr)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[rr][tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  // ---- staging maps
  const float* hsrc[NHV]; int hstep[NHV], hdst[NHV];   // padding lanes read the zero block with step 0: branch-free
#pragma unroll
  for (int e = 0; e < NHV; ++e) {
    const int f = tid + e * NTHR;
    const int pix = min(f >> 3, NPIX - 1), j = f & 7;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = f < NPIX * 8 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    hsrc[e] = ok ? a.x + ((long)(b * a.H + gy) * a.W + gx) * a.ldX + 4 * j : g_zero4w;
    hstep[e] = ok ? KS : 0;
    hdst[e] = f < NPIX * 8 ? pix * LP + 4 * j : -1;
  }
  const int wn = tid >> 3, wj = tid & 7;      // weight float4 e: row n = wn + 64*e, channels 4*wj..
  hrf_f4 hpre[NHV], wpre[WN];
  auto load_halo = [&](int slab) {
#pragma unroll
    for (int e = 0; e < NHV; ++e) hpre[e] = hrf_ld4(hsrc[e] + slab * hstep[e]);
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int e = 0; e < NHV; ++e)
      if (hdst[e] >= 0) lds_st4(sIn + hdst[e], hpre[e]);
  };
  // all step bookkeeping is wave-uniform and incremental (no division in the loop): a step is (tap, slab), its
  // weight tile sits in ring slot `ring`, its halo offset is hoff = (dy * IW + dx) * LP
  const float* wrow[WN];                      // this thread's weight rows at (tap 0, slab 0)
#pragma unroll
  for (int e = 0; e < WN; ++e) wrow[e] = a.wp + (long)min(n0 + wn + 64 * e, a.N - 1) * a.K + 4 * wj;   // rows past N feed switched-off waves
  const long wtap = (long)a.N * a.K;
  auto load_w = [&](int tap, int slab) {
#pragma unroll
    for (int e = 0; e < WN; ++e) wpre[e] = hrf_ld4(wrow[e] + tap * wtap + slab * KS);
  };
  auto store_w = [&](int ring) {
    float* dst = sB + ring * (NB * WP);
#pragma unroll
    for (int e = 0; e < WN; ++e) lds_st4(dst + (wn + 64 * e) * WP + 4 * (wj ^ (wn & 7)), wpre[e]);
  };
  hrf_f4 fa[2][WN], fb[2][4];
  const float* abase = sIn + (rg * WN * IW + i) * LP + 4 * q;
  const float* bbase = sB + (chg * 64 + i) * WP;
  const int bsw[2] = {4 * (q ^ (i & 7)), 4 * ((q + 4) ^ (i & 7))};        // swizzled chunk of c16 = 0 / 1
  auto read_frags = [&](int hoff, int ring, int c16, int slot) {
    const float* ap = abase + hoff + c16 * 16;
    const float* bp = bbase + ring * (NB * WP) + bsw[c16];
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) fa[slot][rr] = lds_ld4(ap + rr * IW * LP);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) fb[slot][tt] = lds_ld4(bp + tt * 16 * WP);
  };
  auto mma = [&](int slot, int half) {        // D[row = channel][col = pixel]: the weight fragment is the row operand
#pragma unroll
    for (int m = 2 * half; m < 2 * half + 2; ++m)
#pragma unroll
      for (int rr = 0; rr < WN; ++rr)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[rr][tt] = hrf_mfma16(fb[slot][tt][m], fa[slot][rr][m], acc[rr][tt]);
  };

  const int nslab = a.K / KS, S = nslab * 9;
  load_halo(0);
  load_w(0, 0);
  store_halo();
  store_w(0);
  load_w(1, 0);                                // S >= 9
  store_w(1);
  load_w(2, 0);                                // stays in registers until the top of step 0
  __syncthreads();
  // (explicit wait: with nothing pending at the loop header on either path, the waitcnt pass does not have to
  // drain the chunk-1 reads in front of the chunk-0 MFMAs)
  if (wave_on) { read_frags(0, 0, 0, 0); HRF_WAIT_LDS(); }

  int tap = 0, slab = 0, hoff = 0;             // step s
  int dxn = 1, dyn = 0;                        // tap coordinates of step s+1
  int tap3 = 3, slab3 = 0;                     // step s+3 (the weight tile fetched during step s)
  int r0 = 0, r1 = 1, r2 = 2;                  // ring slots of steps s, s+1, s+2
  for (int s = 0; s < S; ++s) {
    if (s + 2 < S) store_w(r2);
    if (s + 3 < S) load_w(tap3, slab3);
    if (tap == 0 && slab + 1 < nslab) load_halo(slab + 1);
    const bool boundary = tap == 8;            // the next step reads a new halo slab
    const int hoffn = (dyn * IW + dxn) * LP;
    if (wave_on) {
      // The fragments of the NEXT chunk are fetched in the middle of the current chunk's MFMAs: LLVM's waitcnt
      // pass drains ALL outstanding LDS reads (lgkmcnt(0)) in front of the first MFMA that needs one of them, so a
      // read must be half a chunk (>= 1000 cycles) old by then.  The scheduling fences pin the order written here
      // (the scheduler otherwise sinks the reads behind the MFMAs and waits with an idle matrix core).
      HRF_SCHED_FENCE();
      mma(0, 0);
      HRF_SCHED_FENCE();
      read_frags(hoff, r0, 1, 1);
      HRF_SCHED_FENCE();
      mma(0, 1);
      mma(1, 0);
      HRF_SCHED_FENCE();
      read_frags(hoffn, r1, 0, 0);              // unconditional; re-fetched below at a slab boundary
      HRF_SCHED_FENCE();
      mma(1, 1);
      HRF_SCHED_FENCE();
    }
    __syncthreads();
    if (boundary && s + 1 < S) {
      store_halo();
      __syncthreads();
      if (wave_on) { read_frags(hoffn, r1, 0, 0); HRF_WAIT_LDS(); }
    }
    tap = boundary ? 0 : tap + 1; slab += boundary ? 1 : 0; hoff = hoffn;
    if (++dxn == 3) { dxn = 0; if (++dyn == 3) dyn = 0; }
    if (++tap3 == 9) { tap3 = 0; ++slab3; }
    const int rt = r0; r0 = r1; r1 = r2; r2 = rt;
  }

  // ---- epilogue: acc[rr][tt][r] = out(pixel (row rg*WN + rr, col i), channel chg*64 + tt*16 + 4q + r)
  if (!wave_on) return;
  const int x = x0 + i;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int ch = n0 + chg * 64 + tt * 16 + 4 * q;
    hrf_f4 bv = hrf_f4{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bv = hrf_ld4(a.bias + ch);
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) {
      const int y = y0 + rg * WN + rr;
      if (y < a.H && x < a.W) {
        float* o = a.y + ((long)(b * a.H + y) * a.W + x) * a.ldY + ch;
        hrf_f4 v = acc[rr][tt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
        if (a.accumulate) {
          const hrf_f4 p = hrf_ld4(o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p[r];
        }
        hrf_st4(o, v);
      }
    }
  }
}

// wp[tap][n][k]: dir 0 (forward operand): n = out channel, k = in channel, wp = w[n][k][tap];
// dir 1 (backward-data operand): n = in channel, k = out channel, wp = w[k][n][8 - tap]
__global__ __launch_bounds__(256) void conv3_pack_kernel(const float* w, int Cout, int Cin, int dir, float* wp) {
  const int N = dir ? Cin : Cout, K = dir ? Cout : Cin;
  const long total = 9L * N * K;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k = (int)(e % K);
    const long r = e / K;
    const int n = (int)(r % N), tap = (int)(r / N);
    wp[e] = dir ? w[((long)k * Cin + n) * 9 + 8 - tap] : w[((long)n * Cin + k) * 9 + tap];
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions: dW[co][ci][tap] += sum_pix dY[pix][co] * X[pix + tap][ci].
// GEMM with K = pixels.  One block owns 128 output channels x a 64-input-channel slab x all 9 taps (its 73.7K
// accumulators live in registers: wave = 4 co tiles x 1 ci tile x 9 taps = 36 MFMA tiles) and walks a strided share
// of the 4x16 pixel tiles: per tile the dY tile [64 px][128 co] and the X halo [6x18 px][64 ci] are staged once
// (double-buffered, next tile's global loads in flight during the MFMAs) and every X fragment is reused by the 4 co
// tiles, every dY fragment by the 9 taps: 13 LDS reads per 36 MFMAs.  Every activation byte is read by
// (Cout/128) resp. (Cin/64) blocks - 8x less re-reading than a 64x64 block tile.  The pixel-split partial sums go to a
// caller-owned scratch [split][tap][co][ci] with plain 64-byte-run stores and a second small kernel folds them into the
// OIHW gradient: scattered fp32 atomics from the fragments (stride 36 bytes between lanes) measured 0.7 ms for 2.4 M
// lane-atomics on MI355X - the atomic units are paced per request line, not per lane.
constexpr int GH = 4, GW = 16, GPX = GH * GW, GIH = GH + 2, GIW = GW + 2, GNP = GIH * GIW;   // 64 px, 108 halo px
constexpr int GM = 128, GN = 64;
constexpr int PY = GM + 16, PX = GN + 16;      // LDS pitches = 16 mod 32: lanes (i, q) of a fragment read hit 32 distinct banks per half

struct W3wArgs {
  const float* dy; int ldD; const float* x; int ldX;
  float* part; float* dbias;          // part: [splits][9][Cout][Cin]
  int B, H, W, Cin, Cout;
  int tilesX, tilesY, tiles, splits, nslab;
};

__global__ __launch_bounds__(NTHR) void wgrad3w_kernel(W3wArgs a) {
  HRF_DYN_SMEM(float, smem);
  float* sY = smem;                       // [2][GPX * PY]
  float* sX = smem + 2 * GPX * PY;        // [2][GNP * PX]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int c = wave & 3, h = wave >> 2;
  const int cog = blockIdx.y / a.nslab, slab = blockIdx.y - cog * a.nslab;
  const int co0 = cog * GM, ci0 = slab * GN;

  hrf_f4 acc[4][9];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[m][t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
  hrf_f4 bsum = hrf_f4{0.f, 0.f, 0.f, 0.f};

  hrf_f4 ypre[4], xpre[4];
  auto load_tile = [&](int tile) {
    int t = tile;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY; const int b = t / a.tilesY;
    const int y0 = ty * GH, x0 = tx * GW;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = tid + e * NTHR;
      const int p = f >> 5, j = f & 31;
      const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
      const bool ok = gy < a.H && gx < a.W;
      ypre[e] = hrf_ld4(ok ? a.dy + ((long)(b * a.H + gy) * a.W + gx) * a.ldD + co0 + 4 * j : g_zero4w);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = tid + e * NTHR;
      const int p = min(f >> 4, GNP - 1), j = f & 15;
      const int py = p / GIW, px = p - py * GIW;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = f < GNP * 16 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      xpre[e] = hrf_ld4(ok ? a.x + ((long)(b * a.H + gy) * a.W + gx) * a.ldX + ci0 + 4 * j : g_zero4w);
    }
  };
  auto store_tile = [&](int buf) {
    float* dY = sY + buf * (GPX * PY);
    float* dX = sX + buf * (GNP * PX);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = tid + e * NTHR;
      lds_st4(dY + (f >> 5) * PY + 4 * (f & 31), ypre[e]);
#pragma unroll
      for (int r = 0; r < 4; ++r) bsum[r] += ypre[e][r];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = tid + e * NTHR;
      if (f < GNP * 16) lds_st4(dX + (f >> 4) * PX + 4 * (f & 15), xpre[e]);
    }
  };

  int tile = blockIdx.x, buf = 0;
  if (tile < a.tiles) { load_tile(tile); store_tile(0); }
  __syncthreads();
  for (; tile < a.tiles; tile += a.splits) {
    const int nxt = tile + a.splits;
    if (nxt < a.tiles) load_tile(nxt);
    const float* ay = sY + buf * (GPX * PY) + q * PY + (4 * h) * 16 + i;
    const float* bx = sX + buf * (GNP * PX) + q * PX + c * 16 + i;
#pragma unroll
    for (int kk = 0; kk < GPX / 4; ++kk) {
      const int row = kk >> 2, colb = 4 * (kk & 3);
      float av[4], bv[9];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = ay[4 * kk * PY + m * 16];
#pragma unroll
      for (int t = 0; t < 9; ++t) bv[t] = bx[((row + t / 3) * GIW + colb + t % 3) * PX];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[m][t] = hrf_mfma16(av[m], bv[t], acc[m][t]);
    }
    if (nxt < a.tiles) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- merge: acc[m][t][r] = dW[co0 + (4h + m)*16 + 4q + r][ci0 + c*16 + i][t]
  {
    float* pb = a.part + (long)blockIdx.x * 9 * a.Cout * a.Cin + ci0 + c * 16 + i;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pb[((long)t * a.Cout + co0 + (4 * h + m) * 16 + 4 * q + r) * a.Cin] = acc[m][t][r];
  }
  if (a.dbias != nullptr && slab == 0) {      // (uniform) column sums of the dY tiles this block staged
    __syncthreads();
    float* sb = smem;                          // [16][128]
    const int j = tid & 31, g = tid >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) sb[g * GM + 4 * j + r] = bsum[r];
    __syncthreads();
    if (tid < GM) {
      float t = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < 16; ++g2) t += sb[g2 * GM + tid];
      hrf_atomic_add(a.dbias + co0 + tid, t);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Row GEMM for the wide 1x1 convolutions (the HRFPN reduction convolution sum(in_channels) -> 256, hrfpn.py:53-58,85, and
// its backward-data): y[m][n] (+)= bias[n] + sum_k x[m][k] * wp[n][k], K % 16 == 0, N % 16 == 0 (the caller pads with
// zero columns / zero weight rows).  Same machinery as the 3x3 engine with a 16-deep step: block = 128 rows x up to
// 256 channels, both operand tiles in a 2-slot LDS ring (61 KB: the fragments of step s+1 are fetched in the middle
// of step s's MFMAs, so slot s%2 is free again when step s starts), global loads three steps ahead.
constexpr int GK = 16, GLP = GK + 4, GROWS = 128;

struct RgArgs {
  const float* x; int ldX; const float* wp; const float* bias;
  float* y; int ldY; int accumulate;
  long M; int K, N;
};

template <int WN>
__global__ __launch_bounds__(NTHR) void rowgemm_kernel(RgArgs a) {
  constexpr int NB = WN * 64, SLOT = (GROWS + NB) * GLP;
  constexpr int NWV = (NB * 4 + NTHR - 1) / NTHR;      // weight float4 per thread
  HRF_DYN_SMEM(float, smem);                            // [2][SLOT]: x tile rows first, then the weight rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int chg = wave % WN, rg = wave / WN;
  const long m0 = (long)blockIdx.x * GROWS;
  const int n0 = blockIdx.y * NB;
  bool tile_on[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) tile_on[tt] = n0 + chg * 64 + tt * 16 < a.N;

  hrf_f4 acc[WN][4];
#pragma unroll
  for (int rr = 0; rr < WN; ++rr)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[rr][tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  // staging: x tile = 128 rows x 4 float4 (one per thread); weight tile = NB rows x 4 float4
  const long xr = m0 + (tid >> 2) < a.M ? m0 + (tid >> 2) : a.M - 1;          // tail rows re-read the last row, never stored
  const float* xsrc = a.x + xr * a.ldX + 4 * (tid & 3);
  const int xdst = (tid >> 2) * GLP + 4 * (tid & 3);
  const float* wsrc[NWV]; int wdst[NWV];
#pragma unroll
  for (int e = 0; e < NWV; ++e) {
    const int f = tid + e * NTHR, n = min(f >> 2, NB - 1);
    wsrc[e] = a.wp + (long)min(n0 + n, a.N - 1) * a.K + 4 * (f & 3);
    wdst[e] = f < NB * 4 ? (GROWS + n) * GLP + 4 * (f & 3) : -1;
  }
  hrf_f4 xpre, wpre[NWV];
  auto load_tile = [&](int s) {
    xpre = hrf_ld4(xsrc + s * GK);
#pragma unroll
    for (int e = 0; e < NWV; ++e) wpre[e] = hrf_ld4(wsrc[e] + s * GK);
  };
  auto store_tile = [&](int slot) {
    float* d = smem + slot * SLOT;
    lds_st4(d + xdst, xpre);
#pragma unroll
    for (int e = 0; e < NWV; ++e)
      if (wdst[e] >= 0) lds_st4(d + wdst[e], wpre[e]);
  };
  hrf_f4 fa[2][WN], fb[2][4];
  const float* abase = smem + (rg * 16 * WN + i) * GLP + 4 * q;
  const float* bbase = smem + (GROWS + chg * 64 + i) * GLP + 4 * q;
  auto read_frags = [&](int slot, int set) {
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) fa[set][rr] = lds_ld4(abase + slot * SLOT + rr * 16 * GLP);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) fb[set][tt] = lds_ld4(bbase + slot * SLOT + tt * 16 * GLP);
  };
  auto mma = [&](int set, int half) {
#pragma unroll
    for (int m = 2 * half; m < 2 * half + 2; ++m)
#pragma unroll
      for (int rr = 0; rr < WN; ++rr)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          if (tile_on[tt]) acc[rr][tt] = hrf_mfma16(fb[set][tt][m], fa[set][rr][m], acc[rr][tt]);
  };

  const int S = a.K / GK;
  load_tile(0);
  store_tile(0);
  if (S > 1) { load_tile(1); store_tile(1); }
  if (S > 2) load_tile(2);
  __syncthreads();
  read_frags(0, 0);
  HRF_WAIT_LDS();
  __syncthreads();                                      // slot 0 is rewritten in step 0: every wave holds its fragments first
  // one step: MFMAs of step s on fragment set `set`, with the tile of step s+2 written into slot s%2 (its fragments
  // are in registers already) and the fragments of step s+1 fetched from the other slot half-way
  auto step = [&](int s, int set) {
    HRF_SCHED_FENCE();
    mma(set, 0);
    HRF_SCHED_FENCE();
    if (s + 2 < S) store_tile(s & 1);
    if (s + 3 < S) load_tile(s + 3);
    read_frags((s + 1) & 1, set ^ 1);          // complete since the previous barrier (stale but valid after the last step)
    HRF_SCHED_FENCE();
    mma(set, 1);
    HRF_SCHED_FENCE();
    __syncthreads();
  };
  for (int s = 0; s < S; s += 2) {
    step(s, 0);
    if (s + 1 < S) step(s + 1, 1);
  }

  // ---- epilogue: acc[rr][tt][r] = y(row m0 + rg*16*WN + rr*16 + i, channel n0 + chg*64 + tt*16 + 4q + r)
  // an odd S leaves the last accumulators in the same registers either way (acc is shared by both sets)
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    if (!tile_on[tt]) continue;
    const int ch = n0 + chg * 64 + tt * 16 + 4 * q;
    hrf_f4 bv = hrf_f4{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bv = hrf_ld4(a.bias + ch);
#pragma unroll
    for (int rr = 0; rr < WN; ++rr) {
      const long m = m0 + rg * 16 * WN + rr * 16 + i;
      if (m < a.M) {
        float* o = a.y + m * a.ldY + ch;
        hrf_f4 v = acc[rr][tt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
        if (a.accumulate) {
          const hrf_f4 p = hrf_ld4(o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p[r];
        }
        hrf_st4(o, v);
      }
    }
  }
}

// wp[n][k], n < Np, k < Kp: dir 0: w[n][k] (n < Cout, k < Cin); dir 1: w[k][n] (n < Cin, k < Cout); zero elsewhere
__global__ __launch_bounds__(256) void rowgemm_pack_kernel(const float* w, int Cout, int Cin, int dir, int Np, int Kp, float* wp) {
  const long total = (long)Np * Kp;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k = (int)(e % Kp), n = (int)(e / Kp);
    float v = 0.f;
    if (dir == 0) { if (n < Cout && k < Cin) v = w[(long)n * Cin + k]; }
    else if (n < Cin && k < Cout) v = w[(long)k * Cin + n];
    wp[e] = v;
  }
}

template <int WN>
int launch_rg(const RgArgs& a, void* stream) {
  constexpr size_t smem = (size_t)2 * (GROWS + WN * 64) * GLP * sizeof(float);
#ifndef HRF_EMUL
  static std::atomic<unsigned> lds_set{0u};
  if (hrf_dyn_lds_once(lds_set, reinterpret_cast<const void*>(&rowgemm_kernel<WN>), (int)smem) != HRF_OK) return HRF_ERR_LAUNCH;
#endif
  const dim3 grid(hrf_cdiv(a.M, GROWS), hrf_cdiv(a.N, WN * 64));
  HRF_LAUNCH((rowgemm_kernel<WN>), grid, dim3(NTHR), smem, stream, a);
  return hrf_check_launch();
}

// dw[co][ci][t] += sum_s part[s][t][co][ci]: one thread per (t, co, ci); reads coalesced over ci, the split loop
// unrolled so that 8 loads are in flight per thread
__global__ __launch_bounds__(256) void wgrad3w_fold_kernel(const float* part, int splits, int Cout, int Cin, float* dw) {
  const long n = (long)Cout * Cin, total = 9 * n;
  const long f = (long)blockIdx.x * 256 + threadIdx.x;
  if (f >= total) return;
  const int t = (int)(f / n);
  const long e = f - (long)t * n;
  const float* p = part + f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int sp = 0;
  for (; sp + 4 <= splits; sp += 4) {
    s0 += p[(long)sp * total]; s1 += p[(long)(sp + 1) * total]; s2 += p[(long)(sp + 2) * total]; s3 += p[(long)(sp + 3) * total];
  }
  for (; sp < splits; ++sp) s0 += p[(long)sp * total];
  dw[e * 9 + t] += (s0 + s1) + (s2 + s3);
}

int g_force_wn = 0, g_wsplit = 0;

int wgrad3w_splits(int B, int H, int W, int Cin, int Cout) {
  const long tiles = (long)hrf_cdiv(W, GW) * hrf_cdiv(H, GH) * B;
  const int chan_blocks = (Cout / GM) * (Cin / GN);
  const int splits = g_wsplit > 0 ? g_wsplit : hrf_cdiv(256, chan_blocks);
  return (int)(splits < tiles ? splits : tiles);
}

template <int WN>
int launch_w(C3wArgs a, void* stream) {
  constexpr size_t smem = (size_t)(NPIX * LP + 3 * WN * 64 * WP) * sizeof(float);
#ifndef HRF_EMUL
  static std::atomic<unsigned> lds_set{0u};
  if (hrf_dyn_lds_once(lds_set, reinterpret_cast<const void*>(&conv3w_kernel<WN>), (int)smem) != HRF_OK) return HRF_ERR_LAUNCH;
#endif
  const dim3 grid(a.tilesX * a.tilesY * a.B, hrf_cdiv(a.N, WN * 64));
  HRF_LAUNCH((conv3w_kernel<WN>), grid, dim3(NTHR), smem, stream, a);
  return hrf_check_launch();
}

}  // namespace

extern "C" int hrf_conv3_pack(const float* w, int Cout, int Cin, int dir, float* wp, void* stream) {
  const long total = 9L * Cout * Cin;
  if (total <= 0) return HRF_OK;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  HRF_LAUNCH(conv3_pack_kernel, dim3(grid), dim3(256), 0, stream, w, Cout, Cin, dir, wp);
  return hrf_check_launch();
}

extern "C" int hrf_conv3_packed(const float* x, int ldX, const float* wp, const float* bias, float* y, int ldY,
                                int accumulate, int B, int H, int W, int K, int N, void* stream) {
  if (K <= 0 || N <= 0 || K % KS != 0 || N % 64 != 0) return HRF_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0) return HRF_OK;
  C3wArgs a{x, ldX, wp, bias, y, ldY, accumulate, B, H, W, K, N, hrf_cdiv(W, TW), hrf_cdiv(H, TH)};
  const long tiles = (long)a.tilesX * a.tilesY * B;
  // fewer tiles: fewer 64-channel groups per block (more blocks, shorter serial depth of each; a step of the
  // 1-group variant is bound by the weight-prefetch latency, so the 2-group one is preferred while it fills the CUs)
  int wn = (N % 256 == 0 && tiles >= 128) ? 4 : ((N % 128 == 0 && tiles >= 32) ? 2 : 1);
  if (g_force_wn == 1 || (g_force_wn == 4 && N % 256 == 0) || (g_force_wn == 2 && N % 128 == 0)) wn = g_force_wn;
  return wn == 4 ? launch_w<4>(a, stream) : (wn == 2 ? launch_w<2>(a, stream) : launch_w<1>(a, stream));
}

extern "C" int hrf_rowgemm_pack(const float* w, int Cout, int Cin, int dir, int Np, int Kp, float* wp, void* stream) {
  const long total = (long)Np * Kp;
  if (total <= 0) return HRF_OK;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  HRF_LAUNCH(rowgemm_pack_kernel, dim3(grid), dim3(256), 0, stream, w, Cout, Cin, dir, Np, Kp, wp);
  return hrf_check_launch();
}

extern "C" int hrf_rowgemm(const float* x, int ldX, const float* wp, const float* bias, float* y, int ldY, int accumulate,
                           long M, int K, int N, void* stream) {
  if (K <= 0 || N <= 0 || K % GK != 0 || N % 16 != 0) return HRF_ERR_ARG;
  if (M <= 0) return HRF_OK;
  RgArgs a{x, ldX, wp, bias, y, ldY, accumulate, M, K, N};
  const long mtiles = (M + GROWS - 1) / GROWS;
  int wn = (N > 128 && mtiles >= 128) ? 4 : (N > 64 ? 2 : 1);
  if (g_force_wn == 1 || g_force_wn == 2 || g_force_wn == 4) wn = g_force_wn;
  return wn == 4 ? launch_rg<4>(a, stream) : (wn == 2 ? launch_rg<2>(a, stream) : launch_rg<1>(a, stream));
}

extern "C" long hrf_conv3_wgrad_wide_scratch(int B, int H, int W, int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % GN != 0 || Cout % GM != 0 || B <= 0 || H <= 0 || W <= 0) return 0;
  return (long)wgrad3w_splits(B, H, W, Cin, Cout) * 9 * Cout * Cin;
}

extern "C" int hrf_conv3_wgrad_wide(const float* dy, int ldD, const float* x, int ldX, int B, int H, int W, int Cin,
                                    int Cout, float* dw, float* dbias, float* scratch, void* stream) {
  if (Cin <= 0 || Cout <= 0 || Cin % GN != 0 || Cout % GM != 0 || scratch == nullptr) return HRF_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0) return HRF_OK;
  W3wArgs a{dy, ldD, x, ldX, scratch, dbias, B, H, W, Cin, Cout, hrf_cdiv(W, GW), hrf_cdiv(H, GH), 0, 0, Cin / GN};
  a.tiles = a.tilesX * a.tilesY * B;
  const int chan_blocks = (Cout / GM) * a.nslab;
  a.splits = wgrad3w_splits(B, H, W, Cin, Cout);
  constexpr size_t smem = (size_t)(2 * GPX * PY + 2 * GNP * PX) * sizeof(float);
#ifndef HRF_EMUL
  static std::atomic<unsigned> lds_set{0u};
  if (hrf_dyn_lds_once(lds_set, reinterpret_cast<const void*>(&wgrad3w_kernel), (int)smem) != HRF_OK) return HRF_ERR_LAUNCH;
#endif
  HRF_LAUNCH(wgrad3w_kernel, dim3(a.splits, chan_blocks), dim3(NTHR), smem, stream, a);
  const int fgrid = hrf_cdiv(9L * Cout * Cin, 256);
  HRF_LAUNCH(wgrad3w_fold_kernel, dim3(fgrid), dim3(256), 0, stream, (const float*)scratch, a.splits, Cout, Cin, dw);
  return hrf_check_launch();
}

// (reached through hrf_debug_knob only: not exported)
extern "C" __attribute__((visibility("hidden"))) int hrf_conv3w_knob(int key, int value) {
  if (key == 0) { g_force_wn = value; return HRF_OK; }
  if (key == 2) { g_wsplit = value; return HRF_OK; }
  return HRF_ERR_ARG;
}
