// Eval-mode CrossFFN in ONE launch (gfx950, fp32): the whole second half of an HRFormerBlock / fusion block with FROZEN
// BatchNorm statistics,
//
//   out = x' + GELU(BN3(fc3( GELU(BN2(dw3x3( GELU(BN1(fc1( LN_2(x') ))) ))) )))
//
// hrformer.py:267-295 (CrossFFN.forward: 1x1 expansion -> BN -> GELU -> depthwise 3x3 -> BN -> GELU -> 1x1 projection ->
// BN -> GELU) + :351 (norm2) + :371-372 (residual; DropPath is the identity in eval).  In training the three BatchNorms need
// batch statistics - grid-wide dependencies that ARE the kernel boundaries of the training path (fc1 | dw | fc3).  With
// frozen statistics every BatchNorm is a per-channel affine and nothing is left that crosses a spatial tile except the 1-pixel
// halo of the depthwise convolution: a workgroup owns a 4 x 16 pixel tile, recomputes LN_2 + fc1 + BN1 + GELU on the
// 6 x 18 halo, and the 4C-wide hidden tensor - 8.8 MB written and read twice per 18-channel block of the 96 x 160 map on the
// per-op route - never leaves the chip.
//
// The hidden dimension is walked in chunks of HC channels (72 for the 18-multiples of HRFuser-T / STF, 78 for HRFuser-B):
// the depthwise convolution does not mix channels and the projection SUMS over them, so a chunk is
//   fc1 (MFMA: rows of W1 x halo pixels) -> LDS [108 halo pixels][HC]
//   depthwise 3x3 + BN2 + GELU evaluated by each lane for exactly the (pixel, hidden channel) values it feeds to the
//   projection's MFMA as B operand (9 LDS reads + 9 broadcast weight reads per value; no second hidden tile, no transposition)
//   fc3 partial sums into the wave's accumulators (MFMA: rows of W3 x the wave's 16 pixels).
// Wave w owns output row w of the tile; LDS: LN_2 rows [108][C+1] + hidden chunk [108][HC+1] + depthwise weights.
#include "hrf_common.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int FE_TH = 4, FE_TW = 16, FE_HH = FE_TH + 2, FE_HW = FE_TW + 2, FE_NPH = FE_HH * FE_HW;   // 108 halo pixels
constexpr int FE_PT = (FE_NPH + 15) / 16;                                                            // 7 halo pixel tiles
constexpr bool FE_WL(int C) { return C <= 36; }

__device__ float g_fe_zero4[4] = {0.f, 0.f, 0.f, 0.f};

// 4 consecutive elements p[off .. off+3], `nvalid` of them exist (<= 0: none); full: the whole 16-wide slab is valid (uniform)
__device__ __forceinline__ hrf_f4 fe_ld(bool full, const float* p, long off, int nvalid) {
  if (full) return hrf_ld4(nvalid > 0 ? p + off : g_fe_zero4);
  hrf_f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = *(e < nvalid ? p + off + e : g_fe_zero4);
  return r;
}

template <int C, int HC>
__global__ __launch_bounds__(256) void ffn_eval_kernel(hrf_ffn_eval_t a) {
  constexpr int PC = C + 1, HCP = (HC + 3) & ~3, PH = HCP + 4;      // PH: 16-byte rows (one ds_read_b128 = 4 consecutive hidden channels)
  constexpr int CT = (C + 15) / 16, HT = (HC + 15) / 16, KS1 = (C + 15) / 16, KS3 = (HC + 15) / 16;
  // WL (narrow blocks: one or two hidden chunks at the two finest branches, where the launch has hundreds of tiles): the
  // chunk's W1 / W3 rows and the fc1 bias / BN1 affine are staged in LDS once per chunk, zero-padded to whole 16-wide MFMA tiles
  // ([80][PW1], [CT*16][PW3], 16-byte rows): every fragment is then one ds_read_b128 and no global round trip sits inside the
  // pixel-tile and slab loops (v1 fetched them from L2 per pixel tile: 22 us for the 18-channel block of the 96 x 160 map)
  constexpr bool WL = FE_WL(C);
  constexpr int PW1 = ((C + 3) & ~3) + 4, PW3 = ((HC + 3) & ~3) + 4;
  HRF_DYN_SMEM(float, smem);
  float* sXn = smem;                       // [108][PC] x' rows of the halo -> LN_2 rows
  float* sH = sXn + FE_NPH * PC;           // [108][PH] GELU(BN1(fc1)) of the current hidden chunk (0 outside the image)
  float* sWd = sH + FE_NPH * PH;           // [9][HCP] depthwise taps of the chunk, tap-major (pad channels zero)
  float* sP2 = sWd + 9 * HCP;              // [3][HCP] depthwise bias | BN2 scale | BN2 shift
  float* sW1 = sP2 + 3 * HCP;              // (WL) [HT*16][PW1]
  float* sW3 = sW1 + (WL ? HT * 16 * PW1 : 0);                              // (WL) [CT*16][PW3]
  float* sP1 = sW3 + (WL ? CT * 16 * PW3 : 0);                              // (WL) [3][HT*16] fc1 bias | BN1 scale | BN1 shift
  __shared__ int sIn[FE_NPH];              // 1: halo pixel lies inside the image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  int t = blockIdx.x;
  const int tilesX = (a.W + FE_TW - 1) / FE_TW, tilesY = (a.H + FE_TH - 1) / FE_TH;
  const int tx = t % tilesX; t /= tilesX;
  const int ty = t % tilesY; const int b = t / tilesY;
  const int y0 = ty * FE_TH, x0 = tx * FE_TW;

  // ---- halo rows of x' (zeros outside the image), all loads of a thread before its first store
  constexpr int NE = (FE_NPH * C + 255) / 256;
  {
    float v[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u, ec = e < FE_NPH * C ? e : 0;
      const int p = ec / C, c = ec - p * C;
      const int py = p / FE_HW, px = p - py * FE_HW;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool in = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const float xv = a.x[in ? ((long)(b * a.H + gy) * a.W + gx) * C + c : 0];
      v[u] = in ? xv : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u;
      if (e < FE_NPH * C) { const int p = e / C; sXn[p * PC + (e - p * C)] = v[u]; }
    }
    if (tid < FE_NPH) {
      const int py = tid / FE_HW, px = tid - py * FE_HW;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      sIn[tid] = ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) ? 1 : 0;
    }
  }
  __syncthreads();
  // ---- LayerNorm_2 in place: two lanes per pixel, two-pass statistics (as hrf_ln_stats)
  {
    const int p = tid >> 1, part = tid & 1;
    const int pc = p < FE_NPH ? p : 0;
    float* row = sXn + pc * PC;
    float s = 0.f;
    for (int c = part; c < C; c += 2) s += row[c];
    s += __shfl_xor(s, 1);
    const float mean = s * (1.0f / (float)C);
    float qq = 0.f;
    for (int c = part; c < C; c += 2) { const float d = row[c] - mean; qq = fmaf(d, d, qq); }
    qq += __shfl_xor(qq, 1);
    const float rstd = hrf_rsqrt_nr(qq * (1.0f / (float)C) + a.ln_eps);
    if (p < FE_NPH)
      for (int c = part; c < C; c += 2) row[c] = fmaf((row[c] - mean) * rstd, a.ln_g[c], a.ln_b[c]);
  }
  __syncthreads();

  hrf_f4 acc3[CT];
#pragma unroll
  for (int tt = 0; tt < CT; ++tt) acc3[tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};
  const int p0 = (wave + 1) * FE_HW + (j + 1);                      // halo index of this lane's output pixel (row wave, column j)

#pragma unroll 1
  for (int hc0 = 0; hc0 < a.hidden; hc0 += HC) {
    // depthwise parameters of the chunk
    for (int e = tid; e < 9 * HCP; e += 256) {
      const int tap = e / HCP, hc = e - tap * HCP;
      const float w = a.wd[(long)(hc0 + (hc < HC ? hc : 0)) * 9 + tap];
      sWd[e] = hc < HC ? w : 0.f;
    }
    for (int e = tid; e < HCP; e += 256) {
      const bool v = e < HC;
      const int ec = hc0 + (v ? e : 0);
      const float bb = a.bd[ec], ss = a.s2[ec], tt = a.t2[ec];
      sP2[e] = v ? bb : 0.f; sP2[HCP + e] = v ? ss : 0.f; sP2[2 * HCP + e] = v ? tt : 0.f;
    }
    if (WL) {
      for (int e = tid; e < HT * 16 * (PW1 / 4); e += 256) {           // W1 rows hc0 .. hc0+HC-1 (rows / columns beyond: zero)
        const int n = e / (PW1 / 4), kb = 4 * (e - n * (PW1 / 4));
        hrf_st4(sW1 + n * PW1 + kb, fe_ld(kb + 4 <= C, a.w1, (long)(hc0 + n) * C + kb, n < HC ? C - kb : 0));
      }
      for (int e = tid; e < CT * 16 * (PW3 / 4); e += 256) {           // W3[c][hc0 .. hc0+HC-1]
        const int n = e / (PW3 / 4), kb = 4 * (e - n * (PW3 / 4));
        hrf_st4(sW3 + n * PW3 + kb, fe_ld(kb + 4 <= HC, a.w3, (long)n * a.hidden + hc0 + kb, n < C ? HC - kb : 0));
      }
      for (int e = tid; e < HT * 16; e += 256) {
        const bool v = e < HC;
        const int ec = hc0 + (v ? e : 0);
        const float bb = a.b1[ec], ss = a.s1[ec], tt = a.t1[ec];
        sP1[e] = v ? bb : 0.f; sP1[HT * 16 + e] = v ? ss : 0.f; sP1[2 * HT * 16 + e] = v ? tt : 0.f;
      }
      __syncthreads();
    }
    // ---- fc1 + BN1 + GELU on the halo: pixel tiles round-robin over the waves
#pragma unroll 1
    for (int pt = wave; pt < FE_PT; pt += 4) {
      const int p = 16 * pt + j, pcl = p < FE_NPH ? p : 0;
      hrf_f4 acc[HT];
#pragma unroll
      for (int tt = 0; tt < HT; ++tt) {                              // bias rows into the accumulators
        const int nb = 16 * tt + 4 * q;
        acc[tt] = WL ? hrf_ld4(sP1 + nb) : fe_ld(16 * (tt + 1) <= HC, a.b1, hc0 + nb, HC - nb);
      }
      const float* xrow = sXn + pcl * PC;
#pragma unroll (FE_WL(C) ? 4 : 2)
      for (int s = 0; s < KS1; ++s) {
        const int kbase = 16 * s + 4 * q, kval = C - kbase;
        const bool kfull = 16 * (s + 1) <= C;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float xv = xrow[r < kval ? kbase + r : 0]; bv[r] = r < kval ? xv : 0.f; }
        hrf_f4 wv[HT];
#pragma unroll
        for (int tt = 0; tt < HT; ++tt) {
          const int n = 16 * tt + j;                                  // (A operand: row index = lane & 15)
          wv[tt] = WL ? hrf_ld4(sW1 + n * PW1 + (kbase < PW1 ? kbase : 0)) : fe_ld(kfull, a.w1, (long)(hc0 + n) * C + kbase, n < HC ? kval : 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tt = 0; tt < HT; ++tt) acc[tt] = hrf_mfma16(wv[tt][r], bv[r], acc[tt]);
      }
      const bool inimg = sIn[pcl] != 0;
#pragma unroll
      for (int tt = 0; tt < HT; ++tt) {
        const int nb = 16 * tt + 4 * q;
        const hrf_f4 sc = WL ? hrf_ld4(sP1 + HT * 16 + nb) : fe_ld(16 * (tt + 1) <= HC, a.s1, hc0 + nb, HC - nb);
        const hrf_f4 sh = WL ? hrf_ld4(sP1 + 2 * HT * 16 + nb) : fe_ld(16 * (tt + 1) <= HC, a.t1, hc0 + nb, HC - nb);
        hrf_f4 h4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h = hrf_gelu(fmaf(acc[tt][r], sc[r], sh[r]));
          h4[r] = (inimg && nb + r < HC) ? h : 0.f;                   // zero padding of the depthwise input; pad channels zero
        }
        if (p < FE_NPH && nb < HCP) hrf_st4(sH + p * PH + nb, h4);
      }
    }
    __syncthreads();
    // ---- depthwise 3x3 + BN2 + GELU for the values this lane feeds to the projection, fc3 partial sums
#pragma unroll (FE_WL(C) ? 5 : 1)
    for (int s = 0; s < KS3; ++s) {
      const int kb = 16 * s + 4 * q;
      hrf_f4 wv[CT];
#pragma unroll
      for (int tt = 0; tt < CT; ++tt) {
        const int n = 16 * tt + j;
        wv[tt] = WL ? hrf_ld4(sW3 + n * PW3 + kb) : fe_ld(16 * (s + 1) <= HC, a.w3, (long)n * a.hidden + hc0 + kb, n < C ? HC - kb : 0);
      }
      // the lane's 4 consecutive hidden channels kb .. kb+3 (pad channels: zero weights, zero affine -> GELU(0) = 0)
      const int kbc = kb < HCP ? kb : 0;
      const float* hp = sH + p0 * PH + kbc;
      hrf_f4 sum = hrf_ld4(sP2 + kbc);
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const hrf_f4 hh = hrf_ld4(hp + (dy * FE_HW + dx) * PH), ww = hrf_ld4(sWd + ((dy + 1) * 3 + dx + 1) * HCP + kbc);
#pragma unroll
          for (int r = 0; r < 4; ++r) sum[r] = fmaf(hh[r], ww[r], sum[r]);
        }
      const hrf_f4 sc2 = hrf_ld4(sP2 + HCP + kbc), sh2 = hrf_ld4(sP2 + 2 * HCP + kbc);
      float hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g = hrf_gelu(fmaf(sum[r], sc2[r], sh2[r]));
        hv[r] = (kb + r < HC) ? g : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tt = 0; tt < CT; ++tt) acc3[tt] = hrf_mfma16(wv[tt][r], hv[r], acc3[tt]);
    }
    __syncthreads();                                                  // the next chunk overwrites sH / sWd / sP2
  }

  // ---- BN3 + GELU + residual: lane (j, q) holds channels 16 tt + 4 q + r of pixel (y0 + wave, x0 + j)
  const int gy = y0 + wave, gx = x0 + j;
  if (gy < a.H && gx < a.W) {
    const long pix = (long)(b * a.H + gy) * a.W + gx;
#pragma unroll
    for (int tt = 0; tt < CT; ++tt) {
      const int cb = 16 * tt + 4 * q, nval = C - cb;
      const bool full = 16 * (tt + 1) <= C;
      const hrf_f4 b3 = fe_ld(full, a.b3, cb, nval), sc = fe_ld(full, a.s3, cb, nval), sh = fe_ld(full, a.t3, cb, nval);
      const hrf_f4 xr = fe_ld(full, a.x, pix * C + cb, nval);
      hrf_f4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = xr[r] + hrf_gelu(fmaf(acc3[tt][r] + b3[r], sc[r], sh[r]));
      float* dst = a.out + pix * C + cb;
      if (full) hrf_st4(dst, o);
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r < nval) dst[r] = o[r];
      }
    }
  }
}

template <int C, int HC>
int fe_launch(const hrf_ffn_eval_t& a, void* stream) {
  constexpr int HT = (HC + 15) / 16, CT = (C + 15) / 16, PW1 = ((C + 3) & ~3) + 4, PW3 = ((HC + 3) & ~3) + 4;
  constexpr size_t wl = FE_WL(C) ? (size_t)HT * 16 * PW1 + (size_t)CT * 16 * PW3 + 3 * HT * 16 : 0;
  constexpr int HCP = (HC + 3) & ~3;
  constexpr size_t smem = ((size_t)FE_NPH * (C + 1) + (size_t)FE_NPH * (HCP + 4) + 12 * HCP + wl) * sizeof(float);
#ifndef HRF_EMUL
  static std::atomic<unsigned> lds_set{0u};
  if (hrf_dyn_lds_once(lds_set, reinterpret_cast<const void*>(&ffn_eval_kernel<C, HC>), (int)smem) != HRF_OK) return HRF_ERR_LAUNCH;
#endif
  const int tiles = a.B * ((a.H + FE_TH - 1) / FE_TH) * ((a.W + FE_TW - 1) / FE_TW);
  HRF_LAUNCH((ffn_eval_kernel<C, HC>), dim3(tiles), dim3(256), (unsigned)smem, stream, a);
  return hrf_check_launch();
}

}  // namespace

extern "C" int hrf_ffn_eval_supported(int C, int hidden) {
  if (hidden != 4 * C) return 0;
  // (the kernel body is generic in C / HC - 72 / 144 / 78 / 156 were built and measured: with 36 or 12 tiles per launch and 4 - 8
  // serial hidden chunks per tile they lose to the per-op launches by 2 - 6x, DESIGN.md section 11 - only the widths that pay are instantiated)
  return (C == 18 || C == 36) ? 1 : 0;
}

extern "C" int hrf_ffn_eval(const hrf_ffn_eval_t* p, void* stream) {
  if (p == nullptr || !hrf_ffn_eval_supported(p->C, p->hidden)) return HRF_ERR_ARG;
  const hrf_ffn_eval_t& a = *p;
  if (a.x == nullptr || a.out == nullptr || a.ln_g == nullptr || a.ln_b == nullptr || a.w1 == nullptr || a.b1 == nullptr ||
      a.s1 == nullptr || a.t1 == nullptr || a.wd == nullptr || a.bd == nullptr || a.s2 == nullptr || a.t2 == nullptr ||
      a.w3 == nullptr || a.b3 == nullptr || a.s3 == nullptr || a.t3 == nullptr) return HRF_ERR_ARG;
  if (a.B <= 0 || a.H <= 0 || a.W <= 0) return HRF_OK;
  if (a.C == 18) return fe_launch<18, 72>(a, stream);
  return fe_launch<36, 72>(a, stream);
}
