// 7x7 windowed multi-head attention core (self- and cross-attention) for gfx950, fp32.
//
// Replaces, per (window, head): the centre zero-pad + window partition, q@k^T + relative position
// bias, softmax, attn@v, window merge and de-pad of
//   WindowMSA.forward / LocalWindowSelfAttention.forward   hrformer.py:96-131 / 184-236
//   WindowMCA.forward / MultiWindowCrossAttention.forward  hrfuser_hrformer_based.py:106-151 / 189-248
// The Q/K/V projections are 1x1 convs on the MFMA engine (conv_engine.hip) over the UNPADDED
// NHWC map; the window partition is pure index math here and is never materialised.  Padded
// tokens are exact zeros AFTER LayerNorm in the reference, so their projections equal the
// projection bias: padded keys/values are synthesised from (k_bias, v_bias) and - as in the
// reference (with_pad_mask=False) - are NOT masked.
//
// Mapping: one wave64 per (window, head); lane i = query row i (49 of 64 lanes active).  K/V rows
// of the window are staged once in LDS (rows padded to a multiple of 4 floats, 16 B aligned) and
// read as wave-wide ds_read_b128 broadcasts (conflict-free); the whole 49-wide logit row lives in
// the lane's registers (fully unrolled), so softmax needs no cross-lane traffic and the 49
// dot-product chains are independent (ILP hides the LDS latency inside a single wave).
// head_dim is 18 (HRFuser-T) / 39 (HRFuser-B): with 49x49xD contractions padded to 64x64x20 an
// MFMA tile would waste > 55 % of its lanes, so the contraction runs as per-lane FMA chains at the
// same fp32 rate the f32 MFMA has ("attention core").
#include "hrf_common.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int NT = 49;      // tokens per window (7x7)

struct alignas(16) V4 { float x, y, z, w; };

struct AttnArgs {
  const float* q; int ldq, qoff;
  const float* k; int ldk, koff;
  const float* v; int ldv, voff;
  const float* kpad; const float* vpad;      // (C): key / value of a padded token (= projection bias)
  const float* rpb;                          // (169, heads) relative position bias table
  float* o; int ldo;
  int B, H, W, heads, nWh, nWw, pt, pl;
  float scale;
  // backward only
  const float* dout; int lddo;
  float* dq; int lddq, dqoff;
  float* dk; int lddk, dkoff;
  float* dv; int lddv, dvoff;
  float* dkpad; float* dvpad; float* drpb; long copy_stride;
  int iters;
};

__device__ __forceinline__ int tok_pixel(const AttnArgs& a, int b, int wy, int wx, int t) {
  const int ty = t / 7, tx = t - 7 * ty;
  const int py = wy * 7 + ty - a.pt, px = wx * 7 + tx - a.pl;
  if ((unsigned)py < (unsigned)a.H && (unsigned)px < (unsigned)a.W) return (b * a.H + py) * a.W + px;
  return -1;
}

// dot(q[0..DP), row[0..DP)) + init, row read as DP/4 aligned 16-byte LDS broadcasts
template <int DP>
__device__ __forceinline__ float dot_row(const float* q, const float* row, float init) {
  const V4* r4 = reinterpret_cast<const V4*>(row);
  float acc = init;
#pragma unroll
  for (int c = 0; c < DP / 4; ++c) {
    const V4 v = r4[c];
    acc = fmaf(q[4 * c + 0], v.x, acc); acc = fmaf(q[4 * c + 1], v.y, acc);
    acc = fmaf(q[4 * c + 2], v.z, acc); acc = fmaf(q[4 * c + 3], v.w, acc);
  }
  return acc;
}
template <int DP>
__device__ __forceinline__ void axpy_row(float* o, float p, const float* row) {
  const V4* r4 = reinterpret_cast<const V4*>(row);
#pragma unroll
  for (int c = 0; c < DP / 4; ++c) {
    const V4 v = r4[c];
    o[4 * c + 0] = fmaf(p, v.x, o[4 * c + 0]); o[4 * c + 1] = fmaf(p, v.y, o[4 * c + 1]);
    o[4 * c + 2] = fmaf(p, v.z, o[4 * c + 2]); o[4 * c + 3] = fmaf(p, v.w, o[4 * c + 3]);
  }
}

constexpr int SP = 65;      // pitch of the per-wave [key j][query i] score planes: lane = i or lane = j both <= 2-way

// Forward.  The 49-wide logit row of a query lives in a per-wave LDS plane ([j][i], pitch 65) instead
// of 49 registers of a fully unrolled body: the loops stay rolled (7 x unrolled-7), the kernel is
// ~3 KB instead of 16 KB - at 20 us per launch the cold instruction cache was the main cost.
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_fwd_kernel(AttnArgs a) {
  constexpr int DP = (D + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float sK[WAVES][NT * DP];
  __shared__ __attribute__((aligned(16))) float sV[WAVES][NT * DP];
  __shared__ float sS[WAVES][NT * SP];
  __shared__ float sT[WAVES][176];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = blockIdx.y;
  const int nwin = a.B * a.nWh * a.nWw;
  const int win = blockIdx.x * WAVES + wave;
  const bool active = win < nwin;
  const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
  if (active) {
    for (int e = lane; e < NT * DP; e += 64) {
      const int j = e / DP, d = e - j * DP;
      const int pix = tok_pixel(a, b, wy, wx, j);
      const bool dv = d < D;
      const long pc = pix >= 0 ? pix : 0;       // unconditional clamped loads, select afterwards
      const int col = h * D + (dv ? d : 0);
      const float kx = a.k[pc * a.ldk + a.koff + col], kp = a.kpad[col];
      const float vx = a.v[pc * a.ldv + a.voff + col], vp = a.vpad[col];
      sK[wave][e] = dv ? (pix >= 0 ? kx : kp) : 0.f;
      sV[wave][e] = dv ? (pix >= 0 ? vx : vp) : 0.f;
    }
    for (int e = lane; e < 169; e += 64) sT[wave][e] = a.rpb[e * a.heads + h];
  }
  __syncthreads();
  const int i = lane;
  const int pix = (active && i < NT) ? tok_pixel(a, b, wy, wx, i) : -1;
  if (pix < 0) return;                       // padded / idle query rows produce no output
  float q[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) q[d] = d < D ? a.q[(long)pix * a.ldq + a.qoff + h * D + d] * a.scale : 0.f;
  const int yi = i / 7, xi = i - 7 * yi;
  const int bias0 = (yi + 6) * 13 + (xi + 6);
  const float* Kw = sK[wave];
  const float* Vw = sV[wave];
  const float* Tw = sT[wave];
  float* Sw = sS[wave];
  float m = -3.0e38f;
#pragma unroll 1
  for (int jy = 0; jy < 7; ++jy) {
#pragma unroll
    for (int jx = 0; jx < 7; ++jx) {
      const int j = jy * 7 + jx;
      const float sv = dot_row<DP>(q, Kw + j * DP, Tw[bias0 - jy * 13 - jx]);
      Sw[j * SP + i] = sv;
      m = fmaxf(m, sv);
    }
  }
  float l = 0.f, o[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) o[d] = 0.f;
#pragma unroll 1
  for (int jy = 0; jy < 7; ++jy) {
#pragma unroll
    for (int jx = 0; jx < 7; ++jx) {
      const int j = jy * 7 + jx;
      const float p = __expf(Sw[j * SP + i] - m);
      l += p;
      axpy_row<DP>(o, p, Vw + j * DP);
    }
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < D; ++d) a.o[(long)pix * a.ldo + h * D + d] = o[d] * inv;
}

// Backward.  Pass A (lane = query i) leaves the softmax P and dS = P*(dP - D_i) in two LDS planes;
// pass B (lane = key j) only READS them (no recomputation of logits / exp), the relative-position
// bias gradient is a gather over the dS plane (no LDS atomics: ds_add_f32 runs at ~1 lane/clk),
// pad-key gradients are wave sums.  All loops rolled (7 x 7-unrolled): ~6 KB of code, < 128 VGPRs.
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_kernel(AttnArgs a) {
  constexpr int DP = (D + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float sK[WAVES][NT * DP];
  __shared__ __attribute__((aligned(16))) float sV[WAVES][NT * DP];
  __shared__ __attribute__((aligned(16))) float sQ[WAVES][NT * DP];
  __shared__ __attribute__((aligned(16))) float sG[WAVES][NT * DP];        // dO rows
  __shared__ float sP[WAVES][NT * SP];
  __shared__ float sD[WAVES][NT * SP];
  __shared__ float sT[WAVES][176];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = blockIdx.y;
  const int nwin = a.B * a.nWh * a.nWw;
  for (int e = lane; e < 176; e += 64) sT[wave][e] = a.rpb[(e < 169 ? e : 0) * a.heads + h];
  const float* Kw = sK[wave];
  const float* Vw = sV[wave];
  const float* Qw = sQ[wave];
  const float* Gw = sG[wave];
  const float* Tw = sT[wave];
  float* Pw = sP[wave];
  float* Dw = sD[wave];
  const int li = lane < NT ? lane : 0;
  const int yl = li / 7, xl = li - 7 * yl;
  float bin[3] = {0.f, 0.f, 0.f};              // dRPB bins lane, lane+64, lane+128
  float padk = 0.f, padv = 0.f;                // lane d < D: pad-key / pad-value gradient of channel d

  for (int it = 0; it < a.iters; ++it) {
    const int win = (it * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool active = win < nwin;
    const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
    __syncthreads();                           // previous iteration's readers are done
    if (active) {
      for (int e = lane; e < NT * DP; e += 64) {
        const int j = e / DP, d = e - j * DP;
        const int pix = tok_pixel(a, b, wy, wx, j);
        const bool dv = d < D;
        const int col = h * D + (dv ? d : 0);
        const long pc = pix >= 0 ? pix : 0;     // unconditional clamped loads, select afterwards
        const float kx = a.k[pc * a.ldk + a.koff + col], kp = a.kpad[col];
        const float vx = a.v[pc * a.ldv + a.voff + col], vp = a.vpad[col];
        const float qx = a.q[pc * a.ldq + a.qoff + col] * a.scale, gx = a.dout[pc * a.lddo + col];
        sK[wave][e] = dv ? (pix >= 0 ? kx : kp) : 0.f;
        sV[wave][e] = dv ? (pix >= 0 ? vx : vp) : 0.f;
        sQ[wave][e] = (dv && pix >= 0) ? qx : 0.f;
        sG[wave][e] = (dv && pix >= 0) ? gx : 0.f;
      }
    }
    __syncthreads();
    const int pix = (active && lane < NT) ? tok_pixel(a, b, wy, wx, lane) : -1;
    // ---- pass A: lane = query row i -> P, dS planes and dQ_i
    if (active && lane < NT) {
      const int i = lane;
      float q[DP], g[DP];
#pragma unroll
      for (int d = 0; d < DP; ++d) { q[d] = Qw[i * DP + d]; g[d] = Gw[i * DP + d]; }
      const int bias0 = (yl + 6) * 13 + (xl + 6);
      float m = -3.0e38f;
#pragma unroll 1
      for (int jy = 0; jy < 7; ++jy) {
#pragma unroll
        for (int jx = 0; jx < 7; ++jx) {
          const int j = jy * 7 + jx;
          const float sv = dot_row<DP>(q, Kw + j * DP, Tw[bias0 - jy * 13 - jx]);
          Pw[j * SP + i] = sv;
          Dw[j * SP + i] = dot_row<DP>(g, Vw + j * DP, 0.f);
          m = fmaxf(m, sv);
        }
      }
      float l = 0.f, acc = 0.f;
#pragma unroll 7
      for (int j = 0; j < NT; ++j) {
        const float p = __expf(Pw[j * SP + i] - m);
        Pw[j * SP + i] = p;
        l += p;
        acc = fmaf(p, Dw[j * SP + i], acc);
      }
      const float inv = 1.0f / l, Dl = acc * inv;
      float dq[DP];
#pragma unroll
      for (int d = 0; d < DP; ++d) dq[d] = 0.f;
#pragma unroll 1
      for (int jy = 0; jy < 7; ++jy) {
#pragma unroll
        for (int jx = 0; jx < 7; ++jx) {
          const int j = jy * 7 + jx;
          const float p = Pw[j * SP + i] * inv;
          const float ds = p * (Dw[j * SP + i] - Dl);
          Pw[j * SP + i] = p;
          Dw[j * SP + i] = ds;
          axpy_row<DP>(dq, ds, Kw + j * DP);
        }
      }
      if (pix >= 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) a.dq[(long)pix * a.lddq + a.dqoff + h * D + d] = dq[d] * a.scale;
      }
    }
    __syncthreads();
    // ---- pass B: lane = key column j -> dK_j, dV_j from the stored planes
    float dk[DP], dv[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    const bool keyl = active && lane < NT;
    if (keyl) {
      const int j = lane;
#pragma unroll 1
      for (int iy = 0; iy < 7; ++iy) {
#pragma unroll
        for (int ix = 0; ix < 7; ++ix) {
          const int i = iy * 7 + ix;
          axpy_row<DP>(dk, Dw[j * SP + i], Qw + i * DP);
          axpy_row<DP>(dv, Pw[j * SP + i], Gw + i * DP);
        }
      }
      if (pix >= 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          a.dk[(long)pix * a.lddk + a.dkoff + h * D + d] = dk[d];
          a.dv[(long)pix * a.lddv + a.dvoff + h * D + d] = dv[d];
        }
      }
    }
    // padded keys: their gradient flows to the projection bias only (wave sums, boundary windows only)
    const bool padl = keyl && pix < 0;
    if (__any(padl)) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float sk = hrf_wave_sum(padl ? dk[d] : 0.f), sv = hrf_wave_sum(padl ? dv[d] : 0.f);
        if (lane == d) { padk += sk; padv += sv; }
      }
    }
    // dRPB[(yi-yj+6)*13 + (xi-xj+6)] += dS[i][j]: gather over the dS plane
    if (active) {
#pragma unroll
      for (int kbin = 0; kbin < 3; ++kbin) {
        const int e = lane + 64 * kbin;
        if (e < 169) {
          const int dy = e / 13 - 6, dx = e - (e / 13) * 13 - 6;
          const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? 6 - dy : 6;
          const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? 6 - dx : 6;
          float sacc = 0.f;
          for (int yj = y0; yj <= y1; ++yj)
            for (int xj = x0; xj <= x1; ++xj)
              sacc += Dw[(yj * 7 + xj) * SP + (yj + dy) * 7 + xj + dx];
          bin[kbin] += sacc;
        }
      }
    }
  }
  // one atomic per bin per wave into the (replicated) parameter-gradient accumulators
  const long cp = (long)((blockIdx.x * WAVES + wave) % HRF_STAT_COPIES) * a.copy_stride;
#pragma unroll
  for (int kbin = 0; kbin < 3; ++kbin) {
    const int e = lane + 64 * kbin;
    if (e < 169) hrf_atomic_add(&a.drpb[cp + e * a.heads + h], bin[kbin]);
  }
  if (lane < D) {
    hrf_atomic_add(&a.dkpad[cp + h * D + lane], padk);
    hrf_atomic_add(&a.dvpad[cp + h * D + lane], padv);
  }
}

inline void window_geom(AttnArgs& a) {
  a.nWh = (a.H + 6) / 7; a.nWw = (a.W + 6) / 7;
  a.pt = (a.nWh * 7 - a.H) / 2; a.pl = (a.nWw * 7 - a.W) / 2;   // centre pad: top/left = pad//2
}

}  // namespace

#define HRF_ATTN_DISPATCH(KERN, WV_SMALL, WV_BIG, GRIDX)                                             \
  switch (D) {                                                                                       \
    case 8:  HRF_LAUNCH((KERN<8, WV_SMALL>), dim3(GRIDX(WV_SMALL), heads), dim3(WV_SMALL * 64), 0, stream, a); break;  \
    case 16: HRF_LAUNCH((KERN<16, WV_SMALL>), dim3(GRIDX(WV_SMALL), heads), dim3(WV_SMALL * 64), 0, stream, a); break; \
    case 18: HRF_LAUNCH((KERN<18, WV_SMALL>), dim3(GRIDX(WV_SMALL), heads), dim3(WV_SMALL * 64), 0, stream, a); break; \
    case 32: HRF_LAUNCH((KERN<32, WV_BIG>), dim3(GRIDX(WV_BIG), heads), dim3(WV_BIG * 64), 0, stream, a); break;       \
    case 39: HRF_LAUNCH((KERN<39, WV_BIG>), dim3(GRIDX(WV_BIG), heads), dim3(WV_BIG * 64), 0, stream, a); break;       \
    default: return HRF_ERR_ARG;                                                                     \
  }

extern "C" int hrf_window_attn_fwd(const float* q, int ldq, int qoff, const float* k, int ldk, int koff,
                                   const float* v, int ldv, int voff, const float* kpad, const float* vpad,
                                   const float* rpb, float* o, int ldo, int B, int H, int W, int C, int heads,
                                   void* stream) {
  if (heads <= 0 || C % heads) return HRF_ERR_ARG;
  const int D = C / heads;
  AttnArgs a{};
  a.q = q; a.ldq = ldq; a.qoff = qoff; a.k = k; a.ldk = ldk; a.koff = koff; a.v = v; a.ldv = ldv; a.voff = voff;
  a.kpad = kpad; a.vpad = vpad; a.rpb = rpb; a.o = o; a.ldo = ldo; a.B = B; a.H = H; a.W = W; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)D);
  window_geom(a);
  const int nwin = B * a.nWh * a.nWw;
  if (nwin <= 0) return HRF_OK;
#define HRF_FWD_GRID(WV) hrf_cdiv(nwin, WV)
  HRF_ATTN_DISPATCH(attn_fwd_kernel, 2, 2, HRF_FWD_GRID)
  return hrf_check_launch();
}

extern "C" int hrf_window_attn_bwd(const float* q, int ldq, int qoff, const float* k, int ldk, int koff,
                                   const float* v, int ldv, int voff, const float* kpad, const float* vpad,
                                   const float* rpb, const float* dout, int lddo,
                                   float* dq, int lddq, int dqoff, float* dk, int lddk, int dkoff,
                                   float* dv, int lddv, int dvoff, float* dkpad, float* dvpad, float* drpb,
                                   long copy_stride, int B, int H, int W, int C, int heads, void* stream) {
  if (heads <= 0 || C % heads) return HRF_ERR_ARG;
  const int D = C / heads;
  AttnArgs a{};
  a.q = q; a.ldq = ldq; a.qoff = qoff; a.k = k; a.ldk = ldk; a.koff = koff; a.v = v; a.ldv = ldv; a.voff = voff;
  a.kpad = kpad; a.vpad = vpad; a.rpb = rpb; a.B = B; a.H = H; a.W = W; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)D);
  a.dout = dout; a.lddo = lddo; a.dq = dq; a.lddq = lddq; a.dqoff = dqoff; a.dk = dk; a.lddk = lddk; a.dkoff = dkoff;
  a.dv = dv; a.lddv = lddv; a.dvoff = dvoff; a.dkpad = dkpad; a.dvpad = dvpad; a.drpb = drpb; a.copy_stride = copy_stride;
  window_geom(a);
  const int nwin = B * a.nWh * a.nWw;
  if (nwin <= 0) return HRF_OK;
  // windows per wave: enough blocks to fill 256 CUs, few enough that the dRPB flush stays cheap
#define HRF_BWD_GRID(WV) bwd_grid(nwin, WV, heads, &a.iters)
  auto bwd_grid = [](int nw, int wv, int hd, int* iters) {
    int gx = hrf_cdiv(nw, wv);
    const int cap = hrf_cdiv(512, hd);
    if (gx > cap) gx = cap;
    *iters = hrf_cdiv(nw, gx * wv);
    return gx;
  };
  HRF_ATTN_DISPATCH(attn_bwd_kernel, 2, 2, HRF_BWD_GRID)
  return hrf_check_launch();
}
