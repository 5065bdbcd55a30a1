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
// Mapping: one wave64 per (window, head); lane i = query row i (49 of 64 lanes active), K/V rows
// of the window are staged once in LDS and read as wave-wide broadcasts (conflict-free); the
// softmax row lives in one lane, so the row max/sum need no cross-lane traffic at all.
#include "hrf_common.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int NT = 49;      // tokens per window (7x7)

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
  float* dkpad; float* dvpad; float* drpb;
  int iters;
};

__device__ __forceinline__ int tok_pixel(const AttnArgs& a, int b, int wy, int wx, int t) {
  const int ty = t / 7, tx = t - 7 * ty;
  const int py = wy * 7 + ty - a.pt, px = wx * 7 + tx - a.pl;
  if ((unsigned)py < (unsigned)a.H && (unsigned)px < (unsigned)a.W) return (b * a.H + py) * a.W + px;
  return -1;
}

template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_fwd_kernel(AttnArgs a) {
  __shared__ float sK[WAVES][NT * D];
  __shared__ float sV[WAVES][NT * D];
  __shared__ float sT[WAVES][176];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = blockIdx.y;
  const int nwin = a.B * a.nWh * a.nWw;
  const int win = blockIdx.x * WAVES + wave;
  const bool active = win < nwin;
  const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
  if (active) {
    for (int e = lane; e < NT * D; e += 64) {
      const int j = e / D, d = e - j * D;
      const int pix = tok_pixel(a, b, wy, wx, j);
      sK[wave][e] = pix >= 0 ? a.k[(long)pix * a.ldk + a.koff + h * D + d] : a.kpad[h * D + d];
      sV[wave][e] = pix >= 0 ? a.v[(long)pix * a.ldv + a.voff + h * D + d] : a.vpad[h * D + d];
    }
    for (int e = lane; e < 169; e += 64) sT[wave][e] = a.rpb[e * a.heads + h];
  }
  __syncthreads();
  const int i = lane;
  const int pix = (active && i < NT) ? tok_pixel(a, b, wy, wx, i) : -1;
  if (pix < 0) return;                       // padded / idle query rows produce no output
  float q[D];
#pragma unroll
  for (int d = 0; d < D; ++d) q[d] = a.q[(long)pix * a.ldq + a.qoff + h * D + d] * a.scale;
  const int yi = i / 7, xi = i - 7 * yi;
  const int bias0 = (yi + 6) * 13 + (xi + 6);
  const float* Kw = sK[wave];
  const float* Vw = sV[wave];
  const float* Tw = sT[wave];
  float m = -3.0e38f;
  for (int j = 0; j < NT; ++j) {
    const int yj = j / 7, xj = j - 7 * yj;
    float s = Tw[bias0 - yj * 13 - xj];
#pragma unroll
    for (int d = 0; d < D; ++d) s = fmaf(q[d], Kw[j * D + d], s);
    m = fmaxf(m, s);
  }
  float l = 0.f, o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  for (int j = 0; j < NT; ++j) {
    const int yj = j / 7, xj = j - 7 * yj;
    float s = Tw[bias0 - yj * 13 - xj];
#pragma unroll
    for (int d = 0; d < D; ++d) s = fmaf(q[d], Kw[j * D + d], s);
    const float p = expf(s - m);
    l += p;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = fmaf(p, Vw[j * D + d], o[d]);
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < D; ++d) a.o[(long)pix * a.ldo + h * D + d] = o[d] * inv;
}

template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_kernel(AttnArgs a) {
  __shared__ float sK[WAVES][NT * D];
  __shared__ float sV[WAVES][NT * D];
  __shared__ float sQ[WAVES][NT * D];
  __shared__ float sG[WAVES][NT * D];        // dO rows
  __shared__ float sT[WAVES][176];
  __shared__ float sdT[WAVES][176];
  __shared__ float sM[WAVES][64], sL[WAVES][64], sDl[WAVES][64];
  __shared__ float sPadK[WAVES][D], sPadV[WAVES][D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = blockIdx.y;
  const int nwin = a.B * a.nWh * a.nWw;
  for (int e = lane; e < 176; e += 64) { sdT[wave][e] = 0.f; sT[wave][e] = e < 169 ? a.rpb[e * a.heads + h] : 0.f; }
  for (int e = lane; e < D; e += 64) { sPadK[wave][e] = 0.f; sPadV[wave][e] = 0.f; }
  const float* Kw = sK[wave];
  const float* Vw = sV[wave];
  const float* Qw = sQ[wave];
  const float* Gw = sG[wave];
  const float* Tw = sT[wave];
  const int li = lane < NT ? lane : 0;
  const int yl = li / 7, xl = li - 7 * yl;

  for (int it = 0; it < a.iters; ++it) {
    const int win = (it * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool active = win < nwin;
    const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
    __syncthreads();                           // previous iteration's readers are done
    if (active) {
      for (int e = lane; e < NT * D; e += 64) {
        const int j = e / D, d = e - j * D;
        const int pix = tok_pixel(a, b, wy, wx, j);
        const int col = h * D + d;
        sK[wave][e] = pix >= 0 ? a.k[(long)pix * a.ldk + a.koff + col] : a.kpad[col];
        sV[wave][e] = pix >= 0 ? a.v[(long)pix * a.ldv + a.voff + col] : a.vpad[col];
        sQ[wave][e] = pix >= 0 ? a.q[(long)pix * a.ldq + a.qoff + col] * a.scale : 0.f;
        sG[wave][e] = pix >= 0 ? a.dout[(long)pix * a.lddo + col] : 0.f;
      }
    }
    __syncthreads();
    const int pix = (active && lane < NT) ? tok_pixel(a, b, wy, wx, lane) : -1;
    // ---- pass A: lane = query row i -> softmax stats, D_i, dQ_i, dRPB
    if (active && lane < NT) {
      const int i = lane;
      float q[D], g[D];
#pragma unroll
      for (int d = 0; d < D; ++d) { q[d] = Qw[i * D + d]; g[d] = Gw[i * D + d]; }
      const int bias0 = (yl + 6) * 13 + (xl + 6);
      float m = -3.0e38f;
      for (int j = 0; j < NT; ++j) {
        const int yj = j / 7, xj = j - 7 * yj;
        float s = Tw[bias0 - yj * 13 - xj];
#pragma unroll
        for (int d = 0; d < D; ++d) s = fmaf(q[d], Kw[j * D + d], s);
        m = fmaxf(m, s);
      }
      float l = 0.f, acc = 0.f;
      for (int j = 0; j < NT; ++j) {
        const int yj = j / 7, xj = j - 7 * yj;
        float s = Tw[bias0 - yj * 13 - xj], dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(q[d], Kw[j * D + d], s); dp = fmaf(g[d], Vw[j * D + d], dp); }
        const float p = expf(s - m);
        l += p; acc = fmaf(p, dp, acc);
      }
      const float inv = 1.0f / l, Dl = acc * inv;
      float dq[D];
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] = 0.f;
      for (int j = 0; j < NT; ++j) {
        const int yj = j / 7, xj = j - 7 * yj;
        const int bidx = bias0 - yj * 13 - xj;
        float s = Tw[bidx], dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(q[d], Kw[j * D + d], s); dp = fmaf(g[d], Vw[j * D + d], dp); }
        const float ds = expf(s - m) * inv * (dp - Dl);
#pragma unroll
        for (int d = 0; d < D; ++d) dq[d] = fmaf(ds, Kw[j * D + d], dq[d]);
        hrf_atomic_add(&sdT[wave][bidx], ds);     // distinct bins across lanes for a fixed j
      }
      sM[wave][i] = m; sL[wave][i] = inv; sDl[wave][i] = Dl;
      if (pix >= 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) a.dq[(long)pix * a.lddq + a.dqoff + h * D + d] = dq[d] * a.scale;
      }
    }
    __syncthreads();
    // ---- pass B: lane = key column j -> dK_j, dV_j
    if (active && lane < NT) {
      const int j = lane;
      float kj[D], vj[D], dk[D], dv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) { kj[d] = Kw[j * D + d]; vj[d] = Vw[j * D + d]; dk[d] = 0.f; dv[d] = 0.f; }
      const int sub = yl * 13 + xl;
      for (int i = 0; i < NT; ++i) {
        const int yi = i / 7, xi = i - 7 * yi;
        float s = Tw[(yi + 6) * 13 + (xi + 6) - sub], dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(Qw[i * D + d], kj[d], s); dp = fmaf(Gw[i * D + d], vj[d], dp); }
        const float p = expf(s - sM[wave][i]) * sL[wave][i];
        const float ds = p * (dp - sDl[wave][i]);
#pragma unroll
        for (int d = 0; d < D; ++d) { dk[d] = fmaf(ds, Qw[i * D + d], dk[d]); dv[d] = fmaf(p, Gw[i * D + d], dv[d]); }
      }
      if (pix >= 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          a.dk[(long)pix * a.lddk + a.dkoff + h * D + d] = dk[d];
          a.dv[(long)pix * a.lddv + a.dvoff + h * D + d] = dv[d];
        }
      } else {                                     // padded key: gradient flows to the projection bias only
#pragma unroll
        for (int d = 0; d < D; ++d) { hrf_atomic_add(&sPadK[wave][d], dk[d]); hrf_atomic_add(&sPadV[wave][d], dv[d]); }
      }
    }
  }
  __syncthreads();
  for (int e = lane; e < 169; e += 64) hrf_atomic_add(&a.drpb[e * a.heads + h], sdT[wave][e]);
  for (int e = lane; e < D; e += 64) {
    hrf_atomic_add(&a.dkpad[h * D + e], sPadK[wave][e]);
    hrf_atomic_add(&a.dvpad[h * D + e], sPadV[wave][e]);
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
  HRF_ATTN_DISPATCH(attn_fwd_kernel, 4, 4, HRF_FWD_GRID)
  return hrf_check_launch();
}

extern "C" int hrf_window_attn_bwd(const float* q, int ldq, int qoff, const float* k, int ldk, int koff,
                                   const float* v, int ldv, int voff, const float* kpad, const float* vpad,
                                   const float* rpb, const float* dout, int lddo,
                                   float* dq, int lddq, int dqoff, float* dk, int lddk, int dkoff,
                                   float* dv, int lddv, int dvoff, float* dkpad, float* dvpad, float* drpb,
                                   int B, int H, int W, int C, int heads, void* stream) {
  if (heads <= 0 || C % heads) return HRF_ERR_ARG;
  const int D = C / heads;
  AttnArgs a{};
  a.q = q; a.ldq = ldq; a.qoff = qoff; a.k = k; a.ldk = ldk; a.koff = koff; a.v = v; a.ldv = ldv; a.voff = voff;
  a.kpad = kpad; a.vpad = vpad; a.rpb = rpb; a.B = B; a.H = H; a.W = W; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)D);
  a.dout = dout; a.lddo = lddo; a.dq = dq; a.lddq = lddq; a.dqoff = dqoff; a.dk = dk; a.lddk = lddk; a.dkoff = dkoff;
  a.dv = dv; a.lddv = lddv; a.dvoff = dvoff; a.dkpad = dkpad; a.dvpad = dvpad; a.drpb = drpb;
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
  HRF_ATTN_DISPATCH(attn_bwd_kernel, 4, 2, HRF_BWD_GRID)
  return hrf_check_launch();
}
