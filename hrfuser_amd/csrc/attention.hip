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
// Mapping: one 256-thread block per (window, head); all contractions run on v_mfma_f32_16x16x4_f32 with the 49 tokens
// padded to 64 (attn_fwd_mfma_kernel / attn_bwd_mfma_kernel below).
#include "hrf_common.h"
#include "hrf_group.h"
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

constexpr int SP = 51;      // pitch of the [key j][query i < 49] dS plane of the backward kernel (odd)

// Staging for the MFMA kernels: [64][P] tiles of Q (scaled), K, V (and dO), zero outside the 49 x D payload.  All global
// loads of a thread are issued before the first LDS store (one round trip instead of one per loop iteration).
template <int D, int P, bool BWD, int ROWS = 64>
__device__ __forceinline__ void stage_tiles(const AttnArgs& a, int b, int wy, int wx, int h, float* sQ, float* sK, float* sV,
                                            float* sG) {
  for (int e = threadIdx.x; e < ROWS * P; e += 256) {
    const int j = e / P, d = e - j * P;
    if (j >= NT || d >= D) { sQ[e] = 0.f; sK[e] = 0.f; sV[e] = 0.f; if (BWD) sG[e] = 0.f; }
  }
  constexpr int NE = (NT * D + 255) / 256;
  float kx[NE], vx[NE], qx[NE], gx[NE];
  int dst[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = threadIdx.x + 256 * u;
    const int ec = e < NT * D ? e : 0;
    const int j = ec / D, d = ec - j * D;
    const int pix = tok_pixel(a, b, wy, wx, j);
    const long pc = pix >= 0 ? pix : 0;           // unconditional clamped loads, select afterwards
    const int col = h * D + d;
    const float k0 = a.k[pc * a.ldk + a.koff + col], kp = a.kpad[col];
    const float v0 = a.v[pc * a.ldv + a.voff + col], vp = a.vpad[col];
    const float q0 = a.q[pc * a.ldq + a.qoff + col];
    const float g0 = BWD ? a.dout[pc * a.lddo + col] : 0.f;
    kx[u] = pix >= 0 ? k0 : kp;
    vx[u] = pix >= 0 ? v0 : vp;
    qx[u] = pix >= 0 ? q0 * a.scale : 0.f;
    gx[u] = pix >= 0 ? g0 : 0.f;
    dst[u] = e < NT * D ? j * P + d : -1;
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    if (dst[u] >= 0) {
      sK[dst[u]] = kx[u]; sV[dst[u]] = vx[u]; sQ[dst[u]] = qx[u];
      if (BWD) sG[dst[u]] = gx[u];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA forward: both contractions of a (window, head) on v_mfma_f32_16x16x4_f32.  The 49 tokens are padded to 64; wave w
// owns queries 16w..16w+15.  The score tile is produced TRANSPOSED (row operand = key rows of K, column operand = the
// wave's query rows of Q): lane (i, q) then holds S[query 16w+i][keys 16t + 4q + r], i.e. one query COLUMN per lane
// with its 64 keys spread over the four lanes of equal i - the softmax row reduction is 16 in-register values plus two
// wave shuffles (xor 16, 32), and the probabilities already sit in the A-operand position of the second contraction
// (contraction slot q of MFMA (t, r) <-> key 16t + 4q + r), so P never moves: O = P V is 16 x ceil(D/16) MFMAs whose B
// operand is a V row read from LDS.  K/V/Q tiles are staged once per block (pitch 16*ceil(D/16) + 1).
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(HrfGroup<AttnArgs> grp) {
  const AttnArgs& a = grp.sel();
  constexpr int KS = (D + 3) / 4;             // contraction steps of Q K^T
  constexpr int DT = (D + 15) / 16;           // 16-wide output tiles of P V
  constexpr int P = DT * 16 + 1;              // LDS pitch (floats)
  __shared__ float sQ[64 * P];
  __shared__ float sK[64 * P];
  __shared__ float sV[64 * P];
  __shared__ float sT[176];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = blockIdx.y;
  const int i = lane & 15, q = lane >> 4;
  const int win = blockIdx.x;
  const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
  stage_tiles<D, P, false>(a, b, wy, wx, h, sQ, sK, sV, nullptr);
  for (int e = threadIdx.x; e < 169; e += 256) sT[e] = a.rpb[e * a.heads + h];
  __syncthreads();

  // ---- S^T tiles: acc[t][r] = S[query 16w + i][key 16t + 4q + r]
  hrf_f4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
  const float* qrow = sQ + (16 * wave + i) * P + q;
  const float* krow = sK + i * P + q;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const float qv = qrow[4 * kk];              // (columns D .. 16*DT-1 are zero in LDS)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float kv = krow[16 * t * P + 4 * kk];
      acc[t] = hrf_mfma16(kv, qv, acc[t]);
    }
  }
  // ---- relative position bias, padding of the key axis, softmax over the query's row
  const int qi = 16 * wave + i;
  const int qc = qi < NT ? qi : 0;
  const int yi = qc / 7, xi = qc - 7 * yi;
  float m = -3.0e38f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * t + 4 * q + r;
      const int jc = j < NT ? j : 0;
      const int yj = jc / 7, xj = jc - 7 * yj;
      const float sv = j < NT ? acc[t][r] + sT[(yi - yj + 6) * 13 + (xi - xj + 6)] : -3.0e38f;
      acc[t][r] = sv;
      m = fmaxf(m, sv);
    }
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = __expf(acc[t][r] - m);          // exp(-huge) == 0 for the 15 padding keys
      acc[t][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  // ---- O = P V: contraction slot q of MFMA (t, r) is key 16t + 4q + r - the lane's own probability
  hrf_f4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* vrow = sV + (16 * t + 4 * q + r) * P + i;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = hrf_mfma16(acc[t][r], vrow[16 * dt], o[dt]);
    }
  // o[dt][r] = unnormalised O[query 16w + 4q + r][d = 16 dt + i]; the row sums live in the lanes with i = query % 16
  const float inv = 1.0f / l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float invr = __shfl(inv, 4 * q + r);
    const int qo = 16 * wave + 4 * q + r;
    const int px = qo < NT ? tok_pixel(a, b, wy, wx, qo) : -1;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = 16 * dt + i;
      if (px >= 0 && d < D) a.o[(long)px * a.ldo + h * D + d] = o[dt][r] * invr;
    }
  }
}

// MFMA backward.  Every 49x49 quantity is produced twice, once per orientation, because an MFMA result can feed the
// next contraction without moving only as the operand whose contraction index is spread over the lane quartets:
//   query-column orientation (wave = 16 queries; lane (i, q) holds X[query 16w+i][key 16t+4q+r]): scores -> softmax
//     statistics (m, 1/l), dP = dO V^T, D = sum_j P dP, dS; contraction over KEYS: dQ = dS K;
//   key-column orientation (wave = 16 keys; lane holds X[query 16t+4q+r][key 16w+i]): P and dS again from the stored
//     row statistics; contraction over QUERIES: dV = P^T dO, dK = dS^T Q - complete rows per wave, no cross-wave sums.
// 176 + MFMAs per wave instead of ~1200 dependent FMAs per lane.  dS also goes to an LDS plane for the relative-
// position-bias gather; dK / dV rows pass through LDS so that the padded keys of boundary windows are summed in the
// block before they reach the projection-bias gradients.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(HrfGroup<AttnArgs> grp) {
  const AttnArgs& a = grp.sel();
  constexpr int KS = (D + 3) / 4, DT = (D + 15) / 16, P = DT * 16 + 1;
  // 49-ROW tiles (round 5): the MFMA fragments of token tile 3 read rows 48 .. 63, of which only row 48 exists - those reads are
  // clamped to row 48 (finite values; every product they enter is masked: P = dS = 0 for tokens >= 49).  At head_dim 39
  // (HRFuser-B) the block's LDS drops 65 -> 50 KB: three blocks per CU, the 1 288 (window, head) blocks of the 96x160 map run in two
  // rounds instead of three
  __shared__ float sQ[NT * P];
  __shared__ float sK[NT * P];
  __shared__ float sV[NT * P];
  __shared__ float sG[NT * P];                                      // dO rows
  __shared__ float sD[NT * SP];                                     // dS[key][query] plane (relative position bias)
  // dK / dV rows reuse the K / V tiles once the last MFMA has read them: 48 KB of LDS per block at D = 18, i.e. three
  // resident blocks per CU - the 644 windows of the 96x160 map then run in one round instead of two
  float* sX = sK;                                                   // dK rows [49][DT*16]
  float* sY = sV;                                                   // dV rows
  __shared__ float sT[176];
  __shared__ float sM[64], sIL[64], sDl[64];
  __shared__ int sPad[64];                                          // key j is a padded (out-of-image) token
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = blockIdx.y;
  const int i = lane & 15, q = lane >> 4;
  const int win = blockIdx.x;
  const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
  stage_tiles<D, P, true, NT>(a, b, wy, wx, h, sQ, sK, sV, sG);
  const int wrow = min(16 * wave + i, NT - 1);                      // the wave's own token row, clamped (see above)
  for (int e = threadIdx.x; e < 169; e += 256) sT[e] = a.rpb[e * a.heads + h];
  if (threadIdx.x < 64) sPad[threadIdx.x] = (threadIdx.x < NT && tok_pixel(a, b, wy, wx, threadIdx.x) < 0) ? 1 : 0;
  __syncthreads();

  // ================= query-column orientation: wave = queries 16w .. 16w+15
  {
    hrf_f4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { s[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; dp[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; }
    const float* qrow = sQ + wrow * P + q;
    const float* grow = sG + wrow * P + q;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const float qv = qrow[4 * kk], gv = grow[4 * kk];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int tr = t == 3 ? NT - 1 : 16 * t + i;                   // (tile 3: rows 48 .. 63 -> 48)
        s[t] = hrf_mfma16(sK[tr * P + 4 * kk + q], qv, s[t]);
        dp[t] = hrf_mfma16(sV[tr * P + 4 * kk + q], gv, dp[t]);
      }
    }
    const int qi = 16 * wave + i, qc = qi < NT ? qi : 0;
    const int yi = qc / 7, xi = qc - 7 * yi;
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * t + 4 * q + r, jc = j < NT ? j : 0;
        const int yj = jc / 7, xj = jc - 7 * yj;
        const float sv = j < NT ? s[t][r] + sT[(yi - yj + 6) * 13 + (xi - xj + 6)] : -3.0e38f;
        s[t][r] = sv;
        m = fmaxf(m, sv);
      }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[t][r] = __expf(s[t][r] - m); l += s[t][r]; }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    float Dl = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[t][r] *= inv; Dl = fmaf(s[t][r], dp[t][r], Dl); }
    Dl += __shfl_xor(Dl, 16);
    Dl += __shfl_xor(Dl, 32);
    if (q == 0) { sM[qi] = m; sIL[qi] = inv; sDl[qi] = Dl; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = s[t][r] * (dp[t][r] - Dl);
        s[t][r] = ds;
        const int j = 16 * t + 4 * q + r;
        if (j < NT && qi < NT) sD[j * SP + qi] = ds;
      }
    // dQ = dS K (contraction over keys: slot q of MFMA (t, r) is key 16t + 4q + r)
    hrf_f4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* krow = sK + (t == 3 ? NT - 1 : 16 * t + 4 * q + r) * P + i;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = hrf_mfma16(s[t][r], krow[16 * dt], o[dt]);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qo = 16 * wave + 4 * q + r;
      const int px = qo < NT ? tok_pixel(a, b, wy, wx, qo) : -1;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + i;
        if (px >= 0 && d < D) a.dq[(long)px * a.lddq + a.dqoff + h * D + d] = o[dt][r] * a.scale;
      }
    }
  }
  __syncthreads();
  // ================= key-column orientation: wave = keys 16w .. 16w+15
  {
    hrf_f4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { s[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; dp[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; }
    const float* krow = sK + wrow * P + q;
    const float* vrow = sV + wrow * P + q;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const float kv = krow[4 * kk], vv = vrow[4 * kk];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int tr = t == 3 ? NT - 1 : 16 * t + i;
        s[t] = hrf_mfma16(sQ[tr * P + 4 * kk + q], kv, s[t]);                  // S[query 16t+4q+r][key 16w+i]
        dp[t] = hrf_mfma16(sG[tr * P + 4 * kk + q], vv, dp[t]);
      }
    }
    const int kj = 16 * wave + i, kc = kj < NT ? kj : 0;
    const int yj = kc / 7, xj = kc - 7 * yj;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = 16 * t + 4 * q + r, qc = qi < NT ? qi : 0;
        const int yi = qc / 7, xi = qc - 7 * yi;
        const bool ok = kj < NT && qi < NT;
        const float p = ok ? __expf(s[t][r] + sT[(yi - yj + 6) * 13 + (xi - xj + 6)] - sM[qc]) * sIL[qc] : 0.f;
        s[t][r] = p;
        dp[t][r] = ok ? p * (dp[t][r] - sDl[qc]) : 0.f;                         // dS
      }
    // dV = P^T dO, dK = dS^T Q (contraction over queries: slot q of MFMA (t, r) is query 16t + 4q + r)
    hrf_f4 ov[DT], ok_[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { ov[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f}; ok_[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tr = t == 3 ? NT - 1 : 16 * t + 4 * q + r;
        const float* gr = sG + tr * P + i;
        const float* qr = sQ + tr * P + i;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          ov[dt] = hrf_mfma16(s[t][r], gr[16 * dt], ov[dt]);
          ok_[dt] = hrf_mfma16(dp[t][r], qr[16 * dt], ok_[dt]);
        }
      }
    __syncthreads();                                                            // every wave is done reading sK / sV
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ko = 16 * wave + 4 * q + r;                                     // result row = key
      if (ko < NT) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          sX[ko * (DT * 16) + 16 * dt + i] = ok_[dt][r];
          sY[ko * (DT * 16) + 16 * dt + i] = ov[dt][r];
        }
      }
    }
  }
  __syncthreads();
  const long cp = (long)(blockIdx.x % HRF_STAT_COPIES) * a.copy_stride;
  for (int e = threadIdx.x; e < NT * D; e += 256) {
    const int jj = e / D, d = e - jj * D;
    const int px = tok_pixel(a, b, wy, wx, jj);
    if (px >= 0) {
      a.dk[(long)px * a.lddk + a.dkoff + h * D + d] = sX[jj * (DT * 16) + d];
      a.dv[(long)px * a.lddv + a.dvoff + h * D + d] = sY[jj * (DT * 16) + d];
    }
  }
  // padded keys (boundary windows only): their gradient flows to the projection bias.  Wave w sums keys 13w .. 13w+12
  // (lane = channel); one atomic per wave and channel into the replicated accumulators
  if (wy == 0 || wx == 0 || wy == a.nWh - 1 || wx == a.nWw - 1) {
    if (lane < D) {
      const int d = lane;
      float pk = 0.f, pv = 0.f;
      bool anypad = false;
      const int j1 = min(NT, 13 * wave + 13);
      for (int jj = 13 * wave; jj < j1; ++jj) {
        if (sPad[jj] != 0) {
          pk += sX[jj * (DT * 16) + d];
          pv += sY[jj * (DT * 16) + d];
          anypad = true;
        }
      }
      if (anypad) {
        hrf_atomic_add(&a.dkpad[cp + h * D + d], pk);
        hrf_atomic_add(&a.dvpad[cp + h * D + d], pv);
      }
    }
  }
  // dRPB[(yi-yj+6)*13 + (xi-xj+6)] += dS[i][j]: gather over the dS plane.  169 bins x 4 row groups of the key grid
  // (wave-sized work items instead of 169 threads walking up to 49 entries each), partial sums merged in LDS,
  // one atomic per bin and block
  {
    float* sBin = sX;                                               // (dK rows are stored by now)
    __syncthreads();
    for (int it = threadIdx.x; it < 169 * 4; it += 256) {
      const int e = it % 169, part = it / 169;
      const int dy = e / 13 - 6, dx = e - (e / 13) * 13 - 6;
      const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? 6 - dy : 6;
      const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? 6 - dx : 6;
      float sacc = 0.f;
      for (int yj = y0 + part; yj <= y1; yj += 4)
        for (int xj = x0; xj <= x1; ++xj)
          sacc += sD[(yj * 7 + xj) * SP + (yj + dy) * 7 + xj + dx];
      sBin[part * 176 + e] = sacc;
    }
    __syncthreads();
    if (threadIdx.x < 169) {
      const int e = threadIdx.x;
      hrf_atomic_add(&a.drpb[cp + e * a.heads + h], (sBin[e] + sBin[176 + e]) + (sBin[352 + e] + sBin[528 + e]));
    }
  }
}

inline void window_geom(AttnArgs& a) {
  a.nWh = (a.H + 6) / 7; a.nWw = (a.W + 6) / 7;
  a.pt = (a.nWh * 7 - a.H) / 2; a.pl = (a.nWw * 7 - a.W) / 2;   // centre pad: top/left = pad//2
}

}  // namespace

#define HRF_ATTN_DISPATCH(KERN)                                                              \
  switch (D) {                                                                               \
    case 8:  HRF_LAUNCH_G((KERN<8>), dim3(nwin, heads), dim3(256), 0, stream, a); break;       \
    case 16: HRF_LAUNCH_G((KERN<16>), dim3(nwin, heads), dim3(256), 0, stream, a); break;      \
    case 18: HRF_LAUNCH_G((KERN<18>), dim3(nwin, heads), dim3(256), 0, stream, a); break;      \
    case 32: HRF_LAUNCH_G((KERN<32>), dim3(nwin, heads), dim3(256), 0, stream, a); break;      \
    case 39: HRF_LAUNCH_G((KERN<39>), dim3(nwin, heads), dim3(256), 0, stream, a); break;      \
    default: return HRF_ERR_ARG;                                                             \
  }

extern "C" int hrf_window_attn_fwd(const float* q, int ldq, int qoff, const float* k, int ldk, int koff,
                                   const float* v, int ldv, int voff, const float* kpad, const float* vpad,
                                   const float* rpb, float* o, int ldo, int B, int H, int W, int C, int heads,
                                   void* stream) {
  HRF_GROUP_CALL();
  if (heads <= 0 || C % heads) return HRF_ERR_ARG;
  const int D = C / heads;
  AttnArgs a{};
  a.q = q; a.ldq = ldq; a.qoff = qoff; a.k = k; a.ldk = ldk; a.koff = koff; a.v = v; a.ldv = ldv; a.voff = voff;
  a.kpad = kpad; a.vpad = vpad; a.rpb = rpb; a.o = o; a.ldo = ldo; a.B = B; a.H = H; a.W = W; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)D);
  window_geom(a);
  const int nwin = B * a.nWh * a.nWw;
  if (nwin <= 0) return HRF_OK;
  HRF_ATTN_DISPATCH(attn_fwd_mfma_kernel)
  return hrf_check_launch();
}

extern "C" int hrf_window_attn_bwd(const float* q, int ldq, int qoff, const float* k, int ldk, int koff,
                                   const float* v, int ldv, int voff, const float* kpad, const float* vpad,
                                   const float* rpb, const float* dout, int lddo,
                                   float* dq, int lddq, int dqoff, float* dk, int lddk, int dkoff,
                                   float* dv, int lddv, int dvoff, float* dkpad, float* dvpad, float* drpb,
                                   long copy_stride, int B, int H, int W, int C, int heads, void* stream) {
  HRF_GROUP_CALL();
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
  HRF_ATTN_DISPATCH(attn_bwd_mfma_kernel)
  return hrf_check_launch();
}
