// Fused window-attention block (gfx950, fp32): the WHOLE token-local half of an HRFormerBlock / fusion block in ONE
// launch per direction, one 256-thread workgroup per 7x7 window:
//
//   forward   x' = res (+ res2) + drop( out_proj( softmax(q k^T d^-1/2 + RPB) v ) ),   q = LN_q(xq) Wq^T + bq,
//                                                                                      k,v = LN_kv(xkv) W{k,v}^T + b{k,v}
//             and, optionally, the head of the CrossFFN that follows:  h1 = LN_2(x') W1^T + b1  (+ BatchNorm moments of h1)
//
// Replaces, for one HRFormerBlock (hrformer.py:365-373): norm1 (:343) -> LocalWindowSelfAttention.forward (:184-236:
// centre zero-pad, window partition) -> WindowMSA.forward (:96-131: qkv Linear, q k^T + relative position bias, softmax,
// attn v, out_proj) -> window merge / de-pad -> residual add (:369) -> norm2 (:351) -> CrossFFN.layers[0] (:268, the 1x1
// expansion convolution);  for one modality of a fusion block (hrfuser_hrformer_based.py:305-317): norm1[k] / norm2[k]
// (:279-280) -> MultiWindowCrossAttention.forward (:189-248) -> WindowMCA.forward (:106-151: q/k/v Linear, attention,
// out_proj, Dropout) -> DropPath + the two residual adds (:311-313) and, after the last modality, norm3 (:291) + the
// CrossFFN head.  Everything is local to a window except the CrossFFN's BatchNorm, so nothing but x' (and h1) ever
// leaves the chip: q / k / v / the attention output live in LDS, LayerNorm rows and softmax rows in registers.
//
// Layout: four token-major LDS tiles [64][C+1] (49 tokens padded to 64): X (source rows -> LayerNorm'd rows -> x' -> LN_2
// rows), Q (scaled q -> attention output), K, V.  Every GEMM is "rows of W x this wave's 16 tokens" on
// v_mfma_f32_16x16x4_f32: the A fragment comes straight from the (L2-resident) weight matrix with one 16-byte load per
// four MFMAs (k permuted identically on both operands, as in lin_engine.hip), the B fragment from the LDS tile; wave w
// owns tokens 16w..16w+15 for the projections and queries 16w..16w+15 for the attention core (which is the MFMA core of
// attention.hip reading head h at column offset h*D).  Tokens outside the image are exact zeros AFTER LayerNorm in the
// reference, i.e. their q/k/v equal the projection biases: the LDS rows are zeroed, the GEMM does the rest.
#include "hrf_common.h"
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int NTOK = 49;

__device__ float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};

// p[idx] when ok, else 0 - as an UNCONDITIONAL load from a clamped index followed by a select.  `ok ? p[idx] : 0.f` with a
// per-lane condition keeps the load inside an exec-masked branch with its own s_waitcnt: one DEPENDENT LDS round trip per
// element (round 5: the 18-channel backward kernel carried 404 such branches and 834 wait points).  LLVM re-creates the branch
// from the select (it sinks a load whose only use is one arm of a select) unless the loaded value passes through something it
// cannot move: HRF_KEEP, an empty non-volatile asm (no instruction, no ordering, free to schedule).
#ifdef HRF_EMUL
#define HRF_KEEP(x) ((void)0)
#define AB_FENCE() ((void)0)
#else
#define HRF_KEEP(x) asm("" : "+v"(x))
// scheduling fence: the machine scheduler may not move anything across (it hoists the LDS loads of several independent small
// GEMMs in front of the first one - 72 live registers for the three dn GEMMs of the self-attention epilogue - and spills)
#define AB_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
__device__ __forceinline__ float ldz(const float* p, int idx, bool ok) {
  float v = p[ok ? idx : 0];
  HRF_KEEP(v);
  return ok ? v : 0.f;
}

// 4 consecutive elements p[off..off+3], `nvalid` of them exist (<= 0: none); full = the whole 16-wide group row is valid
__device__ __forceinline__ hrf_f4 ld_sel(bool full, const float* p, long off, int nvalid) {
  if (full) return hrf_ld4(nvalid > 0 ? p + off : g_zero4);
  hrf_f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = *(e < nvalid ? p + off + e : g_zero4);
  return r;
}

__device__ __forceinline__ int ab_tok_pixel(const hrf_attn_block_t& a, int b, int wy, int wx, int t) {
  const int ty = t / 7, tx = t - 7 * ty;
  const int py = wy * 7 + ty - a.pt, px = wx * 7 + tx - a.pl;
  if ((unsigned)py < (unsigned)a.H && (unsigned)px < (unsigned)a.W) return (b * a.H + py) * a.W + px;
  return -1;
}

// A 16-deep contraction step issues four MFMAs, MFMA r taking k = 16 s + 4 q + r of lane group q.  With K = 18 (36 + 2 ...) the
// last step's r >= K - 16 s touch nothing but the zero padding for EVERY q: skipped (2 of 8 MFMAs per tile at 18 channels).
// Needs the step index at compile time: the loops that use it are fully unrolled or tested on the unrolled body.
#ifndef AB_NO_SKIPK
#define AB_SKIPK(kmin, K) ((kmin) >= (K))
#else
#define AB_SKIPK(kmin, K) false
#endif
// acc[t] (t < NT) += W[n0 + 16t + i][.] . rows[tok0 + j][.] over K: D[n][token], lane (j, q) ends up holding the four
// output channels n0 + 16t + 4q + r of token tok0 + j.  W is [N][K] row-major in global memory, rows an LDS tile.
template <int K, int NT>
__device__ __forceinline__ void wave_gemm(const float* W, int n0, int N, const float* rows, int pitch, int tok0, int lane,
                                          hrf_f4* acc) {
  const int i = lane & 15, q = lane >> 4;
  constexpr int NS = (K + 15) / 16;
  const float* brow = rows + (tok0 + i) * pitch;
#pragma unroll 2
  for (int s = 0; s < NS; ++s) {
    const int kbase = 16 * s + 4 * q, kval = K - kbase;
    const bool kfull = 16 * (s + 1) <= K;
    hrf_f4 wv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + i;
      wv[t] = ld_sel(kfull, W, (long)n * K + kbase, n < N ? kval : 0);
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = ldz(brow, kbase + r, r < kval);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (NS <= 2 && AB_SKIPK(16 * s + r, K)) continue;     // every k = 16s + 4q + r of this step lies beyond K: exact zeros (NS <= 2: s is a compile-time value)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = hrf_mfma16(wv[t][r], bv[r], acc[t]);
    }
  }
}

// bias rows into the accumulators (D = A*B + C): acc[t][r] = bias[n0 + 16t + 4q + r]
template <int NT>
__device__ __forceinline__ void acc_bias(const float* bias, int n0, int N, int lane, hrf_f4* acc) {
  const int q = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int nb = n0 + 16 * t + 4 * q;
    acc[t] = ld_sel(n0 + 16 * (t + 1) <= N, bias, nb, bias != nullptr ? N - nb : 0);
  }
}

// acc_bias from a staged LDS vector: acc[t] = sB[n0 + 16t + 4q .. +3] (zero beyond lim, a multiple of 4)
template <int NT>
__device__ __forceinline__ void acc_bias_l(const float* sB, int n0, int lim, int lane, hrf_f4* acc) {
  const int q = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int nb = n0 + 16 * t + 4 * q;
    const hrf_f4 v = hrf_ld4(sB + (nb < lim ? nb : 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = nb < lim ? v[r] : 0.f;
  }
}

// Stage the 49 rows of a window (zeros for tokens outside the image and for rows 49..63) and LayerNorm them in place.
// Four lanes cooperate on one token; all global loads of a thread are issued before its first LDS store.
// `tail` (uniform): the rows are FORMED here - x = x[.] + rs * GELU(sTs[c] * traw[.] + sTs[C + c]), the CrossFFN tail of the
// preceding block (hrformer.py:371-372) - and written to xout for the residual add of this launch and for the backward.
template <int C>
__device__ __forceinline__ void stage_ln_rows(const hrf_attn_block_t& a, const float* x, const float* gam, const float* bet,
                                              int b, int wy, int wx, float* sX, const int* sPix,
                                              const float* traw = nullptr, const float* sTs = nullptr, float rs = 1.f,
                                              float* xout = nullptr) {
  constexpr int PC = C + 1;
  constexpr int NE = (NTOK * C + 255) / 256;
  float v[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = threadIdx.x + 256 * u;
    const int ec = e < NTOK * C ? e : 0;
    const int j = ec / C, c = ec - j * C;
    const int pix = sPix[j];
    const float x0 = x[(long)(pix >= 0 ? pix : 0) * C + c];
    v[u] = pix >= 0 ? x0 : 0.f;
  }
  if (traw != nullptr) {                                            // (uniform)
    float rw[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = threadIdx.x + 256 * u;
      const int ec = e < NTOK * C ? e : 0;
      const int j = ec / C, c = ec - j * C;
      const int pix = sPix[j];
      rw[u] = traw[(long)(pix >= 0 ? pix : 0) * C + c];
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = threadIdx.x + 256 * u;
      const int ec = e < NTOK * C ? e : 0;
      const int j = ec / C, c = ec - j * C;
      const int pix = sPix[j];
      const float xv = v[u] + rs * hrf_gelu(fmaf(rw[u], sTs[c], sTs[C + c]));
      v[u] = pix >= 0 ? xv : 0.f;
      if (e < NTOK * C && pix >= 0) xout[(long)pix * C + c] = xv;
    }
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (e < NTOK * C) { const int j = e / C; sX[j * PC + (e - j * C)] = v[u]; }
  }
  __syncthreads();
  const int t = threadIdx.x >> 2, part = threadIdx.x & 3;           // token, quarter of its channels
  const float* row = sX + t * PC;
  float s = 0.f;
  for (int c = part; c < C; c += 4) s += row[c];
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
  const float mean = s * (1.0f / (float)C);
  float q = 0.f;
  for (int c = part; c < C; c += 4) { const float d = row[c] - mean; q = fmaf(d, d, q); }
  q += __shfl_xor(q, 1); q += __shfl_xor(q, 2);
  const float rstd = hrf_rsqrt_nr(q * (1.0f / (float)C) + a.ln_eps);
  const bool real = sPix[t] >= 0;                                   // (rows 49..63 and out-of-image tokens stay zero)
  float* wrow = sX + t * PC;
  for (int c = part; c < C; c += 4) { const float y = fmaf((row[c] - mean) * rstd, gam[c], bet[c]); wrow[c] = real ? y : 0.f; }
}

// ---- weights staged in LDS (widths 18 / 36): every GEMM of the block then reads both operands from LDS and the only
// global round trip of a workgroup is the batch of loads at its start.  Tile layout [N][PW], PW = K rounded up to 4
// (16-byte rows), pad columns zero.
template <int K, int PW, int NTHR = 256>
__device__ __forceinline__ void stage_weight(const float* W, int N, float* sW) {
  const int n4 = N * (PW / 4);
  for (int e = threadIdx.x; e < n4; e += NTHR) {
    const int n = e / (PW / 4), kb = 4 * (e - n * (PW / 4));
    const hrf_f4 v = ld_sel(kb + 4 <= K, W, (long)n * K + kb, K - kb);
    hrf_st4(sW + n * PW + kb, v);
  }
}

// wave_gemm with the weight tile in LDS
template <int K, int PW, int NT>
__device__ __forceinline__ void wave_gemm_l(const float* sW, int n0, int N, const float* brow, int lane, hrf_f4* acc) {
  const int i = lane & 15, q = lane >> 4;
  constexpr int NS = (K + 15) / 16;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int kbase = 16 * s + 4 * q;
    const bool kin = kbase < PW;
    hrf_f4 wv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + i;
      const hrf_f4 w = hrf_ld4(sW + (n < N ? n : 0) * PW + (kin ? kbase : 0));
      const bool ok = kin && n < N;
#pragma unroll
      for (int r = 0; r < 4; ++r) wv[t][r] = ok ? w[r] : 0.f;
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = ldz(brow, kbase + r, kbase + r < K);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (AB_SKIPK(16 * s + r, K)) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = hrf_mfma16(wv[t][r], bv[r], acc[t]);
    }
  }
}

// wave_gemm_t with the weight tile in LDS: acc[t] += sum_n brow[n] * sW[n][k0 + 16t + i]
template <int N, int PW, int NT>
__device__ __forceinline__ void wave_gemm_tl(const float* sW, int K, const float* brow, int lane, hrf_f4* acc) {
  const int i = lane & 15, q = lane >> 4;
  constexpr int NS = (N + 15) / 16;
#ifndef HRF_TL_UNROLL
#define HRF_TL_UNROLL 2
#endif
#pragma unroll HRF_TL_UNROLL
  for (int s = 0; s < NS; ++s) {
    const int nbase = 16 * s + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (NS <= 2 && AB_SKIPK(16 * s + r, N)) continue;     // contraction index 16s + 4q + r >= N for every q
      const bool nv = nbase + r < N;
      const float bv = ldz(brow, nbase + r, nv);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int k = 16 * t + i;
        float w = sW[(nv ? nbase + r : 0) * PW + (k < K ? k : 0)];
        HRF_KEEP(w);
        acc[t] = hrf_mfma16((nv && k < K) ? w : 0.f, bv, acc[t]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------- forward
struct AbFwdArgs {
  hrf_attn_block_t a;
  hrf_bn_fin_t fin;     // BatchNorm of the preceding block's CrossFFN tail finalised on load (stats != null)
};

// K-steps of the token contractions whose four rows (tokens 16t + 4q + r, q = 0..3) all lie beyond the 49 tokens of a window:
// t = 3, r = 1..3 (tokens 49 ... 63).  Their operands are exact zeros (P and dS of pad keys / queries): skipped, 13 of 16 steps remain
#define AB_PADK(t, r) ((t) == 3 && (r) > 0)
template <int C, int HEADS>
__global__ __launch_bounds__(256) void attn_block_fwd_kernel(HrfGroup<AbFwdArgs> grp) {
  const hrf_attn_block_t& a = grp.sel().a;
  const hrf_bn_fin_t& tfin = grp.sel().fin;
  constexpr int D = C / HEADS, PC = C + 1, CT = (C + 15) / 16, PW = (C + 3) & ~3;
  constexpr int KSD = (D + 3) / 4, DT = (D + 15) / 16;
  constexpr int TILE = 64 * PC;
  constexpr bool WL = C <= 36;            // weights staged in LDS (8*C*PW floats in front of the tiles)
  HRF_DYN_SMEM(float, smem);
  float* sW1 = smem;                      // [4C][PW] w1, [C][PW] wo / wq / wk / wv  (WL only)
  float* sWo = sW1 + 4 * C * PW;
  float* sWq = sWo + C * PW;
  float* sWk = sWq + C * PW;
  float* sWv = sWk + C * PW;
  float* sX = smem + (WL ? 8 * C * PW : 0);   // source rows -> LayerNorm'd rows -> x' -> LN_2(x')
  float* sQ = sX + TILE;                  // scaled q -> attention output o
  float* sK = sQ + TILE;
  float* sV = sK + TILE;                  // (+32 floats of slack behind it: V fragments are read 16 columns wide)
  float* sT = sV + TILE + 32;             // [HEADS][176] relative position bias of every head
  float* sStat = sK;                      // [4][2][4C] BatchNorm moments of h1 per wave: aliases K / V once attention is done
  __shared__ int sPix[64];
  __shared__ float sTs[2 * C];            // scale | shift of the tail's BatchNorm
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  const int win = blockIdx.x;
  const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
  const bool tail = a.tail_raw != nullptr;
  if (tid < 64) sPix[tid] = tid < NTOK ? ab_tok_pixel(a, b, wy, wx, tid) : -1;
  if (tail) {
    if (tfin.stats != nullptr) hrf_bn_fin_onload(tfin, sTs, sTs + C, tid, 256, blockIdx.x == 0);
    else for (int e = tid; e < C; e += 256) { sTs[e] = a.tail_scale[e]; sTs[C + e] = a.tail_shift[e]; }
  }
  for (int e = tid; e < 64 * PC; e += 256) sX[e] = 0.f;
  for (int e = tid; e < HEADS * 176; e += 256) {
    const int h = e / 176, k = e - h * 176;
    sT[e] = k < 169 ? a.rpb[k * HEADS + h] : 0.f;
  }
  if (tid < 32) sV[TILE + tid] = 0.f;
  if (tid < 64) { sQ[tid * PC + C] = 0.f; sK[tid * PC + C] = 0.f; sV[tid * PC + C] = 0.f; }   // pitch column: read (masked) by the last k-step
  if (WL) {                                                         // the block's weights: issued with the first batch of loads
    if (a.w1 != nullptr) stage_weight<C, PW>(a.w1, 4 * C, sW1);
    stage_weight<C, PW>(a.wo, C, sWo);
    stage_weight<C, PW>(a.wq, C, sWq);
    stage_weight<C, PW>(a.wk, C, sWk);
    stage_weight<C, PW>(a.wv, C, sWv);
  }
  __syncthreads();
  const int tok0 = 16 * wave;

  // ---- projections: q from the query source, k / v from the key-value source (the same rows for self-attention)
  if (tail) stage_ln_rows<C>(a, a.tail_res, a.lnq_g, a.lnq_b, b, wy, wx, sX, sPix, a.tail_raw, sTs,
                             a.tail_rowscale != nullptr ? a.tail_rowscale[b] : 1.f, a.x_out);
  else stage_ln_rows<C>(a, a.xq, a.lnq_g, a.lnq_b, b, wy, wx, sX, sPix);
  __syncthreads();
  {
    hrf_f4 acc[CT];
    acc_bias<CT>(a.bq, 0, C, lane, acc);
    if (WL) wave_gemm_l<C, PW, CT>(sWq, 0, C, sX + (tok0 + i) * PC, lane, acc);
    else wave_gemm<C, CT>(a.wq, 0, C, sX, PC, tok0, lane, acc);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int n = 16 * t + 4 * q + r; if (n < C) sQ[(tok0 + i) * PC + n] = acc[t][r] * a.scale; }
  }
  if (a.xkv != a.xq) {                                               // (uniform) cross-attention: re-stage the tile
    __syncthreads();
    stage_ln_rows<C>(a, a.xkv, a.lnkv_g, a.lnkv_b, b, wy, wx, sX, sPix);
    __syncthreads();
  }
  {
    hrf_f4 acc[CT];
    acc_bias<CT>(a.bk, 0, C, lane, acc);
    if (WL) wave_gemm_l<C, PW, CT>(sWk, 0, C, sX + (tok0 + i) * PC, lane, acc);
    else wave_gemm<C, CT>(a.wk, 0, C, sX, PC, tok0, lane, acc);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int n = 16 * t + 4 * q + r; if (n < C) sK[(tok0 + i) * PC + n] = acc[t][r]; }
    acc_bias<CT>(a.bv, 0, C, lane, acc);
    if (WL) wave_gemm_l<C, PW, CT>(sWv, 0, C, sX + (tok0 + i) * PC, lane, acc);
    else wave_gemm<C, CT>(a.wv, 0, C, sX, PC, tok0, lane, acc);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int n = 16 * t + 4 * q + r; if (n < C) sV[(tok0 + i) * PC + n] = acc[t][r]; }
  }
  __syncthreads();

  // ---- attention core per head (attention.hip's MFMA formulation on the packed tiles): S^T tiles, softmax over the
  // query's 49 keys in registers, O = P V with P already in A-operand position; o overwrites the wave's own q rows
  {
    const int qi = tok0 + i, qc = qi < NTOK ? qi : 0;
    const int yi = qc / 7, xi = qc - 7 * yi;
#pragma unroll 1
    for (int h = 0; h < HEADS; ++h) {
      hrf_f4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
      const float* qrow = sQ + (tok0 + i) * PC + h * D + q;
      const float* krow = sK + i * PC + h * D + q;
#pragma unroll
      for (int kk = 0; kk < KSD; ++kk) {
        const float qv = ldz(qrow, 4 * kk, 4 * kk + q < D);          // columns >= D belong to the next head
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = hrf_mfma16(krow[16 * t * PC + 4 * kk], qv, acc[t]);
      }
      const float* bias = sT + h * 176;
      float m = -3.0e38f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * t + 4 * q + r, jc = j < NTOK ? j : 0;
          const int yj = jc / 7, xj = jc - 7 * yj;
          float bj = bias[(yi - yj + 6) * 13 + (xi - xj + 6)];
          HRF_KEEP(bj);
          const float sv = j < NTOK ? acc[t][r] + bj : -3.0e38f;
          acc[t][r] = sv;
          m = fmaxf(m, sv);
        }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float l = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = __expf(acc[t][r] - m); acc[t][r] = p; l += p; }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      hrf_f4 o[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (AB_PADK(t, r)) continue;
          const float* vrow = sV + (16 * t + 4 * q + r) * PC + h * D + i;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[dt] = hrf_mfma16(acc[t][r], vrow[16 * dt], o[dt]);
        }
      const float inv = hrf_rcp(l);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float invr = __shfl(inv, 4 * q + r);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = 16 * dt + i;
          if (d < D) sQ[(tok0 + 4 * q + r) * PC + h * D + d] = o[dt][r] * invr;
        }
      }
    }
  }

  // ---- out_proj + dropout / droppath + residual(s): x' leaves as 16-byte stores, and stays in the accumulators
  HRF_WAVE_SYNC();                                                  // o rows were stored by other lanes of this wave
  const int tok = tok0 + i;
  const int pix = sPix[tok];
  const long pc = pix >= 0 ? pix : 0;
  hrf_f4 xo[CT];
  acc_bias<CT>(a.bo, 0, C, lane, xo);
  if (WL) wave_gemm_l<C, PW, CT>(sWo, 0, C, sQ + (tok0 + i) * PC, lane, xo);
  else wave_gemm<C, CT>(a.wo, 0, C, sQ, PC, tok0, lane, xo);
  {
    const float rs = (a.rowscale != nullptr ? a.rowscale[pc / a.rows_per_sample] : 1.f) * a.mscale;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int nb = 16 * t + 4 * q, nval = C - nb;
      const bool nfull = 16 * (t + 1) <= C;
      const hrf_f4 r1 = ld_sel(nfull, a.res, pc * C + nb, nval);
      const hrf_f4 r2 = ld_sel(nfull, a.res2, pc * C + nb, a.res2 != nullptr ? nval : 0);
      const hrf_f4 mk = ld_sel(nfull, a.mask, pc * C + nb, a.mask != nullptr ? nval : 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = xo[t][r] * rs;
        if (a.mask != nullptr) y *= mk[r];
        xo[t][r] = (r1[r] + r2[r]) + y;
      }
      if (pix >= 0) {
        float* dst = a.out + pc * C + nb;
        if (nfull) hrf_st4(dst, xo[t]);
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r < nval) dst[r] = xo[t][r];
        }
      }
    }
  }
  if (a.w1 == nullptr && a.out_rowstat == nullptr) return;          // (uniform)

  // ---- LayerNorm statistics of the x' row (its C channels live in lanes j, j+16, j+32, j+48), two-pass as ln_stats
  float sm = 0.f;
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) sm += (16 * t + 4 * q + r < C) ? xo[t][r] : 0.f;
  sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
  const float mu = sm * (1.0f / (float)C);
  float sq = 0.f;
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float dd = (16 * t + 4 * q + r < C) ? xo[t][r] - mu : 0.f; sq = fmaf(dd, dd, sq); }
  sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
  const float rstd = hrf_rsqrt_nr(sq * (1.0f / (float)C) + a.out_eps);
  if (a.out_rowstat != nullptr && q == 0 && pix >= 0) { a.out_rowstat[2 * pc] = mu; a.out_rowstat[2 * pc + 1] = rstd; }
  if (a.w1 == nullptr) return;                                      // (uniform)

  // ---- CrossFFN head: h1 = LN_2(x') W1^T + b1 over the wave's own rows of the X tile, BatchNorm moments of h1
  __syncthreads();                                                  // every wave is done with K / V (sStat aliases them)
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int nb = 16 * t + 4 * q;
    const hrf_f4 g2 = ld_sel(16 * (t + 1) <= C, a.ln2_g, nb, C - nb), b2 = ld_sel(16 * (t + 1) <= C, a.ln2_b, nb, C - nb);
#pragma unroll
    for (int r = 0; r < 4; ++r) if (nb + r < C) sX[tok * PC + nb + r] = fmaf((xo[t][r] - mu) * rstd, g2[r], b2[r]);
  }
  HRF_WAVE_SYNC();                                                  // a token's row was stored by four lanes of this wave
  constexpr int N1 = 4 * C, FT = (N1 / 16 >= 9) ? 9 : (N1 + 15) / 16;  // hidden width; 16-channel tiles per pass
  const bool tokv = pix >= 0;
#pragma unroll 1
  for (int n0 = 0; n0 < N1; n0 += 16 * FT) {
    hrf_f4 acc[FT];
    acc_bias<FT>(a.b1, n0, N1, lane, acc);
    if (WL) wave_gemm_l<C, PW, FT>(sW1, n0, N1, sX + tok * PC, lane, acc);
    else wave_gemm<C, FT>(a.w1, n0, N1, sX, PC, tok0, lane, acc);
#pragma unroll
    for (int t = 0; t < FT; ++t) {
      const int nb = n0 + 16 * t + 4 * q;
      if (tokv && nb < N1) hrf_st4(a.h1 + pc * N1 + nb, acc[t]);     // (4C is a multiple of 4: whole groups)
      if (a.stats1 != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = tokv ? acc[t][r] : 0.f;
          const float s1 = hrf_row16_sum(v), s2 = hrf_row16_sum(v * v);
          if (i == 0 && nb + r < N1) { sStat[(wave * 2 + 0) * N1 + nb + r] = s1; sStat[(wave * 2 + 1) * N1 + nb + r] = s2; }
        }
      }
    }
  }
  if (a.stats1 != nullptr) {
    __syncthreads();
    double* st = a.stats1 + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * N1;
    for (int e = tid; e < 2 * N1; e += 256) {
      const int which = e / N1, ch = e - which * N1;
      const float s = (sStat[(0 * 2 + which) * N1 + ch] + sStat[(1 * 2 + which) * N1 + ch]) +
                      (sStat[(2 * 2 + which) * N1 + ch] + sStat[(3 * 2 + which) * N1 + ch]);
      hrf_atomic_add(&st[which * N1 + ch], (double)s);
    }
  }
}

template <int C, int HEADS>
int launch_fwd(const hrf_attn_block_t& a0, const hrf_bn_fin_t& fin, int nwin, void* stream) {
  AbFwdArgs a;
  a.a = a0; a.fin = fin;
  constexpr size_t smem = ((size_t)(C <= 36 ? 8 * C * ((C + 3) & ~3) : 0) + 4 * 64 * (C + 1) + 32 + HEADS * 176) * sizeof(float);
#ifndef HRF_EMUL
  static std::atomic<unsigned> lds_set{0u};
  if (hrf_dyn_lds_once(lds_set, reinterpret_cast<const void*>(&attn_block_fwd_kernel<C, HEADS>), (int)smem) != HRF_OK) return HRF_ERR_LAUNCH;
#endif
  return HRF_LAUNCH_G((attn_block_fwd_kernel<C, HEADS>), dim3(nwin), dim3(256), (unsigned)smem, stream, a);
}


// ---------------------------------------------------------------------------------------------------------- backward
// acc[t] (t < NT) += sum_n rows[tok0 + j][n] * W[n][k0 + 16t + i]: D[k][token] = the adjoint of wave_gemm (contraction over
// the OUTPUT channels n < N of the Linear, W [N][K] row-major); lane (j, q) ends up with input channels k0+16t+4q+r.
template <int N, int NT>
__device__ __forceinline__ void wave_gemm_t(const float* W, int K, int k0, const float* rows, int pitch, int tok0, int lane,
                                            hrf_f4* acc) {
  const int i = lane & 15, q = lane >> 4;
  constexpr int NS = (N + 15) / 16;
  const float* brow = rows + (tok0 + i) * pitch;
#pragma unroll 2
  for (int s = 0; s < NS; ++s) {
    const int nbase = 16 * s + 4 * q;
    float bv[4], wv[NT][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool nv = nbase + r < N;
      bv[r] = ldz(brow, nbase + r, nv);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int k = k0 + 16 * t + i;
        wv[t][r] = *((nv && k < K) ? W + (long)(nbase + r) * K + k : g_zero4);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (NS <= 2 && AB_SKIPK(16 * s + r, N)) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = hrf_mfma16(wv[t][r], bv[r], acc[t]);
    }
  }
}

// Weight-gradient tile sums over the 49 tokens of the window: acc[tk] += sum_tok A[tok][n0 + i] * B(tok)[k0 + 16 tk + j],
// D[n][k]; A / B are LDS tiles, B optionally mapped through a LayerNorm affine (gam / bet in LDS; rows flagged by realf[tok]
// == 0 give 0: tokens outside the image).  Lane (j, q) ends up with dW[n0 + 4q + r][k0 + 16 tk + j].
template <int NTK, bool AFF>
__device__ __forceinline__ void wave_tgemm(const float* sA, int pitchA, int n0, int N, const float* sB, int pitchB, int k0,
                                           int K, const float* gam, const float* bet, const float* realf, int lane, hrf_f4* acc) {
  const int i = lane & 15, q = lane >> 4;
  const bool nv = n0 + i < N;
// (rolled, every token step was a dependent LDS round trip in front of its MFMAs: 13 steps x ~150 cycles per call, four calls
// per wave; unrolled by 4 the workgroup's program went 50.0 -> 46.3 us at 18 channels, 72.2 -> 64.6 us at 36 - a full unroll
// adds nothing: tools/time_ab_phases.py)
#ifndef HRF_TG_UNROLL
#define HRF_TG_UNROLL 4
#endif
#pragma unroll HRF_TG_UNROLL
  for (int tt = 0; tt < 13; ++tt) {
    const int tok = 4 * tt + q;
    const bool tv = tok < NTOK;
    const int tc = tv ? tok : 0;
    const float av = ldz(sA, tc * pitchA + n0 + i, tv && nv);
#pragma unroll
    for (int tk = 0; tk < NTK; ++tk) {
      const int k = k0 + 16 * tk + i;
      const int kc = k < K ? k : 0;
      float bvv = sB[tc * pitchB + kc];
      if (AFF) { float aff = fmaf(bvv, gam[kc], bet[kc]); HRF_KEEP(aff); bvv = realf[tc] != 0.f ? aff : 0.f; }
      else HRF_KEEP(bvv);
      acc[tk] = hrf_mfma16(av, (tv && k < K) ? bvv : 0.f, acc[tk]);
    }
  }
}

// store a finished dW tile: slot[(n0 + 4q + r) * K + k0 + 16 tk + j]
template <int NTK>
__device__ __forceinline__ void store_wtile(float* dst, int n0, int N, int k0, int K, int lane, const hrf_f4* acc) {
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int tk = 0; tk < NTK; ++tk) {
    const int k = k0 + 16 * tk + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * q + r;
      if (n < N && k < K) dst[(long)n * K + k] = acc[tk][r];
    }
  }
}

// stage raw rows of a window, replace them by xhat = (x - mean) * rstd (zeros for tokens outside the image), keep rstd
template <int C>
__device__ __forceinline__ void stage_xhat_rows(const hrf_attn_block_t& a, const float* x, float eps, float* sX, float* sRs,
                                                const int* sPix) {
  constexpr int PC = C + 1;
  constexpr int NE = (NTOK * C + 255) / 256;
  float v[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = threadIdx.x + 256 * u;
    const int ec = e < NTOK * C ? e : 0;
    const int j = ec / C, c = ec - j * C;
    const int pix = sPix[j];
    const float x0 = x[(long)(pix >= 0 ? pix : 0) * C + c];
    v[u] = pix >= 0 ? x0 : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (e < NTOK * C) { const int j = e / C; sX[j * PC + (e - j * C)] = v[u]; }
  }
  __syncthreads();
  const int t = threadIdx.x >> 2, part = threadIdx.x & 3;
  float* row = sX + t * PC;
  float s = 0.f;
  for (int c = part; c < C; c += 4) s += row[c];
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
  const float mean = s * (1.0f / (float)C);
  float qq = 0.f;
  for (int c = part; c < C; c += 4) { const float d = row[c] - mean; qq = fmaf(d, d, qq); }
  qq += __shfl_xor(qq, 1); qq += __shfl_xor(qq, 2);
  const float rstd = hrf_rsqrt_nr(qq * (1.0f / (float)C) + eps);
  const bool real = sPix[t] >= 0;
  for (int c = part; c < C; c += 4) { const float y = (row[c] - mean) * rstd; row[c] = real ? y : 0.f; }
  if (part == 0) sRs[t] = real ? rstd : 0.f;
}

// LayerNorm backward of one token row held in registers (dn[t][r] = d/d(LN output channel 16t+4q+r) of token tok0 + j):
// returns dx in place, stores (sum dn*xhat, sum dn) over the wave's tokens into sPar[0..C) / sPar[C..2C).  Tokens outside
// the image (real == false) carry a constant 0 instead of a LayerNorm output: no gradient, no parameter contribution.
template <int C, int CT>
__device__ __forceinline__ void ln_bwd_rows(hrf_f4* dn, const float* sXh, int pitch, int tok, bool real, float rstd,
                                            const float* gam, float* sPar, int lane) {
  const int i = lane & 15, q = lane >> 4;
  float xh[CT][4], g[CT][4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * t + 4 * q + r;
      const bool kv = k < C && real;
      if (!kv) dn[t][r] = 0.f;
      const int kc = k < C ? k : 0;
      float xv = sXh[tok * pitch + kc], gm = gam[kc];
      HRF_KEEP(xv); HRF_KEEP(gm);
      xh[t][r] = kv ? xv : 0.f;
      g[t][r] = kv ? dn[t][r] * gm : 0.f;
      s1 += g[t][r]; s2 = fmaf(g[t][r], xh[t][r], s2);
    }
  s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
  const float m1 = s1 * (1.0f / (float)C), m2 = s2 * (1.0f / (float)C);
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * t + 4 * q + r;
      const float d = dn[t][r];
      const float pg = hrf_row16_sum(d * xh[t][r]), pb = hrf_row16_sum(d);
      if (i == 0 && k < C) { sPar[k] = pg; sPar[C + k] = pb; }
      dn[t][r] = rstd * (g[t][r] - m1 - xh[t][r] * m2);
    }
}

// phase stamps of workgroup 100 for tools/time_ab_phases.py: compiled in only with -DHRF_AB_TIMING (the stamps overwrite
// the first rows of out_rowstat)
#if defined(HRF_AB_TIMING) && !defined(HRF_EMUL)
#define AB_T(k) do { if (a.out_rowstat != nullptr && blockIdx.x == 100 && threadIdx.x == 0) reinterpret_cast<long long*>(a.out_rowstat)[k] = wall_clock64(); } while (0)
#else
#define AB_T(k)
#endif
struct AbBwdArgs {
  hrf_attn_block_t a;
  hrf_bn_bfin_t bf;     // BatchNorm-backward coefficients of the CrossFFN head derived on load (gstats != null)
};

// Two ROLES per window (round 5).  A workgroup of this kernel used to live ~46 us at 18 channels and a launch lasts as long as
// one workgroup (every window of a 96 x 160 map is resident at once; a workgroup ALONE on a CU takes the same time): what a
// launch costs is the LENGTH of the dependent chain inside a window, not throughput.  That chain carried two kinds of work
// that do not depend on each other - the data gradient (dy1 W1 -> LN_2 backward -> dy -> dO -> attention backward -> dq/dk/dv
// W -> LN_1 backward) and the weight gradients of the window (token contractions over tiles the data path has finished) - and
// an attention core whose two orientations (query columns: softmax statistics, O, dS, dQ; key columns: P and dS again, dV,
// dK) only meet at the row statistics.  Waves 0-3 (role A, wave w = tokens / queries 16w..16w+15) run the data-gradient chain
// and the query-column orientation; NWB more waves (role B) run the key-column orientation beside it and the weight-gradient
// tiles as soon as their operands are final.  Both roles pass the same __syncthreads() sequence (one hardware barrier per
// workgroup).  NWB = 4 (36 channels: one workgroup per CU, 8 waves = 2 per SIMD) or 2 (18 channels: three workgroups per CU
// must stay resident for the 644 windows of the 96 x 160 map, 18 waves = 5 per SIMD = 96 registers; eight waves would need 80
// and spill).  FFN / CROSS / TAIL are compile-time: every instantiation allocates registers for its own path only.
#ifndef AB_MINW18
#define AB_MINW18 5
#endif
#ifndef AB_NWB18
#define AB_NWB18 4          // role-B waves of the 18-channel kernel (0: four waves run both roles; 4: eight waves at 80 registers)
#endif

template <int C, int HEADS, int NWB, bool FFN, bool CROSS, bool TAIL>
__global__ __launch_bounds__(64 * (4 + NWB), (NWB == 0 ? (C <= 18 ? 3 : 1) : (C <= 18 ? (NWB == 4 ? 6 : 5) : 2))) void attn_block_bwd_kernel(HrfGroup<AbBwdArgs> grp) {
  const hrf_attn_block_t& a = grp.sel().a;
  const hrf_bn_bfin_t& bf = grp.sel().bf;
  constexpr int D = C / HEADS, PC = C + 1, CT = (C + 15) / 16, PW = (C + 3) & ~3;
  constexpr int KSD = (D + 3) / 4, DT = (D + 15) / 16;
  constexpr int TILE = 64 * PC, N1 = 4 * C, PH = N1 + 1;
  constexpr bool W1A = N1 * PW <= 2 * TILE;   // w1 fits the (dy, dO) tiles, which are written only after its last use
  constexpr int NT = 64 * (4 + NWB);                                // threads of the workgroup
  // GXP: role A parks its gradient row gx (8 registers that live from the prologue to the LayerNorm_1 epilogue) in memory
  // behind dy / dO and fetches it back in ONE batch in front of the epilogue.  At 80 registers (three 8-wave workgroups per CU)
  // the allocator spilled exactly these values and reloaded them one at a time, each a dependent L2 round trip: 54 us
  // instead of ~30 for the 18-channel window
  constexpr bool GXP = NWB != 0 && C <= 18;
  constexpr int NBW = NWB == 0 ? 4 : NWB, NB = 64 * NBW;            // waves / threads that carry role B (NWB = 0: all four, after A)
  constexpr bool ffn = FFN, cross = CROSS, tail = TAIL;
  HRF_DYN_SMEM(float, smem);
  float* sWo = smem;                      // [C][PW] out_proj
  float* sWq = sWo + C * PW;              // [C][PW] q / k / v projections
  float* sWk = sWq + C * PW;
  float* sWv = sWk + C * PW;
  float* sX = sWv + C * PW + (W1A ? 0 : N1 * PW);   // xhat of the query source (LN_q)
  float* sDY = sX + TILE;                 // dy = d/d(out_proj output) rows
  float* sW1 = W1A ? sDY : sWv + C * PW;  // [4C][PW] CrossFFN expansion weight (see W1A)
  float* sG = sDY + TILE;                 // dO rows (gradient of the attention output)
  float* sO = sG + TILE;                  // xhat of x' (LN_2) first, then the recomputed attention output
  float* sQ = sO + TILE;                  // scaled q
  float* sK = sQ + TILE;                  // k, later dk
  float* sV = sK + TILE;                  // v, later dv
  float* sDQ = sV + TILE;                 // dq rows (already multiplied by the q scale)
  float* sH = sQ;                         // [64][PH] dy1 rows of the CrossFFN head: dead before q / k / v / dq are written
  float* sXkv = sDQ + TILE + 32;          // xhat of the key/value source (cross-attention only)
  __shared__ int sPix[64];
  __shared__ float sReal[64], sRs2[64], sRsQ[64], sRsKV[64], sM[64], sIL[64], sDl[64];
  __shared__ float sT[HEADS * 176];
  __shared__ float sGam[3][C], sBet[3][C];                 // LN_2, LN_q, LN_kv affine parameters
  __shared__ float sPar[4][3][2 * C];                      // per role-A wave: (sum dn*xhat | sum dn) of LN_2, LN_q, LN_kv
  __shared__ float sCo[3 * (4 * C)];                       // BatchNorm-backward coefficients of h1 (cA | cB | cC)
  __shared__ float sTs[2 * C];                             // scale | shift of the preceding block's tail BatchNorm
  __shared__ float sTst[4][2 * C];                         // per role-A wave: (sum tail_du | sum tail_du * tail_raw)
  __shared__ __attribute__((aligned(16))) float sB3[3][(C + 3) & ~3];   // bq | bk | bv (pads zero)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // NWB = 0: four waves, each runs its role-A part and then its role-B part between the same barriers (18 channels: three
  // workgroups per CU must stay resident, which leaves 8 waves 80 registers and 6 waves 96 - both spill)
  const bool roleA = NWB == 0 || wave < 4, roleB = NWB == 0 || wave >= 4;   // (wave-uniform: scalar branches)
  const int wb = NWB == 0 ? wave : wave - 4, lt = NWB == 0 ? tid : tid - 256;   // role B: wave / thread index inside the role
  const int i = lane & 15, q = lane >> 4;
  const int win = blockIdx.x;
  const int wx = win % a.nWw, wy = (win / a.nWw) % a.nWh, b = win / (a.nWw * a.nWh);
  float* slot = a.pslot + (long)blockIdx.x * a.slot_stride;
  if (tid < 64) {
    const int px = tid < NTOK ? ab_tok_pixel(a, b, wy, wx, tid) : -1;
    sPix[tid] = px; sReal[tid] = px >= 0 ? 1.f : 0.f;
  }
  AB_T(0);
  // role A: wave w owns tokens 16w .. 16w+15 (rows of every per-token GEMM, queries of the attention core)
  const int tok0 = 16 * (wave & 3), tok = tok0 + i;
  const int pix = tok < NTOK ? ab_tok_pixel(a, b, wy, wx, tok) : -1;   // (== sPix[tok]; computed so that no load waits for LDS)
  const long pc = pix >= 0 ? pix : 0;
  const bool tokv = pix >= 0;

  // ---- the first batch of global loads goes out BEFORE the tiles are zeroed: incoming gradients (role A: the data path),
  // source rows, dy1 operands, the scalars of later phases; the q / k / v / out weights too where they are one 16-byte
  // group per thread (WE)
  hrf_f4 gx[CT];                                                    // d/d(out row), this lane's 4-channel groups (role A)
#pragma unroll
  for (int t = 0; t < CT; ++t) gx[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
  float rs_out = 1.f;
  if (roleA) {
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int nb = 16 * t + 4 * q;
      gx[t] = ld_sel(16 * (t + 1) <= C, a.gout, pc * C + nb, tokv ? C - nb : 0);
    }
    rs_out = (a.rowscale != nullptr ? a.rowscale[pc / a.rows_per_sample] : 1.f) * a.mscale;
  }
  constexpr bool EARLY = FFN && C <= 18;                            // register budget: dy1 operands join the first batch
  constexpr int NHG = EARLY ? (NTOK * (N1 / 4) + NT - 1) / NT : 1;
  hrf_f4 hdu[NHG], hh1[NHG];
  if (EARLY) {
#pragma unroll
    for (int u = 0; u < NHG; ++u) {
      const int e = tid + NT * u, ec = e < NTOK * (N1 / 4) ? e : 0;
      const int j = ec / (N1 / 4), n = 4 * (ec - j * (N1 / 4));
      const int px = ab_tok_pixel(a, b, wy, wx, j);
      const unsigned pp = px >= 0 ? (unsigned)px * N1 + n : (unsigned)n;
      hdu[u] = hrf_ld4(a.du1 + pp); hh1[u] = hrf_ld4(a.h1 + pp);
    }
  }
  // raw rows of x' (-> sO), the query source (-> sX) and the key/value source (-> sXkv)
  constexpr int NE = (NTOK * C + NT - 1) / NT;
  float v2[NE], vq[NE], vk[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = tid + NT * u, ec = e < NTOK * C ? e : 0;
    const int j = ec / C, c = ec - j * C;
    const int px = ab_tok_pixel(a, b, wy, wx, j);
    const unsigned o = (unsigned)(px >= 0 ? px : 0) * C + c;
    v2[u] = ffn ? a.out[o] : 0.f; vq[u] = a.xq[o]; vk[u] = cross ? a.xkv[o] : 0.f;
    if (px < 0) { v2[u] = 0.f; vq[u] = 0.f; vk[u] = 0.f; }
  }
  constexpr int WG4 = C * (PW / 4);                                 // 16-byte groups of one [C][PW] weight
  constexpr bool WE = 4 * WG4 <= 2 * NT;                            // at most two groups per thread cover wo, wq, wk, wv
  constexpr int WPT = WE ? (4 * WG4 + NT - 1) / NT : 1;
  hrf_f4 wreg[WPT];
  if (WE) {
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int e = tid + NT * u, ec = e < 4 * WG4 ? e : 0;
      const int which = ec / WG4, g = ec - which * WG4;
      const int n = g / (PW / 4), kb = 4 * (g - n * (PW / 4));
      const float* wp = which == 0 ? a.wo : (which == 1 ? a.wq : (which == 2 ? a.wk : a.wv));
      wreg[u] = ld_sel(kb + 4 <= C, wp, (long)n * C + kb, C - kb);
    }
  }
  // zero: the tiles whose pad rows / columns are read (rows 49..63, pitch column)
  for (int e = tid; e < 8 * TILE + 32 + (cross ? TILE : 0); e += NT) sX[e] = 0.f;
  __syncthreads();

  AB_T(1);
  // ---- weights, parameters
  if (ffn) stage_weight<C, PW, NT>(a.w1, N1, sW1);
  if (WE) {
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int e = tid + NT * u;
      if (e < 4 * WG4) {
        const int which = e / WG4, g = e - which * WG4;
        const int n = g / (PW / 4), kb = 4 * (g - n * (PW / 4));
        hrf_st4(sWo + which * (C * PW) + n * PW + kb, wreg[u]);       // (sWo, sWq, sWk, sWv are consecutive)
      }
    }
  } else {
    stage_weight<C, PW, NT>(a.wo, C, sWo);
    stage_weight<C, PW, NT>(a.wq, C, sWq);
    stage_weight<C, PW, NT>(a.wk, C, sWk);
    stage_weight<C, PW, NT>(a.wv, C, sWv);
  }
  for (int e = tid; e < HEADS * 176; e += NT) { const int h = e / 176, k = e - h * 176; sT[e] = k < 169 ? a.rpb[k * HEADS + h] : 0.f; }
  for (int e = tid; e < C; e += NT) {
    sGam[0][e] = ffn ? a.ln2_g[e] : 0.f; sBet[0][e] = ffn ? a.ln2_b[e] : 0.f;
    sGam[1][e] = a.lnq_g[e]; sBet[1][e] = a.lnq_b[e];
    sGam[2][e] = a.lnkv_g[e]; sBet[2][e] = a.lnkv_b[e];
  }
  for (int e = tid; e < 3 * PW; e += NT) {                          // q / k / v biases of the recomputed projections
    const int which = e / PW, c = e - which * PW;
    const float* bp = which == 0 ? a.bq : (which == 1 ? a.bk : a.bv);
    sB3[which][c] = *((bp != nullptr && c < C) ? bp + c : g_zero4);
  }
  for (int e = tid; e < 4 * 3 * 2 * C; e += NT) (&sPar[0][0][0])[e] = 0.f;
  if (tail) for (int e = tid; e < C; e += NT) { sTs[e] = a.tail_scale[e]; sTs[C + e] = a.tail_shift[e]; }
  if (ffn && roleB) {
    // (role B only: the fp64 moments of the on-load finalize - 2 x 8 copies per channel - are never live beside role A's row
    // of incoming gradients; the kernel's register allocation is the larger of the two paths)
    if (bf.gstats != nullptr) hrf_bn_bfin_onload(bf, sCo, sCo + N1, sCo + 2 * N1, lt, NB, blockIdx.x == 0);
    else for (int e = lt; e < N1; e += NB) { sCo[e] = a.cA1[e]; sCo[N1 + e] = a.cB1[e]; sCo[2 * N1 + e] = a.cC1[e]; }
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = tid + NT * u;
    if (e < NTOK * C) {
      const int j = e / C, o = j * PC + (e - j * C);
      sO[o] = v2[u]; sX[o] = vq[u];
      if (cross) sXkv[o] = vk[u];
    }
  }
  AB_T(2);
  __syncthreads();                                                  // sCo, raw rows
  if (ffn) {
    // dy1 rows: BatchNorm backward applied while staging (sH aliases the q / k / v / dq tiles)
    if (EARLY) {
#pragma unroll
      for (int u = 0; u < NHG; ++u) {
        const int e = tid + NT * u;
        if (e < NTOK * (N1 / 4)) {
          const int j = e / (N1 / 4), n = 4 * (e - j * (N1 / 4));
          const bool pv = sPix[j] >= 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float dy1 = fmaf(sCo[n + r], hdu[u][r], fmaf(sCo[N1 + n + r], hh1[u][r], sCo[2 * N1 + n + r]));
            HRF_KEEP(dy1);
            sH[j * PH + n + r] = pv ? dy1 : 0.f;
          }
        }
      }
    } else {
      for (int e = tid; e < NTOK * (N1 / 4); e += NT) {
        const int j = e / (N1 / 4), n = 4 * (e - j * (N1 / 4));
        const int px = sPix[j];
        const long pp = px >= 0 ? px : 0;
        const hrf_f4 du = hrf_ld4(a.du1 + pp * N1 + n), hr = hrf_ld4(a.h1 + pp * N1 + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float dy1 = fmaf(sCo[n + r], du[r], fmaf(sCo[N1 + n + r], hr[r], sCo[2 * N1 + n + r]));
          HRF_KEEP(dy1);
          sH[j * PH + n + r] = px >= 0 ? dy1 : 0.f;
        }
      }
    }
  }
  {
    // xhat rows in place (zeros for tokens outside the image) + rstd: 4 lanes per token, 256 threads per tile.  NWB = 4: role A
    // takes the LN_2 tile (and the key/value source of a cross block), role B the query source; NWB = 2: role A takes all
    const int t = (tid & 255) >> 2, part = tid & 3;
    const bool real = sPix[t] >= 0;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
      if (which == 0 && !ffn) continue;
      if (which == 2 && !cross) continue;
      if (NWB == 4 ? ((which == 1) == roleA) : !roleA) continue;    // (uniform per wave; NWB = 0: every wave is A)
      float* row = (which == 0 ? sO : (which == 1 ? sX : sXkv)) + t * PC;
      const float eps = which == 0 ? a.out_eps : a.ln_eps;
      float s = 0.f;
      for (int c = part; c < C; c += 4) s += row[c];
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
      const float mean = s * (1.0f / (float)C);
      float qq = 0.f;
      for (int c = part; c < C; c += 4) { const float d = row[c] - mean; qq = fmaf(d, d, qq); }
      qq += __shfl_xor(qq, 1); qq += __shfl_xor(qq, 2);
      const float rstd = hrf_rsqrt_nr(qq * (1.0f / (float)C) + eps);
      for (int c = part; c < C; c += 4) { const float y = (row[c] - mean) * rstd; row[c] = real ? y : 0.f; }
      if (part == 0) (which == 0 ? sRs2 : (which == 1 ? sRsQ : sRsKV))[t] = real ? rstd : 0.f;
    }
  }
  __syncthreads();

  // Every phase re-derives its lane coordinates from threadIdx.x behind an opaque copy (and the token's pixel from LDS): kept alive
  // across the whole kernel they are ~12 registers per lane that the allocator, at 80 registers, spilled one by one and
  // reloaded one by one - every reload a dependent L2 round trip on a path whose cost IS its dependent latency
#define AB_LANE()                                                                                   \
  int tid_ = threadIdx.x; HRF_KEEP(tid_);                                                           \
  const int lane = tid_ & 63, i = lane & 15, q = lane >> 4, tok0 = 16 * ((tid_ >> 6) & 3), tok = tok0 + i; \
  (void)lane; (void)i; (void)q; (void)tok0; (void)tok
#define AB_PIX()                                                                                    \
  const int pix = sPix[tok]; const long pc = pix >= 0 ? pix : 0; const bool tokv = pix >= 0;        \
  (void)pc; (void)tokv
  AB_T(3);
  // ================================================================================================================ roles
  // The phases below are written once (lambdas) and stitched together three ways: NWB = 0 - four waves run A then B between
  // the same barriers; NWB > 0 - ONE top-level scalar branch, role A's waves run only A phases and role B's only B phases, each
  // arm with the same __syncthreads() sequence (a hardware barrier counts arrivals, not call sites).  Each role keeps its
  // state in variables of its own: with the roles interleaved in one control flow (round-5 first form) every long-lived value
  // of A - the gradient row gx, the 8 score / dP tiles across the statistics barrier - was live through B's code as well and
  // the register allocation was the SUM of the two (60 - 95 spills at 80 registers; B alone fits in 75, A alone in 80 + its gx).
  constexpr int NT1 = (N1 + 15) / 16;                               // 16-row tiles of d w1 [4C][C] = dy1^T LN_2(x')
  constexpr int NTC = C / 16 + (C % 16 ? 1 : 0);                    // 16-row tiles of a [C][C] weight gradient
  constexpr int VA = (CT + 1) / 2;
  constexpr int lkv = CROSS ? 2 : 1;
  const float* sXk = cross ? sXkv : sX;
  hrf_f4 dnA[CT], sA[4], dpA[4];                                    // role A: d LN_2 output; score / dP tiles (query columns)
  float DlA = 0.f;
  hrf_f4 sB[4], dpB[4], o1B[DT], o2B[DT];                           // role B: score / dP tiles (key columns); dv / dk of a key tile

  // dy rows = gx * dropout mask * scales (the out_proj output enters the residual through Dropout / DropPath), dO = dy Wo
  auto dy_and_dO = [&]() __attribute__((always_inline)) {
    AB_LANE(); AB_PIX();
    const float rs = rs_out;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int nb = 16 * t + 4 * q;
      const hrf_f4 mk = ld_sel(16 * (t + 1) <= C, a.mask, pc * C + nb, a.mask != nullptr ? C - nb : 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = tokv ? gx[t][r] * rs : 0.f;
        if (a.mask != nullptr) y *= mk[r];
        if (nb + r < C) sDY[tok * PC + nb + r] = y;
      }
    }
    HRF_WAVE_SYNC();                                                // a token's dy row was stored by four lanes of this wave
    hrf_f4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
    wave_gemm_tl<C, PW, CT>(sWo, C, sDY + tok * PC, lane, acc);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int k = 16 * t + 4 * q + r; if (k < C) sG[tok * PC + k] = acc[t][r]; }
  };
  auto w1tile = [&](int nt) __attribute__((always_inline)) {
    AB_LANE();
    hrf_f4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
    wave_tgemm<CT, true>(sH, PH, 16 * nt, N1, sO, PC, 0, C, sGam[0], sBet[0], sReal, lane, acc);
    store_wtile<CT>(slot + a.off_w1, 16 * nt, N1, 0, C, lane, acc);
  };
  // ---- phase 3: CrossFFN head backward.  A: d LN_2 output = dy1 W1 | barrier (w1 may occupy the dy / dO tiles) | LayerNorm
  // backward, gx += ., dy, dO.   B: the d w1 tiles (dy1^T LN_2(x')), d b1
  auto A3a = [&]() __attribute__((always_inline)) {
    AB_LANE();
    if (ffn) {
#pragma unroll
      for (int t = 0; t < CT; ++t) dnA[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
      wave_gemm_tl<N1, PW, CT>(sW1, C, sH + tok * PH, lane, dnA);
    }
  };
  auto B3a = [&]() __attribute__((always_inline)) { if (ffn && wb < NT1) w1tile(wb); };
  auto A3b = [&]() __attribute__((always_inline)) {
    AB_LANE(); AB_PIX();
    if (ffn) {
      ln_bwd_rows<C, CT>(dnA, sO, PC, tok, tokv, sRs2[tok], sGam[0], sPar[wave & 3][0], lane);
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) gx[t][r] += tokv ? dnA[t][r] : 0.f;
    }
    dy_and_dO();
    if (GXP) {
      float* park = a.gx_park + ((long)blockIdx.x * 64 + tok) * (CT * 16) + 4 * q;
#pragma unroll
      for (int t = 0; t < CT; ++t) hrf_st4(park + 16 * t, gx[t]);
    }
  };
  auto B3b = [&]() __attribute__((always_inline)) {
    if (ffn) {
      for (int nt = wb + NBW; nt < NT1; nt += NBW) w1tile(nt);
      for (int n = lt; n < N1; n += NB) {                           // d b1 = column sums of dy1
        float sacc = 0.f;
        for (int j = 0; j < NTOK; ++j) sacc += sH[j * PH + n];
        slot[a.off_b1 + n] = sacc;
      }
    }
  };
  // ---- phase 5: recompute the projections q / k / v with the LayerNorm affine applied on read (0 for tokens outside the
  // image).  NWB = 4: A: q and the first channel tiles of v, B: k and the remaining tiles of v (own token tile each);
  // NWB = 0: A: q and v, B: k;  NWB = 2: A: q and v, B wave w: k of the token tiles w and w + 2
  auto proj = [&](const float* sW, const float* bias, const float* xh, int ln, float* dstT, float mul, int t_lo, int t_hi, int tk) __attribute__((always_inline)) {
    AB_LANE();
    hrf_f4 acc[CT];
    acc_bias_l<CT>(bias, 0, PW, lane, acc);
    const bool real = sReal[tk] != 0.f;
    constexpr int NS = (C + 15) / 16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int kbase = 16 * s + 4 * q;
      const bool kin = kbase < PW;
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kbase + r < C ? kbase + r : 0;
        float aff = fmaf(xh[tk * PC + k], sGam[ln][k], sBet[ln][k]);
        HRF_KEEP(aff);
        bv[r] = (kbase + r < C && real) ? aff : 0.f;
      }
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        if (t < t_lo || t >= t_hi) continue;                        // (uniform)
        const int n = 16 * t + i;
        const hrf_f4 w = hrf_ld4(sW + (n < C ? n : 0) * PW + (kin ? kbase : 0));
        const bool ok = kin && n < C;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t] = hrf_mfma16(ok ? w[r] : 0.f, bv[r], acc[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      if (t < t_lo || t >= t_hi) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int n = 16 * t + 4 * q + r; if (n < C) dstT[tk * PC + n] = acc[t][r] * mul; }
    }
  };
  auto A5 = [&]() __attribute__((always_inline)) {
    AB_LANE();
    proj(sWq, sB3[0], sX, 1, sQ, a.scale, 0, CT, tok);
    proj(sWv, sB3[2], sXk, lkv, sV, 1.f, 0, NWB == 4 ? VA : CT, tok);
  };
  auto B5 = [&]() __attribute__((always_inline)) {
    AB_LANE();
    if (NWB == 2) {
#pragma unroll 1
      for (int tt = wb; tt < 4; tt += 2) proj(sWk, sB3[1], sXk, lkv, sK, 1.f, 0, CT, 16 * tt + i);
    } else {
      proj(sWk, sB3[1], sXk, lkv, sK, 1.f, 0, CT, tok);
      if (NWB == 4) proj(sWv, sB3[2], sXk, lkv, sV, 1.f, VA, CT, tok);
    }
  };
  // ---- phase 6: attention backward per head (attention.hip's MFMA formulation on the packed tiles).  A = query columns, B =
  // key columns; the row statistics (max, 1 / sum, D = sum P dP) pass from A to B through LDS at the first barrier of a head.
  // NWB = 4: a B wave owns one key tile and computes its raw scores / dP beside A's softmax, before that barrier; NWB = 0 / 2:
  // behind it (NWB = 2: two key tiles per wave, the second one behind the second barrier - one head only).
  auto A6a = [&](int h) __attribute__((always_inline)) {                                           // wave = queries tok0 .. tok0+15: scores, dP, statistics
    AB_LANE();
    const float* bias = sT + h * 176;
#pragma unroll
    for (int t = 0; t < 4; ++t) { sA[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; dpA[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; }
    const float* qrow = sQ + (tok0 + i) * PC + h * D + q;
    const float* grow = sG + (tok0 + i) * PC + h * D + q;
#pragma unroll
    for (int kk = 0; kk < KSD; ++kk) {
      const bool kv = 4 * kk + q < D;
      const float qv = ldz(qrow, 4 * kk, kv), gv = ldz(grow, 4 * kk, kv);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sA[t] = hrf_mfma16(sK[(16 * t + i) * PC + h * D + 4 * kk + q], qv, sA[t]);
        dpA[t] = hrf_mfma16(sV[(16 * t + i) * PC + h * D + 4 * kk + q], gv, dpA[t]);
      }
    }
    const int qi = tok0 + i, qc = qi < NTOK ? qi : 0;
    const int yi = qc / 7, xi = qc - 7 * yi;
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * t + 4 * q + r, jc = j < NTOK ? j : 0;
        const int yj = jc / 7, xj = jc - 7 * yj;
        float bj = bias[(yi - yj + 6) * 13 + (xi - xj + 6)];
        HRF_KEEP(bj);
        const float sv = j < NTOK ? sA[t][r] + bj : -3.0e38f;
        sA[t][r] = sv;
        m = fmaxf(m, sv);
      }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sA[t][r] = __expf(sA[t][r] - m); l += sA[t][r]; }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = hrf_rcp(l);
    float Dl = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sA[t][r] *= inv; Dl = fmaf(sA[t][r], dpA[t][r], Dl); }
    Dl += __shfl_xor(Dl, 16);
    Dl += __shfl_xor(Dl, 32);
    DlA = Dl;
    if (q == 0) { sM[qi] = m; sIL[qi] = inv; sDl[qi] = Dl; }
  };
  auto A6b = [&](int h) __attribute__((always_inline)) {                                           // O = P V again (for d wo), dS -> plane, dQ = dS K
    AB_LANE();
    float* dsp = a.ds_plane + ((long)blockIdx.x * HEADS + h) * (NTOK * NTOK);   // dS[key][query] of this (window, head)
    const int qi = tok0 + i;
    hrf_f4 o1[DT], o2[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { o1[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f}; o2[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (AB_PADK(t, r)) continue;
        const float* vrow = sV + (16 * t + 4 * q + r) * PC + h * D + i;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o1[dt] = hrf_mfma16(sA[t][r], vrow[16 * dt], o1[dt]);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { const int d = 16 * dt + i; if (d < D) sO[(tok0 + 4 * q + r) * PC + h * D + d] = o1[dt][r]; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = sA[t][r] * (dpA[t][r] - DlA);
        sA[t][r] = ds;
        const int j = 16 * t + 4 * q + r;
        if (j < NTOK && qi < NTOK) dsp[j * NTOK + qi] = ds;
      }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (AB_PADK(t, r)) continue;
        const float* krow = sK + (16 * t + 4 * q + r) * PC + h * D + i;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o2[dt] = hrf_mfma16(sA[t][r], krow[16 * dt], o2[dt]);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qo = tok0 + 4 * q + r;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { const int d = 16 * dt + i; if (d < D) sDQ[qo * PC + h * D + d] = qo < NTOK ? o2[dt][r] * a.scale : 0.f; }
    }
  };
  // key-column scores of key tile kt0: S[query 16t+4q+r][key kt0+i] and dP
  auto key_scores = [&](int h, int kt0) __attribute__((always_inline)) {
    AB_LANE();
#pragma unroll
    for (int t = 0; t < 4; ++t) { sB[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; dpB[t] = hrf_f4{0.f, 0.f, 0.f, 0.f}; }
    const float* krow = sK + (kt0 + i) * PC + h * D + q;
    const float* vrow = sV + (kt0 + i) * PC + h * D + q;
#pragma unroll
    for (int kk = 0; kk < KSD; ++kk) {
      const bool kv = 4 * kk + q < D;
      const float kvv = ldz(krow, 4 * kk, kv), vv = ldz(vrow, 4 * kk, kv);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sB[t] = hrf_mfma16(sQ[(16 * t + i) * PC + h * D + 4 * kk + q], kvv, sB[t]);
        dpB[t] = hrf_mfma16(sG[(16 * t + i) * PC + h * D + 4 * kk + q], vv, dpB[t]);
      }
    }
  };
  // one key tile: P and dS in key-column orientation from the row statistics, dV = P^T dO and dK = dS^T Q
  auto key_tile = [&](int h, int kt0, bool scores) __attribute__((always_inline)) {
    AB_LANE();
    const float* bias = sT + h * 176;
    if (scores) key_scores(h, kt0);
    const int kj = kt0 + i, kc = kj < NTOK ? kj : 0;
    const int yj = kc / 7, xj = kc - 7 * yj;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = 16 * t + 4 * q + r, qc = qi < NTOK ? qi : 0;
        const int yi = qc / 7, xi = qc - 7 * yi;
        const bool ok = kj < NTOK && qi < NTOK;
        float bj = bias[(yi - yj + 6) * 13 + (xi - xj + 6)], mq = sM[qc], ilq = sIL[qc], dlq = sDl[qc];
        HRF_KEEP(bj); HRF_KEEP(mq); HRF_KEEP(ilq); HRF_KEEP(dlq);
        const float p = ok ? __expf(sB[t][r] + bj - mq) * ilq : 0.f;
        sB[t][r] = p;
        dpB[t][r] = ok ? p * (dpB[t][r] - dlq) : 0.f;
      }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { o1B[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f}; o2B[dt] = hrf_f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (AB_PADK(t, r)) continue;
        const float* gr = sG + (16 * t + 4 * q + r) * PC + h * D + i;
        const float* qr = sQ + (16 * t + 4 * q + r) * PC + h * D + i;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          o1B[dt] = hrf_mfma16(sB[t][r], gr[16 * dt], o1B[dt]);     // dV = P^T dO
          o2B[dt] = hrf_mfma16(dpB[t][r], qr[16 * dt], o2B[dt]);    // dK = dS^T Q
        }
      }
  };
  auto key_store = [&](int h, int kt0) __attribute__((always_inline)) {
    AB_LANE();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ko = kt0 + 4 * q + r;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + i;
        if (d < D) { sK[ko * PC + h * D + d] = o2B[dt][r]; sV[ko * PC + h * D + d] = o1B[dt][r]; }
      }
    }
  };
  static_assert(NWB != 2 || HEADS == 1, "two key tiles per role-B wave: one head (the statistics of the head must stay)");
  const int ktB = NWB == 2 ? 16 * wb : 16 * (wave & 3);                        // (first) key tile of a role-B wave
  auto B6a = [&](int h) __attribute__((always_inline)) { if (NWB == 4) key_scores(h, ktB); };
  auto B6b = [&](int h) __attribute__((always_inline)) { key_tile(h, ktB, NWB != 4); };
  auto B6c = [&](int h) __attribute__((always_inline)) {                                           // behind the barrier: every wave is done reading head h of sK / sV
    key_store(h, ktB);
    if (NWB == 2) {
      // the second key tile of this wave: it reads its OWN rows of sK / sV (nobody writes those) and all rows of sQ / sG;
      // role A waits at the barrier behind the head loop
      key_tile(h, 16 * (wb + 2), true);
      key_store(h, 16 * (wb + 2));
    }
  };
  // ---- phases 7 / 8.  Weight-gradient tile: [16 out rows][all input channels]
  auto wtile = [&](int which, int nt) __attribute__((always_inline)) {
    AB_LANE();
    hrf_f4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
    if (which == 0) {
      wave_tgemm<CT, false>(sDY, PC, 16 * nt, C, sO, PC, 0, C, nullptr, nullptr, nullptr, lane, acc);
      store_wtile<CT>(slot + a.off_wo, 16 * nt, C, 0, C, lane, acc);
    } else if (which == 1) {
      wave_tgemm<CT, true>(sDQ, PC, 16 * nt, C, sX, PC, 0, C, sGam[1], sBet[1], sReal, lane, acc);
      store_wtile<CT>(slot + a.off_wq, 16 * nt, C, 0, C, lane, acc);
    } else {
      wave_tgemm<CT, true>(which == 2 ? sK : sV, PC, 16 * nt, C, sXk, PC, 0, C, sGam[lkv], sBet[lkv], sReal, lane, acc);
      store_wtile<CT>(slot + (which == 2 ? a.off_wk : a.off_wv), 16 * nt, C, 0, C, lane, acc);
    }
  };
  auto A7 = [&]() __attribute__((always_inline)) {
    // ---- d LN outputs = dq Wq (+ dk Wk + dv Wv), LayerNorm backward, output gradients
    // (the row offset is formed again from the pixel index, behind an opaque copy: kept alive since the prologue it is a 64-bit
    // pair per lane that the register allocator spills in the variants that sit at their budget)
    AB_LANE(); AB_PIX();
    const int lane7 = lane;
    const long pc7 = pc;
    hrf_f4 tr[CT];                                                  // raw rows of the preceding block's tail (u = sc*raw + sh)
    if (tail) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int nb = 16 * t + 4 * q;
        tr[t] = ld_sel(16 * (t + 1) <= C, a.tail_raw, pc7 * C + nb, tokv ? C - nb : 0);
      }
    }
    hrf_f4 gx7[CT];                                                 // the gradient row again (GXP: parked by A3b)
#pragma unroll
    for (int t = 0; t < CT; ++t)
      gx7[t] = GXP ? hrf_ld4(a.gx_park + ((long)blockIdx.x * 64 + tok) * (CT * 16) + 4 * q + 16 * t) : gx[t];
    hrf_f4 dn[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) dn[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
    wave_gemm_tl<C, PW, CT>(sWq, C, sDQ + tok * PC, lane7, dn);
    if (!cross) {
      wave_gemm_tl<C, PW, CT>(sWk, C, sK + tok * PC, lane7, dn);
      wave_gemm_tl<C, PW, CT>(sWv, C, sV + tok * PC, lane7, dn);
    }
    ln_bwd_rows<C, CT>(dn, sX, PC, tok, tokv, sRsQ[tok], sGam[1], sPar[wave & 3][1], lane7);
    if (a.dq != nullptr && tokv) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int nb = 16 * t + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nb + r < C) {
            float v = dn[t][r] + (a.dq_add_res ? gx7[t][r] : 0.f);
            dn[t][r] = v;                                           // dx of this launch alone (the tail below needs it)
            if (a.dq_acc) v += a.dq[pc7 * C + nb + r];
            a.dq[pc7 * C + nb + r] = v;
          }
        }
      }
    }
    if (tail) {
      // x = tail_res + rs * GELU(u): tail_du = dx * rs * GELU'(u) and its BatchNorm moments (what hrf_act_bwd computed)
      const float rs = a.tail_rowscale != nullptr ? a.tail_rowscale[b] : 1.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int nb = 16 * t + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ch = nb + r < C ? nb + r : 0;
          const float rw = tr[t][r];
          float gg = hrf_gelu_grad(fmaf(rw, sTs[ch], sTs[C + ch]));
          HRF_KEEP(gg);
          const float du = (tokv && nb + r < C) ? dn[t][r] * rs * gg : 0.f;
          if (tokv && nb + r < C) a.tail_du[pc7 * C + nb + r] = du;
          const float m1 = hrf_row16_sum(du), m2 = hrf_row16_sum(du * rw);
          if (i == 0 && nb + r < C) { sTst[wave & 3][nb + r] = m1; sTst[wave & 3][C + nb + r] = m2; }
        }
      }
    }
    if (cross) {
#pragma unroll
      for (int t = 0; t < CT; ++t) dn[t] = hrf_f4{0.f, 0.f, 0.f, 0.f};
      wave_gemm_tl<C, PW, CT>(sWk, C, sK + tok * PC, lane7, dn);
      wave_gemm_tl<C, PW, CT>(sWv, C, sV + tok * PC, lane7, dn);
      ln_bwd_rows<C, CT>(dn, sXkv, PC, tok, tokv, sRsKV[tok], sGam[2], sPar[wave & 3][2], lane7);
      if (a.dkv != nullptr && tokv) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          const int nb = 16 * t + 4 * q;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (nb + r < C) {
              float v = dn[t][r] + (a.dkv_add_res ? gx7[t][r] : 0.f);
              if (a.dkv_acc) v += a.dkv[pc7 * C + nb + r];
              a.dkv[pc7 * C + nb + r] = v;
            }
          }
        }
      }
    }
    if (a.dres != nullptr && tokv) {
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int nb = 16 * t + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nb + r < C) {
            float v = gx7[t][r];
            if (a.dres_acc) v += a.dres[pc7 * C + nb + r];
            a.dres[pc7 * C + nb + r] = v;
          }
        }
      }
    }
    AB_T(8);
    // the out_proj weight gradient (dy^T O) and the four bias gradients behind the data path: role B holds the q / k / v tiles
    for (int nt = wave & 3; nt < NTC; nt += 4) wtile(0, nt);
    // bias gradients = column sums over the 49 tokens (tokens outside the image included: their k / v ARE the biases)
    for (int e = tid & 255; e < 4 * C; e += 256) {
      const int which = e / C, n = e - which * C;
      const float* T = which == 0 ? sDY : (which == 1 ? sDQ : (which == 2 ? sK : sV));
      float sacc = 0.f;
      for (int j = 0; j < NTOK; ++j) sacc += T[j * PC + n];
      slot[(which == 0 ? a.off_bo : (which == 1 ? a.off_bq : (which == 2 ? a.off_bk : a.off_bv))) + n] = sacc;
    }
  };
  // weight gradients of the q / k / v projections: [out tile of 16][all input channels] per wave, round-robin
  auto B8 = [&]() __attribute__((always_inline)) { for (int wt = wb; wt < 3 * NTC; wt += NBW) wtile(1 + wt / NTC, wt % NTC); };

  if (NWB == 0) {                                                   // four waves, both roles between the same barriers
    A3a(); B3a();
    __syncthreads();                                                // w1 (it may occupy the dy / dO tiles): last use by every wave
    A3b(); B3b();
    AB_T(4);
    __syncthreads();                                                // sH / xhat_2 are dead from here on (sH aliases sQ .. sDQ)
    AB_T(5);
    A5(); B5();
    __syncthreads();
    AB_T(6);
#pragma unroll 1
    for (int h = 0; h < HEADS; ++h) {
      A6a(h);
      __syncthreads();                                              // the row statistics of head h
      A6b(h); B6b(h);
      __syncthreads();                                              // every wave is done reading head h of sK / sV
      B6c(h);
    }
    __syncthreads();                                                // dk / dv rows of the last head
    AB_T(7);
    A7(); B8();
  } else if (roleA) {
    A3a();
    __syncthreads();
    A3b();
    AB_T(4);
    __syncthreads();
    AB_T(5);
    A5();
    __syncthreads();
    AB_T(6);
#pragma unroll 1
    for (int h = 0; h < HEADS; ++h) {
      A6a(h);
      __syncthreads();
      A6b(h);
      __syncthreads();
    }
    __syncthreads();
    AB_T(7);
    A7();
  } else {
    B3a();
    __syncthreads();
    B3b();
    __syncthreads();
    B5();
    __syncthreads();
#pragma unroll 1
    for (int h = 0; h < HEADS; ++h) {
      B6a(h);
      __syncthreads();
      B6b(h);
      __syncthreads();
      B6c(h);
    }
    __syncthreads();
    B8();
  }
  __syncthreads();
  AB_T(9);
  if (tail) {
    double* st = a.tail_gstats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * C;
    for (int e = tid; e < 2 * C; e += NT)
      hrf_atomic_add(&st[e], (double)((sTst[0][e] + sTst[1][e]) + (sTst[2][e] + sTst[3][e])));
  }
  // LayerNorm parameter gradients: sum of the four role-A waves' partials
  for (int e = tid; e < 3 * 2 * C; e += NT) {
    const int ln = e / (2 * C), k = e - ln * 2 * C;
    const float v = (sPar[0][ln][k] + sPar[1][ln][k]) + (sPar[2][ln][k] + sPar[3][ln][k]);
    const int off = ln == 0 ? (k < C ? a.off_g2 : a.off_bt2) : (ln == 1 ? (k < C ? a.off_gq : a.off_btq) : (k < C ? a.off_gkv : a.off_btkv));
    if (off >= 0) slot[off + (k < C ? k : k - C)] = v;
  }
  AB_T(10);
}

template <int C, int HEADS, int NWB, bool FFN, bool CROSS, bool TAIL>
int launch_bwd_v(const hrf_attn_block_t& a, const hrf_bn_bfin_t& bf, int nwin, void* stream) {
  constexpr int TILE = 64 * (C + 1), PW = (C + 3) & ~3;
  static_assert(4 * TILE >= 64 * (4 * C + 1), "dy1 rows alias the q / k / v / dq tiles");
  constexpr int W1X = (4 * C * PW <= 2 * TILE) ? 0 : 4 * C * PW;     // w1 aliases the (dy, dO) tiles when it fits
  constexpr size_t smem = ((size_t)4 * C * PW + W1X + 8 * TILE + 32 + (CROSS ? TILE : 0)) * sizeof(float);
#ifndef HRF_EMUL
  static std::atomic<unsigned> lds_set{0u};
  if (hrf_dyn_lds_once(lds_set, reinterpret_cast<const void*>(&attn_block_bwd_kernel<C, HEADS, NWB, FFN, CROSS, TAIL>), (int)smem) != HRF_OK) return HRF_ERR_LAUNCH;
#endif
  AbBwdArgs ab;
  ab.a = a; ab.bf = bf;
  return HRF_LAUNCH_G((attn_block_bwd_kernel<C, HEADS, NWB, FFN, CROSS, TAIL>), dim3(nwin), dim3(64 * (4 + NWB)), (unsigned)smem, stream, ab);
}

// (FFN head present, cross-attention, CrossFFN tail of the preceding block on load) are wave-uniform launch properties: one
// instantiation each - six per width (the tail form exists for self-attention only)
template <int C, int HEADS, int NWB>
int launch_bwd(const hrf_attn_block_t& a, const hrf_bn_bfin_t& bf, int nwin, void* stream) {
  const bool cross = a.xkv != a.xq, ffn = a.w1 != nullptr, tail = a.tail_raw != nullptr;
  if (cross) return ffn ? launch_bwd_v<C, HEADS, NWB, true, true, false>(a, bf, nwin, stream)
                        : launch_bwd_v<C, HEADS, NWB, false, true, false>(a, bf, nwin, stream);
  if (tail) return ffn ? launch_bwd_v<C, HEADS, NWB, true, false, true>(a, bf, nwin, stream)
                       : launch_bwd_v<C, HEADS, NWB, false, false, true>(a, bf, nwin, stream);
  return ffn ? launch_bwd_v<C, HEADS, NWB, true, false, false>(a, bf, nwin, stream)
             : launch_bwd_v<C, HEADS, NWB, false, false, false>(a, bf, nwin, stream);
}

// Relative-position-bias gradient from the dS planes the backward kernel left in memory (a LEAF of the backward graph:
// issued with the deferred weight gradients, off the dependency chain): drpb[(yi-yj+6)*13 + (xi-xj+6)][h] += dS[j][i].
// grid = (window chunks, heads); the planes of a chunk pass through LDS, thread e < 169 owns bin e.
#ifndef HRF_RPB_CHUNKS
#define HRF_RPB_CHUNKS 128          // window chunks (= workgroups per layer and head) of the RPB gradient gather
#endif
__device__ __forceinline__ void rpb_grad_body(const float* ds, int nwin, int heads, float* drpb, long copy_stride) {
  __shared__ float sP[NTOK * NTOK];
  const int h = blockIdx.y, e = threadIdx.x;
  const int dy = e / 13 - 6, dx = e - (e / 13) * 13 - 6;
  const int y0 = dy < 0 ? -dy : 0, y1 = dy > 0 ? 6 - dy : 6;
  const int x0 = dx < 0 ? -dx : 0, x1 = dx > 0 ? 6 - dx : 6;
  float acc = 0.f;
  // a block walks several windows (round 6: one window per block was a bare load -> gather -> 169 atomics chain, 33 K blocks
  // and 5.6 M atomics per step): the NEXT plane is in flight in registers while this one is gathered from LDS
  constexpr int NP = (NTOK * NTOK + 255) / 256;
  float pre[NP];
  auto fetch = [&](int w) {
    const float* p = ds + ((long)w * heads + h) * (NTOK * NTOK);
#pragma unroll
    for (int u = 0; u < NP; ++u) { const int k = threadIdx.x + 256 * u; pre[u] = p[k < NTOK * NTOK ? k : 0]; }
  };
  int w = blockIdx.x;
  if (w < nwin) fetch(w);
  for (; w < nwin; w += gridDim.x) {
#pragma unroll
    for (int u = 0; u < NP; ++u) { const int k = threadIdx.x + 256 * u; if (k < NTOK * NTOK) sP[k] = pre[u]; }
    __syncthreads();
    if (w + (int)gridDim.x < nwin) fetch(w + gridDim.x);
    if (e < 169)
      for (int yj = y0; yj <= y1; ++yj)
        for (int xj = x0; xj <= x1; ++xj) acc += sP[(yj * 7 + xj) * NTOK + (yj + dy) * 7 + xj + dx];
    __syncthreads();
  }
  if (e < 169) hrf_atomic_add(&drpb[(long)(blockIdx.x % HRF_STAT_COPIES) * copy_stride + e * heads + h], acc);
}

__global__ __launch_bounds__(256) void rpb_grad_kernel(const float* ds, int nwin, int heads, float* drpb, long copy_stride) {
  rpb_grad_body(ds, nwin, heads, drpb, copy_stride);
}

// every fused layer of a step in ONE launch (blockIdx.z = layer): seg rows of 5 longs {plane offset (floats) into `planes`,
// windows, heads, accumulator address, copy_stride}; blocks beyond a layer's heads / windows leave at once
__global__ __launch_bounds__(256) void rpb_grad_all_kernel(const float* planes, const long* seg) {
  const long* sg = seg + 5 * blockIdx.z;
  const int nwin = (int)sg[1], heads = (int)sg[2];
  if ((int)blockIdx.y >= heads || (int)blockIdx.x >= nwin) return;   // (uniform)
  rpb_grad_body(planes + sg[0], nwin, heads, reinterpret_cast<float*>(sg[3]), sg[4]);
}

__global__ __launch_bounds__(256) void fold_slots_kernel(const float* slots, const long* seg, const int* map, float* dst) {
  const long* sg = seg + 5 * blockIdx.y;
  const long base = sg[0], nslots = sg[1], stride = sg[2], n = sg[3], moff = sg[4];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int m = map[moff + e];
    if (m < 0) continue;
    const float* p = slots + base + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long k = 0;
    for (; k + 4 <= nslots; k += 4) { s0 += p[k * stride]; s1 += p[(k + 1) * stride]; s2 += p[(k + 2) * stride]; s3 += p[(k + 3) * stride]; }
    for (; k < nslots; ++k) s0 += p[k * stride];
    dst[m] += (s0 + s1) + (s2 + s3);
  }
}

}  // namespace

extern "C" int hrf_attn_block_supported(int C, int heads) {
  return (heads > 0 && C == 18 * heads && (heads == 1 || heads == 2 || heads == 4 || heads == 8)) ? 1 : 0;
}

extern "C" int hrf_attn_block_bwd_supported(int C, int heads) {
  return (heads > 0 && C == 18 * heads && (heads == 1 || heads == 2)) ? 1 : 0;
}

static void ab_geometry(hrf_attn_block_t& a) {
  a.nWh = (a.H + 6) / 7; a.nWw = (a.W + 6) / 7;
  a.pt = (a.nWh * 7 - a.H) / 2; a.pl = (a.nWw * 7 - a.W) / 2;       // centre pad: top/left = pad//2
  a.scale = 1.0f / sqrtf((float)(a.C / a.heads));
  if (a.rows_per_sample <= 0) a.rows_per_sample = a.H * a.W;
}

extern "C" int hrf_attn_block_bwd(const hrf_attn_block_t* p, void* stream) {
  HRF_GROUP_CALL();
  if (p == nullptr || !hrf_attn_block_bwd_supported(p->C, p->heads)) return HRF_ERR_ARG;
  hrf_attn_block_t a = *p;
  if (a.xq == nullptr || a.xkv == nullptr || a.gout == nullptr || a.pslot == nullptr || a.ds_plane == nullptr) return HRF_ERR_ARG;
  if (a.w1 != nullptr && (a.hidden != 4 * a.C || a.h1 == nullptr || a.du1 == nullptr || a.out == nullptr ||
                          (a.bfin1 == nullptr && a.cA1 == nullptr))) return HRF_ERR_ARG;
  if (a.tail_raw != nullptr && (a.xkv != a.xq || a.dq == nullptr || !a.dq_add_res || a.tail_du == nullptr ||
                                a.tail_gstats == nullptr || a.tail_scale == nullptr || a.tail_shift == nullptr)) return HRF_ERR_ARG;
  hrf_bn_bfin_t bf{};
  if (a.w1 != nullptr && a.bfin1 != nullptr) {
    bf = *a.bfin1;
    if (bf.C != 4 * a.C || bf.gstats == nullptr) return HRF_ERR_ARG;
  }
  a.bfin1 = nullptr;
  ab_geometry(a);
  const int nwin = a.B * a.nWh * a.nWw;
  if (nwin <= 0) return HRF_OK;
  if (a.heads == 1 && AB_NWB18 != 0 && a.gx_park == nullptr) return HRF_ERR_ARG;   // 8-wave form: [windows][64][32] floats of scratch
  if (a.heads == 1) return launch_bwd<18, 1, AB_NWB18>(a, bf, nwin, stream);
  return launch_bwd<36, 2, 4>(a, bf, nwin, stream);
}

extern "C" int hrf_rpb_grad(const float* ds_plane, int nwin, int heads, float* drpb, long copy_stride, void* stream) {
  if (nwin <= 0 || heads <= 0) return HRF_OK;
  const int chunks = nwin < HRF_RPB_CHUNKS ? nwin : HRF_RPB_CHUNKS;
  HRF_LAUNCH(rpb_grad_kernel, dim3(chunks, heads), dim3(256), 0, stream, ds_plane, nwin, heads, drpb, copy_stride);
  return hrf_check_launch();
}

extern "C" int hrf_rpb_grad_all(const float* planes, const long* seg, int nseg, int max_nwin, int max_heads, void* stream) {
  if (nseg <= 0 || max_nwin <= 0 || max_heads <= 0) return HRF_OK;
  if (planes == nullptr || seg == nullptr || nseg > 65535 || max_heads > 65535) return HRF_ERR_ARG;
  const int chunks = max_nwin < HRF_RPB_CHUNKS ? max_nwin : HRF_RPB_CHUNKS;
  HRF_LAUNCH(rpb_grad_all_kernel, dim3(chunks, max_heads, nseg), dim3(256), 0, stream, planes, seg);
  return hrf_check_launch();
}

extern "C" int hrf_fold_slots(const float* slots, const long* seg, int nseg, const int* map, float* dst, long max_n, void* stream) {
  if (nseg <= 0 || max_n <= 0) return HRF_OK;
  HRF_LAUNCH(fold_slots_kernel, dim3(hrf_cdiv(max_n, 256), nseg), dim3(256), 0, stream, slots, seg, map, dst);
  return hrf_check_launch();
}

extern "C" int hrf_attn_block_fwd(const hrf_attn_block_t* p, void* stream) {
  HRF_GROUP_CALL();
  if (p == nullptr || !hrf_attn_block_supported(p->C, p->heads)) return HRF_ERR_ARG;
  hrf_attn_block_t a = *p;
  if (a.xq == nullptr || a.xkv == nullptr || a.res == nullptr || a.out == nullptr) return HRF_ERR_ARG;
  if (a.w1 != nullptr && (a.hidden != 4 * a.C || a.h1 == nullptr)) return HRF_ERR_ARG;
  hrf_bn_fin_t fin{};
  if (a.tail_raw != nullptr) {
    // the rows are formed by this launch: self-attention on the buffer it writes them to
    if (a.tail_res == nullptr || a.x_out == nullptr || a.xq != a.x_out || a.xkv != a.x_out || a.res != a.x_out ||
        a.res2 != nullptr) return HRF_ERR_ARG;
    if (a.tail_fin != nullptr) {
      fin = *a.tail_fin;
      if (fin.C != a.C || fin.stats == nullptr) return HRF_ERR_ARG;
    } else if (a.tail_scale == nullptr || a.tail_shift == nullptr) return HRF_ERR_ARG;
  }
  a.tail_fin = nullptr;
  ab_geometry(a);
  const int nwin = a.B * a.nWh * a.nWw;
  if (nwin <= 0) return HRF_OK;
  switch (a.heads) {
    case 1: return launch_fwd<18, 1>(a, fin, nwin, stream);
    case 2: return launch_fwd<36, 2>(a, fin, nwin, stream);
    case 4: return launch_fwd<72, 4>(a, fin, nwin, stream);
    default: return launch_fwd<144, 8>(a, fin, nwin, stream);
  }
}
