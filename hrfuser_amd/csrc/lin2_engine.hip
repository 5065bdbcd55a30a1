// LDS-tiled fp32 MFMA row GEMM with transform-on-load and fused epilogues, for the WIDE stride-1 1x1 convolutions / Linear
// layers (gfx950): min(K, N) >= 64 - every CrossFFN, qkv / out_proj and fuse 1x1 of HRFuser-B (78 ... 2496 channels,
// configs/hrfuser/cascade_rcnn_hrfuser_b_1x_nus_r640_l_r_fusion.py:6-41; hrformer.py:267-295) and the 64 <-> 256 Bottleneck
// convolutions of every stem (resnet.py:263-302).
//
//   lin2_fwd       Y[m][n]  = sum_k tf(X)[m][k] * W[n][k]  + bias + res + res2      (+ BN moments, + LayerNorm row statistics)
//   lin2_bwd_data  dX[m][n] = sum_k bnbwd(dY)[m][k] * W[k][n]    (+= | * act'(.) and BN moments)
//
// Same contract as lin_engine.hip (hrf_lin.h), different data path.  The register-only kernels there re-fetch the weight
// fragments of every 16-pixel tile from L2: at 312 output channels that is 190 MB of weight traffic for 10 MB of
// activations, which held HRFuser-B's GEMMs at 0.16 of the MFMA peak.
//
// Shape of a launch (second version).  These problems are SMALL for the chip: 30 720 rows x 78 x 312 is 1.5 GFLOP = 9.5 us
// of the 256 CUs' MFMA time, i.e. ~120 rows per CU, and K is 5 ... 20 steps of 16.  What a block costs is therefore not its
// MFMAs but its serial latencies (launch, tables, first operand round trip, pipeline fill, epilogue), and the first version
// (128 rows x <= 256 channels, 512 threads, one block per CU by registers) paid them TWICE: N = 312 needed two column
// blocks = 480 blocks on 256 CUs = two rounds, and its prologue fetched tiles 0, 1, 2 in three dependent round trips.  Now:
//   * a block is 64 rows x NB channels with NB chosen so that ONE column block covers N whenever N <= 320 (80 / 160 / 256 /
//     320): 480 row blocks at M = 30 720, all resident at once at two blocks per CU (256 threads, <= 256 VGPRs, <= 80 KB of
//     LDS) - one round, and the two co-resident blocks hide each other's barriers and round trips;
//   * wave = CG row tiles x CT channel tiles (CT = 5 / 8 / 10: up to 80 accumulator registers), both operand tiles of a
//     16-deep K step pass through a 2-slot LDS ring, the contraction index is permuted inside a step (MFMA m takes
//     k = 4q + m) so that every operand fetch is one ds_read_b128 (12 reads per 80 MFMAs in the widest shape);
//   * global loads run PD = 2 or 4 steps ahead of their LDS store in PD register sets (the first version had one set, i.e.
//     one step = ~1 us of cover for a ~2 us round trip), and the prologue issues its PD tiles back to back.
// What the neck's rowgemm lacks and the backbone needs is done where the data passes anyway:
//   * X is transformed while it is staged (BatchNorm affine finalised on load + ReLU / GELU, or LayerNorm), dY gets the
//     BatchNorm backward (cA*dy + cB*y + cC, coefficients derived on load) the same way - the tables live in LDS;
//   * weights are read in the reference layouts ((out, in) rows; the backward stages W[k][n] transposed), ragged K / N
//     are guarded at the loads (no packed copies, no padding);
//   * bias / residual rows, activation' of the producer, per-channel (sum, sum*.) moments and LayerNorm row statistics are
//     applied to the accumulators (D[channel][pixel]: a lane owns 4 consecutive channels of one pixel).
#include <type_traits>
#include "hrf_common.h"
#include "hrf_lin.h"
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int GK = 16, GLP = GK + 4, NW = 4, NTHR = 64 * NW, GROWS = 16 * NW;

// CG channel groups x (NW / CG) row groups of waves; a wave owns RT = CG row tiles x CT channel tiles
template <int CG_, int CT_, int PD_>
struct Tile {
  static constexpr int CG = CG_, CT = CT_, PD = PD_;
  static constexpr int NB = CG * CT * 16, RT = CG, RG = NW / CG;
  static constexpr int SLOT = (GROWS + NB) * GLP;
  static constexpr int NWV = (NB * 4 + NTHR - 1) / NTHR;                 // weight float4 per thread and step
  // epilogue through LDS (one pass per channel group): thread -> (channel quad c4 = tid % Q, row group tid / Q)
  static constexpr int Q = CT * 4, PE = CT * 16 + 4, NRG = NTHR / Q, NR = (GROWS + NRG - 1) / NRG;
  static constexpr int RED = NRG * 2 * CT * 16;                            // per-row-group channel moments
  static_assert(GROWS * PE <= 2 * SLOT, "the transposed accumulator tile must fit the operand ring");
};
typedef Tile<1, 5, 4> TileA;      //  80 channels
typedef Tile<2, 5, 4> TileB;      // 160
typedef Tile<2, 8, 2> TileC;      // 256
typedef Tile<2, 10, 2> TileD;     // 320

#ifdef HRF_EMUL
#define L2_SCHED_FENCE() ((void)0)
#define L2_OPAQUE(v) ((void)0)
#else
#define L2_OPAQUE(v) asm volatile("" : "+v"(v))
#define L2_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// the pipeline stages are lambdas over ~150 live registers: an outlined call would pass them through memory
#define L2_INLINE __attribute__((always_inline))

__device__ float g_zero4l[4] = {0.f, 0.f, 0.f, 0.f};
// phase stamps of one workgroup for tools/time_lin2_phases.py: compiled in only with -DHRF_L2_TIMING
#if defined(HRF_L2_TIMING) && !defined(HRF_EMUL)
__device__ long long g_l2_t[64];
#define L2_T(k) do { if (blockIdx.x == 100 && blockIdx.y == 0 && threadIdx.x == 0) g_l2_t[k] = wall_clock64(); } while (0)
#else
#define L2_T(k)
#endif
// 0: 1 = use this engine whenever the shape is supported, 2 = never (tests / A-B); 1: forced tile (1..4 = 80 / 160 / 256 / 320 channels);
// 2: minimum number of blocks for the automatic dispatch (0 = default)
static int g_l2_knob[4] = {0, 0, 0, 0};

__device__ __forceinline__ hrf_f4 l2_ld4(const float* p) {
#ifdef HRF_EMUL
  return hrf_ld4(p);
#else
  return *reinterpret_cast<const hrf_f4*>(p);      // ds_read_b128 (16-byte aligned by construction)
#endif
}
__device__ __forceinline__ void l2_st4(float* p, hrf_f4 v) {
#ifdef HRF_EMUL
  hrf_st4(p, v);
#else
  *reinterpret_cast<hrf_f4*>(p) = v;
#endif
}
// 4 consecutive floats p[0..3] of which the first `nvalid` exist (the others read as 0); unaligned 16-byte load when all
// four exist, selected ADDRESSES otherwise (never a load under a branch: hrf_lin / DESIGN 2)
__device__ __forceinline__ hrf_f4 gl_ld4(const float* p, int nvalid) {
  if (nvalid >= 4) return hrf_ld4(p);
  hrf_f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = *(e < nvalid ? p + e : g_zero4l);
  return r;
}

// row[idx .. idx+3] of a row of `len` >= 4 floats, elements at or beyond `len` read as 0 (idx may lie beyond the row): ONE
// unconditional 16-byte load from a clamped index, then register selects - a load under a (per-lane) condition gets its own
// dependent round trip, which is what a ragged K / N would cost in every step of the pipeline otherwise
__device__ __forceinline__ hrf_f4 ld4_ragged(const float* row, int idx, int len) {
  const int ic = idx > len - 4 ? len - 4 : idx;
  const int d = idx - ic;                                // 0: aligned with the request; 1..3: shifted; >= 4: nothing valid
  const hrf_f4 v = hrf_ld4(row + ic);
  hrf_f4 o;
  o[0] = d == 0 ? v[0] : (d == 1 ? v[1] : (d == 2 ? v[2] : (d == 3 ? v[3] : 0.f)));
  o[1] = d == 0 ? v[1] : (d == 1 ? v[2] : (d == 2 ? v[3] : 0.f));
  o[2] = d == 0 ? v[2] : (d == 1 ? v[3] : 0.f);
  o[3] = d == 0 ? v[3] : 0.f;
  return o;
}

// ---- epilogue through LDS.  The MFMA result layout (lane = pixel, 4 registers = 4 channels, 16 pixels of 16 DIFFERENT rows
// per store instruction) makes every epilogue access a 64-byte piece per row, and its fully unrolled form was 10 000
// instructions (80 KB of code, cold in the instruction cache at every launch) for the 20 tiles of the widest shape.  The
// accumulators of one channel group are written to the (now idle) operand ring as a [64 rows][CT*16 channels] tile; then
// thread (c4, rg) walks rows rg, rg + NRG, ... of channel quad c4: consecutive lanes = consecutive 16-byte pieces of a row
// (coalesced loads of the epilogue operands and stores), the per-channel moments are plain per-thread sums, and the code is
// one short body.
template <class T>
__device__ __forceinline__ void acc_to_lds(float* sE, const hrf_f4 (*acc)[T::CT], int rg, int lane) {
  const int i = lane & 15, q = lane >> 4;
#pragma unroll
  for (int rr = 0; rr < T::RT; ++rr)
#pragma unroll
    for (int tt = 0; tt < T::CT; ++tt) l2_st4(sE + ((rg * T::RT + rr) * 16 + i) * T::PE + tt * 16 + 4 * q, acc[rr][tt]);
}
// per-thread channel sums (s1 | s2 of the thread's 4 channels over its rows) -> sRed -> one atomic per channel and block
template <class T>
__device__ __forceinline__ void moments_out(float* sRed, double* stats, int N, int chbase, int c4, int rgp, bool active,
                                            const float* s1, const float* s2) {
  constexpr int CW = T::CT * 16;
  if (active) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sRed[(rgp * 2 + 0) * CW + 4 * c4 + r] = s1[r];
      sRed[(rgp * 2 + 1) * CW + 4 * c4 + r] = s2[r];
    }
  }
  __syncthreads();
  double* st = stats + (size_t)(blockIdx.x % HRF_STAT_COPIES) * 2 * N;
  for (int e = threadIdx.x; e < 2 * CW; e += NTHR) {
    const int which = e / CW, c = e - which * CW;
    if (chbase + c < N) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < T::NRG; ++g) s += sRed[(g * 2 + which) * CW + c];
      hrf_atomic_add(&st[which * N + chbase + c], (double)s);
    }
  }
}

template <int V> using IC = std::integral_constant<int, V>;

// ------------------------------------------------------------------------------------------------------------ forward
template <class T, int TF>
__global__ __launch_bounds__(NTHR, 2) void lin2_fwd_kernel(HrfGroup<LinFwdArgs> grp) {
  const LinFwdArgs& a = grp.sel();
  constexpr int CG = T::CG, CT = T::CT, PD = T::PD, NB = T::NB, RT = T::RT, SLOT = T::SLOT, NWV = T::NWV;
  constexpr bool TBLS = TF != HRF_TF_NONE;
  HRF_DYN_SMEM(float, smem);                            // [2][SLOT] ring | [2][KT] per-k tables | reduction scratch
  const int KT = (a.K + GK - 1) & ~(GK - 1);
  float* sTab = smem + 2 * SLOT;                        // scale | shift (BatchNorm affine or LayerNorm gamma / beta)
  float* sRed = sTab + (TBLS ? 2 * KT : 0);             // [RG][2][NB] moments, or [GROWS][CG] row sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: tile predicates become scalar branches
  const int i = lane & 15, q = lane >> 4;
  const int chg = wave % CG, rg = wave / CG;
  const long m0 = (long)blockIdx.x * GROWS;
  const int n0 = blockIdx.y * NB;
  L2_T(0);

  // ---- per-k transform tables in LDS (finalised on load when the producer's BatchNorm is handed over as moments)
  if (TBLS) {
    if (TF != HRF_TF_LN && a.fin.stats != nullptr) {
      hrf_bn_fin_onload(a.fin, sTab, sTab + KT, tid, NTHR, blockIdx.x == 0 && blockIdx.y == 0);
    } else {
      for (int k = tid; k < a.K; k += NTHR) { sTab[k] = a.tf_scale[k]; sTab[KT + k] = a.tf_shift[k]; }
    }
    for (int k = a.K + tid; k < KT; k += NTHR) { sTab[k] = 0.f; sTab[KT + k] = 0.f; }
  }

  hrf_f4 acc[RT][CT];
#pragma unroll
  for (int rr = 0; rr < RT; ++rr)
#pragma unroll
    for (int tt = 0; tt < CT; ++tt) acc[rr][tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  // staging: x tile = 64 rows x 4 float4 (one per thread); weight tile = NB rows x 4 float4
  const long xr = m0 + (tid >> 2) < a.M ? m0 + (tid >> 2) : a.M - 1;          // tail rows re-read the last row, never stored
  const int xdst = (tid >> 2) * GLP + 4 * (tid & 3);
  float mean = 0.f, rstd = 1.f;
  if (TF == HRF_TF_LN) { mean = a.tf_rowstat[2 * xr]; rstd = a.tf_rowstat[2 * xr + 1]; }
  // operand rows as 32-bit element offsets from the (wave-uniform) base pointers: one register per row instead of two
  const unsigned xoff = (unsigned)(xr * a.ldX);
  const int wn0 = tid >> 2, wq = tid & 3;              // this thread stages the k-quarter wq of weight rows wn0 + 64 e
  unsigned woff[NWV];
#pragma unroll
  for (int e = 0; e < NWV; ++e) {
    // rows beyond N: row 0 is staged instead - their accumulators are never stored and are zeroed before the moments
    const int n = n0 + wn0 + 64 * e;
    woff[e] = (unsigned)(n < a.N ? n : 0) * (unsigned)a.K;
  }
  const int wdst0 = (GROWS + wn0) * GLP + 4 * wq;
  hrf_f4 xpre[PD], wpre[PD][NWV];
  auto load_tile = [&](auto SETC, int s) L2_INLINE {
    constexpr int SET = decltype(SETC)::value;
    const int kb = s * GK + 4 * (tid & 3);
    if ((s + 1) * GK <= a.K) {                           // (uniform) whole step inside the rows: plain 16-byte loads
      xpre[SET] = hrf_ld4(a.x + (size_t)(xoff + kb));
#pragma unroll
      for (int e = 0; e < NWV; ++e) wpre[SET][e] = hrf_ld4(a.w + (size_t)(woff[e] + kb));
    } else {
      xpre[SET] = ld4_ragged(a.x + (size_t)xoff, kb, a.K);
#pragma unroll
      for (int e = 0; e < NWV; ++e) wpre[SET][e] = ld4_ragged(a.w + (size_t)woff[e], kb, a.K);
    }
  };
  auto store_tile = [&](auto SETC, int s) L2_INLINE {    // tile s: register set SET -> LDS slot s & 1
    constexpr int SET = decltype(SETC)::value;
    float* d = smem + (s & 1) * SLOT;
    const int kb = s * GK + 4 * (tid & 3);
    hrf_f4 v = xpre[SET];
    if (TBLS) {
      const hrf_f4 sc = l2_ld4(sTab + kb), sh = l2_ld4(sTab + KT + kb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = TF == HRF_TF_LN ? fmaf((v[r] - mean) * rstd, sc[r], sh[r]) : fmaf(v[r], sc[r], sh[r]);
        v[r] = TF == HRF_TF_AFFINE_RELU ? fmaxf(u, 0.f) : (TF == HRF_TF_AFFINE_GELU ? hrf_gelu(u) : u);
      }
    }
    l2_st4(d + xdst, v);
#pragma unroll
    for (int e = 0; e < NWV; ++e)
      if ((e + 1) * NTHR <= NB * 4 || tid + e * NTHR < NB * 4) l2_st4(d + wdst0 + e * 64 * GLP, wpre[SET][e]);
  };
  hrf_f4 fa[RT], fb[CT];
  const float* abase = smem + (rg * 16 * RT + i) * GLP + 4 * q;
  const float* bbase = smem + (GROWS + chg * CT * 16 + i) * GLP + 4 * q;
  auto read_frags = [&](int slot) L2_INLINE {
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) fa[rr] = l2_ld4(abase + slot * SLOT + rr * 16 * GLP);
#pragma unroll
    for (int tt = 0; tt < CT; ++tt) fb[tt] = l2_ld4(bbase + slot * SLOT + tt * 16 * GLP);
  };
  bool tile_on[CT];
#pragma unroll
  for (int tt = 0; tt < CT; ++tt) tile_on[tt] = n0 + (chg * CT + tt) * 16 < a.N;
  const bool all_on = tile_on[CT - 1];                  // (tiles switch off from the top: the common case is all of them)
  // ALL: every channel tile of this wave is inside N - the whole K loop is compiled twice, so that the common case carries
  // no per-tile predicate (and no accumulator copies at the joins of a predicated version)
  auto mma = [&](auto ALL) L2_INLINE {
    if (decltype(ALL)::value) {
#pragma unroll
      for (int tt = 0; tt < CT; ++tt)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int rr = 0; rr < RT; ++rr) acc[rr][tt] = hrf_mfma16(fb[tt][m], fa[rr][m], acc[rr][tt]);
    } else {
#pragma unroll
      for (int tt = 0; tt < CT - 1; ++tt) {
        if (!tile_on[tt]) break;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int rr = 0; rr < RT; ++rr) acc[rr][tt] = hrf_mfma16(fb[tt][m], fa[rr][m], acc[rr][tt]);
      }
    }
  };

  const int S = (a.K + GK - 1) / GK;
  if (TBLS) __syncthreads();                            // tables complete before the first transform
  L2_T(1);
  // prologue: the first PD tiles in ONE batch of loads
  load_tile(IC<0>{}, 0);
  if (S > 1) load_tile(IC<1 % PD>{}, 1);
  if (PD > 2) {
    if (S > 2) load_tile(IC<2 % PD>{}, 2);
    if (S > 3) load_tile(IC<3 % PD>{}, 3);
  }
  store_tile(IC<0>{}, 0);
  if (S > PD) load_tile(IC<0>{}, PD);
  __syncthreads();
  L2_T(2);
  // step s (J = s % PD): tile s+1 goes from its register set to the other LDS slot (every wave finished reading that slot
  // before the barrier that ended step s-1), the set is refilled with tile s+1+PD, then the MFMAs of tile s
  auto step = [&](auto ALL, auto JC, int s) L2_INLINE {
    constexpr int J1 = (decltype(JC)::value + 1) % PD;
    L2_SCHED_FENCE();
    if (s + 1 < S) store_tile(IC<J1>{}, s + 1);
    if (s + 1 + PD < S) load_tile(IC<J1>{}, s + 1 + PD);
    read_frags(s & 1);
    L2_SCHED_FENCE();
    mma(ALL);
    L2_SCHED_FENCE();
    __syncthreads();
    L2_T(3 + (s < 40 ? s : 40));
  };
  auto kloop = [&](auto ALL) L2_INLINE {
    for (int s = 0; s < S; s += PD) {
      step(ALL, IC<0>{}, s);
      if (s + 1 < S) step(ALL, IC<1 % PD>{}, s + 1);
      if (PD > 2) {
        if (s + 2 < S) step(ALL, IC<2 % PD>{}, s + 2);
        if (s + 3 < S) step(ALL, IC<3 % PD>{}, s + 3);
      }
    }
  };
  if (all_on) kloop(std::true_type{}); else kloop(std::false_type{});

  // ---- epilogue (see acc_to_lds): bias / residual rows, store, BatchNorm moments of the output
  {
    constexpr int Q = T::Q, PE = T::PE, NRG = T::NRG, NR = T::NR;
    float* sE = smem;
    const int c4 = tid % Q, rgp = tid / Q;
    const bool active = rgp < NRG;
#pragma unroll 1
    for (int pass = 0; pass < CG; ++pass) {
      const int chbase = n0 + pass * CT * 16, ch = chbase + 4 * c4;
      const int nval = active ? a.N - ch : 0;
      // operand rows of this thread first (they do not depend on the accumulators): one batch of loads
      // (absent operands are skipped by UNIFORM branches: a load of the zero block per element would be 88 loads per thread
      // and pass - more than the 63 a wave can have in flight, i.e. a stall at memory latency, 6.8 us in the first version)
      hrf_f4 bv = hrf_f4{0.f, 0.f, 0.f, 0.f};
      if (a.bias != nullptr) bv = gl_ld4(a.bias + ch, nval);
      hrf_f4 r1[NR], r2[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) { r1[j] = hrf_f4{0.f, 0.f, 0.f, 0.f}; r2[j] = r1[j]; }
      if (a.res != nullptr) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const long m = m0 + rgp + j * NRG;
          const bool rv = rgp + j * NRG < GROWS && m < a.M;
          r1[j] = gl_ld4(a.res + (rv ? m : 0) * a.ldR + ch, rv ? nval : 0);
        }
      }
      if (a.res2 != nullptr) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const long m = m0 + rgp + j * NRG;
          const bool rv = rgp + j * NRG < GROWS && m < a.M;
          r2[j] = gl_ld4(a.res2 + (rv ? m : 0) * a.ldR + ch, rv ? nval : 0);
        }
      }
      L2_T(44 + 4 * pass);
      if (pass > 0) __syncthreads();                     // the previous pass's readers are done with the tile
      if (chg == pass) acc_to_lds<T>(sE, acc, rg, lane);
      __syncthreads();
      L2_T(45 + 4 * pass);
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int row = rgp + j * NRG;
        const long m = m0 + row;
        if (row < GROWS && m < a.M && nval > 0) {
          hrf_f4 v = l2_ld4(sE + row * PE + 4 * c4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = r < nval ? v[r] + bv[r] + r1[j][r] + r2[j][r] : 0.f;
            s1[r] += v[r]; s2[r] = fmaf(v[r], v[r], s2[r]);
          }
          float* o = a.y + m * a.ldY + a.yoff + ch;
          if (nval >= 4) hrf_st4(o, v);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r < nval) o[r] = v[r];
          }
        }
      }
      L2_T(46 + 4 * pass);
      if (a.stats != nullptr) moments_out<T>(sRed, a.stats, a.N, chbase, c4, rgp, active, s1, s2);
      L2_T(47 + 4 * pass);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ backward data
// dX[m][n] = sum_k d(m, k) * W[k][n],  d = BNB ? cA[k]*dy + cB[k]*yraw + cC[k] : dy.  The weight tile [n][k-step] is staged
// TRANSPOSED from the (out = k, in = n) rows of W.  epi: * act'(sc[n]*xraw + sh[n]) and the (sum du, sum du*xraw) moments.
template <class T, bool BNB>
__global__ __launch_bounds__(NTHR, 2) void lin2_bwd_data_kernel(HrfGroup<LinBwdDataArgs> grp) {
  const LinBwdDataArgs& a = grp.sel();
  constexpr int CG = T::CG, CT = T::CT, PD = T::PD, NB = T::NB, RT = T::RT, SLOT = T::SLOT, NWV = T::NWV;
  HRF_DYN_SMEM(float, smem);
  const int KT = (a.K + GK - 1) & ~(GK - 1);
  float* sTab = smem + 2 * SLOT;                        // cA | cB | cC
  float* sRed = sTab + (BNB ? 3 * KT : 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int chg = wave % CG, rg = wave / CG;
  const long m0 = (long)blockIdx.x * GROWS;
  const int n0 = blockIdx.y * NB;
  L2_T(0);

  if (BNB) {
    if (a.bfin.gstats != nullptr) {
      hrf_bn_bfin_onload(a.bfin, sTab, sTab + KT, sTab + 2 * KT, tid, NTHR, blockIdx.x == 0 && blockIdx.y == 0);
    } else {
      for (int k = tid; k < a.K; k += NTHR) { sTab[k] = a.cA[k]; sTab[KT + k] = a.cB[k]; sTab[2 * KT + k] = a.cC[k]; }
    }
    for (int k = a.K + tid; k < KT; k += NTHR) { sTab[k] = 0.f; sTab[KT + k] = 0.f; sTab[2 * KT + k] = 0.f; }
  }

  hrf_f4 acc[RT][CT];
#pragma unroll
  for (int rr = 0; rr < RT; ++rr)
#pragma unroll
    for (int tt = 0; tt < CT; ++tt) acc[rr][tt] = hrf_f4{0.f, 0.f, 0.f, 0.f};

  const long xr = m0 + (tid >> 2) < a.M ? m0 + (tid >> 2) : a.M - 1;
  const unsigned doff = (unsigned)(xr * a.ldD + a.doff);
  const int xdst = (tid >> 2) * GLP + 4 * (tid & 3);
  // weight staging: thread -> (kk = tid & 15, n = 4 * (tid >> 4) + 64 e): a float4 of W[k][n .. n+3], written to LDS rows
  // n .. n+3 at column kk
  const int wkk = tid & 15, wn0 = 4 * (tid >> 4);
  hrf_f4 dpre[PD], ypre[PD], wpre[PD][NWV];
  auto load_tile = [&](auto SETC, int s) L2_INLINE {
    constexpr int SET = decltype(SETC)::value;
    const int kb = s * GK + 4 * (tid & 3);
    if ((s + 1) * GK <= a.K) {                           // (uniform)
      dpre[SET] = hrf_ld4(a.dy + (size_t)(doff + kb));
      if (BNB) ypre[SET] = hrf_ld4(a.yraw + (size_t)(doff + kb));
    } else {
      dpre[SET] = ld4_ragged(a.dy + (size_t)doff, kb, a.K);
      if (BNB) ypre[SET] = ld4_ragged(a.yraw + (size_t)doff, kb, a.K);
    }
    const int k = s * GK + wkk;
    const float* wrow = a.w + (size_t)((unsigned)(k < a.K ? k : a.K - 1) * (unsigned)a.N);   // rows beyond K: row K-1, zeroed at the store
    // the column of this thread, hidden from the loop-invariant code motion: hoisted, the clamp / shift state of ld4_ragged
    // for every e (10 registers) was spilled to scratch, and a scratch reload waits for ALL outstanding prefetch loads
    int nn = n0 + wn0;
    L2_OPAQUE(nn);
#pragma unroll
    for (int e = 0; e < NWV; ++e) wpre[SET][e] = ld4_ragged(wrow, nn + 64 * e, a.N);
  };
  auto store_tile = [&](auto SETC, int s) L2_INLINE {
    constexpr int SET = decltype(SETC)::value;
    float* d = smem + (s & 1) * SLOT;
    const int kb = s * GK + 4 * (tid & 3);
    hrf_f4 v = dpre[SET];
    if (BNB) {
      const hrf_f4 ca = l2_ld4(sTab + kb), cb = l2_ld4(sTab + KT + kb), cc = l2_ld4(sTab + 2 * KT + kb);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf(ca[r], v[r], fmaf(cb[r], ypre[SET][r], cc[r]));
    }
    l2_st4(d + xdst, v);
    const float km = s * GK + wkk < a.K ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < NWV; ++e)
      if ((e + 1) * NTHR <= NB * 4 || tid + e * NTHR < NB * 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[(GROWS + wn0 + 64 * e + r) * GLP + wkk] = wpre[SET][e][r] * km;
      }
  };
  hrf_f4 fa[RT], fb[CT];
  const float* abase = smem + (rg * 16 * RT + i) * GLP + 4 * q;
  const float* bbase = smem + (GROWS + chg * CT * 16 + i) * GLP + 4 * q;
  auto read_frags = [&](int slot) L2_INLINE {
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) fa[rr] = l2_ld4(abase + slot * SLOT + rr * 16 * GLP);
#pragma unroll
    for (int tt = 0; tt < CT; ++tt) fb[tt] = l2_ld4(bbase + slot * SLOT + tt * 16 * GLP);
  };
  bool tile_on[CT];
#pragma unroll
  for (int tt = 0; tt < CT; ++tt) tile_on[tt] = n0 + (chg * CT + tt) * 16 < a.N;
  const bool all_on = tile_on[CT - 1];
  auto mma = [&](auto ALL) L2_INLINE {
    if (decltype(ALL)::value) {
#pragma unroll
      for (int tt = 0; tt < CT; ++tt)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int rr = 0; rr < RT; ++rr) acc[rr][tt] = hrf_mfma16(fb[tt][m], fa[rr][m], acc[rr][tt]);
    } else {
#pragma unroll
      for (int tt = 0; tt < CT - 1; ++tt) {
        if (!tile_on[tt]) break;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int rr = 0; rr < RT; ++rr) acc[rr][tt] = hrf_mfma16(fb[tt][m], fa[rr][m], acc[rr][tt]);
      }
    }
  };

  const int S = (a.K + GK - 1) / GK;
  if (BNB) __syncthreads();
  L2_T(1);
  load_tile(IC<0>{}, 0);
  if (S > 1) load_tile(IC<1 % PD>{}, 1);
  if (PD > 2) {
    if (S > 2) load_tile(IC<2 % PD>{}, 2);
    if (S > 3) load_tile(IC<3 % PD>{}, 3);
  }
  store_tile(IC<0>{}, 0);
  if (S > PD) load_tile(IC<0>{}, PD);
  __syncthreads();
  L2_T(2);
  auto step = [&](auto ALL, auto JC, int s) L2_INLINE {
    constexpr int J1 = (decltype(JC)::value + 1) % PD;
    L2_SCHED_FENCE();
    if (s + 1 < S) store_tile(IC<J1>{}, s + 1);
    if (s + 1 + PD < S) load_tile(IC<J1>{}, s + 1 + PD);
    read_frags(s & 1);
    L2_SCHED_FENCE();
    mma(ALL);
    L2_SCHED_FENCE();
    __syncthreads();
    L2_T(3 + (s < 40 ? s : 40));
  };
  auto kloop = [&](auto ALL) L2_INLINE {
    for (int s = 0; s < S; s += PD) {
      step(ALL, IC<0>{}, s);
      if (s + 1 < S) step(ALL, IC<1 % PD>{}, s + 1);
      if (PD > 2) {
        if (s + 2 < S) step(ALL, IC<2 % PD>{}, s + 2);
        if (s + 3 < S) step(ALL, IC<3 % PD>{}, s + 3);
      }
    }
  };
  if (all_on) kloop(std::true_type{}); else kloop(std::false_type{});

  // ---- epilogue (see acc_to_lds): * act'(sc*xraw + sh) and the (sum du, sum du*xraw) moments, or += dx
  {
    constexpr int Q = T::Q, PE = T::PE, NRG = T::NRG, NR = T::NR;
    float* sE = smem;
    const int c4 = tid % Q, rgp = tid / Q;
    const bool active = rgp < NRG;
    const bool mom = a.epi == 1 && a.stats != nullptr;
#pragma unroll 1
    for (int pass = 0; pass < CG; ++pass) {
      const int chbase = n0 + pass * CT * 16, ch = chbase + 4 * c4;
      const int nval = active ? a.N - ch : 0;
      hrf_f4 sc = hrf_f4{0.f, 0.f, 0.f, 0.f}, sh = sc;
      if (a.epi == 1) { sc = gl_ld4(a.tf_scale + ch, nval); sh = gl_ld4(a.tf_shift + ch, nval); }
      hrf_f4 xv[NR];                                     // xraw rows (epi) or the previous dx rows (accumulate): one batch of loads
#pragma unroll
      for (int j = 0; j < NR; ++j) xv[j] = hrf_f4{0.f, 0.f, 0.f, 0.f};
      if (a.epi == 1 || a.accumulate) {                  // (uniform)
        const float* src = a.epi == 1 ? a.xraw : a.dx;
        const long ld = a.epi == 1 ? a.ldXr : a.ldDx;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const long m = m0 + rgp + j * NRG;
          const bool rv = rgp + j * NRG < GROWS && m < a.M;
          xv[j] = gl_ld4(src + (rv ? m : 0) * ld + ch, rv ? nval : 0);
        }
      }
      L2_T(44 + 4 * pass);
      if (pass > 0) __syncthreads();
      if (chg == pass) acc_to_lds<T>(sE, acc, rg, lane);
      __syncthreads();
      L2_T(45 + 4 * pass);
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      auto rows = [&](auto ACT) L2_INLINE {               // ACT: 0 = += dx (or plain), 1 = ReLU', 2 = GELU', 3 = identity'
        constexpr int AC = decltype(ACT)::value;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int row = rgp + j * NRG;
          const long m = m0 + row;
          if (row < GROWS && m < a.M && nval > 0) {
            hrf_f4 v = l2_ld4(sE + row * PE + 4 * c4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (AC == 0) v[r] += xv[j][r];
              else if (AC == 1) v[r] = fmaf(xv[j][r], sc[r], sh[r]) > 0.f ? v[r] : 0.f;
              else if (AC == 2) v[r] *= hrf_gelu_grad(fmaf(xv[j][r], sc[r], sh[r]));
              v[r] = r < nval ? v[r] : 0.f;
              s1[r] += v[r]; s2[r] = fmaf(v[r], xv[j][r], s2[r]);
            }
            float* o = a.dx + m * a.ldDx + ch;
            if (nval >= 4) hrf_st4(o, v);
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (r < nval) o[r] = v[r];
            }
          }
        }
      };
      if (a.epi != 1) rows(IC<0>{});
      else if (a.act == HRF_ACT_RELU) rows(IC<1>{});
      else if (a.act == HRF_ACT_GELU) rows(IC<2>{});
      else rows(IC<3>{});
      L2_T(46 + 4 * pass);
      if (mom) moments_out<T>(sRed, a.stats, a.N, chbase, c4, rgp, active, s1, s2);
      L2_T(47 + 4 * pass);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------- dispatch
// tile id: 1 = 80, 2 = 160, 3 = 256, 4 = 320 channels per block
inline int tile_nb(int id) { return id == 1 ? 80 : (id == 2 ? 160 : (id == 3 ? 256 : 320)); }
inline int pick_tile(int N) {
  if (g_l2_knob[1] >= 1 && g_l2_knob[1] <= 4) return g_l2_knob[1];      // tests: force a block width
  if (N <= 80) return 1;
  if (N <= 160) return 2;
  if (N <= 256) return 3;
  if (N <= 320) return 4;
  // several column blocks: the width that pads N the least (ties: the wider one, fewer re-reads of X)
  const int p4 = ((N + 319) / 320) * 320, p3 = ((N + 255) / 256) * 256;
  return p3 < p4 ? 3 : 4;
}
inline size_t smem_bytes(int id, int ntab, int K, int red) {
  const int KT = (K + GK - 1) & ~(GK - 1);
  return ((size_t)2 * (GROWS + tile_nb(id)) * GLP + (size_t)ntab * KT + red) * sizeof(float);
}
inline bool wide_enough(long M, int K, int N) {
  if (g_l2_knob[0] == 2) return false;
  if (K < 16 || N < 16 || M <= 0) return false;
  if (g_l2_knob[0] == 1) return true;
  // a block owns 64 rows: with too few blocks the register-only kernels (one wave per 16 rows) fill the chip better
  const int id = pick_tile(N);
  const long blocks = ((M + GROWS - 1) / GROWS) * ((N + tile_nb(id) - 1) / tile_nb(id));
  // (measured on HRFuser-B, tools/time_lin2_phases.py: a step of this pipeline costs ~0.7 us on top of its MFMAs, so the 120 ... 240
  // blocks of the 48x80 branch lose to lin_engine.hip - 44.7 -> 61 us for 624 -> 156 - while the 480+ of the 96x160 branch win)
  // (the 48x80 branch wins here only in its "expand" form - K = 156 -> N = 624: few steps, wide blocks: 42.2 -> 36.2 us forward,
  // 53.9 -> 41.9 us for the data gradient of fc3)
  const long minb = g_l2_knob[2] > 0 ? g_l2_knob[2] : ((K <= 160 && N >= 320) ? 200 : 400);
  return (K < N ? K : N) >= 64 && blocks >= minb;
}

template <class KERN>
int set_smem(KERN kern, size_t smem) {
#ifndef HRF_EMUL
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return HRF_ERR_LAUNCH;
#endif
  return HRF_OK;
}
constexpr size_t SMEM_MAX = 160 * 1024;

template <class T, int TF>
int launch_fwd(const LinFwdArgs& a, int id, void* stream) {
  const size_t smem = smem_bytes(id, TF != HRF_TF_NONE ? 2 : 0, a.K, T::RED);
  if (smem > SMEM_MAX) return -1;
  static size_t granted = 0;
  if (smem > granted) { if (set_smem(&lin2_fwd_kernel<T, TF>, smem) != HRF_OK) return HRF_ERR_LAUNCH; granted = smem; }
  const dim3 grid(hrf_cdiv(a.M, GROWS), hrf_cdiv(a.N, T::NB));
  return HRF_LAUNCH_G((lin2_fwd_kernel<T, TF>), grid, dim3(NTHR), (unsigned)smem, stream, a);
}
template <class T>
int launch_fwd_tf(const LinFwdArgs& a, int id, void* stream) {
  switch (a.tf_mode) {
    case HRF_TF_NONE: return launch_fwd<T, HRF_TF_NONE>(a, id, stream);
    case HRF_TF_AFFINE: return launch_fwd<T, HRF_TF_AFFINE>(a, id, stream);
    case HRF_TF_AFFINE_RELU: return launch_fwd<T, HRF_TF_AFFINE_RELU>(a, id, stream);
    case HRF_TF_AFFINE_GELU: return launch_fwd<T, HRF_TF_AFFINE_GELU>(a, id, stream);
    case HRF_TF_LN: return launch_fwd<T, HRF_TF_LN>(a, id, stream);
    default: return -1;
  }
}
template <class T, bool BNB>
int launch_bwd(const LinBwdDataArgs& a, int id, void* stream) {
  const size_t smem = smem_bytes(id, BNB ? 3 : 0, a.K, T::RED);
  if (smem > SMEM_MAX) return -1;
  static size_t granted = 0;
  if (smem > granted) { if (set_smem(&lin2_bwd_data_kernel<T, BNB>, smem) != HRF_OK) return HRF_ERR_LAUNCH; granted = smem; }
  const dim3 grid(hrf_cdiv(a.M, GROWS), hrf_cdiv(a.N, T::NB));
  return HRF_LAUNCH_G((lin2_bwd_data_kernel<T, BNB>), grid, dim3(NTHR), (unsigned)smem, stream, a);
}
template <class T>
int launch_bwd_b(const LinBwdDataArgs& a, int id, void* stream) {
  return a.cA != nullptr ? launch_bwd<T, true>(a, id, stream) : launch_bwd<T, false>(a, id, stream);
}

}  // namespace

// -1: shape not served by this engine (the caller falls back to lin_engine.hip)
int hrf_lin2_fwd_launch(const LinFwdArgs& a, void* stream) {
  if (!wide_enough(a.M, a.K, a.N)) return -1;
  // LayerNorm row statistics of the output are emitted by the register-only kernels (one wave holds whole rows there); the
  // caller's fallback serves such launches (out_proj + residual followed by a LayerNorm: N = the block width, K = N)
  if (a.ln_out != nullptr && g_l2_knob[0] != 1) return -1;
  if ((long)a.M * a.ldX >= (1L << 32)) return -1;         // operand rows are addressed through 32-bit element offsets
  // the on-load finalize of a BatchNorm is limited to HRF_FIN_MAXC channels; beyond it scale / shift arrive as arrays
  if (a.tf_mode != HRF_TF_NONE && a.tf_mode != HRF_TF_LN && a.fin.stats != nullptr && a.K > HRF_FIN_MAXC) return -1;
  const int id = pick_tile(a.N);
  switch (id) {
    case 1: return launch_fwd_tf<TileA>(a, id, stream);
    case 2: return launch_fwd_tf<TileB>(a, id, stream);
    case 3: return launch_fwd_tf<TileC>(a, id, stream);
    default: return launch_fwd_tf<TileD>(a, id, stream);
  }
}
bool hrf_lin2_fwd_emits_ln(const LinFwdArgs&) { return false; }
int hrf_lin2_bwd_data_launch(const LinBwdDataArgs& a, void* stream) {
  if (!wide_enough(a.M, a.K, a.N)) return -1;
  if (a.cA != nullptr && a.bfin.gstats != nullptr && a.K > HRF_FIN_MAXC) return -1;
  if ((long)a.M * a.ldD + a.doff >= (1L << 32)) return -1;   // 32-bit element offsets (see the forward)
  const int id = pick_tile(a.N);
  switch (id) {
    case 1: return launch_bwd_b<TileA>(a, id, stream);
    case 2: return launch_bwd_b<TileB>(a, id, stream);
    case 3: return launch_bwd_b<TileC>(a, id, stream);
    default: return launch_bwd_b<TileD>(a, id, stream);
  }
}
#if defined(HRF_L2_TIMING) && !defined(HRF_EMUL)
// the stamps of the last launch (synchronises the device): tools/time_lin2_phases.py
extern "C" int hrf_lin2_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_l2_t), sizeof(long long) * 64) == hipSuccess ? HRF_OK : HRF_ERR_LAUNCH;
}
#endif
// (reached through hrf_debug_knob only: not exported)
extern "C" __attribute__((visibility("hidden"))) int hrf_lin2_knob(int key, int value) {
  if (key < 0 || key >= 4) return HRF_ERR_ARG;
  g_l2_knob[key] = value;
  return HRF_OK;
}
