// Internal interface between conv_engine.hip (C-ABI entry points) and lin_engine.hip (LDS-free
// "row GEMM" kernels for stride-1 1x1 convolutions / Linear layers on channel-contiguous rows).
#pragma once
#include "../../include/hrfuser_hip.h"

struct LinFwdArgs {
  const float* x; int ldX;                       // [M][ldX], K = Cin valid per row
  const float* w; const float* bias;             // [N][K], [N]
  float* y; int ldY; int yoff;                   // [M][ldY] at column offset yoff
  const float* res; const float* res2; int ldR;  // optional residual rows [M][ldR]
  int tf_mode; const float* tf_scale; const float* tf_shift; const float* tf_rowstat;
  double* stats;                                 // [HRF_STAT_COPIES][2*N] or null
  float* ln_out; float ln_eps;                   // optional LayerNorm (mean, rstd) of the output rows
  hrf_bn_fin_t fin;                              // fin.stats != null: BatchNorm of X finalised on load (tf_mode 1..3)
  int M, K, N;
};

struct LinBwdDataArgs {
  const float* dy; int ldD; int doff; const float* yraw;     // [M][ldD] at column offset doff, K = Cout
  const float* cA; const float* cB; const float* cC;         // BatchNorm-backward coefficients (nullable)
  const float* w;                                            // [K = Cout][N = Cin]
  float* dx; int ldDx; int accumulate;                       // [M][ldDx]
  int epi; const float* xraw; int ldXr; const float* tf_scale; const float* tf_shift; int act;
  double* stats;                                             // [HRF_STAT_COPIES][2*N] or null
  hrf_bn_bfin_t bfin;                                        // bfin.gstats != null: cA/cB/cC derived on load
  int M, K, N;
};

// Return HRF_OK after enqueueing the launch, or -1 when the shape is not supported (caller falls
// back to the LDS-tiled implicit-GEMM kernels).
int hrf_lin_fwd_launch(const LinFwdArgs& a, void* stream);
bool hrf_lin_fwd_emits_ln(const LinFwdArgs& a);       // true when the launch above covers whole rows per wave
int hrf_lin_bwd_data_launch(const LinBwdDataArgs& a, void* stream);
// lin2_engine.hip: the same contract on an LDS-tiled data path, for wide problems (min(K, N) >= 64, M >= 1024); tried first
int hrf_lin2_fwd_launch(const LinFwdArgs& a, void* stream);
bool hrf_lin2_fwd_emits_ln(const LinFwdArgs& a);
int hrf_lin2_bwd_data_launch(const LinBwdDataArgs& a, void* stream);

// ---- conv3_engine.hip: 3x3 / stride-1 / pad-1 convolution (forward and backward-data) on NHWC rows,
// input halo tile staged ONCE per channel slab (no im2col re-reads), weights streamed through LDS.
struct Conv3Args {
  const float* in; int ldIn;                 // fwd: x rows; bwd: dY rows (column offset already added)
  const float* in2;                          // bwd + BatchNorm-backward: raw conv output (same indexing) or null
  const float* t0; const float* t1; const float* t2;   // fwd: tf_scale, tf_shift, - ; bwd: cA, cB, cC
  int tf_mode;                               // fwd: HRF_TF_* applied while staging the halo
  const float* w; int wCin;                  // OIHW weights, wCin = the convolution's Cin
  const float* bias;
  float* out; int ldOut; int ooff;
  const float* res; const float* res2; int ldR;
  int accumulate, epi; const float* xraw; int ldXr; const float* esc; const float* esh; int act;
  double* stats;                             // [HRF_STAT_COPIES][2*Cout] or null
  hrf_bn_fin_t fin;                          // forward: BatchNorm of the input finalised on load (stats != null)
  hrf_bn_bfin_t bfin;                        // backward: BatchNorm-backward coefficients derived on load (gstats != null)
  int B, H, W, Cin, Cout;                    // output grid; channels of `in` / of `out`
  int Hs, Ws;                                // stride-2 backward only: grid of `in` (the conv's output)
  int tilesX, tilesY;
};
int hrf_conv3_fwd_launch(const Conv3Args& a, void* stream);
int hrf_conv3_bwd_data_launch(const Conv3Args& a, void* stream);
int hrf_conv3s2_bwd_data_launch(const Conv3Args& a, void* stream);   // stride-2 conv, parity-class blocks

// ---- conv3x_engine.hip: the same contract on tap-major PACKED weights (hrf_conv3x_pack), v_mfma_f32_32x32x2, two 4-wave blocks
// per CU; Cout > 32.
// MODE 0: forward, stride 1 (in = x, transform on load, bias / residual epilogue, (sum, sum of squares) moments)
// MODE 1: backward data of the stride-1 convolution (in = dY [+ BatchNorm backward on load], pack dir 1, tap 8 - t)
// MODE 2: backward data of the stride-2 convolution: the block's tile is a tile of the SOURCE grid (dY), its outputs the four
//         parity classes (Y % 2, X % 2) of the 2 TH x 32 pixel patch of dX above it
struct C3xArgs {
  const float* in; int ldIn;                 // fwd: x rows; bwd: dY rows (column offset already added)
  const float* in2;                          // bwd + BatchNorm backward: raw conv output (same indexing) or null
  const float* t0; const float* t1; const float* t2;   // fwd: tf_scale, tf_shift, - ; bwd: cA, cB, cC
  int tf_mode;
  const float* wp; int Np, Kp;               // [9][Np][Kp]
  const float* bias;
  float* out; int ldOut; int ooff;
  const float* res; const float* res2; int ldR;
  int accumulate, epi; const float* xraw; int ldXr; const float* esc; const float* esh; int act;
  double* stats;
  hrf_bn_fin_t fin;
  hrf_bn_bfin_t bfin;
  int B, H, W, Cin, Cout;                    // output grid; channels of `in` / of `out`
  int Hs, Ws;                                // grid of `in` (MODE 2: the convolution's output grid; otherwise = H, W)
  int tilesX, tilesY, ntiles, per_xcd;
  int cols_per_block;                        // 64-channel output groups one block walks over its staged halo
};

int hrf_conv3x_fwd_launch(const C3xArgs& a, void* stream);
int hrf_conv3x_bwd_data_launch(const C3xArgs& a, void* stream);
int hrf_conv3xs2_fwd_launch(const C3xArgs& a, void* stream);          // stride-2 forward: 9 x 33 source patch in four parity planes
int hrf_conv3xs2_bwd_data_launch(const C3xArgs& a, void* stream);     // stride-2 conv: one block walks the four parity classes
