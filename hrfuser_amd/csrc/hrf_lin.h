// Internal interface between conv_engine.hip (C-ABI entry points) and lin_engine.hip (LDS-free
// "row GEMM" kernels for stride-1 1x1 convolutions / Linear layers on channel-contiguous rows).
#pragma once

struct LinFwdArgs {
  const float* x; int ldX;                       // [M][ldX], K = Cin valid per row
  const float* w; const float* bias;             // [N][K], [N]
  float* y; int ldY; int yoff;                   // [M][ldY] at column offset yoff
  const float* res; const float* res2; int ldR;  // optional residual rows [M][ldR]
  int tf_mode; const float* tf_scale; const float* tf_shift; const float* tf_rowstat;
  double* stats;                                 // [HRF_STAT_COPIES][2*N] or null
  int M, K, N;
};

struct LinBwdDataArgs {
  const float* dy; int ldD; int doff; const float* yraw;     // [M][ldD] at column offset doff, K = Cout
  const float* cA; const float* cB; const float* cC;         // BatchNorm-backward coefficients (nullable)
  const float* w;                                            // [K = Cout][N = Cin]
  float* dx; int ldDx; int accumulate;                       // [M][ldDx]
  int epi; const float* xraw; int ldXr; const float* tf_scale; const float* tf_shift; int act;
  double* stats;                                             // [HRF_STAT_COPIES][2*N] or null
  int M, K, N;
};

// Return HRF_OK after enqueueing the launch, or -1 when the shape is not supported (caller falls
// back to the LDS-tiled implicit-GEMM kernels).
int hrf_lin_fwd_launch(const LinFwdArgs& a, void* stream);
int hrf_lin_bwd_data_launch(const LinBwdDataArgs& a, void* stream);
