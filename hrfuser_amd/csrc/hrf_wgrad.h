// Weight-gradient problem descriptors shared by the pixel-major kernel (conv_engine.hip) and the LDS-tiled kernel for wide
// 1x1 problems (wgrad_tiled.hip).
#pragma once

struct WgradDenseArgs {
  const float* dy; int ldD; int doff; const float* yraw;
  const float* cA; const float* cB; const float* cC;
  const float* x; int ldX;
  const float* tf_scale; const float* tf_shift; const float* tf_rowstat;   // rowstat != null: LayerNorm
  float* dw; float* dbias;
  int Cout, Cin, Mpix, chunk;
  int H, W, Ho, Wo, stride, gyc;   // TAP (3x3) only: input / output grids, stride, channel groups per tap
  int gx, gy, sp;                  // logical grid (co groups, n groups, pixel splits), see the XCD mapping of the kernels
};

// Weight gradients are LEAVES of the backward graph: up to WGMAX independent problems of the same kernel variant are
// issued as ONE launch (hrf_wgrad_group_begin/_end): a 13 MB problem alone cannot fill 256 CUs for longer than its own
// ramp-up, 16 of them back to back do.  Block b of the launch belongs to problem p with bstart[p] <= b < bstart[p+1]
// (every problem's block count is a multiple of 8, so the XCD-aware mapping of the kernels is unchanged).
constexpr int WGMAX = 16;
struct WgradGroup {
  int nprob;
  int bstart[WGMAX + 1];
  WgradDenseArgs p[WGMAX];
};

// wgrad_tiled.hip: plan (fills gx / gy / sp / chunk of `d`; false: the problem stays with the pixel-major kernel) and launch
bool hrf_wgrad_tiled_plan(WgradDenseArgs& d, bool bnb, int act, bool queued, int knob, int& key, int& nblocks);
int hrf_wgrad_tiled_launch(int key, const WgradGroup& g, int total_blocks, void* stream);
