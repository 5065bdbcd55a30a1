// How should many workgroups add their BatchNorm moments into one small accumulator on gfx950?
// Every moment-producing launch of the step ends with one fp64 atomic per channel and workgroup into HRF_STAT_COPIES replicated
// accumulators (copy = block % 4): N blocks -> N / 4 same-line atomics, serialised at the memory side (~25 ns each).
// Variants timed here, N blocks x 256 threads, C doubles per copy (one wave per block issues the atomics):
//   agent4    device-scope atomics, 4 copies (what the library does)
//   agent8x   device-scope atomics, copy = the block's XCC id (8 copies)
//   wg8x      WORKGROUP-scope atomics (no sc1: performed in the XCD's own L2), copy = XCC id - every copy is only ever touched
//             from one XCD, so the L2-local read-modify-writes are atomic among all of its writers; the dirty bytes reach memory
//             with the end-of-kernel write-back
//   none      no atomics (floor of the launch)
// Each launch is preceded by a memset node; 20 launches per captured graph, replayed; sums are verified.
// build: hipcc --offload-arch=gfx950 -O3 atomics_scope.hip -o atomics_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7; }   // HW_REG_XCC_ID[3:0]

template <int MODE>
__global__ __launch_bounds__(256) void k(double* acc, int C, const float* src, float* sink) {
  // a little streaming work so that blocks do not all retire in the same cycle
  float v = src[(blockIdx.x * 256 + threadIdx.x) & 0xFFFF];
  if (v == 123.f) sink[0] = v;
  if (threadIdx.x < C) {
    const double val = 1.0 + threadIdx.x;
    if (MODE == 0) unsafeAtomicAdd(&acc[(blockIdx.x & 3) * C + threadIdx.x], val);
    if (MODE == 1) unsafeAtomicAdd(&acc[xcc_id() * C + threadIdx.x], val);
    if (MODE == 2) __hip_atomic_fetch_add(&acc[xcc_id() * C + threadIdx.x], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 4) unsafeAtomicAdd(&acc[(blockIdx.x & 15) * C + threadIdx.x], val);
    if (MODE == 5) __hip_atomic_fetch_add(&acc[(blockIdx.x & 7) * C + threadIdx.x], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <int MODE>
double run(const char* name, int N, int C, double* acc, const float* src, float* sink, int copies) {
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 20; ++i) {
    hipMemsetAsync(acc, 0, sizeof(double) * 16 * C, s);
    hipLaunchKernelGGL(k<MODE>, dim3(N), dim3(256), 0, s, acc, C, src, sink);
  }
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> h(16 * C);
  hipMemcpy(h.data(), acc, sizeof(double) * 16 * C, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int c = 0; c < C; ++c) {
    double t = 0;
    for (int kx = 0; kx < 16; ++kx) t += h[kx * C + c];
    if (MODE != 3 && t != (double)N * (1.0 + c)) ok = false;
  }
  int used = 0;
  for (int kx = 0; kx < 16; ++kx) if (h[kx * C] != 0.0) ++used;
  const double us = ms * 1e3 / 200.0;
  printf("%-8s N=%4d C=%3d: %6.2f us per (memset + launch)  sums %s  copies used %d\n", name, N, C, us, MODE == 3 ? "-" : (ok ? "OK" : "WRONG"), used);
  (void)copies;
  return us;
}

int main() {
  double* acc; hipMalloc(&acc, sizeof(double) * 16 * 1024);
  float* src; hipMalloc(&src, 4 << 16); hipMemset(src, 0, 4 << 16);
  float* sink; hipMalloc(&sink, 64);
  for (int N : {240, 480, 960, 1920}) {
    for (int C : {36, 144}) {
      run<3>("none", N, C, acc, src, sink, 0);
      run<0>("agent4", N, C, acc, src, sink, 4);
      run<4>("agent16", N, C, acc, src, sink, 16);
      run<1>("agent8x", N, C, acc, src, sink, 8);
      run<2>("wg8x", N, C, acc, src, sink, 8);
      run<5>("wg8mod", N, C, acc, src, sink, 8);
    }
  }
  return 0;
}
