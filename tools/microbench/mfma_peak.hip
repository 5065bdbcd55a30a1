// fp32 MFMA issue-rate ceiling on gfx950: 16x16x4 and 32x32x2, 1/2/4 waves per SIMD, 16 independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024) void k16(float* out, int iters, float a, float b) {
  f4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = f4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(1024) void k32(float* out, int iters, float a, float b) {
  f16v acc[4];
  for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int kind = 0; kind < 2; ++kind) {
      const int threads = 256 * wps, blocks = 256 * 2;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
        else hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf = kind == 0 ? 64.0 * 2048 : 32.0 * 4096;     // flop per lane-iteration group (per wave)
        const double flop = (double)blocks * (threads / 64) * iters * mf;
        if (rep) printf("%s waves/SIMD=%d blocks=%d: %.2f ms  %.1f TFLOP/s\n", kind ? "32x32x2" : "16x16x4", wps, blocks, ms, flop / ms / 1e9);
      }
    }
  }
  return 0;
}
