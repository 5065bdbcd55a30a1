// does executed CODE SIZE (cold instruction cache per launch) set the floor of tiny kernels?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Unroll { template <class F> static __device__ __forceinline__ void run(F f) { f(N - 1); Unroll<N - 1>::run(f); } };
template <> struct Unroll<0> { template <class F> static __device__ __forceinline__ void run(F) {} };

// straight-line: N*32 distinct fma instructions (different immediates so nothing folds)
template <int N>
__global__ void k_straight(float* p, float s) {
  float a = p[threadIdx.x], b = s;
  Unroll<N * 32>::run([&](int i) { a = fmaf(a, b, (float)i * 0.37f + 1.0f); b = fmaf(b, 0.999f, (float)i * 0.11f); });
  p[threadIdx.x + blockIdx.x * blockDim.x] = a + b;
}
// same dynamic instruction count, loop of 32
__global__ void k_loop(float* p, float s, int n) {
  float a = p[threadIdx.x], b = s;
  for (int j = 0; j < n; ++j) {
#pragma unroll
    for (int i = 0; i < 32; ++i) { a = fmaf(a, b, (float)i * 0.37f + 1.0f); b = fmaf(b, 0.999f, (float)i * 0.11f); }
  }
  p[threadIdx.x + blockIdx.x * blockDim.x] = a + b;
}
template <class F> float timeit(F f, int n = 200) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / n;
}
int main() {
  float* p; hipMalloc(&p, 64 << 20); hipMemset(p, 0, 64 << 20);
  for (int nb : {1, 256, 2048}) {
    printf("blocks %d x 256 threads\n", nb);
    printf("  straight   1 (~0.5 KB): %.2f us   loop: %.2f us\n", timeit([&] { k_straight<1><<<nb, 256>>>(p, 1.f); }), timeit([&] { k_loop<<<nb, 256>>>(p, 1.f, 1); }));
    printf("  straight   8 (~4 KB)  : %.2f us   loop: %.2f us\n", timeit([&] { k_straight<8><<<nb, 256>>>(p, 1.f); }), timeit([&] { k_loop<<<nb, 256>>>(p, 1.f, 8); }));
    printf("  straight  32 (~16 KB) : %.2f us   loop: %.2f us\n", timeit([&] { k_straight<32><<<nb, 256>>>(p, 1.f); }), timeit([&] { k_loop<<<nb, 256>>>(p, 1.f, 32); }));
    printf("  straight 128 (~64 KB) : %.2f us   loop: %.2f us\n", timeit([&] { k_straight<128><<<nb, 256>>>(p, 1.f); }), timeit([&] { k_loop<<<nb, 256>>>(p, 1.f, 128); }));
  }
  return 0;
}
