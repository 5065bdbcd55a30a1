// micro-benchmark: cost of same-address global atomics on MI355X (device vs workgroup scope)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }
// every block: `per` wave-wide atomics (64 consecutive floats each) on copy (blockIdx % copies)
template <int SCOPE>
__global__ void k_atom(float* p, int per, int copies, int stride) {
  float* q = p + (size_t)(blockIdx.x % copies) * stride;
  if (threadIdx.x < 64)
    for (int i = 0; i < per; ++i) {
      if (SCOPE == 0) unsafeAtomicAdd(q + i * 64 + threadIdx.x, 1.0f);
      else if (SCOPE == 1) __hip_atomic_fetch_add(q + i * 64 + threadIdx.x, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(q + i * 64 + threadIdx.x, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_atom_d(double* p, int per, int copies, int stride) {
  double* q = p + (size_t)(blockIdx.x % copies) * stride;
  if (threadIdx.x < 64)
    for (int i = 0; i < per; ++i) unsafeAtomicAdd(q + i * 64 + threadIdx.x, 1.0);
}
// plain stores of per-block partials
__global__ void k_store(float* p, int per) {
  float* q = p + (size_t)blockIdx.x * per * 64;
  if (threadIdx.x < 64) for (int i = 0; i < per; ++i) q[i * 64 + threadIdx.x] = 1.0f;
}
__global__ void k_xcc(int* out) {
  if (threadIdx.x == 0) { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); out[blockIdx.x] = x; }
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
  float* p; CK(hipMalloc(&p, 64 << 20)); CK(hipMemset(p, 0, 64 << 20));
  double* pd = (double*)p;
  printf("empty launch: %.2f us\n", timeit([&] { k_empty<<<1, 64>>>(p); }));
  printf("empty launch 1024 blocks: %.2f us\n", timeit([&] { k_empty<<<1024, 256>>>(p); }));
  for (int nb : {32, 128, 512, 2048}) {
    for (int per : {1, 8}) {
      for (int copies : {1, 8, 16, 64}) {
        float t0 = timeit([&] { k_atom<0><<<nb, 64>>>(p, per, copies, 4096); });
        float t1 = timeit([&] { k_atom<1><<<nb, 64>>>(p, per, copies, 4096); });
        float t2 = timeit([&] { k_atom<2><<<nb, 64>>>(p, per, copies, 4096); });
        float t3 = timeit([&] { k_atom_d<<<nb, 64>>>(pd, per, copies, 4096); });
        printf("blocks %5d per %d copies %2d: unsafe %.2f  wg-scope %.2f  agent %.2f  f64 %.2f us\n", nb, per, copies, t0, t1, t2, t3);
      }
    }
    printf("blocks %5d store 8x64: %.2f us\n", nb, timeit([&] { k_store<<<nb, 64>>>(p, 8); }));
  }
  // correctness of workgroup-scope atomics across XCDs
  CK(hipMemset(p, 0, 1 << 20)); 
  k_atom<1><<<2048, 64>>>(p, 1, 1, 4096); CK(hipDeviceSynchronize());
  float h[4]; CK(hipMemcpy(h, p, 16, hipMemcpyDeviceToHost));
  printf("wg-scope sum over 2048 blocks = %.1f (expect 2048)\n", h[0]);
  CK(hipMemset(p, 0, 1 << 20));
  k_atom<0><<<2048, 64>>>(p, 1, 1, 4096); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, p, 16, hipMemcpyDeviceToHost));
  printf("unsafe sum over 2048 blocks = %.1f (expect 2048)\n", h[0]);
  int* xi = (int*)(p + (1 << 20));
  k_xcc<<<64, 64>>>(xi); CK(hipDeviceSynchronize());
  std::vector<int> hx(64); CK(hipMemcpy(hx.data(), xi, 256, hipMemcpyDeviceToHost));
  printf("XCC_ID of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %d", hx[i] & 0xf); printf("\n");
  return 0;
}
