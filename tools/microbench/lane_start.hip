// When do the sibling lanes of a captured multi-stream hipGraph START?  The rocprofv3 timeline of the training step shows the lanes of a
// stage beginning 150-700 us apart although they fork from the same point.  L lanes x N dependent kernels of ~T us (256 blocks x 256
// threads of dependent FMAs, far from filling the chip), forked from and joined into an origin stream:
//   (a) captured into one hipGraph and replayed, (b) the same launches issued eagerly on L streams.
// A trailing marker kernel per lane records wall_clock64 at its first and last kernel; printed relative to the fork.
//   hipcc --offload-arch=gfx950 -O2 -o lane_start lane_start.hip && ./lane_start
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float* p, int n, long long* stamp) {
  if (stamp != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *stamp = wall_clock64();
  float v = p[threadIdx.x & 63];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) p[1] = v;
}
__global__ void mark(long long* stamp) { if (threadIdx.x == 0) *stamp = wall_clock64(); }
int run(int L, int N, int iters, bool graph, int blocks) {
  float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  long long* st; CK(hipMalloc(&st, sizeof(long long) * (2 * L + 1)));
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<hipStream_t> lane(L); std::vector<hipEvent_t> ev(L + 1);
  for (auto& t : lane) CK(hipStreamCreate(&t));
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto issue = [&]() {
    hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, st + 2 * L);           // the fork point
    hipEventRecord(ev[L], s);
    for (int l = 0; l < L; ++l) {
      hipStreamWaitEvent(lane[l], ev[L], 0);
      for (int i = 0; i < N; ++i)
        hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, lane[l], d, iters, i == 0 ? st + 2 * l : nullptr);
      hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, lane[l], st + 2 * l + 1);
      hipEventRecord(ev[l], lane[l]); hipStreamWaitEvent(s, ev[l], 0);
    }
  };
  hipGraph_t g; hipGraphExec_t ge = nullptr;
  if (graph) {
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    issue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  }
  for (int w = 0; w < 4; ++w) { if (graph) CK(hipGraphLaunch(ge, s)); else issue(); }
  CK(hipStreamSynchronize(s));
  std::vector<long long> h(2 * L + 1);
  CK(hipMemcpy(h.data(), st, sizeof(long long) * (2 * L + 1), hipMemcpyDeviceToHost));
  printf("%-6s L=%d N=%d blocks=%d:", graph ? "graph" : "eager", L, N, blocks);
  for (int l = 0; l < L; ++l) printf("  lane %d first +%6.1f us, done +%7.1f us", l, (h[2 * l] - h[2 * L]) / 100.0, (h[2 * l + 1] - h[2 * L]) / 100.0);
  printf("\n");
  return 0;
}
int main() {
  for (int blocks : {256, 1024}) {
    for (int L : {2, 3, 4}) { run(L, 12, 4000, true, blocks); run(L, 12, 4000, false, blocks); }
  }
  return 0;
}
