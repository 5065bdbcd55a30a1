// What does ONE kernel node of a replayed hipGraph cost on this stack, and does the size of its kernarg segment matter?
//   hipcc --offload-arch=gfx950 -O2 -o graph_nodes graph_nodes.hip && ./graph_nodes
// Linear chains and 4-way forked chains of N trivial kernels with a 64-byte and a 2.5 KB kernarg struct; prints the host time
// of hipGraphLaunch and the time until the graph has finished (per node).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Small { float* p; int n; int pad[13]; };
struct Big { float* p; int n; int pad[640]; };
template <class A> __global__ void k(A a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += 1.f; }
template <class A> __global__ void kw(A a) {   // ~5 us of work on 256 blocks
  float v = a.p[threadIdx.x & 63];
  for (int i = 0; i < a.n; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) a.p[1] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class A, bool WORK>
int run(const char* name, int N, int lanes) {
  float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<hipStream_t> side(lanes); std::vector<hipEvent_t> ev(lanes + 1);
  for (auto& t : side) CK(hipStreamCreate(&t));
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  A a{}; a.p = d; a.n = 2000;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  if (lanes <= 1) {
    for (int i = 0; i < N; ++i) { if (WORK) hipLaunchKernelGGL(kw<A>, dim3(256), dim3(256), 0, s, a); else hipLaunchKernelGGL(k<A>, dim3(1), dim3(64), 0, s, a); }
  } else {
    CK(hipEventRecord(ev[lanes], s));
    for (int l = 0; l < lanes; ++l) {
      CK(hipStreamWaitEvent(side[l], ev[lanes], 0));
      for (int i = 0; i < N / lanes; ++i) { if (WORK) hipLaunchKernelGGL(kw<A>, dim3(256), dim3(256), 0, side[l], a); else hipLaunchKernelGGL(k<A>, dim3(1), dim3(64), 0, side[l], a); }
      CK(hipEventRecord(ev[l], side[l])); CK(hipStreamWaitEvent(s, ev[l], 0));
    }
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  double host = 0, tot = 0; const int R = 20;
  for (int r = 0; r < R; ++r) {
    const double t0 = now(); CK(hipGraphLaunch(ge, s)); const double t1 = now(); CK(hipStreamSynchronize(s)); const double t2 = now();
    host += t1 - t0; tot += t2 - t0;
  }
  printf("%-28s N=%d lanes=%d kernarg=%4zu B: hipGraphLaunch %.3f ms, until done %.3f ms = %.2f us per node\n", name, N, lanes, sizeof(A),
         host / R * 1e3, tot / R * 1e3, tot / R * 1e6 / N);
  return 0;
}
int main() {
  run<Small, false>("empty kernels", 1000, 1);
  run<Big, false>("empty kernels", 1000, 1);
  run<Small, false>("empty kernels", 1000, 4);
  run<Big, false>("empty kernels", 1000, 4);
  run<Small, true>("~5 us kernels", 1000, 1);
  run<Big, true>("~5 us kernels", 1000, 1);
  run<Small, true>("~5 us kernels", 1000, 4);
  run<Big, true>("~5 us kernels", 1000, 4);
  return 0;
}
