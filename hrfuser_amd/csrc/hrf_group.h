// Multi-problem launches: ONE launch for the equal-shape problems of sibling sensor streams.
//
// HRFuser runs the camera stream's finest branch and the M modality streams (lidar, radar, gated) through layers of
// IDENTICAL shape with different weights (hrfuser_hrformer_based.py:536-544,564-565,585-586).  At 2 images per GPU each
// of those launches fills a fraction of the 256 CUs and lives 10-60 us, so issuing them one by one - or on one HIP
// stream per sensor - leaves the chip time-sharing three latency-bound chains.  Here every hot kernel takes its
// arguments as HrfGroup<Args> = an array of up to HRF_GROUP_MAX problems in the kernarg segment and blockIdx.z selects
// the problem; a single launch is the n = 1 case of the same code object (no second instantiation to keep warm in the
// instruction cache).
//
// Host side: between hrf_group_begin() and hrf_group_end(stream) (include/hrfuser_hip.h) the launches of the C-ABI calls
// are queued instead of issued; hrf_group_end merges queued launches of the same kernel instantiation and launch
// geometry that sit at the same position of DIFFERENT calls (the caller brackets mutually independent calls only; the
// launches of one call keep their order) and issues them on `stream`.
#pragma once
#include <cstring>
#include "hrf_rt.h"

// Problems per launch.  The PRODUCT build uses 1: on MI355X the merged launches measured SLOWER than one HIP stream per sensor
// (same box, HRFuser-T training step: 15.9 ms on separate lanes; 17.2 ms with the stems merged, 17.9 ms with stems, transitions
// and the modality stages merged - lock-step sensor streams hit the same resource at the same time and a kernel boundary
// becomes a barrier for all of them), and the 4x kernarg segment alone cost 0.9 ms per step (a multi-stream hipGraph copies
// every node's kernarg on the host at each replay: tools/microbench/graph_nodes.hip, 2.8 -> 3.7 us per node at 2.5 KB).
// -DHRF_GROUP_MAX=4 (HRF_EXTRA_FLAGS, and the CPU emulator build of the tests) compiles the multi-problem form.
#ifndef HRF_GROUP_MAX
#define HRF_GROUP_MAX 1
#endif

template <class A>
struct HrfGroup {
  A p[HRF_GROUP_MAX];
#if HRF_GROUP_MAX == 1
  __device__ __forceinline__ const A& sel() const { return p[0]; }
#else
  __device__ __forceinline__ const A& sel() const { return p[blockIdx.z]; }
#endif
};

typedef int (*hrf_grp_issue_t)(const void* kern, dim3 grid, dim3 block, unsigned smem, void* stream,
                               const unsigned char* const* argv, int n);
bool hrf_grp_collecting();
// HRF_GROUP_CALL() at the top of every groupable C-ABI entry point: the launches queued until the scope ends belong to one
// call (entry points that call other entry points stay ONE call: only the outermost scope counts)
void hrf_grp_enter();
void hrf_grp_leave();
struct HrfGrpCallScope {
  HrfGrpCallScope() { hrf_grp_enter(); }
  ~HrfGrpCallScope() { hrf_grp_leave(); }
};
void hrf_grp_push(const void* kern, dim3 grid, dim3 block, unsigned smem, const void* args, unsigned argsize,
                  hrf_grp_issue_t issue);
#define HRF_GROUP_CALL() HrfGrpCallScope hrf_grp_call_scope_

template <class A>
int hrf_grp_issue(const void* kern, dim3 grid, dim3 block, unsigned smem, void* stream, const unsigned char* const* argv, int n) {
  HrfGroup<A> g;
  for (int i = 0; i < HRF_GROUP_MAX; ++i) std::memcpy(static_cast<void*>(&g.p[i]), argv[i < n ? i : 0], sizeof(A));
  grid.z = (unsigned)n;
  typedef void (*K)(HrfGroup<A>);
  K k = reinterpret_cast<K>(const_cast<void*>(kern));
  HRF_LAUNCH(k, grid, block, smem, stream, g);
  return hrf_check_launch();
}

// launch `kern` for problem `a` now, or queue it when a group is being collected
template <class A>
int hrf_launch_grp(void (*kern)(HrfGroup<A>), dim3 grid, dim3 block, unsigned smem, void* stream, const A& a) {
  if (hrf_grp_collecting()) {
    hrf_grp_push(reinterpret_cast<const void*>(kern), grid, block, smem, &a, (unsigned)sizeof(A), &hrf_grp_issue<A>);
    return HRF_OK;
  }
  const unsigned char* one = reinterpret_cast<const unsigned char*>(&a);
  return hrf_grp_issue<A>(reinterpret_cast<const void*>(kern), grid, block, smem, stream, &one, 1);
}
#define HRF_LAUNCH_G(kern, grid, block, smem, stream, a) hrf_launch_grp((kern), grid, block, smem, stream, a)
