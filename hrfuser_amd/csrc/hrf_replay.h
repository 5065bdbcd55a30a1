// Launch recorder / multi-threaded replayer ("lanes as HIP streams, without hipGraph").
//
// Why: a hipGraph with more than one stream replays through a host-driven slow path on this ROCm
// (measured: ~4.3 us of synchronous host time per node, 13-16 ms per training step of ~3 000 nodes,
// tools/microbench/graph_launch_ub.py), which had become the floor of the step.  Every kernel of
// the step is launched through HRF_LAUNCH, so the library records (kernel, grid, block, argument
// bytes, stream) of one eager step plus the fork/join points of the lane schedule, and replays the
// per-stream lists from one host thread per stream; cross-stream ordering uses device-side events
// (hipEventRecord / hipStreamWaitEvent), the host threads only order the ISSUE of those two calls.
#pragma once
#ifndef HRF_EMUL
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>

namespace hrf_rp {

struct Cmd {
  int type;                       // 0 kernel, 1 event record, 2 event wait, 3 memset
  const void* func;
  dim3 grid, block;
  unsigned smem;
  int ev;
  void* ptr; int value; size_t bytes;
  std::vector<char> blob;         // argument bytes, each argument at offs[i]
  std::vector<unsigned> offs;
};

bool recording();
void push(void* stream, Cmd&& c);

template <class T>
inline void pack_one(Cmd& c, const T& v) {
  const size_t al = alignof(T) < 8 ? 8 : alignof(T);
  size_t off = (c.blob.size() + al - 1) / al * al;
  c.blob.resize(off + sizeof(T));
  std::memcpy(c.blob.data() + off, &v, sizeof(T));
  c.offs.push_back((unsigned)off);
}
inline void pack(Cmd&) {}
template <class T, class... R>
inline void pack(Cmd& c, const T& v, const R&... r) { pack_one(c, v); pack(c, r...); }

template <class K, class... A>
inline void record_launch(K kern, dim3 grid, dim3 block, size_t smem, void* stream, const A&... args) {
  Cmd c{};
  c.type = 0; c.func = reinterpret_cast<const void*>(kern); c.grid = grid; c.block = block; c.smem = (unsigned)smem;
  pack(c, args...);
  push(stream, std::move(c));
}

}  // namespace hrf_rp
#endif
