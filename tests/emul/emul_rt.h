// CPU fiber emulator for the HRFuser HIP kernels — TEST INFRASTRUCTURE ONLY.
//
// Compiles the unmodified kernel sources (hrfuser_amd/csrc/*.hip) with g++ -DHRF_EMUL and runs
// each workgroup as a set of cooperative fibers (ucontext), one fiber per work-item:
//   * __syncthreads()      -> block-wide fiber barrier (deadlock = divergent barrier -> abort)
//   * __shfl*/MFMA         -> wave-wide (64 lanes) exchange through a staging buffer
//   * __shared__           -> static thread_local (one instance per worker = per running block)
//   * atomics              -> real host atomics (blocks run on several OS threads)
// It exists so kernel index math, barrier placement, shuffle convergence and the fp32 MFMA
// fragment layout can be debugged and sanitised without a GPU.  The product package never loads
// the emulation library; the GPU parity tests (-m gpu) are the parity proof.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;

namespace hrf_emul {
struct Fiber;
struct Tid { unsigned x, y, z; };
extern thread_local Tid* cur_tid;
extern thread_local dim3 bidx, bdim, gdim;
extern thread_local int cur_lane;            // lane id of the running fiber within its wave
void sync_block();
void sync_wave();
void* wave_buf();                            // 64 x 16 bytes staging area of the current wave
void* dyn_smem();
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
}  // namespace hrf_emul

#define threadIdx (*hrf_emul::cur_tid)
#define blockIdx (hrf_emul::bidx)
#define blockDim (hrf_emul::bdim)
#define gridDim (hrf_emul::gdim)

#define HRF_DYN_SMEM(T, name) T* name = reinterpret_cast<T*>(hrf_emul::dyn_smem())
#define HRF_LAUNCH(kern, grid, block, smem, stream, ...) \
  hrf_emul::launch(grid, block, smem, [&]() { kern(__VA_ARGS__); })

inline void __syncthreads() { hrf_emul::sync_block(); }
// lanes of a wave run as separate fibers between rendezvous points: an intra-wave LDS hand-off needs one here
#define HRF_WAVE_SYNC() hrf_emul::sync_wave()

template <class T>
inline T hrf_emul_exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  char* buf = static_cast<char*>(hrf_emul::wave_buf());
  std::memcpy(buf + 16 * hrf_emul::cur_lane, &v, sizeof(T));
  hrf_emul::sync_wave();
  T r;
  std::memcpy(&r, buf + 16 * (src_lane & 63), sizeof(T));
  hrf_emul::sync_wave();
  return r;
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
  int lane = hrf_emul::cur_lane;
  return hrf_emul_exchange(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  return hrf_emul_exchange(v, hrf_emul::cur_lane ^ mask);
}
template <class T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
  int lane = hrf_emul::cur_lane;
  int src = lane + (int)delta;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hrf_emul_exchange(v, src);
}

struct hrf_f4 {
  float d[4];
  float& operator[](int i) { return d[i]; }
  const float& operator[](int i) const { return d[i]; }
};
inline hrf_f4 hrf_ld4(const float* p) { hrf_f4 v; std::memcpy(v.d, p, 16); return v; }
inline void hrf_st4(float* p, hrf_f4 v) { std::memcpy(p, v.d, 16); }
// v_mfma_f32_16x16x4_f32 semantics: D = A(16x4) * B(4x16) + C, exact fp32 fmaf chain in k order.
// lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; result reg r of lane l = D[(l>>4)*4+r][l&15].
inline hrf_f4 hrf_mfma16(float a, float b, hrf_f4 c) {
  char* buf = static_cast<char*>(hrf_emul::wave_buf());
  int lane = hrf_emul::cur_lane;
  float ab[2] = {a, b};
  std::memcpy(buf + 16 * lane, ab, 8);
  hrf_emul::sync_wave();
  auto A = [&](int i, int k) { float v; std::memcpy(&v, buf + 16 * (k * 16 + i), 4); return v; };
  auto B = [&](int k, int j) { float v; std::memcpy(&v, buf + 16 * (k * 16 + j) + 4, 4); return v; };
  hrf_f4 d = c;
  int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(A(row, k), B(k, col), acc);
    d[r] = acc;
  }
  hrf_emul::sync_wave();
  return d;
}

inline void hrf_atomic_add(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  do {
    float f; std::memcpy(&f, &old, 4); f += v; std::memcpy(&nw, &f, 4);
  } while (!__atomic_compare_exchange_n(ip, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
inline void hrf_atomic_add(double* p, double v) {
  uint64_t* ip = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  do {
    double f; std::memcpy(&f, &old, 8); f += v; std::memcpy(&nw, &f, 8);
  } while (!__atomic_compare_exchange_n(ip, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
inline float atomicAdd(float* p, float v) { hrf_atomic_add(p, v); return 0.f; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

inline bool __any(bool p) {
  int v = p ? 1 : 0;
  for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m);
  return v != 0;
}
inline float hrf_row16_sum(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // wave-uniform by construction at the call sites
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
using std::max;
using std::min;

// host-side HIP API subset used by the launchers
inline int hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }
