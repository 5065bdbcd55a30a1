// Peer-to-peer SyncBN exchange over xGMI (gfx950) - no communicator, one launch per exchange, on the lane that needs it.
//
// What is exchanged: the BatchNorm moments of a layer (torch.nn.SyncBatchNorm as the reference configs build it:
// configs/_base_/models/cascade_rcnn_hrfuser_fpn_nus_clr_fusion.py:2, norm_cfg type SyncBN): per layer 2*C doubles plus this
// rank's sample count, forward (sum y, sum y^2) and backward (sum du, sum du*y).  330-374 layers x 2 directions per step, each a
// dependency of the very next kernel of its lane - a latency problem (288 B .. 23 KB per message), not a bandwidth one.
//
// Through RCCL every exchange is a collective of ONE communicator, and the collectives of a communicator must execute in the
// same order on every rank: the lanes (HIP streams) of the step had to hop to the main lane for every exchange (2(k-1)
// cross-stream edges of ~10 us each; +4.4 ms per step on ONE rank, profiles/r03_bench_forced_rccl.json).  Here there is
// nothing to order: every BatchNorm layer owns a slot in an inbox that each rank exposes to its peers (hipIpcGetMemHandle /
// hipIpcOpenMemHandle; fine-grained device memory), and ONE small kernel per exchange
//     1. folds the rank's replicated moment copies (what hrf_bn_pack did),
//     2. stores the result into the slot [source = this rank][generation parity][layer] of every PEER's inbox, fences at
//        system scope and releases a flag = the step generation next to it,
//     3. spins (acquire, system scope, with a time-out) until the flags of all peers show this generation,
//     4. adds the contributions in rank order (bit-identical totals on every rank) into `packed` - the buffer the consumers'
//        finalize-on-load prologues read, exactly where the all-reduced buffer used to be.
// The generation is a device-side step counter (hrf_p2p_tick, one launch per step), so a captured hipGraph replays it; slots
// are double-buffered by its parity.  A rank cannot overwrite a slot its peer has not consumed yet: it would first need the
// peer's contribution to a later exchange of the same step, and the gradient exchange at the end of a step is a barrier.
//
// A peer that does not arrive within the time-out (default: NCCL's 30 minutes, hrfuser_amd/p2p.py) is an ERROR, never a
// result: the launch writes (source, slot) into the error word and fills its part of `packed` with NaN, and every later
// exchange of the process does the same without waiting - the statistics of a step that lost an exchange are poison, the loss
// turns NaN at once and the host raises at its next step boundary (P2PExchange.poll / check).  Nothing is ever reduced from
// an inbox whose flag did not arrive.
//
// Emulator (HRF_EMUL, tests only): inboxes are POSIX shared-memory objects, so two emulator PROCESSES run the same protocol
// (tests/test_dp_gloo.py); a wait with timeout_ticks <= 0 fails at once there (one launch at a time inside one process).
#include <cstdlib>
#include <cstring>
#ifdef HRF_EMUL
#include <cstdio>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <map>
#include <mutex>
#include <string>
#endif
#include "hrf_common.h"
#include "../../include/hrfuser_hip.h"

namespace {

constexpr int PX_MAX = 8;        // layers per launch (one workgroup each)

struct P2pArgs {
  hrf_p2p_t ctx;
  const double* src[PX_MAX]; int C[PX_MAX]; double rows[PX_MAX]; long slot_off[PX_MAX]; int slot_id[PX_MAX];
  int off[PX_MAX]; int roff[PX_MAX];       // where the layer's sums / count go in `packed`
  int has_rows, phase;
};

#ifdef HRF_EMUL
__device__ inline void px_store(double* p, double v) { __atomic_store(p, &v, __ATOMIC_RELAXED); }
__device__ inline double px_load(const double* p) { double v; __atomic_load(p, &v, __ATOMIC_RELAXED); return v; }
__device__ inline void px_flag_release(long long* p, long long v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
__device__ inline long long px_flag_acquire(const long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
__device__ inline void px_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
__device__ inline long long px_clock() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (long long)t.tv_sec * 100000000LL + t.tv_nsec / 10; }
__device__ inline void px_sleep() { sched_yield(); }
__device__ inline double px_nan() { return __builtin_nan(""); }
#else
// every access to an inbox is a system-scope operation: the data lives in (possibly remote) fine-grained memory that another
// GPU writes while this kernel runs
__device__ __forceinline__ void px_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double px_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void px_flag_release(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ long long px_flag_acquire(const long long* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void px_fence() { __threadfence_system(); }
__device__ __forceinline__ long long px_clock() { return (long long)wall_clock64(); }
__device__ __forceinline__ void px_sleep() { __builtin_amdgcn_s_sleep(8); }
__device__ __forceinline__ double px_nan() { return __builtin_nan(""); }
#endif

__global__ __launch_bounds__(256) void p2p_exchange_kernel(P2pArgs a, double* packed) {
  const hrf_p2p_t& x = a.ctx;
  const int e = blockIdx.x, tid = threadIdx.x;
  const int C2 = 2 * a.C[e], len = C2 + (a.has_rows ? 1 : 0);
  const long long gen = (long long)*x.gen;
  const long par = (long)(gen & 1);
  const long mine = ((long)x.rank * 2 + par) * x.slot_doubles + a.slot_off[e];
  // this rank's own contribution never travels: it is folded again (the same arithmetic, bit for bit) when the totals are
  // formed, so a one-rank group costs what hrf_bn_pack cost and an N-rank one waits for its N - 1 peers only
  auto fold = [&](int c) -> double {
    if (c >= C2) return a.rows[e];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < HRF_STAT_COPIES; ++k) s += a.src[e][(size_t)k * C2 + c];
    return s;
  };
  if ((a.phase & 1) && x.world > 1) {
    for (int c = tid; c < len; c += 256) {
      const double s = fold(c);
      for (int p = 0; p < x.world; ++p)
        if (p != x.rank) px_store(x.inbox[p] + mine + c, s);
    }
    px_fence();                                   // the data before the flag, system-wide
    __syncthreads();
    if (tid < x.world && tid != x.rank)
      px_flag_release(reinterpret_cast<long long*>(x.flags[tid]) + ((long)x.rank * 2 + par) * x.nslots + a.slot_id[e], gen);
  }
  if (!(a.phase & 2)) return;
  __shared__ int sBad;                             // a source of this layer never arrived (now, or earlier in this process)
  if (tid == 0) sBad = 0;
  __syncthreads();
  if (tid < x.world && tid != x.rank) {
    const long long* f = reinterpret_cast<const long long*>(x.flags[x.rank]) + ((long)tid * 2 + par) * x.nslots + a.slot_id[e];
    const long long t0 = px_clock();
    // an exchange that already timed out in this process is not waited for again: the step is lost - its result is NaN below
    bool bad = x.err != nullptr && px_flag_acquire(reinterpret_cast<const long long*>(x.err)) != 0;
    while (!bad && px_flag_acquire(f) != gen) {
#ifdef HRF_EMUL
      if (x.timeout_ticks <= 0) bad = true;         // one launch at a time inside one process: nobody could deliver it
#endif
      px_sleep();
      if (x.timeout_ticks > 0 && px_clock() - t0 > x.timeout_ticks) bad = true;
      if (bad && x.err != nullptr)                  // the peer never arrived: flag it, do not hang the GPU
        px_flag_release(reinterpret_cast<long long*>(x.err), ((long long)(tid + 1) << 32) | (long long)(a.slot_id[e] + 1));
    }
    if (bad) sBad = 1;
  }
  __syncthreads();
  if (x.world > 1) px_fence();
  const bool poison = sBad != 0;
  const double* in = x.inbox[x.rank];
  for (int c = tid; c < len; c += 256) {
    double s = 0.0;
    for (int r = 0; r < x.world; ++r)             // rank order on every rank: bit-identical totals everywhere
      s += r == x.rank ? fold(c) : px_load(in + ((long)r * 2 + par) * x.slot_doubles + a.slot_off[e] + c);
    packed[c < C2 ? a.off[e] + c : a.roff[e]] = poison ? px_nan() : s;
  }
}

__global__ void p2p_tick_kernel(long long* gen) { *gen += 1; }

}  // namespace

extern "C" int hrf_p2p_tick(long* gen, void* stream) {
  if (gen == nullptr) return HRF_ERR_ARG;
  HRF_LAUNCH(p2p_tick_kernel, dim3(1), dim3(1), 0, stream, reinterpret_cast<long long*>(gen));
  return hrf_check_launch();
}

extern "C" int hrf_p2p_exchange(const hrf_p2p_t* ctx, const double* const* stats, const int* C, int n, const double* rows,
                                const long* slot_off, const int* slot_id, double* packed, int phase, void* stream) {
  if (n <= 0) return HRF_OK;
  if (ctx == nullptr || stats == nullptr || C == nullptr || slot_off == nullptr || slot_id == nullptr || packed == nullptr) return HRF_ERR_ARG;
  if (ctx->world < 1 || ctx->world > HRF_P2P_MAX_RANKS || ctx->rank < 0 || ctx->rank >= ctx->world || ctx->gen == nullptr) return HRF_ERR_ARG;
  for (int p = 0; p < ctx->world; ++p) if (ctx->inbox[p] == nullptr || ctx->flags[p] == nullptr) return HRF_ERR_ARG;
  int tail = 0;
  for (int k = 0; k < n; ++k) {
    if (C[k] <= 0 || slot_id[k] < 0 || slot_id[k] >= ctx->nslots || slot_off[k] < 0 ||
        slot_off[k] + 2 * C[k] + 1 > ctx->slot_doubles) return HRF_ERR_ARG;
    tail += 2 * C[k];
  }
  int off = 0;
  for (int b = 0; b < n; b += PX_MAX) {
    P2pArgs a{};
    a.ctx = *ctx;
    a.has_rows = rows != nullptr ? 1 : 0;
    a.phase = phase <= 0 ? 3 : phase;
    const int m = n - b < PX_MAX ? n - b : PX_MAX;
    for (int k = 0; k < m; ++k) {
      a.src[k] = stats[b + k]; a.C[k] = C[b + k]; a.rows[k] = rows != nullptr ? rows[b + k] : 0.0;
      a.slot_off[k] = slot_off[b + k]; a.slot_id[k] = slot_id[b + k];
      a.off[k] = off; off += 2 * C[b + k];
      a.roff[k] = tail + b + k;
    }
    HRF_LAUNCH(p2p_exchange_kernel, dim3(m), dim3(256), 0, stream, a, packed);
  }
  return hrf_check_launch();
}

// ---- inbox memory: fine-grained device memory that peers map through IPC handles (one process per GPU; two processes on one
// GPU work as well, which is how the build container's single-GPU box tests the protocol).  Emulator: a POSIX shared-memory
// object per inbox, the "IPC handle" is its name + size.
#ifdef HRF_EMUL
namespace {
struct ShmEnt { size_t bytes; std::string name; bool owner; };
std::mutex g_shm_mu;
std::map<void*, ShmEnt> g_shm;
int g_shm_seq = 0;
}  // namespace
#endif

extern "C" int hrf_p2p_alloc(long bytes, void** ptr, void* handle64) {
  if (ptr == nullptr || bytes <= 0) return HRF_ERR_ARG;
#ifdef HRF_EMUL
  std::lock_guard<std::mutex> lk(g_shm_mu);
  char name[48];
  snprintf(name, sizeof name, "/hrf_p2p_%d_%d", (int)getpid(), g_shm_seq++);
  (void)shm_unlink(name);
  const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return HRF_ERR_LAUNCH;
  if (ftruncate(fd, bytes) != 0) { close(fd); shm_unlink(name); return HRF_ERR_LAUNCH; }
  void* p = mmap(nullptr, (size_t)bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { shm_unlink(name); return HRF_ERR_LAUNCH; }
  std::memset(p, 0, (size_t)bytes);
  g_shm[p] = ShmEnt{(size_t)bytes, name, true};
  if (handle64 != nullptr) {
    std::memset(handle64, 0, 64);
    std::memcpy(handle64, name, strlen(name) + 1);
    const long long nb = bytes;
    std::memcpy(static_cast<char*>(handle64) + 48, &nb, 8);
  }
  *ptr = p;
  return HRF_OK;
#else
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) return HRF_ERR_LAUNCH;
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return HRF_ERR_LAUNCH; }
  if (handle64 != nullptr) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipFree(p); return HRF_ERR_LAUNCH; }
    std::memcpy(handle64, &h, 64);
  }
  *ptr = p;
  return HRF_OK;
#endif
}

extern "C" int hrf_p2p_open(const void* handle64, void** ptr) {
  if (handle64 == nullptr || ptr == nullptr) return HRF_ERR_ARG;
#ifdef HRF_EMUL
  char name[48];
  std::memcpy(name, handle64, 48);
  name[47] = 0;
  long long nb = 0;
  std::memcpy(&nb, static_cast<const char*>(handle64) + 48, 8);
  if (name[0] != '/' || nb <= 0) return HRF_ERR_ARG;
  const int fd = shm_open(name, O_RDWR, 0600);
  if (fd < 0) return HRF_ERR_LAUNCH;
  void* p = mmap(nullptr, (size_t)nb, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return HRF_ERR_LAUNCH;
  std::lock_guard<std::mutex> lk(g_shm_mu);
  g_shm[p] = ShmEnt{(size_t)nb, name, false};
  *ptr = p;
  return HRF_OK;
#else
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return HRF_ERR_LAUNCH; }
  *ptr = p;
  return HRF_OK;
#endif
}

#ifdef HRF_EMUL
static int shm_release(void* ptr) {
  if (ptr == nullptr) return HRF_OK;
  std::lock_guard<std::mutex> lk(g_shm_mu);
  auto it = g_shm.find(ptr);
  if (it == g_shm.end()) return HRF_ERR_ARG;
  munmap(ptr, it->second.bytes);
  if (it->second.owner) shm_unlink(it->second.name.c_str());
  g_shm.erase(it);
  return HRF_OK;
}
#endif

extern "C" int hrf_p2p_close(void* ptr) {
#ifdef HRF_EMUL
  return shm_release(ptr);
#else
  return (ptr == nullptr || hipIpcCloseMemHandle(ptr) == hipSuccess) ? HRF_OK : HRF_ERR_LAUNCH;
#endif
}

extern "C" int hrf_p2p_free(void* ptr) {
#ifdef HRF_EMUL
  return shm_release(ptr);
#else
  return (ptr == nullptr || hipFree(ptr) == hipSuccess) ? HRF_OK : HRF_ERR_LAUNCH;
#endif
}
