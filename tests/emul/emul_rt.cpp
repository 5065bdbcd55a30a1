// Fiber scheduler of the CPU kernel emulator (see emul_rt.h).  TEST INFRASTRUCTURE ONLY.
#include "emul_rt.h"

#include <ucontext.h>

#include <atomic>
#include <thread>
#include <vector>

namespace hrf_emul {

enum State { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, FINISHED = 3 };

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  State state = FINISHED;
  Tid tid{0, 0, 0};
  int lane = 0, wave = 0;
};

static constexpr size_t kStack = 192 * 1024;
static constexpr int kMaxThreads = 1024;

struct Worker {
  std::vector<Fiber> fibers;
  ucontext_t sched;
  const std::function<void()>* body = nullptr;
  int running = -1;
  std::vector<unsigned char> smem;
  std::vector<unsigned char> wavebuf;   // (kMaxThreads/64) x 64 x 16 B
  Worker() : fibers(kMaxThreads), smem(160 * 1024 + 64), wavebuf((kMaxThreads / 64) * 64 * 16) {}
  ~Worker() { for (auto& f : fibers) std::free(f.stack); }
};

thread_local Tid* cur_tid = nullptr;
thread_local dim3 bidx, bdim, gdim;
thread_local int cur_lane = 0;
static thread_local Worker* W = nullptr;

static void yield_with(State s) {
  Fiber& f = W->fibers[W->running];
  f.state = s;
  swapcontext(&f.ctx, &W->sched);
}
void sync_block() { yield_with(WAIT_BLOCK); }
void sync_wave() { yield_with(WAIT_WAVE); }
void* wave_buf() { return W->wavebuf.data() + (size_t)W->fibers[W->running].wave * 64 * 16; }
void* dyn_smem() {
  uintptr_t p = reinterpret_cast<uintptr_t>(W->smem.data());
  return reinterpret_cast<void*>((p + 15) & ~uintptr_t(15));
}

static void fiber_entry() {
  (*W->body)();
  W->fibers[W->running].state = FINISHED;
  // returning resumes uc_link (= scheduler)
}

static void run_block(int nthreads) {
  Worker& w = *W;
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = w.fibers[t];
    if (!f.stack) f.stack = static_cast<char*>(std::malloc(kStack));
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &w.sched;
    makecontext(&f.ctx, fiber_entry, 0);
    f.state = RUNNABLE;
    f.tid.x = t % bdim.x;
    f.tid.y = (t / bdim.x) % bdim.y;
    f.tid.z = t / (bdim.x * bdim.y);
    f.lane = t & 63;
    f.wave = t >> 6;
  }
  const int nwaves = (nthreads + 63) / 64;
  for (;;) {
    bool ran = false;
    for (int t = 0; t < nthreads; ++t) {
      Fiber& f = w.fibers[t];
      if (f.state != RUNNABLE) continue;
      ran = true;
      w.running = t;
      cur_tid = &f.tid;
      cur_lane = f.lane;
      swapcontext(&w.sched, &f.ctx);
    }
    // release wave-level rendezvous
    bool released = false;
    for (int wv = 0; wv < nwaves; ++wv) {
      int lo = wv * 64, hi = std::min(nthreads, lo + 64), waiting = 0, live = 0;
      for (int t = lo; t < hi; ++t) {
        if (w.fibers[t].state != FINISHED) ++live;
        if (w.fibers[t].state == WAIT_WAVE) ++waiting;
      }
      if (live > 0 && waiting == live) {
        for (int t = lo; t < hi; ++t)
          if (w.fibers[t].state == WAIT_WAVE) w.fibers[t].state = RUNNABLE;
        released = true;
      }
    }
    int live = 0, waiting = 0;
    for (int t = 0; t < nthreads; ++t) {
      if (w.fibers[t].state != FINISHED) ++live;
      if (w.fibers[t].state == WAIT_BLOCK) ++waiting;
    }
    if (live == 0) return;
    if (waiting == live) {
      for (int t = 0; t < nthreads; ++t)
        if (w.fibers[t].state == WAIT_BLOCK) w.fibers[t].state = RUNNABLE;
      released = true;
    }
    if (!ran && !released) {
      std::fprintf(stderr,
                   "hrf_emul: DEADLOCK in block (%u,%u,%u): divergent __syncthreads / wave "
                   "op (live=%d, at block barrier=%d)\n", bidx.x, bidx.y, bidx.z, live, waiting);
      std::abort();
    }
  }
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads <= 0 || nthreads > kMaxThreads || smem_bytes > 160 * 1024) {
    std::fprintf(stderr, "hrf_emul: bad launch config (%d threads, %zu smem)\n", nthreads, smem_bytes);
    std::abort();
  }
  const long nblocks = (long)grid.x * grid.y * grid.z;
  if (nblocks <= 0) return;
  int nworkers = (int)std::min<long>(nblocks, std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
  const char* env = std::getenv("HRF_EMUL_THREADS");
  if (env) nworkers = std::max(1, std::min(nworkers, std::atoi(env)));
  std::atomic<long> next{0};
  auto work = [&]() {
    static thread_local Worker* mine = nullptr;
    if (!mine) mine = new Worker();       // kept for the life of the OS thread
    W = mine;
    W->body = &body;
    bdim = block;
    gdim = grid;
    for (;;) {
      long b = next.fetch_add(1);
      if (b >= nblocks) break;
      bidx.x = (unsigned)(b % grid.x);
      bidx.y = (unsigned)((b / grid.x) % grid.y);
      bidx.z = (unsigned)(b / ((long)grid.x * grid.y));
      run_block(nthreads);
    }
  };
  if (nworkers == 1) {
    work();
  } else {
    // persistent pool would be faster; per-launch threads keep the emulator trivially correct
    std::vector<std::thread> pool;
    for (int i = 0; i < nworkers; ++i) pool.emplace_back([&]() {
      work();
      delete W; W = nullptr;              // thread exits: free its fiber stacks
    });
    for (auto& t : pool) t.join();
  }
}

}  // namespace hrf_emul
