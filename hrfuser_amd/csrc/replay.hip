// See hrf_replay.h.  C-ABI: hrf_rec_begin / hrf_rec_sync / hrf_rec_end / hrf_replay / hrf_replay_free,
// hrf_memset (recordable hipMemsetAsync).
#include "hrf_rt.h"
#include "../../include/hrfuser_hip.h"

#ifdef HRF_EMUL
// CPU emulator build (tests): no recording, hrf_memset is a plain memset
#include <cstring>
extern "C" int hrf_rec_begin(void) { return HRF_ERR_ARG; }
extern "C" int hrf_rec_sync(void*, void*) { return HRF_OK; }
extern "C" int hrf_rec_end(void) { return -1; }
extern "C" int hrf_replay(int) { return HRF_ERR_ARG; }
extern "C" int hrf_replay_free(int) { return HRF_ERR_ARG; }
extern "C" int hrf_replay_info(int, int) { return -1; }
extern "C" int hrf_memset(void* ptr, int value, long bytes, void*) { if (bytes > 0) std::memset(ptr, value, (size_t)bytes); return HRF_OK; }
#else
#include "hrf_replay.h"

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace hrf_rp {

struct StreamList { void* stream; std::vector<Cmd> cmds; };
struct Program {
  std::vector<StreamList> lists;
  int nevents = 0;
  std::vector<hipEvent_t> events;
  std::unique_ptr<std::atomic<int>[]> issued;
  long launches = 0;
};

static std::mutex g_mu;
static bool g_rec = false;
static Program* g_cur = nullptr;
static std::map<int, Program*> g_progs;
static int g_next_id = 1;

bool recording() { return g_rec; }

static StreamList& list_of(Program& p, void* stream) {
  for (auto& l : p.lists)
    if (l.stream == stream) return l;
  p.lists.push_back(StreamList{stream, {}});
  return p.lists.back();
}

void push(void* stream, Cmd&& c) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_rec || g_cur == nullptr) return;
  if (c.type == 0) ++g_cur->launches;
  list_of(*g_cur, stream).cmds.push_back(std::move(c));
}

static void run_list(Program* p, StreamList* l, std::atomic<int>* err) {
  hipStream_t s = (hipStream_t)l->stream;
  std::vector<void*> argv;
  for (Cmd& c : l->cmds) {
    hipError_t e = hipSuccess;
    switch (c.type) {
      case 0:
        argv.resize(c.offs.size());
        for (size_t i = 0; i < c.offs.size(); ++i) argv[i] = c.blob.data() + c.offs[i];
        e = hipLaunchKernel(c.func, c.grid, c.block, argv.data(), c.smem, s);
        break;
      case 1:
        e = hipEventRecord(p->events[c.ev], s);
        p->issued[c.ev].store(1, std::memory_order_release);
        break;
      case 2:
        while (p->issued[c.ev].load(std::memory_order_acquire) == 0) {   // the record call of another thread
          if (err->load(std::memory_order_relaxed)) return;
          std::this_thread::yield();
        }
        e = hipStreamWaitEvent(s, p->events[c.ev], 0);
        break;
      case 3:
        e = hipMemsetAsync(c.ptr, c.value, c.bytes, s);
        break;
    }
    if (e != hipSuccess) { err->store(1); return; }
  }
}

}  // namespace hrf_rp

using namespace hrf_rp;

extern "C" int hrf_rec_begin(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rec) return HRF_ERR_ARG;
  g_cur = new Program();
  g_rec = true;
  return HRF_OK;
}

// "everything enqueued so far on `src` happens before what `dst` enqueues next"
extern "C" int hrf_rec_sync(void* src, void* dst) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_rec || g_cur == nullptr) return HRF_OK;
  const int ev = g_cur->nevents++;
  Cmd r{}; r.type = 1; r.ev = ev;
  Cmd w{}; w.type = 2; w.ev = ev;
  list_of(*g_cur, src).cmds.push_back(std::move(r));
  list_of(*g_cur, dst).cmds.push_back(std::move(w));
  return HRF_OK;
}

// returns a program id (> 0), or -1
extern "C" int hrf_rec_end(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_rec || g_cur == nullptr) return -1;
  g_rec = false;
  Program* p = g_cur;
  g_cur = nullptr;
  p->events.resize(p->nevents);
  for (auto& e : p->events)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
  p->issued.reset(new std::atomic<int>[p->nevents > 0 ? p->nevents : 1]);
  const int id = g_next_id++;
  g_progs[id] = p;
  return id;
}

extern "C" int hrf_replay_info(int prog, int what) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_progs.find(prog);
  if (it == g_progs.end()) return -1;
  Program* p = it->second;
  if (what == 0) return (int)p->launches;
  if (what == 1) return (int)p->lists.size();
  if (what == 2) return p->nevents;
  return -1;
}

// Enqueue the whole program (returns when everything is enqueued, not when it has executed).
extern "C" int hrf_replay(int prog) {
  Program* p;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_progs.find(prog);
    if (it == g_progs.end()) return HRF_ERR_ARG;
    p = it->second;
  }
  for (int i = 0; i < p->nevents; ++i) p->issued[i].store(0, std::memory_order_relaxed);
  std::atomic<int> err{0};
  std::vector<std::thread> th;
  for (size_t i = 1; i < p->lists.size(); ++i) th.emplace_back(run_list, p, &p->lists[i], &err);
  if (!p->lists.empty()) run_list(p, &p->lists[0], &err);
  for (auto& t : th) t.join();
  return err.load() ? HRF_ERR_LAUNCH : HRF_OK;
}

extern "C" int hrf_replay_free(int prog) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_progs.find(prog);
  if (it == g_progs.end()) return HRF_ERR_ARG;
  for (auto& e : it->second->events) (void)hipEventDestroy(e);
  delete it->second;
  g_progs.erase(it);
  return HRF_OK;
}

extern "C" int hrf_memset(void* ptr, int value, long bytes, void* stream) {
  if (bytes <= 0) return HRF_OK;
  if (recording()) {
    Cmd c{};
    c.type = 3; c.ptr = ptr; c.value = value; c.bytes = (size_t)bytes;
    push(stream, std::move(c));
  }
  return hipMemsetAsync(ptr, value, (size_t)bytes, (hipStream_t)stream) == hipSuccess ? HRF_OK : HRF_ERR_LAUNCH;
}
#endif  // HRF_EMUL
