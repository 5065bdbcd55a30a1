// Queue + merge logic of the multi-problem launches (hrf_group.h); C-ABI: hrf_group_begin / hrf_group_end /
// hrf_group_count (include/hrfuser_hip.h).
#include <vector>
#include "hrf_group.h"
#include "../../include/hrfuser_hip.h"

namespace {
struct Pending {
  const void* kern;
  dim3 grid, block;
  unsigned smem, argsize;
  int call;
  hrf_grp_issue_t issue;
  size_t off;           // of the argument bytes in g_blob
};
thread_local bool g_on = false;
thread_local int g_call = -1;
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_q;
thread_local std::vector<unsigned char> g_blob;
long g_count[3] = {0, 0, 0};   // launches issued by hrf_group_end, problems they carried, hrf_group_end calls

inline bool same(const Pending& a, const Pending& b) {
  return a.kern == b.kern && a.issue == b.issue && a.argsize == b.argsize && a.smem == b.smem &&
         a.grid.x == b.grid.x && a.grid.y == b.grid.y && a.grid.z == b.grid.z &&
         a.block.x == b.block.x && a.block.y == b.block.y && a.block.z == b.block.z;
}
}  // namespace

bool hrf_grp_collecting() { return g_on; }
void hrf_grp_enter() {
  if (g_depth++ == 0 && g_on) ++g_call;
}
void hrf_grp_leave() { --g_depth; }
void hrf_grp_push(const void* kern, dim3 grid, dim3 block, unsigned smem, const void* args, unsigned argsize,
                  hrf_grp_issue_t issue) {
  Pending p;
  p.kern = kern; p.grid = grid; p.block = block; p.smem = smem; p.argsize = argsize;
  p.call = g_call < 0 ? 0 : g_call;
  p.issue = issue;
  p.off = (g_blob.size() + 15) & ~(size_t)15;
  g_blob.resize(p.off + argsize);
  std::memcpy(g_blob.data() + p.off, args, argsize);
  g_q.push_back(p);
}

extern "C" int hrf_group_begin(void) {
  g_q.clear();
  g_blob.clear();
  g_call = -1;
  g_on = true;
  return HRF_OK;
}

extern "C" int hrf_group_end(void* stream) {
  g_on = false;
  ++g_count[2];
  const int n = (int)g_q.size();
  if (n == 0) return HRF_OK;
  const int ncalls = g_q.back().call + 1;
  std::vector<std::vector<int>> calls(ncalls);
  size_t maxlen = 0;
  for (int i = 0; i < n; ++i) {
    calls[g_q[i].call].push_back(i);
    maxlen = calls[g_q[i].call].size() > maxlen ? calls[g_q[i].call].size() : maxlen;
  }
  std::vector<char> done(n, 0);
  int rc = HRF_OK;
  // position k of every call, in call order: the launches of ONE call keep their order; launches of different calls are
  // independent by contract and merge when kernel instantiation and launch geometry are identical
  for (size_t k = 0; k < maxlen; ++k) {
    for (int c = 0; c < ncalls; ++c) {
      if (calls[c].size() <= k || done[calls[c][k]]) continue;
      const int i = calls[c][k];
      const unsigned char* argv[HRF_GROUP_MAX];
      int m = 0;
      argv[m++] = g_blob.data() + g_q[i].off;
      done[i] = 1;
      if (g_q[i].grid.z == 1) {
        for (int c2 = c + 1; c2 < ncalls && m < HRF_GROUP_MAX; ++c2) {
          if (calls[c2].size() <= k) continue;
          const int j = calls[c2][k];
          if (done[j] || !same(g_q[i], g_q[j])) continue;
          argv[m++] = g_blob.data() + g_q[j].off;
          done[j] = 1;
        }
      }
      const int r = g_q[i].issue(g_q[i].kern, g_q[i].grid, g_q[i].block, g_q[i].smem, stream, argv, m);
      if (r != HRF_OK) rc = r;
      ++g_count[0];
      g_count[1] += m;
    }
  }
  g_q.clear();
  g_blob.clear();
  return rc;
}

extern "C" long hrf_group_count(int what) {
  if (what == 3) return HRF_GROUP_MAX;                 // problems per launch this library was compiled for (1: pass-through)
  return (what >= 0 && what < 3) ? g_count[what] : -1;
}
