// prof.hip -- optional per-kernel timing INSIDE the C entries (sl_prof_enable / sl_prof_dump).
//
// bench.py reports the live duration and the algorithmic bytes / flops of every hand-written kernel.  Timing them from
// Python (an event pair around each ctypes call) forces the kernel-by-kernel call path -- but the steps that are timed
// run the one-call layer entries (sl_sage_fwd / sl_sage_bwd_chain ...), whose kernels differ (GEMM epilogue fusions).
// With the scopes here every sl_* entry records a HIP-event pair around its own launch on the stream it launches on,
// whatever the call path: the measured path IS the timed path.  Off (the default) a scope is one branch.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace shadow {
namespace {

struct Rec {
  std::string name;
  hipEvent_t e0, e1;
  double bytes, flops;
};

std::mutex g_mu;
std::atomic<bool> g_on{false};
uint32_t g_gen = 0;                            // bumped whenever g_recs is cleared: a scope that straddles the clear drops its end event
std::vector<Rec> g_recs;
std::map<std::string, uint64_t> g_counts;      // launches per kernel class (all of them; only the first kMaxPerName carry events)
std::map<std::string, uint32_t> g_timed;
constexpr uint32_t kMaxPerName = 512;          // keep the number of live HIP events bounded on long runs

}  // namespace

bool prof_enabled() { return g_on.load(std::memory_order_relaxed); }

ProfScope::ProfScope(const char *name, double bytes, double flops, hipStream_t st) : idx(-1), gen(0), st(st) {
  if (!g_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on.load()) return;
  g_counts[name]++;
  if (g_timed[name] >= kMaxPerName) return;
  Rec r;
  r.name = name; r.bytes = bytes; r.flops = flops;
  if (hipEventCreate(&r.e0) != hipSuccess) return;
  if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
  (void)hipEventRecord(r.e0, st);
  g_timed[name]++;
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
  gen = g_gen;
}

ProfScope::~ProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (gen == g_gen && idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].e1, st);
}


// ---- fill_words (common.h): 16 bytes per lane and iteration, one workgroup per CU-slot at most
__global__ void __launch_bounds__(256) fill_words_kernel(uint32_t *__restrict__ dst, uint32_t value, size_t nwords) {
  const size_t nvec = nwords / 4;
  uint4 *d4 = reinterpret_cast<uint4 *>(dst);
  const uint4 v4 = make_uint4(value, value, value, value);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) d4[i] = v4;
  for (size_t i = nvec * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) dst[i] = value;
}

int fill_words(void *dst, uint32_t value, size_t nwords, hipStream_t st) {
  if (nwords == 0) return SG_OK;
  if (((uintptr_t)dst & 15u) != 0) {           // (not 16-byte aligned: the runtime's own fill)
    SHD_HIP(hipMemsetD32Async((hipDeviceptr_t)dst, (int)value, nwords, st));
    return SG_OK;
  }
  const size_t blocks = (nwords / 4 + 255) / 256;
  hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)std::min<size_t>(std::max<size_t>(blocks, 1), 2048)), dim3(256), 0, st,
                     reinterpret_cast<uint32_t *>(dst), value, nwords);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

}  // namespace shadow

using namespace shadow;

extern "C" int sl_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int prev = g_on.load() ? 1 : 0;
  if (on > 0 && !prev) {
    g_gen++;
    for (auto &r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_recs.clear(); g_counts.clear(); g_timed.clear();
  }
  if (on >= 0) g_on.store(on != 0);
  return prev;
}

// One line per kernel class: "name\tlaunches\ttimed\ttotal_ms\tbytes_sum\tflops_sum\n" (the sums over the timed launches).
// Returns the number of bytes the text needs (incl. the terminating 0); writes at most `cap`.  Waits for the events.
extern "C" size_t sl_prof_dump(char *buf, size_t cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  struct Agg { uint32_t n = 0; double ms = 0, by = 0, fl = 0; };
  std::map<std::string, Agg> agg;
  for (auto &r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    Agg &a = agg[r.name];
    a.n++; a.ms += ms; a.by += r.bytes; a.fl += r.flops;
  }
  std::string out;
  char line[512];
  for (auto &kv : agg) {
    snprintf(line, sizeof(line), "%s\t%llu\t%u\t%.6f\t%.0f\t%.0f\n", kv.first.c_str(), (unsigned long long)g_counts[kv.first], kv.second.n,
             kv.second.ms, kv.second.by, kv.second.fl);
    out += line;
  }
  if (buf && cap) {
    const size_t n = std::min(cap - 1, out.size());
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}
