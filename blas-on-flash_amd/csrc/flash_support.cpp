// flash_support.cpp -- small host-side services shared by the level-3 pipelines: the per-device
// call lock, copy streams, the pinned-block cache, roctx ranges and the NUMA placement of the
// I/O threads.  (Declared in flash_common.h / bof_internal.h.)
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <ctype.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <thread>
#include <mutex>
#include <string>
#include <sys/syscall.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "bof_hip.h"
#include "bof_internal.h"
#include "flash_common.h"

namespace bof {

static std::recursive_mutex g_call_mu[64];
std::recursive_mutex &device_call_mutex() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void) hipGetLastError(); dev = 0; }
  return g_call_mu[dev & 63];
}



std::atomic<void (*)(void *)> g_crash_dump_fn{nullptr};
std::atomic<void *> g_crash_dump_arg{nullptr};

// ---- per-device cache of call-lifetime HBM blocks (flash_common.h) -----------------------------------------
namespace {
struct DevCache {
  std::mutex mu;
  std::multimap<size_t, void *> free_[64];
  std::unordered_map<void *, std::pair<int, size_t>> live;
  size_t cached[64] = {};
} g_devc;
constexpr size_t kDevCacheCap = 8ull << 30;
}  // namespace
int dev_cache_alloc(void **p, size_t bytes) {
  int dev = 0;
  BOF_HIP_TRY(hipGetDevice(&dev));
  dev &= 63;
  if (bytes == 0) bytes = 4;
  {
    std::lock_guard<std::mutex> lk(g_devc.mu);
    auto it = g_devc.free_[dev].lower_bound(bytes);
    if (it != g_devc.free_[dev].end() && it->first <= std::max(2 * bytes, bytes + (8u << 20))) {
      *p = it->second;
      g_devc.live[*p] = std::make_pair(dev, it->first);
      g_devc.cached[dev] -= it->first;
      g_devc.free_[dev].erase(it);
      return BOF_OK;
    }
  }
  const size_t rounded = (bytes + 4095) / 4096 * 4096;
  hipError_t e = hipMalloc(p, rounded);
  if (e != hipSuccess) {          // the cache may be what is in the way
    (void) hipGetLastError();
    dev_cache_release();
    e = hipMalloc(p, rounded);
  }
  if (e == hipErrorOutOfMemory) {
    (void) hipGetLastError();
    set_error("out of HBM: a call-lifetime block of " + std::to_string(rounded >> 20) + " MiB (the block cache was emptied first)");
    return BOF_ENOMEM;
  }
  if (e != hipSuccess) return hip_fail(e, "hipMalloc (call-lifetime block)");
  std::lock_guard<std::mutex> lk(g_devc.mu);
  g_devc.live[*p] = std::make_pair(dev, rounded);
  return BOF_OK;
}
void dev_cache_free(void *p) {
  if (!p) return;
  int dev = -1;
  size_t sz = 0;
  {
    std::lock_guard<std::mutex> lk(g_devc.mu);
    auto it = g_devc.live.find(p);
    if (it != g_devc.live.end()) {
      dev = it->second.first;
      sz = it->second.second;
      g_devc.live.erase(it);
      if (g_devc.cached[dev] + sz <= kDevCacheCap) {
        g_devc.free_[dev].emplace(sz, p);
        g_devc.cached[dev] += sz;
        return;
      }
    }
  }
  if (dev >= 0) {
    DeviceScope ds(dev);
    (void) hipFree(p);
  } else {
    (void) hipFree(p);
  }
}
void dev_cache_release() {
  std::vector<std::pair<int, void *>> drop;
  {
    std::lock_guard<std::mutex> lk(g_devc.mu);
    for (int d = 0; d < 64; d++) {
      for (auto &kv : g_devc.free_[d]) drop.emplace_back(d, kv.second);
      g_devc.free_[d].clear();
      g_devc.cached[d] = 0;
    }
  }
  for (auto &x : drop) {
    DeviceScope ds(x.first);
    (void) hipFree(x.second);
  }
}

// ---- persistent launcher threads (flash_common.h) ------------------------------------------------------------
namespace {
struct Launcher {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::pair<std::function<void()>, std::shared_ptr<LaunchJob>>> q;
};
std::mutex g_launch_mu;
thread_local bool t_launcher = false;
// (never destroyed: the detached threads wait on these objects until the process ends)
std::map<std::pair<int, int>, Launcher *> &launchers() {
  static auto *m = new std::map<std::pair<int, int>, Launcher *>();
  return *m;
}
}  // namespace
std::shared_ptr<LaunchJob> launch_async(int dev, int rep, std::function<void()> fn) {
  Launcher *L = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_launch_mu);
    Launcher *&slot = launchers()[std::make_pair(dev, rep)];
    if (!slot) {
      slot = new Launcher();
      Launcher *me = slot;
      std::thread([me, dev] {
        t_launcher = true;
        (void) hipSetDevice(dev);
        for (;;) {
          std::pair<std::function<void()>, std::shared_ptr<LaunchJob>> job;
          {
            std::unique_lock<std::mutex> lk2(me->mu);
            me->cv.wait(lk2, [&] { return !me->q.empty(); });
            job = std::move(me->q.front());
            me->q.pop_front();
          }
          job.first();
          { std::lock_guard<std::mutex> lk2(job.second->mu); job.second->done = true; }
          job.second->cv.notify_all();
        }
      }).detach();
    }
    L = slot;
  }
  auto job = std::make_shared<LaunchJob>();
  { std::lock_guard<std::mutex> lk(L->mu); L->q.emplace_back(std::move(fn), job); }
  L->cv.notify_one();
  return job;
}
bool on_launcher_thread() { return t_launcher; }
void launch_wait(const std::shared_ptr<LaunchJob> &job) {
  if (!job) return;
  std::unique_lock<std::mutex> lk(job->mu);
  job->cv.wait(lk, [&] { return job->done; });
}

// ---- BOF_VERIFY table (flash_common.h) ---------------------------------------------------------------------
int Verify::init(int device, size_t capacity, size_t spots) {
  release();
  dev = device;
  DeviceScope scope(dev);
  capacity += 2 * spots;
  BOF_HIP_TRY(hipMalloc((void **) &d_tab, capacity * 2 * sizeof(unsigned long long)));
  BOF_HIP_TRY(hipMemset(d_tab, 0, capacity * 2 * sizeof(unsigned long long)));
  if (spots) BOF_HIP_TRY(hipMalloc((void **) &d_spot, spots * 128 * sizeof(float)));
  spot_cap = spots;
  spot_next.store(0);
  // hipMemset of device memory runs on the null stream and may return before it has executed; the pipelines' streams
  // are non-blocking (not ordered behind the null stream), so without this the zeroing could land AFTER the first sums
  // (seen with eight processes sharing the GPU: device-side sums of the first panels read back as zero)
  BOF_HIP_TRY(hipDeviceSynchronize());
  h_tab = std::vector<std::atomic<uint64_t>>(capacity * 2);
  touched = std::vector<std::atomic<uint8_t>>(capacity);
  for (auto &v : h_tab) v.store(0);
  for (auto &v : touched) v.store(0);
  cap = capacity;
  next.store(0);
  expects.clear();
  on = true;
  return BOF_OK;
}
void Verify::release() {
  if (d_tab || d_spot) {
    DeviceScope scope(dev);
    if (d_tab) (void) hipFree(d_tab);
    if (d_spot) (void) hipFree(d_spot);
  }
  d_tab = nullptr;
  d_spot = nullptr;
  spot_cap = 0;
  on = false;
  cap = 0;
}
int Verify::finish(Counters &cnt, const char *call) {
  if (!on) return BOF_OK;
  const size_t n = std::min(next.load(), cap);
  std::vector<unsigned long long> d(2 * std::max<size_t>(n, 1), 0);
  {
    DeviceScope scope(dev);
    BOF_HIP_TRY(hipMemcpy(d.data(), d_tab, 2 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  auto sums = [&](size_t e, uint64_t out[2]) {
    // an entry is filled on ONE side (all its contributions come from the host or all from the device)
    out[0] = h_tab[2 * e].load() + d[2 * e];
    out[1] = h_tab[2 * e + 1].load() + d[2 * e + 1];
  };
  std::lock_guard<std::mutex> lk(mu);
  uint64_t compared = 0, skipped = 0;
  for (const Expect &x : expects) {
    if (x.a >= n || x.b >= n || !touched[x.a].load() || !touched[x.b].load()) { skipped++; continue; }
    uint64_t sa[2], sb[2];
    sums(x.a, sa);
    sums(x.b, sb);
    compared++;
    if (sa[0] == sb[0] && sa[1] == sb[1]) continue;
    char msg[384];
    snprintf(msg, sizeof(msg),
             "%s: BOF_VERIFY mismatch -- %s (ids %d %d %d): sums %016llx/%016llx against %016llx/%016llx%s",
             call, x.what, x.id0, x.id1, x.id2, (unsigned long long) sa[0], (unsigned long long) sa[1],
             (unsigned long long) sb[0], (unsigned long long) sb[1],
             sa[0] == sb[0] ? " (same words, different places)" : "");
    evt("BOF_VERIFY mismatch", x.id0, x.id1, (uint64_t) x.id2);
    fprintf(stderr, "[bof] %s\n", msg);
    evt_dump(stderr, "BOF_VERIFY mismatch");
    set_error(msg);
    cnt.vchecks += compared;
    return BOF_EVERIFY;
  }
  cnt.vchecks += compared;
  (void) skipped;
  return BOF_OK;
}

long env_long(const char *name, long dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atol(v) : dflt;
}

int resolve_devices(const bof_options &o, std::vector<int> &devs) {
  devs.clear();
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void) hipGetLastError();
    set_error("no HIP device: the flash path has no CPU fallback");
    return BOF_ENODEV;
  }
  if (o.n_devices > 0) {
    if (o.n_devices > BOF_MAX_DEVICES) { set_error("bof_options.n_devices exceeds BOF_MAX_DEVICES"); return BOF_EINVAL; }
    for (int i = 0; i < o.n_devices; i++) devs.push_back(o.devices[i]);
  } else if (const char *e = getenv("BOF_DEVICES")) {
    if (!strcmp(e, "all")) {
      for (int d = 0; d < std::min(count, BOF_MAX_DEVICES); d++) devs.push_back(d);
    } else {
      for (const char *p = e; *p;) {
        char *end = nullptr;
        const long d = strtol(p, &end, 10);
        if (end == p) break;
        devs.push_back((int) d);
        p = end;
        while (*p == ',' || *p == ' ') p++;
      }
    }
    if (devs.size() > BOF_MAX_DEVICES) { set_error("BOF_DEVICES names more than BOF_MAX_DEVICES devices"); return BOF_EINVAL; }
  }
  if (devs.empty()) {
    int cur = 0;
    BOF_HIP_TRY(hipGetDevice(&cur));
    devs.push_back(cur);
  }
  for (int d : devs)
    if (d < 0 || d >= count || d >= 64) {
      set_error("device list: ordinal " + std::to_string(d) + " is not one of the " + std::to_string(count) + " visible devices");
      return BOF_EINVAL;
    }
  return BOF_OK;
}

DeviceCallLock::DeviceCallLock(const std::vector<int> &devs) {
  std::vector<int> u(devs);
  std::sort(u.begin(), u.end());
  u.erase(std::unique(u.begin(), u.end()), u.end());
  for (int d : u) {
    g_call_mu[d & 63].lock();
    held.push_back(&g_call_mu[d & 63]);
  }
}
DeviceCallLock::~DeviceCallLock() {
  for (auto it = held.rbegin(); it != held.rend(); ++it) (*it)->unlock();
}

// ---- per-device pools of call-lifetime HIP events and streams (flash_common.h) ----------------------------------
namespace {
struct HipPools {
  std::mutex mu;
  std::vector<hipEvent_t> ev[64][2];        // [device][timing]
  std::vector<hipStream_t> st[64][2];       // [device][copy priority]
  std::unordered_map<void *, std::pair<int, int>> home;   // handle -> (device, kind)
} g_pools;
}  // namespace
hipError_t pooled_event(hipEvent_t *e, bool timing) {
  int dev = 0;
  hipError_t rc = hipGetDevice(&dev);
  if (rc != hipSuccess) return rc;
  dev &= 63;
  {
    std::lock_guard<std::mutex> lk(g_pools.mu);
    auto &v = g_pools.ev[dev][timing ? 1 : 0];
    if (!v.empty()) { *e = v.back(); v.pop_back(); return hipSuccess; }
  }
  rc = hipEventCreateWithFlags(e, timing ? hipEventDefault : hipEventDisableTiming);
  if (rc != hipSuccess) return rc;
  std::lock_guard<std::mutex> lk(g_pools.mu);
  g_pools.home[(void *) *e] = std::make_pair(dev, timing ? 1 : 0);
  return hipSuccess;
}
void pooled_event_return(hipEvent_t e) {
  if (!e) return;
  std::lock_guard<std::mutex> lk(g_pools.mu);
  auto it = g_pools.home.find((void *) e);
  if (it == g_pools.home.end()) return;       // released meanwhile (bof_flash_release): the handle is gone
  g_pools.ev[it->second.first][it->second.second].push_back(e);
}
hipError_t pooled_stream(hipStream_t *s, bool copy_priority) {
  int dev = 0;
  hipError_t rc = hipGetDevice(&dev);
  if (rc != hipSuccess) return rc;
  dev &= 63;
  {
    std::lock_guard<std::mutex> lk(g_pools.mu);
    auto &v = g_pools.st[dev][copy_priority ? 1 : 0];
    if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; }
  }
  rc = copy_priority ? copy_stream_create(s) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  if (rc != hipSuccess) return rc;
  std::lock_guard<std::mutex> lk(g_pools.mu);
  g_pools.home[(void *) *s] = std::make_pair(dev, copy_priority ? 1 : 0);
  return hipSuccess;
}
void pooled_stream_return(hipStream_t s) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(g_pools.mu);
  auto it = g_pools.home.find((void *) s);
  if (it == g_pools.home.end()) return;
  g_pools.st[it->second.first][it->second.second].push_back(s);
}
void hip_pools_release() {
  std::lock_guard<std::mutex> lk(g_pools.mu);
  for (int d = 0; d < 64; d++) {
    bool any = false;
    for (int q = 0; q < 2; q++) any = any || !g_pools.ev[d][q].empty() || !g_pools.st[d][q].empty();
    if (!any) continue;
    DeviceScope ds(d);
    for (int q = 0; q < 2; q++) {
      for (hipEvent_t e : g_pools.ev[d][q]) { (void) hipEventDestroy(e); g_pools.home.erase((void *) e); }
      for (hipStream_t s : g_pools.st[d][q]) { (void) hipStreamDestroy(s); g_pools.home.erase((void *) s); }
      g_pools.ev[d][q].clear();
      g_pools.st[d][q].clear();
    }
  }
}

hipError_t copy_stream_create(hipStream_t *s) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
    if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest) == hipSuccess) return hipSuccess;
  }
  (void) hipGetLastError();
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// ---- cache of pinned host blocks -------------------------------------------------------------------
namespace {
struct PinnedCache {
  std::mutex mu;
  std::multimap<size_t, void *> free_;          // size -> block
  std::unordered_map<void *, size_t> live;      // handed out
  size_t cached_bytes = 0;
} g_pin;
constexpr size_t kPinnedCacheCap = 4ull << 30;
}  // namespace

int pinned_alloc(void **p, size_t bytes) {
  if (bytes == 0) bytes = 1;
  {
    std::lock_guard<std::mutex> lk(g_pin.mu);
    auto it = g_pin.free_.lower_bound(bytes);
    if (it != g_pin.free_.end() && it->first <= std::max(2 * bytes, bytes + (4u << 20))) {
      *p = it->second;
      g_pin.live[*p] = it->first;
      g_pin.cached_bytes -= it->first;
      g_pin.free_.erase(it);
      return BOF_OK;
    }
  }
  const size_t rounded = (bytes + 4095) / 4096 * 4096;
  hipError_t e = hipHostMalloc(p, rounded, hipHostMallocPortable);
  if (e != hipSuccess) {  // the cache may be what is in the way
    pinned_cache_release();
    e = hipHostMalloc(p, rounded, hipHostMallocPortable);
  }
  if (e != hipSuccess) return hip_fail(e, "hipHostMalloc (pinned staging block)");
  std::lock_guard<std::mutex> lk(g_pin.mu);
  g_pin.live[*p] = rounded;
  return BOF_OK;
}
void pinned_free(void *p) {
  if (!p) return;
  size_t sz = 0;
  {
    std::lock_guard<std::mutex> lk(g_pin.mu);
    auto it = g_pin.live.find(p);
    if (it == g_pin.live.end()) return;
    sz = it->second;
    g_pin.live.erase(it);
    if (g_pin.cached_bytes + sz <= kPinnedCacheCap) {
      g_pin.free_.emplace(sz, p);
      g_pin.cached_bytes += sz;
      return;
    }
  }
  (void) hipHostFree(p);
}
void pinned_cache_release() {
  std::vector<void *> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pin.mu);
    for (auto &kv : g_pin.free_) blocks.push_back(kv.second);
    g_pin.free_.clear();
    g_pin.cached_bytes = 0;
  }
  for (void *b : blocks) (void) hipHostFree(b);
}

// ---- roctx ranges -------------------------------------------------------------------------------
namespace {
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
void roctx_resolve() {
  for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
    void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    g_roctx_push = (roctx_push_fn) dlsym(h, "roctxRangePushA");
    g_roctx_pop = (roctx_pop_fn) dlsym(h, "roctxRangePop");
    if (g_roctx_push && g_roctx_pop) return;
    g_roctx_push = nullptr; g_roctx_pop = nullptr;
  }
}
}  // namespace
void trace_push(const char *name) {
  std::call_once(g_roctx_once, roctx_resolve);
  if (g_roctx_push) (void) g_roctx_push(name);
}
void trace_pop() {
  if (g_roctx_pop) (void) g_roctx_pop();
}

// ---- NUMA placement of the I/O threads -------------------------------------------------------
namespace {
struct InitialMask {
  cpu_set_t set;
  bool ok = false;
  InitialMask() {   // static initialisation: runs on the thread that loads the library
    CPU_ZERO(&set);
    ok = sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0;
  }
} g_initial_mask;
struct NodeCpus { int node = -2; cpu_set_t set; };   // -2 not looked up yet, -1 unknown
std::mutex g_numa_mu;
NodeCpus g_numa[64];

bool parse_cpulist(const char *txt, cpu_set_t *set) {
  CPU_ZERO(set);
  int n = 0;
  const char *p = txt;
  while (*p) {
    char *end = nullptr;
    long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int) c, set); n++; }
    while (*p == ',' || *p == ' ' || *p == '\n') p++;
  }
  return n > 0;
}
bool read_small(const std::string &path, char *buf, size_t cap) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return false;
  const size_t n = fread(buf, 1, cap - 1, f);
  fclose(f);
  buf[n] = 0;
  return n > 0;
}
}  // namespace

int bind_thread_near_device(int dev) {
  const bool enabled = !getenv("BOF_NUMA_BIND") || atoi(getenv("BOF_NUMA_BIND")) != 0;
  if (!enabled || dev < 0 || dev >= 64) return -1;
  NodeCpus nc;
  {
    std::lock_guard<std::mutex> lk(g_numa_mu);
    NodeCpus &c = g_numa[dev];
    if (c.node == -2) {
      c.node = -1;
      char bus[64] = {0}, buf[4096];
      if (hipDeviceGetPCIBusId(bus, (int) sizeof(bus), dev) == hipSuccess) {
        for (char *q = bus; *q; q++) *q = (char) tolower(*q);
        if (read_small(std::string("/sys/bus/pci/devices/") + bus + "/numa_node", buf, sizeof(buf))) {
          const int node = atoi(buf);
          if (node >= 0 &&
              read_small("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, sizeof(buf)) &&
              parse_cpulist(buf, &c.set))
            c.node = node;
        }
      } else {
        (void) hipGetLastError();
      }
    }
    nc = c;
  }
  if (nc.node < 0) return -1;
  // The node's CPU list is intersected with the mask the PROCESS had when the library was loaded
  // (taskset / numactl / a launcher's per-rank pinning must hold for the I/O threads too), not with
  // the creating thread's current mask: that one may have been narrowed to a single core by an
  // OpenMP runtime (a BLAS call made earlier in the process is enough).  An intersection that is
  // empty or a single core falls back to the bare node list; the kernel applies the container's
  // cpuset itself and refuses only an empty result.
  cpu_set_t want;
  CPU_AND(&want, &nc.set, &g_initial_mask.set);
  if (!g_initial_mask.ok || CPU_COUNT(&want) <= 1) want = nc.set;
  if (sched_setaffinity(0, sizeof(want), &want) == 0) return nc.node;
  return -1;
}

}  // namespace bof

// $BOF_CRASH_TRACE=1: a fatal signal inside the process prints the native stack (module + offset per frame:
// resolve with addr2line -e libbof_hip.so) and the event ring before the previous handler (Python's faulthandler,
// the default action) takes over.  Diagnostic for long fuzz runs; not async-signal-safe by the letter, but the
// process is about to die anyway.
#include <execinfo.h>
#include <signal.h>
namespace {
struct sigaction g_prev_segv, g_prev_bus, g_prev_abrt;
void crash_handler(int sig, siginfo_t *info, void *ctx) {
  static std::atomic<int> once{0};
  if (once.fetch_add(1) == 0) {
    void *bt[64];
    const int n = backtrace(bt, 64);
    const char msg[] = "[bof] fatal signal -- native stack (module+offset):\n";
    (void) !write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    if (auto fn = bof::g_crash_dump_fn.load()) fn(bof::g_crash_dump_arg.load());
    bof::evt_dump(stderr, "fatal signal");
  }
  struct sigaction *prev = sig == SIGSEGV ? &g_prev_segv : sig == SIGBUS ? &g_prev_bus : &g_prev_abrt;
  sigaction(sig, prev, nullptr);
  if ((prev->sa_flags & SA_SIGINFO) && prev->sa_sigaction) prev->sa_sigaction(sig, info, ctx);
  else raise(sig);
}
struct CrashTraceInit {
  CrashTraceInit() {
    const char *e = getenv("BOF_CRASH_TRACE");
    if (!e || !e[0] || !strcmp(e, "0")) return;
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = crash_handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigaction(SIGSEGV, &sa, &g_prev_segv);
    sigaction(SIGBUS, &sa, &g_prev_bus);
  }
} g_crash_trace_init;
}  // namespace

extern "C" uint64_t bof_event_dump(const char *path) {
  if (!path || !path[0]) {
    bof::evt_dump(stderr, "bof_event_dump");
  } else {
    FILE *f = fopen(path, "a");
    if (f) {
      bof::evt_dump(f, "bof_event_dump");
      fclose(f);
    }
  }
  return bof::evt_count();
}
