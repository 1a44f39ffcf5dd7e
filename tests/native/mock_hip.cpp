// mock_hip.cpp -- TEST INFRASTRUCTURE ONLY (tests/native/host_pipeline.cpp, run by tests/test_host_sanitizers.py).
// Never part of libbof_hip.so: the product has no CPU path and refuses to run without a GPU.
//
// A stand-in for the HIP runtime (the 29 entry points the level-2 / level-3 host code calls) plus stand-ins for
// the kernel wrappers of the .hip files, so that the WHOLE host side of the library -- tilers, panel hub, tile
// cache, CSR pipeline, device lists, staging, file engines -- can run on a machine without a GPU under
// AddressSanitizer / UBSan / ThreadSanitizer, and, above all, with SEVERAL DISTINCT mock devices: the GPU boxes of
// the pool have one GPU, so the in-process multi-device code is otherwise only ever run with one ordinal
// listed several times, which cannot show a stream, event or buffer used on the wrong device.
//
// "Device memory" is host memory tagged with its device.  Two modes:
//   * immediate (default): every operation executes at once in the calling thread -- only the HOST-side ordering
//     of the pipelines is exercised;
//   * MOCK_HIP_ASYNC=1: every stream is a queue with a worker thread of its own (optionally jittered,
//     MOCK_HIP_JITTER_US), copies and kernel stand-ins run when the queue reaches them, hipEventRecord /
//     hipStreamWaitEvent / the synchronize calls have their HIP meaning (a wait captures the event's latest record
//     at the time of the call).  The work of two streams is then ordered ONLY by the events the library put
//     between them -- and ThreadSanitizer, which follows exactly those edges (queue hand-over, event completion),
//     reports a buffer touched by two streams without such an edge: a stream-ordering race detector for the
//     pipelines, on a machine without a GPU.  hipFree waits for the allocation's device like the real one.
// What real HIP enforces -- or silently gets wrong -- is checked and is fatal in both modes:
//   R1  an event is recorded on a stream of the event's own device;
//   R2  hipStreamWaitEvent on an event that was never recorded (a no-op in HIP: the wait the caller wanted does
//       not happen -- the class of the shared-operand race of round 3);
//   R3  async copies: the device side belongs to the stream's device (or to a peer it has been given access to),
//       the host side is pinned memory (a pageable host side of a LINEAR copy is legal and only counted), the
//       ranges lie inside their allocations;
//   R4  kernel stand-ins: the calling thread's current device is the stream's device and every pointer is that
//       device's memory -- except the sources of sum_partials, which need peer access;
//   R5  streams / events are not used after they were destroyed; nothing is freed twice;
//   R6  (armed by the harness, mock_hip_mark_caller_thread) a kernel is launched by a thread that called the library
//       or by one of its persistent launcher threads, never by a thread created for the call: on the real runtime
//       the launches of such a thread were seen to carry a stale trailing argument (the wrong C tile of round 3,
//       profiles/r4/fuzz_thread_bisect.md) -- this rule is the regression test of that fix.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "bof_hip.h"
#include "bof_internal.h"

namespace bof { bool on_launcher_thread(); }   // flash_support.cpp

namespace {

[[noreturn]] void violation(const char *rule, const std::string &what) {
  fprintf(stderr, "mock_hip: RULE %s violated: %s\n", rule, what.c_str());
  abort();
}

struct Alloc { size_t bytes; int dev; bool host; };   // dev = -1 for pinned host memory
std::mutex g_mu;
std::map<uintptr_t, Alloc> g_allocs;
std::vector<size_t> g_used;                            // bytes allocated per device
std::set<std::pair<int, int>> g_peer;                  // (from, to)
std::atomic<uint64_t> g_kernel_launches{0};
std::atomic<uint64_t> g_launches_on[64];              // per device
struct Reporter {                                      // MOCK_HIP_REPORT=1: one line at exit, for tests that run a driver binary
  ~Reporter() {
    if (!getenv("MOCK_HIP_REPORT")) return;
    fprintf(stderr, "mock_hip: kernel stand-in launches per device:");
    const int n = getenv("MOCK_HIP_DEVICES") ? std::max(1, atoi(getenv("MOCK_HIP_DEVICES"))) : 4;
    for (int d = 0; d < n && d < 64; d++) fprintf(stderr, " %llu", (unsigned long long) g_launches_on[d].load());
    fprintf(stderr, "\n");
  }
} g_reporter;
std::atomic<int64_t> g_live_streams{0}, g_live_events{0};
std::vector<struct MockStream *> &g_streams = *new std::vector<struct MockStream *>();   // alive ones (under g_mu)
// every stream / event ever made: destroyed ones stay allocated so that a later use is reported; reachable from
// here (a vector that is itself never destroyed), so LeakSanitizer does not count them
std::vector<void *> &g_created = *new std::vector<void *>();
thread_local int t_dev = 0;
// R6 (armed by the first mock_hip_mark_caller_thread()): a kernel is launched by a thread that CALLED the library or by
// one of its persistent launcher threads, never by a thread the library created for the call (flash_common.h,
// "persistent launcher threads": on the real runtime such a thread's launches were seen to carry a stale argument)
thread_local bool t_caller = false;
std::atomic<bool> g_r6{false};
thread_local hipError_t t_last = hipSuccess;

int n_devices() {
  static const int n = getenv("MOCK_HIP_DEVICES") ? std::max(1, atoi(getenv("MOCK_HIP_DEVICES"))) : 4;
  return n;
}
size_t capacity() {   // per mock device
  static const size_t c = getenv("MOCK_HIP_HBM_MIB") ? (size_t) atol(getenv("MOCK_HIP_HBM_MIB")) << 20 : (size_t) 1 << 30;
  return c;
}
hipError_t fail(hipError_t e) { t_last = e; return e; }

bool async_mode() {
  static const bool on = getenv("MOCK_HIP_ASYNC") && atoi(getenv("MOCK_HIP_ASYNC")) != 0;
  return on;
}
long jitter_us() {
  static const long j = getenv("MOCK_HIP_JITTER_US") ? atol(getenv("MOCK_HIP_JITTER_US")) : 0;
  return j;
}
// "everything queued on a stream up to a point has run": what an event record, a synchronize call stand for
struct Completion {
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
  void set() { { std::lock_guard<std::mutex> lk(m); done = true; } cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return done; }); }
};
constexpr uint32_t kStreamMagic = 0x5354524du, kEventMagic = 0x45564e54u;
struct MockStream {
  uint32_t magic = kStreamMagic;
  int dev = 0;
  std::atomic<bool> alive{true};
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  std::thread worker;
  unsigned seed = 1;
  void run() {
    for (;;) {
      std::function<void()> op;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) return;
        op = std::move(q.front());
        q.pop_front();
      }
      if (jitter_us() > 0) usleep((useconds_t) (rand_r(&seed) % (unsigned) (jitter_us() + 1)));
      op();
    }
  }
  // in order behind everything queued so far; immediate mode: now
  void enqueue(std::function<void()> op) {
    if (!async_mode()) { op(); return; }
    {
      std::lock_guard<std::mutex> lk(m);
      if (!stop) {
        q.push_back(std::move(op));
        op = nullptr;
      }
    }
    // (a stream being destroyed by another thread -- sync_device of a neighbour's hipFree may still hold it: its
    //  queue has run dry, so "behind everything queued" is now)
    if (op) op(); else cv.notify_one();
  }
  void drain() {
    if (!async_mode()) return;
    auto c = std::make_shared<Completion>();
    enqueue([c] { c->set(); });
    c->wait();
  }
};
struct MockEvent {
  uint32_t magic = kEventMagic;
  int dev = 0;
  std::atomic<bool> alive{true};
  std::mutex m;
  std::shared_ptr<Completion> last;      // the latest record (null: never recorded)
};

// every stream of `dev` (-1: of every device) has run what was queued on it
void sync_device(int dev) {
  if (!async_mode()) return;
  std::vector<MockStream *> list;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (MockStream *s : g_streams)
      if (dev < 0 || s->dev == dev) list.push_back(s);
  }
  // (a stream cannot be destroyed under this: the library never destroys a stream while another thread synchronises
  //  its device -- the per-device call lock)
  for (MockStream *s : list) s->drain();
}

MockStream *S(hipStream_t s, const char *who) {
  if (!s) violation("R5", std::string(who) + ": the null stream is never used by the library's pipelines");
  MockStream *m = reinterpret_cast<MockStream *>(s);
  if (m->magic != kStreamMagic || !m->alive) violation("R5", std::string(who) + ": stream destroyed or not a stream");
  return m;
}
MockEvent *E(hipEvent_t e, const char *who) {
  MockEvent *m = reinterpret_cast<MockEvent *>(e);
  if (!m || m->magic != kEventMagic || !m->alive) violation("R5", std::string(who) + ": event destroyed or not an event");
  return m;
}

// the allocation [p, p + bytes) lies in; fatal if it lies in none or crosses its end
Alloc where(const void *p, size_t bytes, const char *who) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_allocs.upper_bound((uintptr_t) p);
  if (it == g_allocs.begin()) violation("R3", std::string(who) + ": pointer in no mock allocation (pageable memory?)");
  --it;
  if ((uintptr_t) p + bytes > it->first + it->second.bytes)
    violation("R3", std::string(who) + ": range runs past the end of its allocation (or pointer in none)");
  return it->second;
}
bool known(const void *p) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_allocs.upper_bound((uintptr_t) p);
  if (it == g_allocs.begin()) return false;
  --it;
  return (uintptr_t) p < it->first + it->second.bytes;
}
void need_device_mem(const void *p, size_t bytes, int dev, const char *who, bool peer_ok = false) {
  if (bytes == 0) return;
  const Alloc a = where(p, bytes, who);
  if (a.host) violation("R4", std::string(who) + ": host memory where device memory is expected");
  if (a.dev == dev) return;
  bool ok = false;
  if (peer_ok) {
    std::lock_guard<std::mutex> lk(g_mu);
    ok = g_peer.count({dev, a.dev}) != 0;
  }
  if (!ok)
    violation(peer_ok ? "R4 (peer access)" : "R4", std::string(who) + ": memory of device " + std::to_string(a.dev) +
                                                     " used on device " + std::to_string(dev));
}
// a kernel launch: the calling thread's current device must be the stream's
MockStream *kernel_stream(hipStream_t st, const char *who) {
  MockStream *s = S(st, who);
  if (s->dev != t_dev)
    violation("R4", std::string(who) + ": launched with current device " + std::to_string(t_dev) + " on a stream of device " +
                        std::to_string(s->dev));
  if (g_r6.load() && !t_caller && !bof::on_launcher_thread())
    violation("R6", std::string(who) + ": launched by a thread created for the call (neither a caller of the library nor a "
                                       "persistent launcher thread)");
  g_kernel_launches++;
  g_launches_on[s->dev & 63]++;
  return s;
}

}  // namespace

// fault injection: the (n + 1)-th hipMalloc / hipHostMalloc from now on fails once with hipErrorOutOfMemory (-1: off)
std::atomic<long> g_fail_malloc_in{-1}, g_fail_hostmalloc_in{-1};
extern "C" void mock_hip_fail_malloc_after(long n) { g_fail_malloc_in.store(n); }
extern "C" void mock_hip_fail_hostmalloc_after(long n) { g_fail_hostmalloc_in.store(n); }
extern "C" long mock_hip_fail_malloc_pending() { return g_fail_malloc_in.load(); }
static bool inject(std::atomic<long> &ctr) {
  long v = ctr.load();
  while (v >= 0) {
    if (ctr.compare_exchange_weak(v, v - 1)) return v == 0;
  }
  return false;
}
// ... and the (n + 1)-th call of one API kind (1 hipMemcpyAsync / 2DAsync, 2 hipEventRecord, 3 hipStreamWaitEvent,
// 4 hipEventCreateWithFlags, 5 hipStreamCreate*, 6 hipMemsetAsync) fails once with hipErrorUnknown
std::atomic<long> g_fail_api_in{-1};
std::atomic<int> g_fail_api_kind{0};
extern "C" void mock_hip_fail_api_after(int kind, long n) { g_fail_api_kind.store(kind); g_fail_api_in.store(n); }
extern "C" long mock_hip_fail_api_pending() { return g_fail_api_in.load(); }
static bool inject_api(int kind) { return g_fail_api_kind.load() == kind && inject(g_fail_api_in); }
extern "C" uint64_t mock_hip_kernel_launches() { return g_kernel_launches.load(); }
extern "C" void mock_hip_mark_caller_thread() {
  t_caller = true;
  g_r6.store(true);
}
extern "C" uint64_t mock_hip_pageable_h2d_bytes();
extern "C" int64_t mock_hip_live_streams() { return g_live_streams.load(); }
extern "C" int64_t mock_hip_live_events() { return g_live_events.load(); }
extern "C" size_t mock_hip_bytes_in_use(int dev) {
  std::lock_guard<std::mutex> lk(g_mu);
  return dev < (int) g_used.size() ? g_used[(size_t) dev] : 0;
}

// ---- the runtime ---------------------------------------------------------------------------------------------
extern "C" {

hipError_t hipGetDeviceCount(int *count) { *count = n_devices(); return hipSuccess; }
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= n_devices()) return fail(hipErrorInvalidDevice);
  t_dev = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int *d) { *d = t_dev; return hipSuccess; }
hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorInvalidValue: return "invalid argument";
    default: return "mock HIP error";
  }
}
hipError_t hipDeviceSynchronize(void) { sync_device(t_dev); return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *id, int len, int device) {
  snprintf(id, (size_t) len, "0000:%02x:00.0", 0x10 + device);
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer) {
  if (dev < 0 || dev >= n_devices() || peer < 0 || peer >= n_devices()) return fail(hipErrorInvalidDevice);
  *can = dev != peer;
  return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned int) {
  if (peer < 0 || peer >= n_devices() || peer == t_dev) return fail(hipErrorInvalidDevice);
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_peer.insert({t_dev, peer}).second) { t_last = hipErrorPeerAccessAlreadyEnabled; return hipErrorPeerAccessAlreadyEnabled; }
  return hipSuccess;
}

hipError_t hipMalloc(void **p, size_t bytes) {
  if (inject(g_fail_malloc_in)) { *p = nullptr; t_last = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_used.empty()) g_used.assign((size_t) n_devices(), 0);
  if (g_used[(size_t) t_dev] + bytes > capacity()) { *p = nullptr; t_last = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
  void *q = nullptr;
  if (posix_memalign(&q, 4096, std::max<size_t>(bytes, 1))) { t_last = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
  memset(q, 0xA5, bytes);      // fresh HBM is not zero
  g_allocs[(uintptr_t) q] = Alloc{bytes, t_dev, false};
  g_used[(size_t) t_dev] += bytes;
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  {
    int dev = -1;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      auto it = g_allocs.find((uintptr_t) p);
      if (it != g_allocs.end()) dev = it->second.dev;
    }
    if (dev >= 0) sync_device(dev);      // hipFree waits for the device the memory belongs to
  }
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_allocs.find((uintptr_t) p);
  if (it == g_allocs.end() || it->second.host) violation("R5", "hipFree of something hipMalloc did not return (or freed twice)");
  g_used[(size_t) it->second.dev] -= it->second.bytes;
  g_allocs.erase(it);
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int) {
  // (twice in a row: pinned_alloc retries once after emptying its cache)
  static std::atomic<bool> again{false};
  if (inject(g_fail_hostmalloc_in)) { again = true; *p = nullptr; t_last = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
  if (again.exchange(false)) { *p = nullptr; t_last = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
  void *q = nullptr;
  if (posix_memalign(&q, 4096, std::max<size_t>(bytes, 1))) { t_last = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
  std::lock_guard<std::mutex> lk(g_mu);
  g_allocs[(uintptr_t) q] = Alloc{bytes, -1, true};
  *p = q;
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  if (!p) return hipSuccess;
  sync_device(-1);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_allocs.find((uintptr_t) p);
  if (it == g_allocs.end() || !it->second.host) violation("R5", "hipHostFree of something hipHostMalloc did not return (or freed twice)");
  g_allocs.erase(it);
  free(p);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_used.empty()) g_used.assign((size_t) n_devices(), 0);
  *total_b = capacity();
  *free_b = capacity() - std::min(capacity(), g_used[(size_t) t_dev]);
  return hipSuccess;
}

static hipError_t new_stream(hipStream_t *s) {
  if (inject_api(5)) { *s = nullptr; return fail(hipErrorUnknown); }
  MockStream *m = new MockStream();
  m->dev = t_dev;
  m->seed = (unsigned) (uintptr_t) m;
  if (async_mode()) m->worker = std::thread([m] { m->run(); });
  g_live_streams++;
  { std::lock_guard<std::mutex> lk(g_mu); g_created.push_back(m); g_streams.push_back(m); }
  *s = reinterpret_cast<hipStream_t>(m);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { return new_stream(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned int, int) { return new_stream(s); }
hipError_t hipStreamDestroy(hipStream_t s) {
  MockStream *m = S(s, "hipStreamDestroy");
  { std::lock_guard<std::mutex> lk(g_mu); g_streams.erase(std::find(g_streams.begin(), g_streams.end(), m)); }
  if (async_mode()) {       // what is queued still runs (HIP destroys the stream once it is idle)
    { std::lock_guard<std::mutex> lk(m->m); m->stop = true; }
    m->cv.notify_all();
    m->worker.join();
  }
  m->alive = false;         // kept allocated: a later use is reported, not a crash
  g_live_streams--;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { S(s, "hipStreamSynchronize")->drain(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
  if (inject_api(4)) { *e = nullptr; return fail(hipErrorUnknown); }
  MockEvent *m = new MockEvent();
  m->dev = t_dev;
  g_live_events++;
  { std::lock_guard<std::mutex> lk(g_mu); g_created.push_back(m); }
  *e = reinterpret_cast<hipEvent_t>(m);
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) { E(e, "hipEventDestroy")->alive = false; g_live_events--; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (inject_api(2)) return fail(hipErrorUnknown);
  MockEvent *ev = E(e, "hipEventRecord");
  MockStream *st = S(s, "hipEventRecord");
  if (ev->dev != st->dev)
    violation("R1", "event of device " + std::to_string(ev->dev) + " recorded on a stream of device " + std::to_string(st->dev));
  auto c = std::make_shared<Completion>();
  { std::lock_guard<std::mutex> lk(ev->m); ev->last = c; }
  st->enqueue([c] { c->set(); });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  MockEvent *ev = E(e, "hipEventSynchronize");
  std::shared_ptr<Completion> c;
  { std::lock_guard<std::mutex> lk(ev->m); c = ev->last; }
  if (c) c->wait();
  return hipSuccess;
}
// KernelTimer (bof_options.kernel_timing): both events must have been recorded and be complete
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  for (hipEvent_t e : {a, b}) {
    MockEvent *ev = E(e, "hipEventElapsedTime");
    std::shared_ptr<Completion> c;
    { std::lock_guard<std::mutex> lk(ev->m); c = ev->last; }
    if (!c) violation("R2", "hipEventElapsedTime on an event that was never recorded");
    else c->wait();
  }
  *ms = 0.001f;
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int) {
  if (inject_api(3)) return fail(hipErrorUnknown);
  MockStream *st = S(s, "hipStreamWaitEvent");
  MockEvent *ev = E(e, "hipStreamWaitEvent");
  std::shared_ptr<Completion> c;
  { std::lock_guard<std::mutex> lk(ev->m); c = ev->last; }     // the record the event holds NOW
  if (!c)
    violation("R2", "hipStreamWaitEvent on an event that has never been recorded (HIP treats it as complete: no wait happens)");
  st->enqueue([c] { c->wait(); });
  return hipSuccess;
}

std::atomic<uint64_t> g_pageable_h2d{0}, g_pageable_d2h{0};
static void check_copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, int dev, const char *who) {
  if (bytes == 0) return;
  if (kind == hipMemcpyHostToDevice && !known(src)) {
    // pageable source of an async H2D copy: legal (HIP stages it before returning, so it is not asynchronous);
    // counted and reported, the device side still checked
    g_pageable_h2d += bytes;
    need_device_mem(dst, bytes, dev, who);
    return;
  }
  if (kind == hipMemcpyDeviceToHost && !known(dst)) {
    // pageable destination of an async D2H copy: legal too (complete once the stream has been synchronised, which
    // is what ThreadSanitizer then checks the reader against); counted, the device side still checked
    g_pageable_d2h += bytes;
    need_device_mem(src, bytes, dev, who);
    return;
  }
  const Alloc d = where(dst, bytes, who), s = where(src, bytes, who);
  auto dev_ok = [&](const Alloc &a) {
    if (a.host) violation("R3", std::string(who) + ": host memory on the device side of the copy");
    if (a.dev != dev) {
      std::lock_guard<std::mutex> lk(g_mu);
      if (!g_peer.count({dev, a.dev}))
        violation("R3", std::string(who) + ": memory of device " + std::to_string(a.dev) + " copied on a stream of device " +
                            std::to_string(dev) + " without peer access");
    }
  };
  auto host_ok = [&](const Alloc &a) {
    if (!a.host) violation("R3", std::string(who) + ": device memory on the host side of the copy");
  };
  if (kind == hipMemcpyHostToDevice) { dev_ok(d); host_ok(s); }
  else if (kind == hipMemcpyDeviceToHost) { host_ok(d); dev_ok(s); }
  else if (kind == hipMemcpyDeviceToDevice) { dev_ok(d); dev_ok(s); }
  else violation("R3", std::string(who) + ": copy kind the library never uses");
}
uint64_t mock_hip_pageable_h2d_bytes() { return g_pageable_h2d.load() + g_pageable_d2h.load(); }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
  if (inject_api(1)) return fail(hipErrorUnknown);
  MockStream *st = S(s, "hipMemcpyAsync");
  check_copy(dst, src, bytes, kind, st->dev, "hipMemcpyAsync");
  if (kind == hipMemcpyHostToDevice && !known(src)) {     // pageable source: staged before the call returns
    auto staged = std::make_shared<std::vector<char>>((const char *) src, (const char *) src + bytes);
    st->enqueue([dst, staged] { memcpy(dst, staged->data(), staged->size()); });
    return hipSuccess;
  }
  st->enqueue([dst, src, bytes] { memmove(dst, src, bytes); });
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  // the synchronous form may take pageable host memory; the device side must be the current device's
  if (kind == hipMemcpyHostToDevice) need_device_mem(dst, bytes, t_dev, "hipMemcpy");
  else if (kind == hipMemcpyDeviceToHost) need_device_mem(src, bytes, t_dev, "hipMemcpy");
  else violation("R3", "hipMemcpy: copy kind the library never uses");
  memmove(dst, src, bytes);      // synchronous, on no stream of the library's
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind kind, hipStream_t s) {
  if (inject_api(1)) return fail(hipErrorUnknown);
  if (width == 0 || height == 0) return hipSuccess;
  if (dpitch < width || spitch < width) return fail(hipErrorInvalidValue);
  MockStream *st = S(s, "hipMemcpy2DAsync");
  const int dev = st->dev;
  // extents: (height - 1) pitches + one width on either side
  const Alloc d = where(dst, (height - 1) * dpitch + width, "hipMemcpy2DAsync"), sa = where(src, (height - 1) * spitch + width, "hipMemcpy2DAsync");
  const Alloc &dv = kind == hipMemcpyHostToDevice ? d : sa, &hv = kind == hipMemcpyHostToDevice ? sa : d;
  if (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost) violation("R3", "hipMemcpy2DAsync: copy kind the library never uses");
  if (dv.host || dv.dev != dev) violation("R3", "hipMemcpy2DAsync: device side is not memory of the stream's device");
  if (!hv.host) violation("R3", "hipMemcpy2DAsync: host side is not pinned host memory");
  st->enqueue([=] {
    for (size_t r = 0; r < height; r++) memcpy((char *) dst + r * dpitch, (const char *) src + r * spitch, width);
  });
  return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int ddev, const void *src, int sdev, size_t bytes, hipStream_t s) {
  MockStream *st = S(s, "hipMemcpyPeerAsync");
  need_device_mem(dst, bytes, ddev, "hipMemcpyPeerAsync");
  need_device_mem(src, bytes, sdev, "hipMemcpyPeerAsync");
  st->enqueue([dst, src, bytes] { memmove(dst, src, bytes); });
  return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t bytes) {      // synchronous: BOF_VERIFY's table, before anything is queued
  need_device_mem(dst, bytes, t_dev, "hipMemset");
  memset(dst, v, bytes);
  return hipSuccess;
}
hipError_t hipMemset2DAsync(void *dst, size_t pitch, int v, size_t width, size_t height, hipStream_t s) {
  if (inject_api(6)) return fail(hipErrorUnknown);
  MockStream *st = S(s, "hipMemset2DAsync");
  if (!width || !height) return hipSuccess;
  need_device_mem(dst, (height - 1) * pitch + width, st->dev, "hipMemset2DAsync");
  st->enqueue([dst, pitch, v, width, height] {
    for (size_t r = 0; r < height; r++) memset((char *) dst + r * pitch, v, width);
  });
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t s) {
  if (inject_api(6)) return fail(hipErrorUnknown);
  MockStream *st = S(s, "hipMemsetAsync");
  need_device_mem(dst, bytes, st->dev, "hipMemsetAsync");
  st->enqueue([dst, v, bytes] { memset(dst, v, bytes); });
  return hipSuccess;
}

}  // extern "C"

// ---- the kernel wrappers of the .hip files ---------------------------------------------------------------------
// Plain loops with the product kernels' contract (k-ordered fmaf chain per element, alpha * acc + beta * c with
// beta == 0 not reading c), queued on their stream like everything else: pointers are checked and data is read when
// the stream gets there.  They exist to move data through the pipelines; arithmetic parity is the GPU suite's job.
namespace bof {

static hipError_t gemm_any(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, const float *a, int64_t lda,
                           const float *b, int64_t ldb, float beta, float *c, int64_t ldc, const float *u1, const float *v1,
                           const float *u2, const float *v2, hipStream_t st, const char *who, GemmChain ch = GemmChain()) {
  MockStream *ms = kernel_stream(st, who);
  if (m <= 0 || n <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    // row-major view: for 'C' the stored matrices are the transposes
    const bool a_mk = (ta == 'N') == (ord == 'R');     // A stored with m as the outer (row) index
    const bool b_kn = (tb == 'N') == (ord == 'R');     // B stored with k as the outer index
    const int64_t a_rows = a_mk ? m : k, a_cols = a_mk ? k : m, b_rows = b_kn ? k : n, b_cols = b_kn ? n : k;
    if (k > 0) {
      need_device_mem(a, (size_t) ((a_rows - 1) * lda + a_cols) * 4, dev, who);
      need_device_mem(b, (size_t) ((b_rows - 1) * ldb + b_cols) * 4, dev, who);
    }
    const int64_t c_rows = ord == 'R' ? m : n, c_cols = ord == 'R' ? n : m;
    need_device_mem(c, (size_t) ((c_rows - 1) * ldc + c_cols) * 4, dev, who);
    if (u1) {
      need_device_mem(u1, (size_t) m * 4, dev, who); need_device_mem(v1, (size_t) n * 4, dev, who);
      need_device_mem(u2, (size_t) m * 4, dev, who); need_device_mem(v2, (size_t) n * 4, dev, who);
    }
    if (ch.acc_in) need_device_mem(ch.acc_in, (size_t) ((c_rows - 1) * ch.ld_acc + c_cols) * 4, dev, who);
    if (ch.c_in) need_device_mem(ch.c_in, (size_t) ((c_rows - 1) * ch.ld_cin + c_cols) * 4, dev, who);
    for (int64_t i = 0; i < m; i++)
      for (int64_t j = 0; j < n; j++) {
        // (an accumulate chain: the raw sums of the k-ranges before this one come in, and go out unscaled)
        float acc = ch.acc_in ? (ord == 'R' ? ch.acc_in[i * ch.ld_acc + j] : ch.acc_in[j * ch.ld_acc + i]) : 0.f;
        for (int64_t l = 0; l < k; l++)
          acc = fmaf(a_mk ? a[i * lda + l] : a[l * lda + i], b_kn ? b[l * ldb + j] : b[j * ldb + l], acc);
        float *cp = ord == 'R' ? c + i * ldc + j : c + j * ldc + i;
        if (ch.raw_out) { *cp = acc; continue; }
        const float cin = ch.c_in ? (ord == 'R' ? ch.c_in[i * ch.ld_cin + j] : ch.c_in[j * ch.ld_cin + i]) : *cp;
        float r = beta == 0.f ? alpha * acc : fmaf(alpha, acc, beta * cin);
        if (u1) r = fmaf(u2[i], v2[j], fmaf(u1[i], v1[j], r));
        *cp = r;
      }
  });
  return hipSuccess;
}
hipError_t sgemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, const float *a, int64_t lda,
                 const float *b, int64_t ldb, float beta, float *c, int64_t ldc, hipStream_t st) {
  return gemm_any(ord, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, nullptr, nullptr, nullptr, nullptr, st, "sgemm");
}
hipError_t sgemm_chain(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, const float *a, int64_t lda,
                       const float *b, int64_t ldb, float beta, float *c, int64_t ldc, const GemmChain &ch, hipStream_t st) {
  GemmChain c2 = ch;
  if (alpha == 0.f && !ch.raw_out) { c2.acc_in = nullptr; k = 0; }      // the quick return belongs to the final launch
  return gemm_any(ord, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, nullptr, nullptr, nullptr, nullptr, st, "sgemm_chain", c2);
}
hipError_t sgemm_rank1x2(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, const float *a, int64_t lda,
                         const float *b, int64_t ldb, float beta, float *c, int64_t ldc, const float *u1, const float *v1,
                         const float *u2, const float *v2, hipStream_t st) {
  return gemm_any(ord, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, u1, v1, u2, v2, st, "sgemm_rank1x2");
}
// BOF_VERIFY's spot check of a launch (gemm_f32_mfma.hip: spot_capture_kernel / spot_check_kernel), same definition:
// 64 sampled outputs recomputed in the kernels' arithmetic, word sums of wanted and stored values
namespace {
struct SpotPos { int64_t r, c; };
SpotPos spot_position(uint64_t seed, int t, int64_t M, int64_t N) {
  uint64_t h = seed * 0x9E3779B97F4A7C15ull + (uint64_t) (t + 1) * 0xD1B54A32D192ED03ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  SpotPos p{(int64_t) ((h & 0xFFFFFFFFu) % (uint32_t) M), (int64_t) ((h >> 32) % (uint32_t) N)};
  if (t == 0) p = {0, 0};
  if (t == 1) p = {M - 1, N - 1};
  if (t == 2) p = {0, N - 1};
  if (t == 3) p = {M - 1, 0};
  return p;
}
// element (i along m, j along n) of a matrix stored like C
inline int64_t c_at(char ord, int64_t i, int64_t j, int64_t ld) { return ord == 'R' ? i * ld + j : j * ld + i; }
}  // namespace
hipError_t sgemm_spot_capture(const SpotArgs &s, float *save, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "sgemm_spot_capture");
  if (s.m <= 0 || s.n <= 0) return hipSuccess;
  const SpotArgs a = s;
  ms->enqueue([a, save] {
    // (positions are drawn in the row-major core's terms: rows of the stored C)
    const int64_t M = a.ord == 'C' ? a.n : a.m, N = a.ord == 'C' ? a.m : a.n;
    for (int t = 0; t < 64; t++) {
      const SpotPos p = spot_position(a.seed, t, M, N);
      save[t] = a.ch.acc_in ? a.ch.acc_in[p.r * a.ch.ld_acc + p.c] : 0.f;
      const float *ci = a.ch.c_in ? a.ch.c_in + p.r * a.ch.ld_cin + p.c : a.c + p.r * a.ldc + p.c;
      save[64 + t] = (a.beta != 0.f && !a.ch.raw_out) ? *ci : 0.f;
    }
  });
  return hipSuccess;
}
hipError_t sgemm_spot_check(const SpotArgs &s, const float *save, unsigned long long *exp2, unsigned long long *got2, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "sgemm_spot_check");
  if (s.m <= 0 || s.n <= 0) return hipSuccess;
  const SpotArgs a = s;
  ms->enqueue([a, save, exp2, got2] {
    const bool cm = a.ord == 'C';
    const int64_t M = cm ? a.n : a.m, N = cm ? a.m : a.n;
    const bool a_mk = (a.ta == 'N') == (a.ord == 'R'), b_kn = (a.tb == 'N') == (a.ord == 'R');
    for (int t = 0; t < 64; t++) {
      const SpotPos p = spot_position(a.seed, t, M, N);
      const int64_t i = cm ? p.c : p.r, j = cm ? p.r : p.c;       // along m, along n
      float acc = a.ch.acc_in ? save[t] : 0.f;
      if (!a.ch.raw_out && a.alpha == 0.f) acc = 0.f;
      else
        for (int64_t l = 0; l < a.k; l++)
          acc = fmaf(a_mk ? a.a[i * a.lda + l] : a.a[l * a.lda + i], b_kn ? a.b[l * a.ldb + j] : a.b[j * a.ldb + l], acc);
      float want = a.ch.raw_out ? acc : (a.beta == 0.f ? a.alpha * acc : fmaf(a.alpha, acc, a.beta * save[64 + t]));
      if (a.u1) want = fmaf(a.u2[i], a.v2[j], fmaf(a.u1[i], a.v1[j], want));      // (as this file's gemm_any adds them)
      const float got = a.c[p.r * a.ldc + p.c];
      uint32_t we, wg;
      memcpy(&we, &want, 4);
      memcpy(&wg, &got, 4);
      exp2[0] += we; exp2[1] += (unsigned long long) we * (unsigned long long) (t + 1);
      got2[0] += wg; got2[1] += (unsigned long long) wg * (unsigned long long) (t + 1);
    }
  });
  return hipSuccess;
}
hipError_t expand_tile_local(const float *src, float *dst, int64_t len, int64_t blk, int64_t nblk, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "expand_tile_local");
  const int dev = ms->dev;
  ms->enqueue([=] {
    need_device_mem(dst, (size_t) len * 4, dev, "expand_tile_local");
    for (int64_t i = 0; i < len; i++) dst[i] = src[i - std::min(i / blk, nblk - 1) * blk];
  });
  return hipSuccess;
}
hipError_t transpose_f32(const float *in, int64_t ld_in, int64_t rows, int64_t cols, float *out, int64_t ld_out, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "transpose_f32");
  if (rows <= 0 || cols <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    need_device_mem(in, (size_t) ((rows - 1) * ld_in + cols) * 4, dev, "transpose_f32");
    need_device_mem(out, (size_t) ((cols - 1) * ld_out + rows) * 4, dev, "transpose_f32");
    std::vector<float> tmp((size_t) rows * (size_t) cols);       // in and out may be the same buffer
    for (int64_t r = 0; r < rows; r++)
      for (int64_t c = 0; c < cols; c++) tmp[(size_t) (c * rows + r)] = in[r * ld_in + c];
    for (int64_t c = 0; c < cols; c++)
      for (int64_t r = 0; r < rows; r++) out[c * ld_out + r] = tmp[(size_t) (c * rows + r)];
  });
  return hipSuccess;
}
// mkl_scsrmm's naming: A is m x k, B k x n, C m x n
// launch receipts (bof_internal.h): the mock "workgroup" is 4 rows, as in the product's row-major kernels
int64_t scsrmm_receipt_entries(char ord_b, int64_t m) { return ord_b == 'R' ? (m + 3) / 4 : 0; }
hipError_t csr_receipt_check(unsigned *seen, int64_t n, unsigned *flag, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "csr_receipt_check");
  if (!seen || n <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    need_device_mem(seen, (size_t) n * 4, dev, "csr_receipt_check");
    need_device_mem(flag, 4, dev, "csr_receipt_check");
    for (int64_t i = 0; i < n; i++) {
      if (seen[i] != 1u) __atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED);
      seen[i] = 0u;
    }
  });
  return hipSuccess;
}
hipError_t scsrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha, const float *val, const int64_t *col,
                  const int64_t *ptr, const float *b, int64_t ldb, float beta, float *c, int64_t ldc, hipStream_t st,
                  unsigned *seen) {
  MockStream *ms = kernel_stream(st, "scsrmm");
  if (m <= 0 || n <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    if (seen && ord_b == 'R') {
      need_device_mem(seen, (size_t) ((m + 3) / 4) * 4, dev, "scsrmm receipts");
      for (int64_t i = 0; i < (m + 3) / 4; i++) seen[i]++;
    }
    need_device_mem(ptr, (size_t) (m + 1) * 8, dev, "scsrmm");
    const int64_t base = ptr[0], nnz = ptr[m] - base;
    need_device_mem(val, (size_t) nnz * 4, dev, "scsrmm");
    need_device_mem(col, (size_t) nnz * 8, dev, "scsrmm");
    if (k > 0) need_device_mem(b, (size_t) (ord_b == 'R' ? (k - 1) * ldb + n : (n - 1) * ldb + k) * 4, dev, "scsrmm");
    need_device_mem(c, (size_t) (ord_b == 'R' ? (m - 1) * ldc + n : (n - 1) * ldc + m) * 4, dev, "scsrmm");
    for (int64_t i = 0; i < m; i++)
      for (int64_t j = 0; j < n; j++) {
        float acc = 0.f;
        for (int64_t p = ptr[i] - base; p < ptr[i + 1] - base; p++)
          acc = fmaf(val[p], ord_b == 'R' ? b[col[p] * ldb + j] : b[j * ldb + col[p]], acc);
        float *cp = ord_b == 'R' ? c + i * ldc + j : c + j * ldc + i;
        *cp = beta == 0.f ? alpha * acc : fmaf(alpha, acc, beta * *cp);
      }
  });
  return hipSuccess;
}
int64_t scsrgemv_receipt_entries(int64_t m) { return (m + 255) / 256; }
hipError_t scsrgemv(char trans, int64_t m, int64_t n, const float *val, const int64_t *ptr, const int64_t *col, const float *x,
                    float *y, hipStream_t st, unsigned *seen) {
  MockStream *ms = kernel_stream(st, "scsrgemv");
  if (m <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    if (seen) {
      need_device_mem(seen, (size_t) ((m + 255) / 256) * 4, dev, "scsrgemv receipts");
      for (int64_t i = 0; i < (m + 255) / 256; i++) seen[i]++;
    }
    need_device_mem(ptr, (size_t) (m + 1) * 8, dev, "scsrgemv");
    const int64_t base = ptr[0], nnz = ptr[m] - base;
    need_device_mem(val, (size_t) nnz * 4, dev, "scsrgemv");
    need_device_mem(col, (size_t) nnz * 8, dev, "scsrgemv");
    need_device_mem(x, (size_t) (trans == 'N' ? n : m) * 4, dev, "scsrgemv");
    need_device_mem(y, (size_t) (trans == 'N' ? m : n) * 4, dev, "scsrgemv");
    // 'T': the product kernel adds with fp32 atomics, so launches on different streams may share y; here
    // the adds go through relaxed atomics for the same reason
    for (int64_t i = 0; i < m; i++) {
      if (trans == 'N') {
        float acc = 0.f;
        for (int64_t p = ptr[i] - base; p < ptr[i + 1] - base; p++) acc = fmaf(val[p], x[col[p]], acc);
        y[i] = acc;
      } else {
        for (int64_t p = ptr[i] - base; p < ptr[i + 1] - base; p++) {
          uint32_t *slot = reinterpret_cast<uint32_t *>(y + col[p]);
          uint32_t old = __atomic_load_n(slot, __ATOMIC_RELAXED), want;
          do {
            float f;
            memcpy(&f, &old, 4);
            f += val[p] * x[i];
            memcpy(&want, &f, 4);
          } while (!__atomic_compare_exchange_n(slot, &old, want, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
        }
      }
    }
  });
  return hipSuccess;
}
hipError_t sum_partials(float *dst, const float *const *srcs, int n_src, int64_t len, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "sum_partials");
  if (len <= 0 || n_src <= 0) return hipSuccess;
  const int dev = ms->dev;
  const std::vector<const float *> src(srcs, srcs + n_src);      // the caller's array need not outlive the call
  ms->enqueue([=] {
    need_device_mem(dst, (size_t) len * 4, dev, "sum_partials");
    for (const float *p : src) need_device_mem(p, (size_t) len * 4, dev, "sum_partials", /*peer_ok=*/true);
    for (int64_t i = 0; i < len; i++) {
      float acc = src[0][i];
      for (size_t q = 1; q < src.size(); q++) acc += src[q][i];
      dst[i] = acc;
    }
  });
  return hipSuccess;
}
// CSR transposition: stable (source rows ascending inside every output row), offsets of the result 0-based
size_t csrcsc_workspace_bytes(int64_t, int64_t) { return 4096; }
size_t csrgemv_t_workspace_bytes(int64_t, int64_t) { return 4096; }
hipError_t scsrcsc(int64_t m, int64_t n, int64_t nnz, const float *val, const int64_t *ptr, const int64_t *col, float *val_tr,
                   int64_t *ptr_tr, int64_t *col_tr, void *, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "scsrcsc");
  const int dev = ms->dev;
  ms->enqueue([=] {
    need_device_mem(ptr_tr, (size_t) (n + 1) * 8, dev, "scsrcsc");
    if (m > 0) need_device_mem(ptr, (size_t) (m + 1) * 8, dev, "scsrcsc");
    if (nnz > 0) {
      need_device_mem(val, (size_t) nnz * 4, dev, "scsrcsc"); need_device_mem(col, (size_t) nnz * 8, dev, "scsrcsc");
      need_device_mem(val_tr, (size_t) nnz * 4, dev, "scsrcsc"); need_device_mem(col_tr, (size_t) nnz * 8, dev, "scsrcsc");
    }
    const int64_t base = m > 0 ? ptr[0] : 0;
    std::vector<int64_t> fill((size_t) n + 1, 0);
    for (int64_t p = 0; p < nnz; p++) fill[(size_t) col[p] + 1]++;
    for (int64_t j = 0; j < n; j++) fill[(size_t) j + 1] += fill[(size_t) j];
    for (int64_t j = 0; j <= n; j++) ptr_tr[j] = fill[(size_t) j];
    for (int64_t i = 0; i < m; i++)
      for (int64_t p = ptr[i] - base; p < ptr[i + 1] - base; p++) {
        const int64_t q = fill[(size_t) col[p]]++;
        val_tr[q] = val[p];
        col_tr[q] = i;
      }
  });
  return hipSuccess;
}
// the out-of-core transposition's merge of the per-block results (csrcsc_kernels.hip: csc_merge_kernel)
hipError_t csc_merge(int nb, int64_t cw, const int64_t *blk_ptr, const int64_t *seg_base, const int64_t *row0, const int64_t *out_ptr,
                     const float *val_in, const int64_t *col_in, float *val_out, int64_t *col_out, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "csc_merge");
  if (cw <= 0 || nb <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    need_device_mem(blk_ptr, (size_t) nb * (size_t) (cw + 1) * 8, dev, "csc_merge");
    need_device_mem(seg_base, (size_t) nb * 8, dev, "csc_merge");
    need_device_mem(row0, (size_t) nb * 8, dev, "csc_merge");
    need_device_mem(out_ptr, (size_t) cw * 8, dev, "csc_merge");
    for (int64_t c = 0; c < cw; c++) {
      int64_t cursor = out_ptr[c];
      for (int b = 0; b < nb; b++) {
        const int64_t s0 = blk_ptr[(int64_t) b * (cw + 1) + c], e0 = blk_ptr[(int64_t) b * (cw + 1) + c + 1];
        for (int64_t i = s0; i < e0; i++) {
          val_out[cursor + (i - s0)] = val_in[seg_base[b] + i];
          col_out[cursor + (i - s0)] = col_in[seg_base[b] + i] + row0[b];
        }
        cursor += e0 - s0;
      }
    }
  });
  return hipSuccess;
}
hipError_t scsrgemv_t_partitioned(int64_t m, int64_t n, int64_t, const float *val, const int64_t *ptr, const int64_t *col,
                                  const float *x, float *y, void *, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "scsrgemv_t_partitioned");
  if (n <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    need_device_mem(y, (size_t) n * 4, dev, "scsrgemv_t_partitioned");
    for (int64_t j = 0; j < n; j++) y[j] = 0.f;
    const int64_t base = m > 0 ? ptr[0] : 0;
    for (int64_t i = 0; i < m; i++)
      for (int64_t p = ptr[i] - base; p < ptr[i + 1] - base; p++) y[col[p]] += val[p] * x[i];
  });
  return hipSuccess;
}
hipError_t gen_dense(float *, int64_t, int64_t, char, uint64_t, hipStream_t) { return hipErrorUnknown; }
// BOF_VERIFY's device-side sums (gen_kernels.hip: verify_sum_kernel), same definition
hipError_t verify_sum(const void *p, int64_t rows, int64_t row_words, int64_t pitch_words, uint64_t index_base, int64_t t_pitch,
                      unsigned long long *out2, hipStream_t st) {
  MockStream *ms = kernel_stream(st, "verify_sum");
  if (rows <= 0 || row_words <= 0) return hipSuccess;
  const int dev = ms->dev;
  ms->enqueue([=] {
    need_device_mem(p, (size_t) ((rows - 1) * pitch_words + row_words) * 4, dev, "verify_sum");
    need_device_mem(out2, 16, dev, "verify_sum");
    const uint32_t *w = (const uint32_t *) p;
    unsigned long long s1 = 0, s2 = 0;
    for (int64_t r = 0; r < rows; r++)
      for (int64_t c = 0; c < row_words; c++) {
        const unsigned long long v = w[r * pitch_words + c];
        const unsigned long long li = index_base + (unsigned long long) (t_pitch > 0 ? c * t_pitch + r : r * row_words + c);
        s1 += v;
        s2 += v * (li + 1ull);
      }
    __atomic_fetch_add(&out2[0], s1, __ATOMIC_RELAXED);
    __atomic_fetch_add(&out2[1], s2, __ATOMIC_RELAXED);
  });
  return hipSuccess;
}
hipError_t gen_sparse_rows(int64_t, int64_t, int64_t, int64_t, float *, int64_t *, int64_t *, hipStream_t) { return hipErrorUnknown; }

}  // namespace bof
