// flash_common.h -- pieces shared by the level-3 pipelines (flash_runtime.cpp: tile cache; flash_csr.cpp: CSR
// row-block ring, transposition; flash_gemm_panels.cpp: panel pipeline of flash::gemm).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bof_hip.h"
#include "bof_internal.h"
#include "fileio.h"

namespace bof {

struct Counters {
  std::atomic<uint64_t> rd{0}, wr{0}, h2d{0}, d2h{0}, tasks{0}, hits{0}, misses{0}, peer{0};
  std::atomic<uint64_t> klaunch{0}, kns{0};   // KernelTimer: timed tile tasks and their summed durations (ns)
  std::atomic<uint64_t> p2p{0}, vchecks{0};   // device-to-device bytes of a shared operand; BOF_VERIFY pairs compared
  uint64_t ops0[2];  // file_io_ops() when the call began
  Counters() { file_io_ops(&ops0[0], &ops0[1]); }
};

// Durations of the compute launches of one level-3 call (bof_options.kernel_timing = 1), measured where
// the kernels run: a pair of timing events around every tile task / row-block task ON THE STREAM IT IS
// LAUNCHED ON, read back once the call's streams have drained.  What bof_flash_stats.kernel_seconds /
// kernel_launches report, and what bench.py's roofline.achieved is computed from (the figure rocprofv3's
// per-kernel average of the same run must agree with).  Off by default: two markers per launch.
// HIP events and streams a call needs for its own lifetime come out of per-device pools and go back when the call is
// done with them (bof_flash_release destroys the pools): in the steady state a level-3 call creates and destroys no
// HIP object.  The create / destroy churn of hundreds of events per call is what the corrupted handles and the
// wrong tiles of rounds 3-4 correlated with (profiles/r4/fuzz_crash.md) -- the two GEMM paths pooled theirs in round
// 4, the CSR pipelines, the region transfers and KernelTimer follow here.  An object is taken and returned
// on the device it belongs to (the pool remembers); a returned event / stream may still have work pending -- the
// next user re-records the event or queues behind the stream's old work.
hipError_t pooled_event(hipEvent_t *e, bool timing = false);
void pooled_event_return(hipEvent_t e);
hipError_t pooled_stream(hipStream_t *s, bool copy_priority);   // copy_priority: copy_stream_create's kind; else plain non-blocking
void pooled_stream_return(hipStream_t s);
void hip_pools_release();

class KernelTimer {
  std::vector<hipEvent_t> ev;   // begin / end pairs in launch order
  std::vector<char> ended;      // pair i: end() recorded its second event (a pooled event carries an OLD record otherwise)
  std::mutex mu;                // several dispatch streams of one device share a timer
 public:
  bool on = false;
  // returns the launch's pair index through *pair (what end() takes): two dispatchers sharing a timer cannot mix
  // their pairs up
  hipError_t begin(hipStream_t st, size_t *pair = nullptr) {
    if (!on) return hipSuccess;
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = pooled_event(&a, true);
    if (e == hipSuccess) e = pooled_event(&b, true);
    if (e == hipSuccess) e = hipEventRecord(a, st);
    if (e != hipSuccess) {
      if (a) pooled_event_return(a);
      if (b) pooled_event_return(b);
      return e;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (pair) *pair = ev.size() / 2;
    ev.push_back(a);
    ev.push_back(b);
    ended.push_back(0);
    return hipSuccess;
  }
  // the end marker of pair `pair` (default: the most recent begin() -- a timer driven by ONE dispatcher thread)
  hipError_t end(hipStream_t st, size_t pair = (size_t) -1) {
    if (!on) return hipSuccess;
    std::lock_guard<std::mutex> lk(mu);
    if (ev.empty()) return hipSuccess;
    const size_t i = pair == (size_t) -1 ? ev.size() - 1 : 2 * pair + 1;
    if (i >= ev.size()) return hipErrorInvalidValue;
    const hipError_t e = hipEventRecord(ev[i], st);
    if (e == hipSuccess) ended[i / 2] = 1;
    return e;
  }
  // after the streams have been synchronised: adds the pairs that completed to the counters, returns the events
  void collect(Counters &c) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i + 1 < ev.size(); i += 2) {
      float ms = 0.f;
      // a pair whose launch failed between begin() and end() is skipped: its pooled `end` event still holds the
      // record of an earlier call, and the elapsed time against it would be garbage
      if (!ended[i / 2]) {
      } else if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) {
        c.klaunch++;
        c.kns += (uint64_t) ((double) ms * 1e6);
      } else {
        (void) hipGetLastError();
      }
      pooled_event_return(ev[i]);
      pooled_event_return(ev[i + 1]);
    }
    ev.clear();
    ended.clear();
  }
  ~KernelTimer() {
    for (hipEvent_t e : ev) pooled_event_return(e);
  }
};

// Pinned host blocks are expensive to create and destroy (page pinning: ~0.1 s per GB each way),
// so the level-3 pipelines take them from a per-process cache: a freed block is kept (up to
// 4 GiB in total) and handed to the next request of a similar size; bof_flash_release empties it.
int pinned_alloc(void **p, size_t bytes);   // BOF_OK / BOF_EHIP; the block has at least `bytes`
void pinned_free(void *p);
void pinned_cache_release();
// The same for HBM blocks of the CURRENT device that a pipeline needs for the length of one call (CSR row-block
// contexts, x / y of csrgemv): hipMalloc costs 13-36 ms per GiB and hipFree synchronises the device -- the ~20
// of each around a cfg5-size csrgemv were 45-60 ms of a 0.24 s call.  A freed block is kept (up to 8 GiB per
// device) and handed to the next request of a similar size; bof_flash_release empties the cache.
int dev_cache_alloc(void **p, size_t bytes);   // BOF_OK / BOF_EHIP / BOF_ENOMEM
void dev_cache_free(void *p);
void dev_cache_release();

template <class T>
class WorkQueue {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<T> q;
  bool closed = false;

 public:
  void push(const T &v) {
    { std::lock_guard<std::mutex> lk(mu); q.push_back(v); }
    cv.notify_one();
  }
  bool pop(T &out) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return closed || !q.empty(); });
    if (q.empty()) return false;
    out = q.front();
    q.pop_front();
    return true;
  }
  void close() {
    { std::lock_guard<std::mutex> lk(mu); closed = true; }
    cv.notify_all();
  }
};

// Pinned staging ring.  A slot handed out by acquire() is safe to overwrite: the GPU
// copies that last referenced it (mark_busy) have completed.  A slot carries one event per
// device of the call (`cols`): a chunk every device needs is copied out of ONE slot to all of
// them, each copy on that device's own H2D stream (events are device-affine).
class PinnedRing {
  std::vector<void *> slots;
  std::vector<hipEvent_t> ev;      // cols per slot
  std::vector<char> ev_set;
  std::vector<int> ev_dev;         // HIP ordinal of event column j
  std::deque<int> free_;
  std::mutex mu;
  std::condition_variable cv;

 public:
  size_t bytes = 0;
  int count() const { return (int) slots.size(); }
  int cols() const { return (int) ev_dev.size(); }
  // devs: the ordinal of every event column (default: one column on the current device).  A ring that could not
  // be built completely is taken down again before the error is returned.
  int init(int n, size_t nbytes, const std::vector<int> *devs = nullptr) {
    const int rc = build(n, nbytes, devs);
    if (rc) destroy();
    return rc;
  }

 private:
  int build(int n, size_t nbytes, const std::vector<int> *devs) {
    std::vector<int> want;
    if (devs) want = *devs;
    else {
      int cur = 0;
      BOF_HIP_TRY(hipGetDevice(&cur));
      want.push_back(cur);
    }
    if ((int) slots.size() == n && bytes == nbytes && want == ev_dev) {  // reuse a cached ring as is
      free_.clear();
      for (int i = 0; i < n; i++) free_.push_back(i);
      return BOF_OK;
    }
    destroy();
    bytes = nbytes;
    ev_dev = want;
    int cur = 0;
    BOF_HIP_TRY(hipGetDevice(&cur));
    for (int i = 0; i < n; i++) {
      void *p = nullptr;
      const int rc = pinned_alloc(&p, nbytes);
      if (rc) return rc;
      file_buffers_add(p, nbytes);   // io_uring fixed buffer (no effect on the AIO engine)
      slots.push_back(p); free_.push_back(i);
      for (int d : ev_dev) {
        hipEvent_t e;
        BOF_HIP_TRY(hipSetDevice(d));
        const hipError_t he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        (void) hipSetDevice(cur);
        if (he != hipSuccess) return hip_fail(he, "hipEventCreate (pinned ring)");
        ev.push_back(e); ev_set.push_back(0);
      }
    }
    return BOF_OK;
  }

 public:
  void destroy() {
    for (size_t i = 0; i < ev.size(); i++) {
      if (ev_set[i]) (void) hipEventSynchronize(ev[i]);
      (void) hipEventDestroy(ev[i]);
    }
    for (size_t i = 0; i < slots.size(); i++) {
      file_buffers_remove(slots[i]);
      pinned_free(slots[i]);
    }
    slots.clear(); ev.clear(); ev_set.clear(); ev_dev.clear(); free_.clear();
  }
  int acquire() {
    int idx;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !free_.empty(); });
      idx = free_.front();
      free_.pop_front();
    }
    const size_t c = ev_dev.size();
    for (size_t j = 0; j < c; j++)
      if (ev_set[(size_t) idx * c + j]) { (void) hipEventSynchronize(ev[(size_t) idx * c + j]); ev_set[(size_t) idx * c + j] = 0; }
    return idx;
  }
  void release(int idx) {
    { std::lock_guard<std::mutex> lk(mu); free_.push_back(idx); }
    cv.notify_one();
  }
  int mark_busy(int idx, hipStream_t st, int col = 0) {
    const size_t e = (size_t) idx * ev_dev.size() + (size_t) col;
    BOF_HIP_TRY(hipEventRecord(ev[e], st));
    ev_set[e] = 1;
    return BOF_OK;
  }
  hipEvent_t event(int idx, int col = 0) { return ev[(size_t) idx * ev_dev.size() + (size_t) col]; }
  void *ptr(int idx) { return slots[idx]; }
};

static inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Runs the registered releases in reverse order when the call returns, on every path.
struct Cleanup {
  std::vector<std::function<void()>> fns;
  void add(std::function<void()> f) { fns.push_back(std::move(f)); }
  ~Cleanup() {
    for (auto it = fns.rbegin(); it != fns.rend(); ++it) (*it)();
  }
};

// Pins the calling thread to the CPUs of the NUMA node the HIP device hangs off (PCI bus id ->
// /sys/bus/pci/devices/*/numa_node -> node cpulist), intersected with the CPUs the process is
// allowed on.  Reader / writer / flusher threads call it so that page-cache copies and the
// pinned staging rings stay on the GPU's socket.  No-op when the topology is
// unknown or BOF_NUMA_BIND=0.  Returns the node or -1.
int bind_thread_near_device(int dev);
// Named ranges for rocprofv3 --marker-trace (roctx): resolved from librocprofiler-sdk-roctx /
// libroctx64 at first use, silently absent otherwise.  BOF_TRACE=1 additionally prints a
// wall-clock timeline to stderr.
void trace_push(const char *name);
void trace_pop();
struct TraceRange {
  explicit TraceRange(const char *name) { trace_push(name); }
  ~TraceRange() { trace_pop(); }
};
// Stream for H2D / D2H copies: non-blocking and of the HIGHEST priority.  HIP multiplexes its
// streams onto a few hardware queues per priority level; a copy stream that lands on the same
// queue as a compute stream is serialised behind whole-chip tile kernels (a leftover compute
// stream of an earlier level-2 call was enough to turn a 0.575 s cfg2 file run into 0.77 s,
// profiles/r2/e2e_stream_aliasing.txt).  High-priority streams get queues of their own, and the
// D2H blit kernels (0.67 ms per 32 MiB) then go ahead of queued 0.95 ms tile kernels.
hipError_t copy_stream_create(hipStream_t *s);
int device_ready();                                       // BOF_OK or BOF_ENODEV (+ message)
void publish_stats(const Counters &c, double seconds);    // what bof_flash_last_stats reports
void publish_device_stats(const std::vector<bof_flash_stats> &per_device);   // bof_flash_last_device_stats

// The devices a level-3 call runs on: opts->devices, else $BOF_DEVICES ("0,1" / "all"), else the
// calling thread's current device.  Ordinals are checked against hipGetDeviceCount; repeats are kept.
int resolve_devices(const bof_options &o, std::vector<int> &devs);
// Holds the per-device call locks of every distinct ordinal in the list (ascending order) for
// the duration of a level-3 call.
class DeviceCallLock {
  std::vector<std::recursive_mutex *> held;
 public:
  explicit DeviceCallLock(const std::vector<int> &devs);
  ~DeviceCallLock();
};
// Sets the calling thread's device for a scope and puts the previous one back.
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void) hipSetDevice(dev);
  }
  ~DeviceScope() { if (prev >= 0) (void) hipSetDevice(prev); }
};
// environment knob read at every call (never cached): `dflt` when unset or empty
long env_long(const char *name, long dflt);

// what a pipeline's io_error code says (negative errno, or -1000 - hipError_t)
inline std::string io_error_text(int e) {
  if (e == -ETIMEDOUT) return "timed out: no progress within $BOF_STALL_TIMEOUT_S, or a peer rank did not deliver its chunk";
  return e > -1000 ? std::string(strerror(-e)) : "HIP error " + std::to_string(-1000 - e);
}

// ---- always-on event ring ------------------------------------------------------------------------------
// Every hand-over of the level-3 pipelines (call begin / end, chunk read, H2D queued, panel ready, group
// dispatched, D2H queued, D2H complete, chunk written, share-ring produce / consume, verify mismatch) leaves one
// 32-byte record -- steady-clock nanoseconds, thread, a static label, three numbers -- in a process-wide ring
// of the last 4096 events.  Recording is one relaxed fetch_add and four stores, so it is never switched off.
// The ring is written out (oldest first, times relative to the newest call's begin) by StallWatch when a call
// stops moving, by the BOF_VERIFY checks on a mismatch, at the end of every level-3 call when
// $BOF_EVENT_DUMP names a file (appended), and by bof_event_dump(); a stall or a wrong tile therefore comes
// with the last few thousand things the library did, whichever thread did them.
struct EventRec {
  uint64_t t_ns;
  const char *what;     // string literal
  uint32_t tid;
  int32_t a, b;
  uint32_t c;
};
constexpr uint32_t kEventRing = 4096;
// (header-only, C++17 inline variables: the test harnesses link single pieces of the library)
inline EventRec g_evt[kEventRing];
inline std::atomic<uint64_t> g_evt_next{0};
inline std::atomic<uint64_t> g_evt_call_begin_ns{0};
inline uint64_t evt_now_ns() {
  return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void evt(const char *what, int a = 0, int b = 0, uint64_t c = 0) {
  static thread_local const uint32_t tid = (uint32_t) syscall(SYS_gettid);
  const uint64_t i = g_evt_next.fetch_add(1, std::memory_order_relaxed);
  EventRec &r = g_evt[i % kEventRing];
  // relaxed atomic stores, the label last with release: a dump that races with a writer (by design: the ring is
  // never locked) reads each field whole and sees either the old record's label or the new one's
  __atomic_store_n(&r.t_ns, evt_now_ns(), __ATOMIC_RELAXED);
  __atomic_store_n(&r.tid, tid, __ATOMIC_RELAXED);
  __atomic_store_n(&r.a, a, __ATOMIC_RELAXED);
  __atomic_store_n(&r.b, b, __ATOMIC_RELAXED);
  __atomic_store_n(&r.c, (uint32_t) c, __ATOMIC_RELAXED);
  __atomic_store_n(&r.what, what, __ATOMIC_RELEASE);
}
inline void evt_mark_call_begin() { g_evt_call_begin_ns.store(evt_now_ns()); }       // times in a dump are relative to the last of these
inline uint64_t evt_count() { return g_evt_next.load(); }
inline void evt_dump(FILE *f, const char *why) {            // whole ring, oldest first
  const uint64_t end = g_evt_next.load();
  const uint64_t begin = end > kEventRing ? end - kEventRing : 0;
  const uint64_t t0 = g_evt_call_begin_ns.load();
  fprintf(f, "[bof events] %s: last %llu of %llu events (ms relative to the last call's begin; thread; event; a b c)\n",
          why ? why : "", (unsigned long long) (end - begin), (unsigned long long) end);
  for (uint64_t i = begin; i < end; i++) {
    EventRec &src = g_evt[i % kEventRing];
    const char *w = __atomic_load_n(&src.what, __ATOMIC_ACQUIRE);
    if (!w) continue;
    EventRec r;
    r.t_ns = __atomic_load_n(&src.t_ns, __ATOMIC_RELAXED);
    r.tid = __atomic_load_n(&src.tid, __ATOMIC_RELAXED);
    r.a = __atomic_load_n(&src.a, __ATOMIC_RELAXED);
    r.b = __atomic_load_n(&src.b, __ATOMIC_RELAXED);
    r.c = __atomic_load_n(&src.c, __ATOMIC_RELAXED);
    fprintf(f, "[bof events] %12.3f  t%-7u %-34s %d %d %u\n", ((double) r.t_ns - (double) t0) * 1e-6, r.tid, w, r.a, r.b, r.c);
  }
  fflush(f);
}
inline void evt_dump_env(const char *why) {                 // to $BOF_EVENT_DUMP (append) when set
  const char *path = getenv("BOF_EVENT_DUMP");
  if (!path || !path[0]) return;
  FILE *f = fopen(path, "a");
  if (!f) return;
  evt_dump(f, why);
  fclose(f);
}                          // times in a dump are relative to the last of these

// ---- host-confirmed hand-overs (round 5) ----------------------------------------------------------------------
// Every dependency between work that DIFFERENT host threads submit -- a reader's H2D copies and the kernels that
// read them, a panel's last kernels and its write-back, a slot's old occupant and the copy that refills it -- is
// confirmed ON THE HOST by the consuming side (hipEventSynchronize on an event that the producing thread itself
// recorded right behind its own submission) before the dependent work is submitted; the device-side
// hipStreamWaitEvent stays, as a second line.  Why: what the instrumented fuzz of round 5 finally SAW (1 in 936 000
// cases, profiles/r5/fuzz_summary.md) is a kernel that summed an operand tile on one stream and got other words than a
// kernel that summed it on another stream later -- one of them ran before an H2D copy that precedes it in stream /
// event order had landed.  Every such order in the failing configuration crossed host threads (copies submitted by a
// reader thread, the kernel behind them by a launcher thread, the event by whichever reader finished the group); all
// wrong results of rounds 3-4 had cross-thread submission in common too.  The only ordering the pipelines still rely
// on without a host check is the most basic one: operations one thread submits to one stream run in that order.
// Cost: a reader waits ~0.6 ms for its 32 MiB copy before it takes the next request (the disk, not the readers,
// bounds the pipelines).
// ROUND 6: OFF by default again ($BOF_HOST_HANDOVER=1 turns it on).  The events this was built against are not a
// hand-over problem: (1) the stand-alone stress of exactly this pattern (tools/exp/handover_stress.hip) ran 58 million
// hand-overs over every ingredient named above without a wrong word; (2) the round's final fuzz caught three wrong
// csrmm results whose dumps show ONE launch in which the workgroups of one XCD did their neighbours' rows (rows of
// blockIdx w - 1 updated twice, rows of w not at all, for every w = x mod 8), and tools/exp/wg_id_stress.hip -- no
// library, two trivial kernels, eight processes on the GPU -- reproduced that: one launch in ~7 million in which 93
// workgroups of one XCD (blockIdx = 3 mod 8) did not write their rows.  A sum kernel or a tile kernel some of whose
// workgroups run under a wrong ID explains every unexplained event of rounds 3-5 (a tile's two sums differ by a few
// thousand words; a k-block missing from one tile column), and no ordering on the host can prevent it.  It has only
// ever been seen with several processes sharing the GPU (profiles/r6/incident_csrmm/README.md).
inline bool host_handover() {
  static const bool on = env_long("BOF_HOST_HANDOVER", 0) != 0;
  return on;
}
// the consumer's side of a hand-over: host check (when on), then the device-side wait as before
inline hipError_t wait_event_both(hipStream_t st, hipEvent_t ev) {
  if (host_handover()) {
    const hipError_t e = hipEventSynchronize(ev);
    if (e != hipSuccess) return e;
  }
  return hipStreamWaitEvent(st, ev, 0);
}

// ---- persistent launcher threads ------------------------------------------------------------------------------
// Kernel launches are only ever issued by the CALLING thread or by one of these long-lived threads, never by a
// thread created for the call.  Why -- CORRELATED, NOT PROVEN (profiles/r4/fuzz_thread_bisect.md, sections 4-6): with a
// fresh dispatcher thread per call AND hundreds of HIP events created and destroyed per call, 4-9 C tiles per 29 000
// fuzzed multi-slab calls came out wrong (a tile task of flash::kmeans ran with the next task's p_l2sq pointer; a plain
// tile task with a wrong scalar); 0 per 29 000 from the calling thread, 0 from a persistent worker -- and, once the
// events were pooled, 0 per 34 000 from a fresh thread as well: either half of the combination removed makes the wrong
// tiles go away, and the cause inside the runtime was not found.  Both halves stay removed (these threads; pooled
// events and copy streams; rule R6 of the mock runtime).  One later mismatch (a k-block that did not add to C, both
// dispatchers of a repeated ordinal, 1 in ~450 000 cases) is unexplained; what it has in common with every earlier
// one is a REPEATED ORDINAL whose dispatchers fed one compute-stream set from several host threads -- since round 5
// every repetition has its own streams (c_api.hip: stream_rep) and BOF_VERIFY checks every launch at its source
// (consumer-side sums, spot checks: "Round 5" below).  launch_async(dev, rep, fn) runs fn on the persistent thread of
// (device, repetition of the ordinal in the call's device list), created on first use and kept for the life of the
// process; launch_wait joins it.
// what the crash handler ($BOF_CRASH_TRACE=1) prints besides the stack: a callback the running pipeline registers
extern std::atomic<void (*)(void *)> g_crash_dump_fn;      // (the last pipeline to register wins: a diagnostic)
extern std::atomic<void *> g_crash_dump_arg;

struct LaunchJob {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
};
std::shared_ptr<LaunchJob> launch_async(int dev, int rep, std::function<void()> fn);
void launch_wait(const std::shared_ptr<LaunchJob> &job);
// flags of the POOLED events of the two GEMM paths: hipEventDisableTiming, or -- $BOF_EVENT_TIMING=1, an experiment switch
// for the crash site of profiles/r4/fuzz_crash.md (a different record path inside the runtime) -- hipEventDefault
inline unsigned pooled_event_flags() {
  static const unsigned f = env_long("BOF_EVENT_TIMING", 0) ? (unsigned) hipEventDefault : (unsigned) hipEventDisableTiming;
  return f;
}
bool on_launcher_thread();      // (the mock runtime's rule R6: kernels are launched by the caller or by these threads)
// The kernels a thread CREATED FOR THE CALL needs behind its copies (a reader's k-major copy, BOF_VERIFY's sums, the
// transposition of a resident operand): fn runs on the device's auxiliary launcher thread and this thread waits until
// it has returned (the launches are queued, nothing is waited for on the GPU).  Jobs of this lane only enqueue work.
constexpr int kAuxLane = 2 << 20;        // (lanes below 1 << 20: repetitions of an ordinal; 1 << 20: the CSR feeder)
template <class F>
inline hipError_t launch_from_persistent(int dev, F &&fn) {
  if (on_launcher_thread()) return fn();
  hipError_t e = hipSuccess;
  launch_wait(launch_async(dev, kAuxLane, [&] {
    e = hipSetDevice(dev);
    if (e == hipSuccess) e = fn();
  }));
  return e;
}

// ---- BOF_VERIFY: hand-over checksums (bof_options.verify / $BOF_VERIFY) ------------------------------------
// Every object a level-3 GEMM pipeline moves (a row panel; a packed tile) is summed at each hand-over -- in the
// pinned slot after the file read, in HBM behind the H2D copies, in HBM again behind the last kernel that used
// it (before its slot is refilled, or at the end of the call), C in HBM behind its last kernel, in the pinned
// slot behind the D2H copy, and in the file once the call's writes have drained -- as two 64-bit sums (words,
// and words weighted by their logical position in the object; verify_sum_kernel / host_word_sums).  The sums
// of one object at two points must agree; the pairs are compared on the host when the call has drained, the
// first divergence is named (which object, between which two points), the event ring is dumped and the call
// fails with BOF_EVERIFY.  Off by default: it reads every byte several more times.
inline void host_word_sums(const void *p, int64_t rows, int64_t row_words, int64_t pitch_words, uint64_t index_base,
                           uint64_t out[2]) {
  uint64_t s1 = 0, s2 = 0;
  const uint32_t *w = (const uint32_t *) p;
  for (int64_t r = 0; r < rows; r++) {
    const uint32_t *row = w + r * pitch_words;
    uint64_t li = index_base + (uint64_t) (r * row_words) + 1;
    for (int64_t c = 0; c < row_words; c++, li++) {
      s1 += row[c];
      s2 += (uint64_t) row[c] * li;
    }
  }
  out[0] = s1;
  out[1] = s2;
}
inline bool verify_wanted(const bof_options &o) {
  if (o.verify == 1) return true;
  if (o.verify == 2) return false;
  const char *e = getenv("BOF_VERIFY");
  return e && e[0] && strcmp(e, "0") != 0;
}
// Round 5, what the hand-over sums could not see (profiles/r4/fuzz_thread_bisect.md section 6: a k-block that did not
// add to C although every operand sum agreed):
//  * CONSUMER-side sums: every operand panel / tile is summed once more ON THE COMPUTE STREAM in front of each launch
//    that reads it (and a chain's partial sums behind one launch and in front of the next): a missing or mis-targeted
//    cross-stream wait shows as a launch-side sum that differs from the producer-side one;
//  * SPOT CHECKS: 64 outputs of every launch recomputed from its own arguments (bof_internal.h: SpotArgs) -- a
//    launch that ran with other arguments, too early, twice or not at all is named by its (panel, k-range);
//  * POISON: every HBM image is filled with 0xFF words (NaN) before its first byte of the call arrives, so a read
//    ahead of the fill gives NaN instead of a plausible old number.
class Verify {
  struct Expect { size_t a, b; const char *what; int id0, id1, id2; };
  unsigned long long *d_tab = nullptr;          // 2 per entry, in the HBM of `dev`
  float *d_spot = nullptr;                      // 128 saved values per spot check
  std::atomic<size_t> spot_next{0};
  size_t spot_cap = 0;
  std::vector<std::atomic<uint64_t>> h_tab;     // 2 per entry
  std::vector<std::atomic<uint8_t>> touched;    // bit 0: host side filled, bit 1: device side filled
  std::atomic<size_t> next{0};
  size_t cap = 0;
  std::mutex mu;
  std::vector<Expect> expects;
  int dev = 0;

 public:
  static constexpr size_t kNone = (size_t) -1;
  bool on = false;
  ~Verify() { release(); }
  int init(int device, size_t capacity, size_t spots = 0);   // BOF_OK / BOF_EHIP; the table lives on `device`
  // spot check of one launch: before() in front of it on its stream, after() behind it (flash_common.h header above)
  struct Spot { float *save = nullptr; size_t e_exp = kNone, e_got = kNone; };
  hipError_t spot_before(const SpotArgs &a, hipStream_t st, Spot *sp, const char *what, int id0, int id1, int id2) {
    *sp = Spot();
    if (!on || !d_spot) return hipSuccess;
    const size_t i = spot_next.fetch_add(1);
    if (i >= spot_cap) return hipSuccess;
    sp->save = d_spot + 128 * i;
    sp->e_exp = entry();
    sp->e_got = entry();
    if (sp->e_exp == kNone || sp->e_got == kNone) { sp->save = nullptr; return hipSuccess; }
    expect(sp->e_exp, sp->e_got, what, id0, id1, id2);
    return sgemm_spot_capture(a, sp->save, st);
  }
  hipError_t spot_after(const SpotArgs &a, const Spot &sp, hipStream_t st) {
    if (!on || !sp.save) return hipSuccess;
    touched[sp.e_exp].fetch_or(2);
    touched[sp.e_got].fetch_or(2);
    return sgemm_spot_check(a, sp.save, d_tab + 2 * sp.e_exp, d_tab + 2 * sp.e_got, st);
  }
  // fills an HBM range with 0xFF words on `st` (NaN: a read ahead of the fill cannot pass for data)
  hipError_t poison(void *p, size_t bytes, hipStream_t st) {
    if (!on || !p || !bytes) return hipSuccess;
    return launch_from_persistent(dev, [&] { return hipMemsetAsync(p, 0xFF, bytes, st); });
  }
  void release();
  size_t entry() {
    if (!on) return kNone;
    const size_t e = next.fetch_add(1);
    return e < cap ? e : kNone;
  }
  hipError_t on_device(size_t e, const void *p, int64_t rows, int64_t row_words, int64_t pitch_words, uint64_t index_base,
                       int64_t t_pitch, hipStream_t st) {
    if (!on || e == kNone) return hipSuccess;
    touched[e].fetch_or(2);
    return launch_from_persistent(dev, [&] { return verify_sum(p, rows, row_words, pitch_words, index_base, t_pitch, d_tab + 2 * e, st); });
  }
  void on_host(size_t e, const void *p, int64_t rows, int64_t row_words, int64_t pitch_words, uint64_t index_base) {
    if (!on || e == kNone) return;
    uint64_t s[2];
    host_word_sums(p, rows, row_words, pitch_words, index_base, s);
    h_tab[2 * e].fetch_add(s[0], std::memory_order_relaxed);
    h_tab[2 * e + 1].fetch_add(s[1], std::memory_order_relaxed);
    touched[e].fetch_or(1);
  }
  void expect(size_t a, size_t b, const char *what, int id0 = 0, int id1 = 0, int id2 = 0) {
    if (!on || a == kNone || b == kNone) return;
    std::lock_guard<std::mutex> lk(mu);
    expects.push_back(Expect{a, b, what, id0, id1, id2});
  }
  // The device must be idle (the call has drained).  Compares every expected pair both of whose sides were
  // filled; BOF_OK, or BOF_EVERIFY with the first divergence in bof_last_error() and the event ring on stderr.
  int finish(Counters &cnt, const char *call);
};

// Watches one level-3 pipeline: `progress` is any number that changes while the call advances (bytes moved +
// tasks launched).  When it has stood still for $BOF_STALL_TIMEOUT_S seconds (default 600; 0 = off) one line
// goes to stderr and `on_stall` fails the call (-ETIMEDOUT -> BOF_EIO), so that everything parked on the
// pipeline's condition variable returns instead of waiting forever.  The reference's counterpart is the fatal
// exit after five failed submit / reap rounds (src/file_handles/flash_file_handle.cpp:28-76).  A thread stuck
// INSIDE a system call or a HIP call is not released by this; the line on stderr then says how far the call got.
class StallWatch {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;

 public:
  StallWatch(const char *what, std::function<uint64_t()> progress, std::function<void()> on_stall) {
    const long limit = env_long("BOF_STALL_TIMEOUT_S", 600);
    if (limit <= 0) return;
    th = std::thread([this, what, progress, on_stall, limit] {
      using clock = std::chrono::steady_clock;
      std::unique_lock<std::mutex> lk(mu);
      uint64_t last = progress();
      clock::time_point since = clock::now();
      const auto step = std::chrono::milliseconds(std::max<long>(50, std::min<long>(limit * 250, 5000)));
      // (wait_until on the system clock = pthread_cond_timedwait, which ThreadSanitizer understands; wait_for
      //  goes through pthread_cond_clockwait, which GCC 11's does not.  Only the polling step hangs on that
      //  clock: the idle time is measured on the steady one.)
      while (!cv.wait_until(lk, std::chrono::system_clock::now() + step, [this] { return done; })) {
        const uint64_t now = progress();
        if (now != last) { last = now; since = clock::now(); continue; }
        const double idle = std::chrono::duration<double>(clock::now() - since).count();
        if (idle < (double) limit) continue;
        fprintf(stderr, "[bof] %s: no progress for %.0f s (progress counter at %llu): failing the call\n", what, idle,
                (unsigned long long) now);
        evt("stall watchdog fired", 0, 0, now);
        evt_dump(stderr, what);
        on_stall();
        return;
      }
    });
  }
  ~StallWatch() {
    if (!th.joinable()) return;
    { std::lock_guard<std::mutex> lk(mu); done = true; }
    cv.notify_all();
    th.join();
  }
  StallWatch(const StallWatch &) = delete;
  StallWatch &operator=(const StallWatch &) = delete;
};

// BOF_TRACE=1: wall-clock milestones of a level-3 call on stderr (t_begin = the call's start)
inline bool trace_enabled() {
  static const bool on = getenv("BOF_TRACE") != nullptr;
  return on;
}
#define BOF_TRACE_T(label)                                                                       \
  do {                                                                                           \
    if (::bof::trace_enabled())                                                                  \
      fprintf(stderr, "[bof trace] %-28s %8.3f ms\n", label,                                     \
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); \
  } while (0)

// Whole array <-> file with up to n_thr workers over 32 MiB pinned chunks (flash_csr.cpp)
int stream_file(const bof_fptr &f, uint64_t bytes, char *dptr, bool to_device, hipStream_t st,
                bool use_aio, int n_thr, Counters &cnt);


// flash::gemm through whole row panels kept in HBM in FILE layout (flash_gemm_panels.cpp).
// Returns BOF_OK / an error, or +1 when the call is not eligible (layout, budget) and the tile
// cache of flash_runtime.cpp must take it.
// `devs`: the devices the C panels are dealt to (contiguous ranges); kh: flash::kmeans' host vectors.
struct KmeansHost {
  const float *c_l2sq, *p_l2sq, *ones;
  int64_t m, n, n_ones;
};
int flash_gemm_panels(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, float beta,
                      bof_fptr fa, bof_fptr fb, bof_fptr fc, int64_t lda, int64_t ldb, int64_t ldc,
                      const bof_options &o, const std::vector<int> &devs, const KmeansHost *kh = nullptr);
// the three norm vectors of flash::kmeans on the current device (one allocation: free kv->c_l2sq)
int kmeans_upload(const KmeansHost &kh, KmeansVecs *kv);
void panel_resources_release();
void panel_resources_release_device(int dev);   // the cached panel slots of one ordinal (its call lock is held)

}  // namespace bof
