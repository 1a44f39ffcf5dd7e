// handover_stress.hip -- stand-alone stress of the cross-thread hand-over the level-3 pipelines rely on
// (VERDICT r5 "next round" item 2; protects the dependency of /root/reference/src/blas/gemm.cpp:122-127,
// add_parent + in_mem_ptrs, which here is "a tile's H2D copies happen-before the kernel that reads the tile").
//
// The one direct sighting of the event (profiles/r5/fuzz_run1/kmeans_seed9201_failure.txt): a tile summed on the
// H2D stream right behind its copies != the same tile summed later on the compute stream; a few thousand words held
// the slot's PREVIOUS contents.  This program repeats that pattern with self-checking data and nothing else around it:
//
//   reader threads (R of them, ONE h2d stream):   [WAR wait on used[s]] -> H2D copy of generation g into HBM slot s
//                                                 -> [check kernel on the h2d stream] -> record ready[s] -> hand g to
//   dispatcher thread (compute streams):          wait ready[s] on a compute stream -> check kernel -> record used[s]
//
// Every word of generation g has a value that no other generation of the same slot has, so a kernel that runs ahead of
// a copy (or reads stale L2 lines behind an SDMA write) counts the words it found wrong, separately for the kernel on
// the h2d stream and the one on the compute stream.  Knobs = the ingredients VERDICT names:
//   --events fixed|pooled|fresh   per-slot events re-recorded (what the pipelines do) | drawn from a pool and returned
//                                 while their work is still pending (flash_common.h:44-50) | created / destroyed per use
//   --copy 1d|2d|chunks           one linear copy | one hipMemcpy2DAsync (the tile cache's) | 4 linear pieces
//   --wgs N                       workgroups of the check kernel (1: one XCD touches the slot; 8: every XCD a fixed
//                                 eighth -- the same lines land in the same L2 every generation; 64: all mixed)
//   --prior-read 0|1              0: K is large, a slot is cold in every L2 when it is refilled; 1: K = 4 slots
//   --h2d-check 0|1               the producer-side kernel on the h2d stream (submitted by the reader)
//   --launcher 0|1                that kernel and the ready record come from a THIRD thread (the tile cache's launcher)
//   --host-confirm 0|1            the reader waits for its copy on the host before it hands over (BOF_HOST_HANDOVER)
//   --readers R  --streams S  --words W  --seconds T
//   --pipelines P                 P independent pipelines in this process (more HIP streams than hardware queues)
//   --no-wait 1                   self-test of the checker: the dispatcher skips the wait for `ready`
// Build: hipcc --offload-arch=gfx950 -O2 -pthread tools/exp/handover_stress.hip -o tools/exp/handover_stress
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define CK(x)                                                                                                     \
  do {                                                                                                            \
    hipError_t e_ = (x);                                                                                          \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

__host__ __device__ inline uint32_t word_of(uint32_t pat, uint32_t i) { return pat * 0x9E3779B1u + i * 0x85EBCA6Bu + 0x1234567u; }

// counters[0]: wrong words seen, counters[1]: launches that saw any, counters[2..5]: first sighting (gen, index, found, expected)
__global__ void check_slot(const uint32_t *slot, uint32_t n_words, uint32_t pat, uint32_t gen, unsigned long long *counters) {
  const uint32_t per = (n_words + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = min(n_words, lo + per);
  uint32_t bad = 0, first_i = 0, first_v = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const uint32_t v = slot[i];
    if (v != word_of(pat, i)) {
      if (!bad) { first_i = i; first_v = v; }
      bad++;
    }
  }
  if (bad) {
    atomicAdd(&counters[0], (unsigned long long) bad);
    if (atomicAdd(&counters[1], 1ull) == 0) {
      counters[2] = gen;
      counters[3] = first_i;
      counters[4] = first_v;
      counters[5] = word_of(pat, first_i);
    }
  }
}

template <class T>
struct Queue {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<T> q;
  bool closed = false;
  void push(T v) {
    { std::lock_guard<std::mutex> lk(mu); q.push_back(v); }
    cv.notify_one();
  }
  bool pop(T &v) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return closed || !q.empty(); });
    if (q.empty()) return false;
    v = q.front();
    q.pop_front();
    return true;
  }
  void close() {
    { std::lock_guard<std::mutex> lk(mu); closed = true; }
    cv.notify_all();
  }
};

struct EventPool {       // flash_common.h's pool in miniature: a returned event may still have work pending
  std::mutex mu;
  std::vector<hipEvent_t> free_;
  hipEvent_t get() {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!free_.empty()) { hipEvent_t e = free_.back(); free_.pop_back(); return e; }
    }
    hipEvent_t e;
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return e;
  }
  void put(hipEvent_t e) { std::lock_guard<std::mutex> lk(mu); free_.push_back(e); }
};

struct Cfg {
  std::string events = "fixed", copy = "1d";
  int wgs = 8, prior_read = 1, h2d_check = 1, launcher = 0, host_confirm = 0, readers = 2, streams = 2;
  int pipelines = 1;      // pipelines side by side in this process (a device list that repeats its ordinal)
  int no_wait = 0;        // self-test: the dispatcher does NOT wait for `ready` -- the checker must then see stale words
  uint32_t words = 65536;
  double seconds = 20;
};

struct Handed { uint64_t gen; hipEvent_t ready; };

static EventPool g_pool;

struct Result { unsigned long long handovers = 0, a[8] = {0}, b[8] = {0}; double dt = 0; int slots = 0; };

// ONE pipeline: its own h2d stream, compute streams, HBM slots, events, reader / launcher / dispatcher threads.
// --pipelines P runs P of them side by side in this process -- what a device list that repeats an ordinal ([0,0,0]: the
// configuration of the one sighting) does: 3 x (copy streams + compute streams) on ONE device, i.e. more HIP streams
// than the runtime has hardware queues (GPU_MAX_HW_QUEUES, 4 by default), so streams of different pipelines SHARE a
// hardware queue and whatever per-queue state the runtime keeps about pending copies and cache invalidates.
static void run_pipeline(const Cfg &c, const std::vector<uint32_t *> &pin, int P, Result *res) {
  const size_t bytes = (size_t) c.words * 4;
  // prior-read 1: 4 slots, each still (partly) in the L2s that read it when it is refilled; 0: enough slots that
  // 2 x (L2 + Infinity Cache = 32 + 256 MiB) of other slots pass through in between
  const int K = c.prior_read ? 4 : (int) std::max<size_t>(8, (size_t) (640ull << 20) / bytes);
  CK(hipSetDevice(0));
  hipStream_t h2d;
  int least, greatest;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&h2d, hipStreamNonBlocking, greatest));
  std::vector<hipStream_t> comp((size_t) c.streams);
  for (auto &s : comp) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<uint32_t *> slot((size_t) K);
  for (int s = 0; s < K; s++) {
    CK(hipMalloc((void **) &slot[(size_t) s], bytes));
    CK(hipMemset(slot[(size_t) s], 0xFF, bytes));
  }
  unsigned long long *cnt_h2d, *cnt_comp;
  CK(hipMalloc((void **) &cnt_h2d, 64));
  CK(hipMalloc((void **) &cnt_comp, 64));
  CK(hipMemset(cnt_h2d, 0, 64));
  CK(hipMemset(cnt_comp, 0, 64));
  CK(hipDeviceSynchronize());

  std::vector<hipEvent_t> ready_fixed((size_t) K), used((size_t) K);
  for (int s = 0; s < K; s++) {
    CK(hipEventCreateWithFlags(&ready_fixed[(size_t) s], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&used[(size_t) s], hipEventDisableTiming));
  }
  EventPool &pool = g_pool;      // ONE pool per device, shared by the pipelines, as in the library
  // slot s is free for generation g once the dispatcher has SUBMITTED the consumer of generation g - K and recorded
  // used[s] behind it (the device-side WAR wait does the rest)
  std::vector<std::atomic<int64_t>> consumed((size_t) K);
  for (int s = 0; s < K; s++) consumed[(size_t) s].store((int64_t) s - K);
  std::atomic<uint64_t> next_gen{0}, done_gens{0};
  std::atomic<bool> stop{false};
  Queue<Handed> to_launcher, to_dispatcher;
  std::mutex h2d_mu;      // readers take turns on the ONE h2d stream per generation (copies of one generation are not interleaved with another's kernel)

  auto finish_producer = [&](uint64_t g, bool from_launcher) {
    (void) from_launcher;
    const int s = (int) (g % (uint64_t) K);
    if (c.h2d_check)
      hipLaunchKernelGGL(check_slot, dim3((unsigned) c.wgs), dim3(256), 0, h2d, slot[(size_t) s], c.words, (uint32_t) (g % P), (uint32_t) g, cnt_h2d);
    hipEvent_t ev;
    if (c.events == "fixed") ev = ready_fixed[(size_t) s];
    else if (c.events == "pooled") ev = pool.get();
    else CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventRecord(ev, h2d));
    to_dispatcher.push(Handed{g, ev});
  };

  auto reader = [&]() {
    CK(hipSetDevice(0));
    hipEvent_t mine;
    CK(hipEventCreateWithFlags(&mine, hipEventDisableTiming));
    while (!stop.load(std::memory_order_relaxed)) {
      const uint64_t g = next_gen.fetch_add(1);
      const int s = (int) (g % (uint64_t) K);
      while (consumed[(size_t) s].load(std::memory_order_acquire) != (int64_t) g - K) {
        if (stop.load(std::memory_order_relaxed)) return;
        std::this_thread::yield();
      }
      const uint32_t *src = pin[(size_t) (g % P)];
      {
        std::lock_guard<std::mutex> lk(h2d_mu);
        if (g >= (uint64_t) K) CK(hipStreamWaitEvent(h2d, used[(size_t) s], 0));
        if (c.copy == "1d") {
          CK(hipMemcpyAsync(slot[(size_t) s], src, bytes, hipMemcpyHostToDevice, h2d));
        } else if (c.copy == "2d") {
          const size_t w = 1024;      // 256 words per row
          CK(hipMemcpy2DAsync(slot[(size_t) s], w, src, w, w, bytes / w, hipMemcpyHostToDevice, h2d));
        } else {
          const size_t q = bytes / 4;
          for (int j = 0; j < 4; j++)
            CK(hipMemcpyAsync((char *) slot[(size_t) s] + j * q, (const char *) src + j * q, q, hipMemcpyHostToDevice, h2d));
        }
        if (c.host_confirm) CK(hipEventRecord(mine, h2d));
        if (!c.launcher && !c.host_confirm) finish_producer(g, false);
      }
      if (c.host_confirm) {
        CK(hipEventSynchronize(mine));
        if (!c.launcher) { std::lock_guard<std::mutex> lk(h2d_mu); finish_producer(g, false); }
      }
      if (c.launcher) to_launcher.push(Handed{g, nullptr});
    }
  };
  auto launcher = [&]() {
    CK(hipSetDevice(0));
    Handed h;
    while (to_launcher.pop(h)) {
      std::lock_guard<std::mutex> lk(h2d_mu);
      finish_producer(h.gen, true);
    }
  };
  auto dispatcher = [&]() {
    CK(hipSetDevice(0));
    Handed h;
    while (to_dispatcher.pop(h)) {
      const int s = (int) (h.gen % (uint64_t) K);
      hipStream_t st = comp[(size_t) (h.gen % (uint64_t) c.streams)];
      if (!c.no_wait) CK(hipStreamWaitEvent(st, h.ready, 0));
      if (c.events == "pooled") pool.put(h.ready);           // returned while the wait is still pending, as the library does
      else if (c.events == "fresh") CK(hipEventDestroy(h.ready));
      hipLaunchKernelGGL(check_slot, dim3((unsigned) c.wgs), dim3(256), 0, st, slot[(size_t) s], c.words, (uint32_t) (h.gen % P), (uint32_t) h.gen, cnt_comp);
      CK(hipEventRecord(used[(size_t) s], st));
      consumed[(size_t) s].store((int64_t) h.gen, std::memory_order_release);
      done_gens.fetch_add(1, std::memory_order_relaxed);
    }
  };

  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  std::thread tl, td(dispatcher);
  if (c.launcher) tl = std::thread(launcher);
  for (int r = 0; r < c.readers; r++) th.emplace_back(reader);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < c.seconds)
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  stop.store(true);
  for (auto &t : th) t.join();
  to_launcher.close();
  if (c.launcher) tl.join();
  to_dispatcher.close();
  td.join();
  CK(hipDeviceSynchronize());
  res->dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  CK(hipMemcpy(res->a, cnt_h2d, 64, hipMemcpyDeviceToHost));
  CK(hipMemcpy(res->b, cnt_comp, 64, hipMemcpyDeviceToHost));
  res->handovers = done_gens.load();
  res->slots = K;
}

int main(int argc, char **argv) {
  Cfg c;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i], v = argv[i + 1];
    if (k == "--events") c.events = v;
    else if (k == "--copy") c.copy = v;
    else if (k == "--wgs") c.wgs = atoi(v.c_str());
    else if (k == "--prior-read") c.prior_read = atoi(v.c_str());
    else if (k == "--h2d-check") c.h2d_check = atoi(v.c_str());
    else if (k == "--launcher") c.launcher = atoi(v.c_str());
    else if (k == "--host-confirm") c.host_confirm = atoi(v.c_str());
    else if (k == "--readers") c.readers = atoi(v.c_str());
    else if (k == "--streams") c.streams = atoi(v.c_str());
    else if (k == "--words") c.words = (uint32_t) atol(v.c_str());
    else if (k == "--seconds") c.seconds = atof(v.c_str());
    else if (k == "--no-wait") c.no_wait = atoi(v.c_str());
    else if (k == "--pipelines") c.pipelines = atoi(v.c_str());
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  const size_t bytes = (size_t) c.words * 4;
  const int P = 61;                       // patterns (prime: gen % P differs from (gen - K) % P for the K used)
  CK(hipSetDevice(0));
  std::vector<uint32_t *> pin((size_t) P);
  for (int p = 0; p < P; p++) {
    CK(hipHostMalloc((void **) &pin[(size_t) p], bytes, hipHostMallocPortable));
    for (uint32_t i = 0; i < c.words; i++) pin[(size_t) p][i] = word_of((uint32_t) p, i);
  }
  std::vector<Result> results((size_t) c.pipelines);
  std::vector<std::thread> pipes;
  for (int q = 0; q < c.pipelines; q++) pipes.emplace_back(run_pipeline, std::cref(c), std::cref(pin), P, &results[(size_t) q]);
  for (auto &t : pipes) t.join();
  unsigned long long a[8] = {0}, b[8] = {0}, total = 0;
  double dt = 0;
  for (const Result &r : results) {
    total += r.handovers;
    dt = r.dt > dt ? r.dt : dt;
    a[0] += r.a[0]; a[1] += r.a[1]; b[0] += r.b[0]; b[1] += r.b[1];
    if (r.a[1] && !a[2]) for (int i = 2; i < 6; i++) a[i] = r.a[i];
    if (r.b[1] && !b[2]) for (int i = 2; i < 6; i++) b[i] = r.b[i];
  }
  const int K = results[0].slots;
  printf("{\"events\":\"%s\",\"copy\":\"%s\",\"wgs\":%d,\"prior_read\":%d,\"h2d_check\":%d,\"launcher\":%d,\"host_confirm\":%d,"
         "\"no_wait\":%d,\"pipelines\":%d,\"readers\":%d,\"streams\":%d,\"slot_KiB\":%zu,\"slots\":%d,\"seconds\":%.1f,\"handovers\":%llu,\"per_s\":%.0f,"
         "\"h2d_stream_check\":{\"wrong_words\":%llu,\"launches_wrong\":%llu,\"first\":[%llu,%llu,%llu,%llu]},"
         "\"compute_stream_check\":{\"wrong_words\":%llu,\"launches_wrong\":%llu,\"first\":[%llu,%llu,%llu,%llu]}}\n",
         c.events.c_str(), c.copy.c_str(), c.wgs, c.prior_read, c.h2d_check, c.launcher, c.host_confirm, c.no_wait, c.pipelines, c.readers, c.streams,
         bytes >> 10, K, dt, total, total / dt, a[0], a[1], a[2], a[3], a[4], a[5],
         b[0], b[1], b[2], b[3], b[4], b[5]);
  fflush(stdout);
  return (a[0] || b[0]) ? 1 : 0;
}
