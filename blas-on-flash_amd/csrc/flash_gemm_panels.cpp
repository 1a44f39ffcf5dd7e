// flash_gemm_panels.cpp -- flash::gemm on file-resident matrices through ROW PANELS.
//
// What it replaces: the reference moves every 64 MiB tile as 4096 strided 16 KiB requests
// (one iocb per tile row, src/file_handles/flash_file_handle.cpp:444-460) and packs it into
// its own buffer (src/blas/gemm.cpp:117-120).  But a block row of a stored matrix -- `blk`
// stored rows x the full stored width -- is ONE contiguous extent of the file.  With 288 GB of
// HBM per GPU the natural unit of the program cache is therefore that panel, kept in HBM in
// FILE layout (leading dimension = the file's):
//   * reads and writes are a handful of large sequential requests per panel (io_chunk_mib,
//     default 32 MiB) instead of thousands of row requests per tile; every request lands in a
//     pinned staging slot and crosses PCIe as one linear SDMA copy;
//   * no packing anywhere: a launch is `pointer into a panel + leading dimension` (for a
//     k-contiguous operand: into the panel's k-major copy, made once per panel on the H2D stream --
//     Mat::kmajor_copy);
//   * C panels are written back as they complete, while the next panels compute.
//
// Schedule: a list of LAUNCHES (C panel, k-block range), round 5.  The first `group` C panels are the
// ramp: while the resident operand streams in they run k-block by k-block, l-major -- one launch per
// (panel, k-block) over the whole panel width -- so that panel l of the resident operand unlocks
// `group` launches, and `group` is chosen so that those take as long as the panel's read (65536^3 from
// O_DIRECT files: a 1 GiB B panel takes 55 ms to read, a panel's k-block launch 15 ms).  Every later C
// panel has everything resident but its own X panel and runs as ONE launch over the whole K, and is
// written back when it is done, which keeps the tail after the last kernel to one panel.  A chain
// that runs as several launches hands its RAW fp32 accumulators from launch to launch and scales once
// at the end (gemm_f32_mfma.hip, ChainEpi), so every cut of K gives the bits of a single launch over
// the whole K: the C file does not depend on the tile size, the ramp group or the device list, and
// equals what drivers/in_mem_gemm.cpp:63-70 computes (bof_options.gemm_chain = 1: the reference's task
// arithmetic, one rounding per k-block, src/blas/gemm.cpp:122-127 -- always for flash::kmeans, whose
// launches are the reference's tile tasks).
// Let D be the dimension along which
// C is paneled (m for row-major C, n for column-major).  The operand that does not contain D
// ("Y": B for row-major) is needed whole by every group and stays resident; the other one
// ("X": A) is resident too when it is paneled along k, else its panels stream through a small
// ring, one group ahead.  If that working set does not fit opts->hbm_budget, or C's rows are
// not contiguous in its file (ldc != stored width: writing whole panels would clobber what
// lies between the rows), the call is handed to the tile cache of flash_runtime.cpp.
//
// HBM: a resident operand is ONE allocation holding the matrix in file layout (Mat::whole: a launch
// can then run over any range of its rows with one pointer); the streamed operand and C have one
// allocation per ring slot.  All of it is made in first-use order by an allocator thread while the
// first panels are already being read, and stays with the device's PanelResources between calls.
//
// Threads: n_io_threads readers (file -> pinned slot -> H2D), one dispatcher per device (the
// launches on its compute streams; the caller itself for the first device), one flusher per device
// (HBM -> pinned, chunk by chunk) and writers (pinned -> file).  Everything is ordered by hipEvents
// and one condition variable; nothing polls.  Hand-overs between threads are HOST-CONFIRMED
// (flash_common.h): a reader sees its own copy complete before the chunk counts as delivered, the
// flusher sees a group's kernels complete before it copies, a slot's old occupant is seen gone before
// the copy that refills it is submitted; the device-side event waits stay as a second line.
//
// Several devices in ONE process (bof_options.devices / $BOF_DEVICES; the reference runs its
// N_COMPUTE_THR workers behind flash::gemm inside one process, src/scheduler/scheduler.cpp:9-16):
// the C panels are dealt to the devices in contiguous ranges (output row blocks), and
// every device runs the schedule above on ITS slab -- same k-order per output element, so the
// C file is bit-identical to the single-device call.  A panel only one device needs (its C
// panels, the A panels of its rows) is private to it; a panel every device needs (B for
// row-major; A too when it is paneled along k) is SHARED: it is read from the file ONCE into a
// pinned slot and copied from there to every device on that device's own H2D stream, i.e. over
// its own PCIe link -- storage sees A, B and C once whatever the number of devices.  Readers,
// the read ring and the writers serve all devices; panel slots, copy streams, write ring,
// flusher and dispatcher are per device.
//
// One PROCESS per GPU (bof_options.share_world > 1, bof_dist.flash_gemm_row_sharded): every rank
// makes this call on its slab, and a shared panel is read from the file by ONE rank of the node
// (panel l by rank l % share_world), which publishes its chunks in a node-shared staging ring
// (ShareSeg: a ring of chunk slots in POSIX shared memory, futex words per slot); the other ranks' readers
// wait for the chunk and copy it out of the ring instead of reading the file.  All ranks queue
// the shared panels in the same order and readers take requests in queue order, so the earliest
// unpublished chunk is always being read by its owner: no cycle of waits.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <hip/hip_runtime.h>
#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "flash_common.h"
#include "share_ring.h"

namespace bof {
namespace {

// compute launches of the last bof_flash_gemm that took the row-panel path, by kind (bof_flash_last_launch_mix):
// [0] one k-range of a chain (<ChainEpi>: the ramp group), [1] whole-K launches of a whole C panel, [2] whole-K launches
// of a row slice of a C panel; all devices of the call added up
std::atomic<uint64_t> g_mix[3];

struct Panel {
  int64_t r0 = 0, nr = 0;       // stored rows [r0, r0 + nr)
  uint64_t bytes = 0;           // extent: (nr-1)*ld + cols elements
  int state = 0;                // 0 idle, 1 being read, 2 usable            (guarded by mu)
  int remaining = 0;            // chunks whose H2D copy is not enqueued yet  (guarded by mu)
  bool retired = false;         // consumers launched / write-back enqueued   (guarded by mu)
  hipEvent_t ready = nullptr;   // recorded on the H2D stream behind the panel's last copy
  hipEvent_t d2h_done = nullptr;
  std::vector<hipEvent_t> retire_ev;  // what the slot's next occupant must wait for
  // BOF_VERIFY entries of this panel (Verify::kNone = not checked)
  enum { VE_HOST_IN, VE_DEV_IN, VE_DEV_LAST, VE_RECT_IN, VE_T_IN, VE_T_LAST, VE_C_DEV, VE_C_HOST, VE_C_FILE, VE_N };
  size_t ve[VE_N] = {Verify::kNone, Verify::kNone, Verify::kNone, Verify::kNone, Verify::kNone,
                     Verify::kNone, Verify::kNone, Verify::kNone, Verify::kNone};
  bool v_last_done = false;     // its HBM image has had its after-last-use sum queued (guarded by PanelRun::vf_mu)
  bool poisoned = false;        // BOF_VERIFY: its HBM image was filled with NaN ahead of its first chunk (guarded by vf_mu)
};

struct Mat {
  bof_fptr f{-1, 0};            // element 0 of what THIS device uses of the matrix
  int fd = -1;                  // descriptor every request of this call uses (O_DIRECT or twin)
  bool aio = false;
  int rdim = 0, cdim = 0;
  int64_t rows = 0, cols = 0, ld = 0, blk_r = 0, blk_c = 0;
  bool shared = false;          // every device needs every panel: read once, copied to all of them
  int64_t col_base = 0;         // shared operand whose columns run along the C panel dimension:
                                // first stored column of this device's slab
  std::vector<Panel> panels;
  bool natural = false;         // whole matrix resident at its file offsets
  int n_slots = 0;
  size_t slot_bytes = 0, total_bytes = 0;
  // One HBM allocation per panel slot (resident matrices: one per panel), made in first-use order
  // by a thread of its own while the first panels are already being read: hipMalloc costs 13-36 ms
  // per GiB and used to sit in front of the first read (0.85 s of a cold 65536^3 call).  A null
  // entry = not allocated yet; entries are written under PanelHub::mu.
  std::vector<char *> *slots = nullptr;
  int slot_of(int p) const { return natural ? p : p % n_slots; }
  // A RESIDENT OPERAND (A or B kept whole) is ONE allocation holding the matrix in file layout, panel p at its file
  // offset: a launch may then run over any range of its rows -- the whole K of a k-paneled operand, the whole width of
  // a C panel -- with one pointer and the file's leading dimension (round 5; C and the streamed operand keep one
  // allocation per slot).  Allocated like a slot: by alloc_main, in first-use order, kept between calls.
  bool whole = false;
  char **whole_ptr = nullptr, **twhole_ptr = nullptr;
  char *panel_ptr(int p) const {
    if (whole) return *whole_ptr ? *whole_ptr + (uint64_t) panels[(size_t) p].r0 * (uint64_t) ld * 4 : nullptr;
    return (*slots)[(size_t) slot_of(p)];
  }
  // k-major copy of a k-contiguous operand panel ([rows][k] -> [k][rows], one per slot), made on
  // the H2D stream behind the panel's last copy: the tile tasks then take the LDS-DMA kernel
  // (148.6 instead of 145.7 TFLOP/s at 4096^3) exactly as bof_gemm_resident arranges it for
  // resident operands; same tiles, same k-order, same bits.
  bool kmajor_copy = false;
  size_t tslot_bytes = 0;
  std::vector<char *> *tslots = nullptr;
  // (a resident operand's copy is ONE [cols][rows] image, panel p in its columns r0 .. r0 + nr - 1)
  char *tpanel_ptr(int p) const {
    if (whole) return *twhole_ptr ? *twhole_ptr + (uint64_t) panels[(size_t) p].r0 * 4 : nullptr;
    return (*tslots)[(size_t) slot_of(p)];
  }
  int64_t t_ld(int p) const { return whole ? rows : panels[(size_t) p].nr; }
  bool slot_ready(int p) const { return panel_ptr(p) && (!kmajor_copy || tpanel_ptr(p)); }
  uint64_t file_off(int p) const { return f.foffset + (uint64_t) panels[(size_t) p].r0 * (uint64_t) ld * 4; }
  // O_DIRECT kept although some request of the call is not sector aligned (an unaligned leading dimension
  // or file offset, a file whose size is no multiple of a sector): reads fetch the aligned superset, writes
  // send their whole pages direct and the partial edge pages through the page cache (fileio.cpp).  The
  // chunks of a panel are then cut at page-aligned FILE positions, so only a panel's two ends have edges.
  bool widen = false;
  static constexpr uint64_t kPage = 4096;
  int n_chunks(int p, size_t chunk) const { return (int) ((panels[(size_t) p].bytes + chunk - 1) / chunk); }
  // panel-relative [off, off + len) of chunk c
  void chunk_span(int p, int c, size_t chunk, uint64_t *off, uint64_t *len) const {
    const uint64_t bytes = panels[(size_t) p].bytes, b0 = file_off(p);
    const int n = n_chunks(p, chunk);
    auto start = [&](int i) -> uint64_t {
      if (i <= 0) return 0;
      if (i >= n) return bytes;
      const uint64_t s = (uint64_t) i * chunk;
      return widen ? std::min(bytes, std::max<uint64_t>((b0 + s) / kPage * kPage, b0) - b0) : s;
    };
    *off = start(c);
    *len = start(c + 1) - *off;
  }
};

struct ChunkReq { int di, mat, panel, c; uint64_t off, bytes; };   // di < 0: a shared panel (every device)
// One kernel launch of the schedule: C panel pc, its tiles [q0, q1) along C's other dimension, the k-blocks [l0, l1).
// flash::gemm: a whole C panel per launch -- one k-block at a time while the resident operand streams in (the ramp
// group), the whole K afterwards; flash::kmeans: one tile task per launch, as the reference has them.
// d0 .. d1: the launch's part of its C panel along the panel dimension (stored rows of the panel); d1 == 0: all of it.
// A C panel that runs as ONE launch over the whole K is cut into row slices (PanelRun::slices): slice s is copied out
// and written while slice s + 1 is still being multiplied, so what follows the last kernel of a call is the write-back
// of a quarter panel instead of a whole one -- the tail that a fast disk or the page cache exposes.
struct Launch { int pc; int64_t q0, q1, l0, l1; int64_t d0 = 0, d1 = 0; bool fin = false; };      // fin: PanelRun::slice_end
struct WriteReq { int di, wslot; uint64_t file_off, bytes, delta; int panel; bool last; };

// Per device (and per repetition of one ordinal in the device list): the HBM panel slots and the
// write ring, kept between calls.
struct PanelResources {
  int dev = 0;
  PinnedRing wring;
  std::vector<char *> slot[3];            // kept between calls
  size_t slot_bytes[3] = {0, 0, 0};
  std::vector<char *> tslot[2];           // k-major copies of operand panels (A, B)
  size_t tslot_bytes[2] = {0, 0};
  char *whole[2] = {nullptr, nullptr};    // resident operands: the whole matrix in file layout, and its k-major copy
  size_t whole_bytes[2] = {0, 0};
  char *twhole[2] = {nullptr, nullptr};
  size_t twhole_bytes[2] = {0, 0};
  std::vector<char *> acc;                // raw accumulator panels of the ramp group's chains (beta != 0), C slot size
  size_t acc_bytes = 0;
  void drop_whole(int x) {
    if (whole[x]) (void) hipFree(whole[x]);
    whole[x] = nullptr;
    whole_bytes[x] = 0;
  }
  void drop_twhole(int x) {
    if (twhole[x]) (void) hipFree(twhole[x]);
    twhole[x] = nullptr;
    twhole_bytes[x] = 0;
  }
  void drop_acc(size_t keep) {
    for (size_t i = keep; i < acc.size(); i++)
      if (acc[i]) (void) hipFree(acc[i]);
    acc.resize(keep);
  }
  size_t held_bytes() const {
    size_t tot = 0;
    for (int x = 0; x < 2; x++) tot += (whole[x] ? whole_bytes[x] : 0) + (twhole[x] ? twhole_bytes[x] : 0);
    for (char *p : acc)
      if (p) tot += acc_bytes;
    for (int x = 0; x < 3; x++)
      for (char *p : slot[x])
        if (p) tot += slot_bytes[x];
    for (int x = 0; x < 2; x++)
      for (char *p : tslot[x])
        if (p) tot += tslot_bytes[x];
    return tot;
  }
  void drop(int x, size_t keep) {         // free the slots of matrix x from index `keep` on
    for (size_t i = keep; i < slot[x].size(); i++)
      if (slot[x][i]) (void) hipFree(slot[x][i]);
    slot[x].resize(keep);
  }
  void drop_t(int x, size_t keep) {
    for (size_t i = keep; i < tslot[x].size(); i++)
      if (tslot[x][i]) (void) hipFree(tslot[x][i]);
    tslot[x].resize(keep);
  }
  // events and copy streams are kept between calls like the slots (a call creates and destroys no HIP object in
  // the steady state: see GemmResources in flash_runtime.cpp and profiles/r4/fuzz_crash.md)
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
  hipStream_t h2d = nullptr, d2h = nullptr;
  int take_event(hipEvent_t *e) {
    if (ev_next == ev_pool.size()) {
      hipEvent_t n = nullptr;
      BOF_HIP_TRY(hipEventCreateWithFlags(&n, pooled_event_flags()));
      ev_pool.push_back(n);
    }
    *e = ev_pool[ev_next++];
    return BOF_OK;
  }
  void drop_all() {
    DeviceScope ds(dev);
    for (hipEvent_t e : ev_pool) (void) hipEventDestroy(e);
    ev_pool.clear();
    ev_next = 0;
    if (h2d) (void) hipStreamDestroy(h2d);
    if (d2h) (void) hipStreamDestroy(d2h);
    h2d = d2h = nullptr;
    wring.destroy();
    for (int x = 0; x < 3; x++) drop(x, 0);
    for (int x = 0; x < 2; x++) { drop_t(x, 0); drop_whole(x); drop_twhole(x); }
    drop_acc(0);
  }
};
std::mutex g_pres_mu;
std::map<std::pair<int, int>, PanelResources *> g_pres;   // (ordinal, repetition) -> resources
// The read ring serves every device of a call; kept per FIRST (lowest) ordinal of the call's device list: a call
// holds the call locks of all its devices, so two calls that run at the same time (disjoint device lists) never
// share one.
std::map<int, PinnedRing *> g_rring;

int trace_level() {   // BOF_TRACE=1: dispatcher milestones; 2: + every panel read / flush / write
  static const int lvl = getenv("BOF_TRACE") ? std::max(1, atoi(getenv("BOF_TRACE"))) : 0;
  return lvl;
}
bool trace_on() { return trace_level() >= 1; }

// A shared operand's way between the ranks of a share_world > 1 call: the staging ring of share_ring.h plus
// the place of every panel's chunks in the operand's chunk order.
struct ShareSeg {
  ShareRing ring;
  size_t n_chunks = 0;
  std::vector<size_t> first_chunk;    // per panel: index of its first chunk in the operand's chunk order
};

struct PanelHub;

// One device's share of the call: the single-device schedule over its slab of C.
struct PanelRun {
  PanelHub *hub = nullptr;
  int di = 0, dev = 0;          // position in the hub's device list, HIP ordinal
  bof_options o;
  GemmGeometry g;               // the slab's geometry (size along the C panel dimension cut down)
  char ord = 'R', ta = 'N', tb = 'N';
  float alpha = 1.f, beta = 0.f;
  Mat mat[3];
  int xmat = 0, ymat = 1;       // streamed-or-resident operand / always-resident operand
  int dC = 0;
  int64_t NpC = 0, Nq = 0;
  int64_t row_base = 0, col_base = 0;   // first row / column of the slab in the whole C (kmeans vectors)
  bool c_read = false;
  size_t chunk = 32u << 20;
  PanelResources *res = nullptr;
  hipStream_t h2d = nullptr, d2h = nullptr;
  StreamSet *ss = nullptr;
  WorkQueue<int> flush_q;
  std::vector<std::pair<int, int>> order;        // (mat, panel) in order of first use
  std::vector<std::pair<int, int>> alloc_order;  // (mat, slot) still to be allocated, in order of first use
  size_t next_fetch = 0;
  std::vector<Launch> launches;                   // execution order
  std::vector<size_t> group_end;                  // launch index one past each group
  // How a C tile's k-blocks are combined (bof_options.gemm_chain).  false (default): the chain carries its raw
  // accumulators from launch to launch and scales once at the end -- ONE k-ordered fmaf chain per element however K is
  // cut (gemm_f32_mfma.hip, ChainEpi), which is also what lets a C panel whose operands are complete run as a single
  // launch over the whole K.  true: the reference's task arithmetic, C = alpha*A_l*B_l + (l ? 1 : beta)*C per k-block
  // (src/blas/gemm.cpp:122-127, include/tasks/gemm_task.h:87-90) -- also taken for alpha == 0 (cblas_sgemm's quick
  // return: neither operand may be looked at) and by flash::kmeans, whose tasks add their rank-1 terms per block.
  bool ref_chain = false;
  bool need_acc = false;                          // beta != 0 and a chain of several launches: raw sums in res->acc
  std::vector<int64_t> gb;                        // first C panel of each group, then NpC
  int n_groups = 0;
  std::vector<std::vector<hipEvent_t>> group_ev;  // per group: one event per compute stream
  // row slices of the C panels that run as one launch over the whole K: per C panel the slices' last rows and the
  // events recorded behind their launches (empty: the panel is not sliced, the flusher waits for group_ev).  A panel
  // of the ramp group is entered as ONE "slice" of all its rows whose event stands behind the last launch of its chain
  // (Launch::fin), so that it leaves when IT is complete and not when the whole group is ($BOF_PANEL_RAMP_FLUSH).
  std::vector<std::vector<int64_t>> slice_end;
  std::vector<std::vector<hipEvent_t>> slice_ev;
  std::vector<int> group_of;                      // C panel -> group
  KmeansVecs kv{nullptr, nullptr, nullptr};
  bool has_kv = false;
  bool kmeans = false;                            // the call is flash::kmeans (known before the vectors are uploaded)
  Counters cnt;                                   // this device's share of the counters
  KernelTimer ktimer;                             // bof_options.kernel_timing
  Verify vf;                                      // bof_options.verify: hand-over checksums of this device's panels
  std::mutex vf_mu;                               // orders "sum the slot's old panel" before the first copy of the new one
  int verify_setup();
  hipError_t verify_last_use(int x, int p, hipStream_t st);   // caller holds vf_mu
  int verify_finish();
  hipError_t herr = hipSuccess;
  int fail = 0;
  double seconds = 0;

  int plan(const std::vector<int> &all_devs, int reps_of_dev, int64_t full_dC, const bof_fptr fp[3], const GemmGeometry &gfull,
           int64_t p_first);
  int prepare();
  void alloc_main();
  void flusher_main();
  void dispatch();
  void trace(const char *label) const;
  void trace2(const char *what, int x, int p) const {
    if (trace_level() >= 2) {
      char lbl[64];
      snprintf(lbl, sizeof(lbl), "d%d %s %c%d", di, what, "ABC"[x], p);
      trace(lbl);
    }
  }
};

// What the devices of one call share: the fetch / write queues with their thread pools, the read
// ring, the error flag and the one mutex + condition variable everything is ordered by.
struct PanelHub {
  std::vector<std::unique_ptr<PanelRun>> runs;
  PinnedRing *rring = nullptr;
  WorkQueue<ChunkReq> fetch_q;
  WorkQueue<WriteReq> write_q;
  std::vector<char> issued[2];   // shared operand panels whose read has been queued
  ShareSeg seg[2];               // share_world > 1: the shared operands' node-wide staging
  int share_world = 1, share_rank = 0;
  bool peer_bcast = false;       // bof_options.peer_bcast: shared panels device to device
  // $BOF_PANEL_WRITES_AFTER_READS=1 (experiment, round 5): the writers hold C's file writes back until the call's last
  // read request has completed.  On a disk whose mixed read + write rate is BELOW its pure rates (the scratch disk of
  // some leases: reads 21 -> 13 GB/s with writes going at 3) a read phase at the full rate followed by a write phase
  // at the full rate is shorter than the two mixed; only when every C panel has an HBM slot of its own (nothing
  // computes into a slot whose write-back it would have to wait for).
  bool writes_after_reads = false;
  std::atomic<uint64_t> chunks_total{0}, chunks_done{0};
  double share_timeout_s = 120;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> io_error{0};
  Counters cnt;
  std::chrono::steady_clock::time_point t_begin;

  void trace(const char *label) const {
    if (trace_on())
      fprintf(stderr, "[bof trace] %-34s %8.3f ms\n", label,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  }
  void fail_io(int code) {
    {
      std::lock_guard<std::mutex> lk(mu);
      int none = 0;
      io_error.compare_exchange_strong(none, code);
    }
    cv.notify_all();
  }

  // One step of a device's fetch list: queue the chunk requests of its next panel, in first-use
  // order, if the HBM slot is free -- the whole-matrix images always are, a ring slot once its
  // previous occupant has retired.  Strictly in order per device, so the readers always work on
  // what that device needs soonest.  A shared panel is queued once, by whichever device gets to
  // it first, when EVERY device has its slot allocated.  Caller holds mu.
  bool pump_step(PanelRun &R) {
    if (R.next_fetch >= R.order.size()) return false;
    const int x = R.order[R.next_fetch].first, p = R.order[R.next_fetch].second;
    Mat &M = R.mat[x];
    Panel &P = M.panels[(size_t) p];
    const int n_chunks = M.n_chunks(p, R.chunk);
    auto push_chunks = [&](int di) {
      for (int c = 0; c < n_chunks; c++) {
        uint64_t off, len;
        M.chunk_span(p, c, R.chunk, &off, &len);
        fetch_q.push(ChunkReq{di, x, p, c, off, len});
      }
    };
    if (M.shared) {
      if (!issued[x][(size_t) p]) {
        for (auto &Q : runs)
          if (!Q->mat[x].slot_ready(p)) return false;     // a slot is still being allocated (alloc_main pumps again)
        issued[x][(size_t) p] = 1;
        for (auto &Q : runs) {
          Panel &PQ = Q->mat[x].panels[(size_t) p];
          PQ.state = 1;
          PQ.remaining = n_chunks;
        }
        push_chunks(-1);
        cnt.misses++;
      }
      R.next_fetch++;
      return true;
    }
    const int prev = M.natural ? -1 : p - M.n_slots;
    if (prev >= 0 && !M.panels[(size_t) prev].retired) return false;
    if (!M.slot_ready(p)) return false;
    P.state = 1;
    P.remaining = n_chunks;
    push_chunks(R.di);
    cnt.misses++;
    R.next_fetch++;
    return true;
  }
  void pump_fetches() {   // round robin over the devices, one panel each, until nobody can go on
    for (bool progress = true; progress;) {
      progress = false;
      for (auto &R : runs) progress = pump_step(*R) || progress;
    }
  }

  void reader_main(int home);
  void writer_main(int home);
};

void PanelRun::trace(const char *label) const {
  if (!trace_on()) return;
  char lbl[96];
  snprintf(lbl, sizeof(lbl), "d%d %s", di, label);
  hub->trace(lbl);
}

// HBM slots in first-use order; every new slot may unblock the next fetch / the dispatcher
void PanelRun::alloc_main() {
  (void) hipSetDevice(dev);
  for (const auto &as : alloc_order) {
    if (hub->io_error.load()) break;
    TraceRange r("panel slot hipMalloc");
    char *p = nullptr, *tp = nullptr;
    Mat &M = mat[as.first];
    hipError_t e = hipSuccess;
    if (M.whole) {     // (as.second = -1) the resident operand's one image, and its k-major copy
      if (!*M.whole_ptr) e = hipMalloc((void **) &p, res->whole_bytes[as.first]);
      if (e == hipSuccess && M.kmajor_copy && !*M.twhole_ptr) e = hipMalloc((void **) &tp, res->twhole_bytes[as.first]);
    } else {
      if (!(*M.slots)[(size_t) as.second]) e = hipMalloc((void **) &p, M.slot_bytes);
      if (e == hipSuccess && M.kmajor_copy && !(*M.tslots)[(size_t) as.second]) e = hipMalloc((void **) &tp, M.tslot_bytes);
    }
    if (e != hipSuccess) {
      (void) hipGetLastError();
      if (p) (void) hipFree(p);
      hub->fail_io(e == hipErrorOutOfMemory ? -ENOMEM : -1000 - (int) e);
      break;
    }
    {
      std::lock_guard<std::mutex> lk(hub->mu);
      // (the copy before the raw image: a usable raw image implies its copy's)
      if (M.whole) {
        if (tp) *M.twhole_ptr = tp;
        if (p) *M.whole_ptr = p;
      } else {
        if (tp) (*M.tslots)[(size_t) as.second] = tp;
        if (p) (*M.slots)[(size_t) as.second] = p;
      }
      hub->pump_fetches();
    }
    hub->cv.notify_all();
  }
  trace("HBM panel slots allocated");
}

// file -> pinned slot -> HBM of the device(s) that need the chunk
void PanelHub::reader_main(int home) {
  PanelRun &H = *runs[(size_t) home % runs.size()];
  (void) hipSetDevice(H.dev);
  (void) bind_thread_near_device(H.dev);
  ChunkReq rq;
  while (fetch_q.pop(rq)) {
    PanelRun &R0 = *runs[(size_t) std::max(rq.di, 0)];
    Mat &M0 = R0.mat[rq.mat];
    const int ps = rring->acquire();
    int rc = 0;
    ShareSeg *sg = rq.di < 0 && share_world > 1 && seg[rq.mat].ring.base ? &seg[rq.mat] : nullptr;
    const size_t ci = sg ? sg->first_chunk[(size_t) rq.panel] + (size_t) rq.c : 0;     // the chunk's index in the operand
    const bool from_peer = sg && rq.panel % share_world != share_rank;
    uint64_t delta = 0;                 // where the chunk's first byte sits in the pinned slot
    char *const slot = (char *) rring->ptr(ps);
    if (from_peer) {
      // another rank of the node reads this panel from the file: its chunk, out of the staging ring
      if (!io_error.load()) {
        TraceRange r("panel chunk from a peer");
        rc = sg->ring.consume(ci, slot, rq.bytes, share_timeout_s, io_error);
        if (rc == -ECANCELED) rc = 0;    // this call is failing for another reason already
      }
      if (rc) fail_io(rc);
      cnt.peer += rq.bytes;
    } else {
      if (!io_error.load()) {
        TraceRange r("panel chunk read");
        evt("panel chunk read begin", rq.mat, rq.panel, (uint64_t) rq.c);
        if (M0.widen) rc = file_read_widened(M0.fd, M0.file_off(rq.panel) + rq.off, rq.bytes, slot, &delta, M0.aio);
        else rc = file_sread(M0.fd, M0.file_off(rq.panel) + rq.off, 0, 1, rq.bytes, slot, M0.aio);
      }
      if (sg) {   // publish (or tell the peers that it will not come)
        const bool ok = !rc && !io_error.load();
        const int prc = sg->ring.produce(ci, ok ? slot + delta : nullptr, rq.bytes, share_world, share_timeout_s, io_error);
        if (prc && prc != -ECANCELED && !rc) rc = prc;
      }
      evt("panel chunk read end", rq.mat, rq.panel, (uint64_t) rq.c);
      if (rc) fail_io(rc);
      cnt.rd += rq.bytes;
      if (rq.di >= 0) R0.cnt.rd += rq.bytes;
    }
    const size_t d0 = rq.di < 0 ? 0 : (size_t) rq.di, d1 = rq.di < 0 ? runs.size() : (size_t) rq.di + 1;
    hipError_t e = hipSuccess;
    // Shared chunk with bof_options.peer_bcast: over PCIe to ONE device only -- the panel's home, dealt round robin --
    // and from its HBM to the others device to device (hipMemcpyPeerAsync: xGMI on the 8-GPU node), each on the
    // receiving device's own copy stream behind the home copy's event.  The host then feeds every shared byte
    // once instead of once per device.  Resident (never refilled) panels only: the home image is the source.
    const bool bcast = peer_bcast && rq.di < 0 && runs.size() > 1 && runs[0]->mat[rq.mat].natural;
    const size_t home = bcast ? (size_t) rq.panel % runs.size() : 0;
    for (size_t pass = 0; pass < (bcast ? 2u : 1u) && e == hipSuccess; pass++)
    for (size_t d = d0; d < d1 && e == hipSuccess; d++) {
      if (bcast && ((pass == 0) != (d == home))) continue;       // pass 0: the home device; pass 1: its peers
      PanelRun &R = *runs[d];
      Mat &M = R.mat[rq.mat];
      e = hipSetDevice(R.dev);
      const int prev = M.natural ? -1 : rq.panel - M.n_slots;
      if (prev >= 0)  // WAR: the slot's previous occupant (its events were recorded before it retired)
        for (hipEvent_t w : M.panels[(size_t) prev].retire_ev)
          if (e == hipSuccess) e = wait_event_both(R.h2d, w);
      std::unique_lock<std::mutex> vlk(R.vf_mu, std::defer_lock);
      if (R.vf.on && !rc) {
        R.vf.on_host(M.panels[(size_t) rq.panel].ve[Panel::VE_HOST_IN], slot + delta, 1, (int64_t) (rq.bytes / 4), 0, rq.off / 4);
        // self-test of the instrumentation ($BOF_VERIFY_INJECT=1): damage one word between the sum and the copy
        if (rq.mat == 1 && rq.panel == 0 && rq.c == 0 && env_long("BOF_VERIFY_INJECT", 0) == 1) ((uint32_t *) (slot + delta))[3] ^= 0x00400000u;
        // the slot's old panel is summed once more behind its last kernel and BEFORE any chunk of the new one
        // lands (whichever reader comes first queues it; the lock keeps its copy behind the sum)
        vlk.lock();
        if (prev >= 0 && e == hipSuccess) e = R.verify_last_use(rq.mat, prev, R.h2d);
        // ... and the image is poisoned (0xFF words: NaN) before the first byte of the new panel: whoever reads it
        // ahead of its fill -- a missing wait, a stale pointer -- gets NaN instead of a plausible old number
        Panel &PN = M.panels[(size_t) rq.panel];
        if (!PN.poisoned && e == hipSuccess) {
          PN.poisoned = true;
          e = R.vf.poison(M.panel_ptr(rq.panel), PN.bytes, R.h2d);
          if (e == hipSuccess && M.kmajor_copy && !M.whole) e = R.vf.poison(M.tpanel_ptr(rq.panel), (size_t) PN.nr * (size_t) M.cols * 4, R.h2d);
          // (the sum of the old panel and the poison were queued by the launcher thread, the copy below comes from
          //  this one: host-confirmed so that the copy cannot overtake them)
          if (e == hipSuccess && host_handover()) e = hipStreamSynchronize(R.h2d);
        }
      }
      if (bcast && d != home) {
        PanelRun &Hm = *runs[home];
        if (e == hipSuccess) e = wait_event_both(R.h2d, rring->event(ps, Hm.di));     // the home copy of this chunk
        if (e == hipSuccess && !rc)
          e = hipMemcpyPeerAsync(M.panel_ptr(rq.panel) + rq.off, R.dev, Hm.mat[rq.mat].panel_ptr(rq.panel) + rq.off, Hm.dev,
                                 rq.bytes, R.h2d);
        if (e == hipSuccess && rring->mark_busy(ps, R.h2d, R.di)) e = hipErrorUnknown;   // (an event behind the peer copy: confirmed below)
        cnt.p2p += rq.bytes;
        R.cnt.p2p += rq.bytes;
      } else {
        if (e == hipSuccess && !rc)
          e = hipMemcpyAsync(M.panel_ptr(rq.panel) + rq.off, slot + delta, rq.bytes, hipMemcpyHostToDevice, R.h2d);
        if (e == hipSuccess && rring->mark_busy(ps, R.h2d, R.di)) e = hipErrorUnknown;   // or the slot would be refilled under the copy
        cnt.h2d += rq.bytes;
        R.cnt.h2d += rq.bytes;
      }
      if (vlk.owns_lock()) vlk.unlock();
      // host-confirmed hand-over (flash_common.h): THIS thread saw its own copy of the chunk complete on this device
      // before the chunk counts as delivered (the home copy of a broadcast chunk before its peers' copies start)
      if (host_handover() && e == hipSuccess && !rc && bcast && pass == 0) e = hipEventSynchronize(rring->event(ps, R.di));
    }
    if (host_handover() && e == hipSuccess && !rc)       // (all devices' copies are in flight; now each is confirmed)
      for (size_t d = d0; d < d1 && e == hipSuccess; d++)
        if (!(bcast && d == home)) e = hipEventSynchronize(rring->event(ps, runs[d]->di));
    rring->release(ps);
    chunks_done++;
    // the chunk is delivered: whoever brings a panel's count to zero on a device finishes the panel there -- the
    // kernels behind its last copy (k-major copy, BOF_VERIFY's sums), `ready`, and (host-confirmed hand-over) the
    // wait for all of that -- OUTSIDE the hub's mutex, and only then shows the panel to the dispatcher
    std::vector<size_t> finish;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t d = d0; d < d1; d++)
        if (--runs[d]->mat[rq.mat].panels[(size_t) rq.panel].remaining == 0) finish.push_back(d);
    }
    for (size_t d : finish) {
      PanelRun &R = *runs[d];
      Mat &M = R.mat[rq.mat];
      Panel &P = M.panels[(size_t) rq.panel];
      if (e == hipSuccess) e = hipSetDevice(R.dev);
      // (launched by a PERSISTENT launcher thread, not by this reader, which was created for the call: flash_common.h,
      //  "persistent launcher threads"; this thread waits for the launches to be queued, then records `ready` behind them)
      if (e == hipSuccess) e = R.vf.on_device(P.ve[Panel::VE_DEV_IN], M.panel_ptr(rq.panel), 1, (int64_t) (P.bytes / 4), 0, 0, 0, R.h2d);
      if (e == hipSuccess && M.kmajor_copy) {
        e = R.vf.on_device(P.ve[Panel::VE_RECT_IN], M.panel_ptr(rq.panel), P.nr, M.cols, M.ld, 0, 0, R.h2d);
        if (e == hipSuccess)
          e = launch_from_persistent(R.dev, [&] {
            return transpose_f32((const float *) M.panel_ptr(rq.panel), M.ld, P.nr, M.cols, (float *) M.tpanel_ptr(rq.panel),
                                 M.t_ld(rq.panel), R.h2d);
          });
        if (e == hipSuccess)
          e = R.vf.on_device(P.ve[Panel::VE_T_IN], M.tpanel_ptr(rq.panel), M.cols, P.nr, M.t_ld(rq.panel), 0, M.cols, R.h2d);
      }
      if (e == hipSuccess) e = hipEventRecord(P.ready, R.h2d);
      if (e == hipSuccess && host_handover() && (M.kmajor_copy || R.vf.on)) e = hipEventSynchronize(P.ready);
      {
        std::lock_guard<std::mutex> lk(mu);
        // a failed copy / record must be visible BEFORE the panel is: the dispatcher tests io_error right
        // after it sees state 2, and a `ready` that was never recorded would make its wait a no-op
        if (e != hipSuccess || rc) { int none = 0; io_error.compare_exchange_strong(none, rc ? rc : -1000 - (int) e); }
        P.state = 2;
      }
      evt("panel in HBM (ready recorded)", rq.mat, rq.panel, (uint64_t) R.di);
      R.trace2("read + H2D done:", rq.mat, rq.panel);
    }
    if (e != hipSuccess) fail_io(-1000 - (int) e);
    cv.notify_all();
  }
}

// HBM -> pinned ring, chunk by chunk, for every finished C panel; then the panel's slot is
// free for a later one.
void PanelRun::flusher_main() {
  (void) hipSetDevice(dev);
  (void) bind_thread_near_device(dev);
  int pc;
  while (flush_q.pop(pc)) {
    Mat &C = mat[2];
    Panel &P = C.panels[(size_t) pc];
    hipError_t e = hipSuccess;
    evt("C panel handed to the flusher", pc, di);
    // a panel multiplied in row slices leaves slice by slice: a chunk waits for the slices that cover its rows only
    const std::vector<int64_t> &send = slice_end[(size_t) pc];
    size_t slices_waited = 0;
    auto wait_slices_through = [&](uint64_t last_byte) {       // every slice that holds a byte <= last_byte of the panel
      const int64_t last_row = (int64_t) (last_byte / ((uint64_t) C.ld * 4));
      while (e == hipSuccess && slices_waited < send.size() && (slices_waited == 0 || send[slices_waited - 1] <= last_row)) {
        e = wait_event_both(d2h, slice_ev[(size_t) pc][slices_waited]);      // (the dispatcher recorded it; this thread copies)
        slices_waited++;
      }
    };
    if (send.empty() || vf.on) {
      for (hipEvent_t w : group_ev[(size_t) group_of[(size_t) pc]])
        if (e == hipSuccess) e = wait_event_both(d2h, w);      // (the dispatcher recorded them; this thread copies)
      slices_waited = send.size();
    }
    if (e == hipSuccess) e = vf.on_device(P.ve[Panel::VE_C_DEV], C.panel_ptr(pc), 1, (int64_t) (P.bytes / 4), 0, 0, 0, d2h);
    const int nc = C.n_chunks(pc, chunk);
    for (int c = 0; c < nc && e == hipSuccess && !hub->io_error.load(); c++) {
      uint64_t off, len;
      C.chunk_span(pc, c, chunk, &off, &len);
      wait_slices_through(off + len - 1);
      if (e != hipSuccess) break;
      // a widened C panel lands in the pinned slot at its file offset modulo the page (file_write_split)
      const uint64_t delta = C.widen ? (C.file_off(pc) + off) % Mat::kPage : 0;
      const int ws = res->wring.acquire();
      evt("C chunk D2H queued", pc, c, (uint64_t) ws);
      e = hipMemcpyAsync((char *) res->wring.ptr(ws) + delta, C.panel_ptr(pc) + off, len, hipMemcpyDeviceToHost, d2h);
      if (e == hipSuccess) e = hipEventRecord(res->wring.event(ws), d2h);
      if (e != hipSuccess) { res->wring.release(ws); break; }
      cnt.d2h += len;
      hub->cnt.d2h += len;
      hub->write_q.push(WriteReq{di, ws, C.file_off(pc) + off, len, delta, pc, c == nc - 1});
    }
    if (e == hipSuccess) e = hipEventRecord(P.d2h_done, d2h);
    if (e != hipSuccess) hub->fail_io(-1000 - (int) e);
    trace2("D2H queued:", 2, pc);
    {
      std::lock_guard<std::mutex> lk(hub->mu);
      P.retire_ev.assign(1, P.d2h_done);
      P.retired = true;
      hub->pump_fetches();
    }
    hub->cv.notify_all();
  }
}

void PanelHub::writer_main(int home) {
  PanelRun &H = *runs[(size_t) home % runs.size()];
  (void) hipSetDevice(H.dev);
  (void) bind_thread_near_device(H.dev);
  WriteReq rq;
  while (write_q.pop(rq)) {
    PanelRun &R = *runs[(size_t) rq.di];
    if (writes_after_reads) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return io_error.load() || chunks_done.load() >= chunks_total.load(); });
    }
    hipError_t e = hipEventSynchronize(R.res->wring.event(rq.wslot));
    if (e != hipSuccess) fail_io(-1000 - (int) e);
    evt("C chunk D2H complete, write begin", rq.panel, rq.wslot, rq.file_off >> 20);
    if (R.vf.on && e == hipSuccess)
      R.vf.on_host(R.mat[2].panels[(size_t) rq.panel].ve[Panel::VE_C_HOST], (char *) R.res->wring.ptr(rq.wslot) + rq.delta, 1,
                   (int64_t) (rq.bytes / 4), 0, (rq.file_off - R.mat[2].file_off(rq.panel)) / 4);
    if (R.vf.on && rq.panel == 0 && rq.file_off == R.mat[2].file_off(0) && env_long("BOF_VERIFY_INJECT", 0) == 2)
      ((uint32_t *) ((char *) R.res->wring.ptr(rq.wslot) + rq.delta))[5] ^= 0x00400000u;      // ($BOF_VERIFY_INJECT=2)
    int rc = 0;
    if (!io_error.load()) {
      TraceRange r("panel chunk write");
      if (R.mat[2].widen)
        rc = file_write_split(R.mat[2].fd, rq.file_off, rq.bytes, (char *) R.res->wring.ptr(rq.wslot) + rq.delta, R.mat[2].aio);
      else
        rc = file_swrite(R.mat[2].fd, rq.file_off, 0, 1, rq.bytes, R.res->wring.ptr(rq.wslot), R.mat[2].aio);
    }
    evt("C chunk write end", rq.panel, rq.wslot, rq.file_off >> 20);
    if (rc) fail_io(rc);
    cnt.wr += rq.bytes;
    R.cnt.wr += rq.bytes;
    R.res->wring.release(rq.wslot);
    if (rq.last) R.trace2("last chunk written:", 2, rq.panel);
  }
}

// One descriptor mode per file per call: O_DIRECT only if EVERY request of the call is sector
// aligned; otherwise every request goes through the buffered twin.  Mixing the two on one file
// lets a direct write and a buffered write of neighbouring regions meet in one page.
bool requests_aligned(const Mat &M, size_t chunk) {
  const uint64_t A = file_is_direct(M.f.fd) ? file_dio_align(M.f.fd) : 512;
  bool aligned = (M.f.foffset % A) == 0 && (chunk % A) == 0;
  for (const Panel &P : M.panels)
    aligned = aligned && (P.bytes % A) == 0 && (((uint64_t) P.r0 * (uint64_t) M.ld * 4) % A) == 0;
  return aligned;
}
void pick_descriptor(Mat &M, bool aligned, bool use_odirect) {
  M.fd = M.f.fd;
  M.aio = false;
  M.widen = false;
  if (file_is_direct(M.f.fd)) {
    if (aligned && use_odirect) M.aio = true;
    else if (aligned) M.aio = false;               // direct descriptor, synchronous requests
    else if (use_odirect && file_dio_align(M.f.fd) <= Mat::kPage && env_long("BOF_UNALIGNED_DIRECT", 1) != 0) {
      M.aio = true;                                // O_DIRECT kept: widened reads, page-split writes
      M.widen = true;
    } else M.fd = file_buffered_fd(M.f.fd);
  }
}

// ---- plan of one device's slab ------------------------------------------------------------------
// gfull: the whole problem; p_first: first C panel of the slab (this->NpC panels); budget from the
// device's free HBM (divided by the number of times the ordinal appears in the device list).
// Returns BOF_OK, +1 (not eligible: the tile cache must take the call) or an error.
int PanelRun::plan(const std::vector<int> &all_devs, int reps_of_dev, int64_t full_dC, const bof_fptr fp[3],
                   const GemmGeometry &gfull, int64_t p_first) {
  (void) all_devs;
  const int64_t Nk = g.nblk[1];
  chunk = (size_t) std::max(1, o.io_chunk_mib) << 20;
  c_read = beta != 0.0f;
  size_t free_b = 0, total_b = 0;
  BOF_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  free_b += res->held_bytes();  // what we already hold counts as free
  free_b /= (size_t) std::max(1, reps_of_dev);
  size_t budget = o.hbm_budget > 0 ? (size_t) o.hbm_budget : (size_t) (free_b * 0.8);
  budget = std::min(budget, (size_t) (free_b * 0.95));
  dC = g.rdim[2];                              // 0: C paneled along m, 2: along n
  xmat = dC == 0 ? 0 : 1;
  ymat = 1 - xmat;
  NpC = g.nblk[dC];
  Nq = dC == 0 ? g.nblk[2] : g.nblk[0];        // C tiles per panel
  // (bof_options.gemm_chain / PanelRun::ref_chain: see the struct)
  ref_chain = kmeans || o.gemm_chain == 1 || alpha == 0.0f;
  // raw accumulator panels: only a chain of several launches on a C that is read (beta != 0) needs them -- one per C
  // panel of the ramp group, of a C slot's size
  need_acc = !ref_chain && c_read && Nk > 1;
  auto fits = [&](const bof_panel_plan &p2) { return p2.eligible != 0; };      // (plan_panels counts the accumulator panels in)
  bof_panel_plan pl = plan_panels(g, budget, 1, full_dC, need_acc);
  if (!fits(pl)) return 1;
  {
    // size of the ramp group (see the header): read time of one panel of the resident operand
    // over the time of the tile tasks one C panel contributes per such panel.  The rates are
    // assumptions (a datacenter NVMe under O_DIRECT; page cache + PCIe otherwise; the fp32 MFMA
    // tile rate) -- bof_options.panel_group / BOF_PANEL_GROUP override.
    int64_t want = o.panel_group > 0 ? o.panel_group : env_long("BOF_PANEL_GROUP", 0);
    if (want <= 0) {
      const double t_io = (double) pl.slot_bytes[ymat] / (o.use_odirect ? 16e9 : 40e9);
      const double t_task = std::max(2.0 * (double) g.blk[0] * (double) g.blk[1] * (double) g.blk[2] / 140e12, 15e-6);
      want = (int64_t) std::ceil(t_io / (t_task * (double) Nq));
      // at most half of the C panels: a ramp group's panels all complete -- and start their
      // write-back -- together at its end, and on a device-bound problem (cfg2 from O_DIRECT
      // files: 0.70 s with 2-4 of 8 panels, 0.74-0.80 with 5, 0.76 with 1) that burst must not
      // wait for most of the reads
      want = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, 8), NpC / 2));
    }
    for (int64_t G = std::min(want, NpC); G > 1; G--) {
      const bof_panel_plan p2 = plan_panels(g, budget, G, full_dC, need_acc);
      if (fits(p2)) { pl = p2; break; }
    }
  }
  const int64_t group = pl.first_group;
  const int64_t slab0 = p_first * gfull.blk[dC];     // first element of the slab along dC
  row_base = dC == 0 ? slab0 : 0;
  col_base = dC == 2 ? slab0 : 0;
  for (int x = 0; x < 3; x++) {
    Mat &M = mat[x];
    M.f = fp[x];
    M.rdim = g.rdim[x]; M.cdim = g.cdim[x];
    M.ld = g.ld[x];
    M.blk_r = g.blk[M.rdim]; M.blk_c = g.blk[M.cdim];
    M.rows = g.size[M.rdim]; M.cols = g.size[M.cdim];
    // a matrix that does not contain the C panel dimension (Y) or contains it along its stored
    // columns (X paneled along k) is needed by every device panel by panel
    M.shared = x < 2 && M.rdim != dC;
    if (x == 2 || M.rdim == dC) {
      M.f.foffset += (uint64_t) slab0 * (uint64_t) M.ld * 4;      // this device's stored rows
    } else if (M.cdim == dC) {
      M.cols = gfull.size[dC];                                    // whole panels; the slab is a column range
      M.blk_c = gfull.blk[dC];
      M.col_base = slab0;
    }
    M.panels.resize((size_t) pl.n_panels[x]);
    for (int64_t p = 0; p < pl.n_panels[x]; p++) {
      Panel &P = M.panels[(size_t) p];
      P.r0 = p * M.blk_r;
      P.nr = (p == pl.n_panels[x] - 1) ? M.rows - P.r0 : M.blk_r;
      P.bytes = ((uint64_t) (P.nr - 1) * (uint64_t) M.ld + (uint64_t) M.cols) * 4;
    }
    M.slot_bytes = (size_t) pl.slot_bytes[x];
    M.total_bytes = (size_t) (((uint64_t) (M.rows - 1) * (uint64_t) M.ld + (uint64_t) M.cols) * 4);
    M.natural = pl.resident[x] != 0;
    M.n_slots = (int) pl.n_slots[x];
  }

  // ---- launch list in execution order, panels in first-use order ------------------------------
  group_of.assign((size_t) NpC, 0);
  slice_end.assign((size_t) NpC, std::vector<int64_t>());
  slice_ev.assign((size_t) NpC, std::vector<hipEvent_t>());
  std::vector<std::vector<char>> seen(3);
  for (int x = 0; x < 3; x++) seen[x].assign(mat[x].panels.size(), 0);
  gb.push_back(0);
  for (int64_t pc = group; pc < NpC; pc++) gb.push_back(pc);
  gb.push_back(NpC);
  auto add_launch = [&](int64_t pc, int64_t q0, int64_t q1, int64_t l0, int64_t l1) {
    launches.push_back(Launch{(int) pc, q0, q1, l0, l1});
    // the panels it touches, in the order A, B, C: a matrix paneled along k contributes l0 .. l1-1, one paneled
    // along C's other dimension q0 .. q1-1, one paneled along the C panel dimension the panel pc
    for (int x = 0; x < 3; x++) {
      const int rd = mat[x].rdim;
      const int64_t p0 = rd == dC ? pc : (rd == 1 ? l0 : q0), p1 = rd == dC ? pc + 1 : (rd == 1 ? l1 : q1);
      for (int64_t pp = p0; pp < p1; pp++) {
        if (seen[x][(size_t) pp]) continue;
        seen[x][(size_t) pp] = 1;
        if (x < 2 || c_read) order.emplace_back(x, (int) pp);
      }
    }
  };
  for (size_t gx = 0; gx + 1 < gb.size(); gx++, n_groups++) {
    const int64_t G0 = gb[gx], G1 = gb[gx + 1];
    for (int64_t pc = G0; pc < G1; pc++) group_of[(size_t) pc] = n_groups;
    if (kmeans) {
      // flash::kmeans: the reference's tile tasks, l-major inside the group
      for (int64_t l = 0; l < Nk; l++)
        for (int64_t pc = G0; pc < G1; pc++)
          for (int64_t q = 0; q < Nq; q++) add_launch(pc, q, q + 1, l, l + 1);
    } else if (gx == 0 || ref_chain) {
      // the ramp group (the resident operand is still streaming in): k-block by k-block, l-major, so that panel l of
      // it unlocks work on every C panel of the group; the reference's chain keeps that cut for all groups
      // Order inside the group: l-major (every C panel of the group takes k-block l, then l + 1): the panels' first-use
      // order -- the order they are READ in -- is then A0 B0 A1 .. A(G-1) B1 B2 ...  $BOF_PANEL_RAMP_ORDER=1 orders the
      // launches by anti-diagonals (C panel + k-block) instead, so that the reads alternate between the operands from
      // the start (A0 B0 | B1 A1 | B2 A2 ...) and n panels of each enable n^2 launches early (VERDICT r5 item 4's
      // wavefront).  Measured in round 6 (profiles/r6/sched_ab.md): 65536^3 from O_DIRECT files 4.21-4.24 s against
      // 4.10-4.15 s l-major -- the launches of one diagonal stand behind its first one, which waits for the newest B
      // panel, and the dispatcher issues in list order -- so l-major stays the default.
      if (env_long("BOF_PANEL_RAMP_ORDER", 0) != 0) {
        for (int64_t d = 0; d < Nk + (G1 - G0) - 1; d++)
          for (int64_t pc = G0; pc < G1; pc++) {
            const int64_t l = d - (pc - G0);
            if (l >= 0 && l < Nk) add_launch(pc, 0, Nq, l, l + 1);
          }
      } else {
        // $BOF_PANEL_RAMP_K k-blocks per launch of the ramp group (default 2 since round 6; the reference's chain:
        // always 1).  A launch over two k-blocks waits for two panels of the streaming operand and halves the raw-sum
        // round trips and launch boundaries of the group: inside the cfg2 steps the tile kernel runs at 0.957 of peak
        // against 0.951 with one k-block per launch, the steps themselves (disk-bound) and cfg2 from the page cache
        // (0.515-0.53 s either way) do not move (profiles/r6/sched/README.md).
        const int64_t R = ref_chain ? 1 : std::max<int64_t>(1, env_long("BOF_PANEL_RAMP_K", 2));
        for (int64_t l = 0; l < Nk; l += R)
          for (int64_t pc = G0; pc < G1; pc++) add_launch(pc, 0, Nq, l, std::min(Nk, l + R));
      }
      // The group's C panels complete one after the other in its last k-range (l-major: panel G0 first, a launch
      // apart each): every panel's last launch carries an event of its own and the flusher takes the panel when THAT
      // has run, instead of waiting for the group's last kernel -- cfg2 from O_DIRECT files: the first C panel starts
      // its write-back three launches (44 ms) earlier, while A's later panels are still being read
      // ($BOF_PANEL_RAMP_FLUSH=0: the round-5 behaviour, every panel of the group waits for group_ev).
      if (G1 - G0 > 1 && env_long("BOF_PANEL_RAMP_FLUSH", 1) != 0)
        for (size_t t2 = group_end.empty() ? 0 : group_end.back(); t2 < launches.size(); t2++) {
          Launch &L2 = launches[t2];
          if (L2.l1 != Nk || L2.q0 != 0 || L2.q1 != Nq) continue;
          L2.fin = true;
          slice_end[(size_t) L2.pc].assign(1, mat[2].panels[(size_t) L2.pc].nr);
        }
    } else {
      // behind the ramp everything a C panel needs but its own streamed panel is resident: ONE launch over the
      // whole K -- no C round trip between the k-blocks, an eighth of the launch boundaries
      // ... the LAST C panel of the slab in row slices, so that its write-back starts before its last kernel has
      // ended: what follows the last kernel of the call is then the write-back of a quarter panel.  Only the last:
      // a launch boundary costs ~0.3 ms (the spread of the workgroups' finishing times in a launch's last round), and
      // four slices of every panel took back the round-6 kernel's gain on the whole-K launches (rocprofv3: 4 x 15.10 ms
      // against 57.3 ms for the panel in one launch).  $BOF_PANEL_SLICES (default 4; 1 = off), $BOF_PANEL_SLICES_ALL=1
      // (every whole-K panel: tests), $BOF_PANEL_SLICE_ROWS (rounding unit of a slice, default 256: whole tile rows)
      const bool slice_all = env_long("BOF_PANEL_SLICES_ALL", 0) != 0;
      for (int64_t pc = G0; pc < G1; pc++) {
        const int64_t nr = mat[2].panels[(size_t) pc].nr;
        const int64_t want = (slice_all || pc == NpC - 1) ? std::max<int64_t>(1, env_long("BOF_PANEL_SLICES", 4)) : 1;
        const int64_t unit = std::max<int64_t>(1, env_long("BOF_PANEL_SLICE_ROWS", 256));      // (tests: small panels)
        const int64_t rows = std::max<int64_t>(unit, (nr / want + unit - 1) / unit * unit);
        if (want <= 1 || rows >= nr) { add_launch(pc, 0, Nq, 0, Nk); continue; }
        for (int64_t d0 = 0; d0 < nr; d0 += rows) {
          add_launch(pc, 0, Nq, 0, Nk);
          launches.back().d0 = d0;
          launches.back().d1 = std::min(nr, d0 + rows);
          slice_end[(size_t) pc].push_back(launches.back().d1);
        }
      }
    }
    group_end.push_back(launches.size());
  }

  // ---- HBM: kept from an earlier call when it has this call's size; what is missing is allocated by alloc_main
  // in first-use order while the pipeline already runs.  A resident operand is ONE allocation (Mat::whole).
  for (int x = 0; x < 3; x++) {
    Mat &M = mat[x];
    M.whole = x < 2 && M.natural;
    M.whole_ptr = x < 2 ? &res->whole[x] : nullptr;
    M.twhole_ptr = x < 2 ? &res->twhole[x] : nullptr;
    if (M.whole) {
      res->drop(x, 0);
      const size_t want_b = round_up(M.total_bytes, 2u << 20);
      if (res->whole_bytes[x] != want_b) {
        res->drop_whole(x);
        res->whole_bytes[x] = want_b;
      }
      M.slots = &res->slot[x];
      continue;
    }
    if (x < 2) res->drop_whole(x);
    if (res->slot_bytes[x] != M.slot_bytes) {
      res->drop(x, 0);
      res->slot_bytes[x] = M.slot_bytes;
    }
    const size_t want = (size_t) (M.natural ? (int) M.panels.size() : M.n_slots);
    if (res->slot[x].size() > want) res->drop(x, want);
    res->slot[x].resize(want, nullptr);
    M.slots = &res->slot[x];
  }
  {
    // k-major copies (see Mat::kmajor_copy): for an operand stored k-contiguous whose panels' row counts keep the
    // copy's leading dimension a multiple of 4 (vector loads), whose tiles are each used by >= 4 tile tasks, and
    // whose copies still fit the budget.  bof_options.panel_kmajor (1 off, 2 on, 3 on without the reuse
    // condition: tests) / BOF_PANEL_KMAJOR (0, 1, 2).
    const int kmode = o.panel_kmajor > 0 ? o.panel_kmajor - 1 : (int) env_long("BOF_PANEL_KMAJOR", 1);
    size_t extra = need_acc ? (size_t) group * mat[2].slot_bytes : 0;
    for (int x = 0; x < 2; x++) {
      Mat &M = mat[x];
      const int64_t reuse = x == xmat ? Nq : NpC;
      // (round 6: an x-major operand goes through sgemm_tile256_dmax_kernel as it is -- swizzled LDS-DMA straight from
      //  its rows, in every layout -- so the copy is only made where that kernel cannot run: rows that are not 16-byte
      //  aligned in the image (leading dimension not a multiple of 4), or $BOF_GEMM_DMAX=0)
      const bool direct = env_long("BOF_GEMM_DMAX", 1) != 0 && M.ld % 4 == 0;
      bool ok = kmode > 0 && M.cdim == 1 && M.cols % 4 == 0 && (kmode > 1 || reuse >= 4) && !direct;
      int64_t max_nr = 0;
      for (const Panel &P : M.panels) {
        ok = ok && P.nr % 4 == 0;
        max_nr = std::max(max_nr, P.nr);
      }
      if (M.whole) {
        const size_t tb = round_up((size_t) M.rows * (size_t) M.cols * 4, 2u << 20);
        if (ok && pl.need_bytes + extra + tb > budget) ok = false;
        M.kmajor_copy = ok;
        res->drop_t(x, 0);
        res->tslot_bytes[x] = 0;
        if (!ok || res->twhole_bytes[x] != tb) {
          res->drop_twhole(x);
          res->twhole_bytes[x] = ok ? tb : 0;
        }
        if (ok) extra += tb;
        M.tslots = &res->tslot[x];
        continue;
      }
      res->drop_twhole(x);
      M.tslot_bytes = round_up((size_t) max_nr * (size_t) M.cols * 4, 2u << 20);
      const size_t n = (size_t) (M.natural ? (int) M.panels.size() : M.n_slots);
      if (ok && pl.need_bytes + extra + n * M.tslot_bytes > budget) ok = false;
      M.kmajor_copy = ok;
      if (!ok || res->tslot_bytes[x] != M.tslot_bytes) {
        res->drop_t(x, 0);
        res->tslot_bytes[x] = ok ? M.tslot_bytes : 0;
      }
      if (ok) {
        extra += n * M.tslot_bytes;
        if (res->tslot[x].size() > n) res->drop_t(x, n);
        res->tslot[x].resize(n, nullptr);
      }
      M.tslots = &res->tslot[x];
    }
  }
  {
    // what alloc_main has to provide, in order of first use ((x, -1): a resident operand's one image)
    std::vector<std::vector<char>> listed(3);
    for (int x = 0; x < 3; x++) listed[x].assign(std::max<size_t>(1, res->slot[x].size()), 0);
    auto want_panel = [&](int x, int64_t pp) {
      Mat &M = mat[x];
      if (M.whole) {
        if (listed[x][0]) return;
        listed[x][0] = 1;
        if (!*M.whole_ptr || (M.kmajor_copy && !*M.twhole_ptr)) alloc_order.emplace_back(x, -1);
        return;
      }
      const int sl = M.slot_of((int) pp);
      if (listed[x][(size_t) sl]) return;
      listed[x][(size_t) sl] = 1;
      if (!(*M.slots)[(size_t) sl] || (M.kmajor_copy && !(*M.tslots)[(size_t) sl])) alloc_order.emplace_back(x, sl);
    };
    for (const Launch &L : launches)
      for (int x = 0; x < 3; x++) {
        const int rd = mat[x].rdim;
        const int64_t p0 = rd == dC ? L.pc : (rd == 1 ? L.l0 : L.q0), p1 = rd == dC ? L.pc + 1 : (rd == 1 ? L.l1 : L.q1);
        for (int64_t pp = p0; pp < p1; pp++) want_panel(x, pp);
      }
  }
  return BOF_OK;
}

// events, streams, the write ring (the device is current)
int PanelRun::prepare() {
  res->ev_next = 0;       // this call's events come out of the resources' pool, from its start
  for (int x = 0; x < 3; x++)
    for (auto &P : mat[x].panels) {
      int rc0 = res->take_event(&P.ready);
      if (!rc0 && x == 2) rc0 = res->take_event(&P.d2h_done);
      if (rc0) return rc0;
    }
  // A 4096^2 tile launch is 256 workgroups = the whole chip, so more than two compute streams only
  // interleave whole-chip kernels of different chains and starve the copy queues: measured on
  // cfg2 files (page cache) 0.70 s with 4 streams, 0.58 s with 2, 0.59 s with 1
  // (profiles/r2/e2e_sweep_*.txt).  bof_options.panel_streams / BOF_PANEL_STREAMS override.
  // Round 6: flash::gemm's panel launches go to ONE compute stream by default -- a panel launch is 512-2048
  // workgroups, i.e. 2-8 rounds of the whole chip, and the same-lease A/B of the headline (profiles/r6/ab_headline:
  // 0.831 s median with two streams, 0.824-0.868 s with one) shows no difference; with one stream the per-launch
  // kernel timing is one kernel alone on the chip, and bench.py measures the library's default instead of a special
  // configuration.  flash::kmeans keeps two (its tile tasks are 256 workgroups each).
  const long senv = o.panel_streams > 0 ? o.panel_streams : env_long("BOF_PANEL_STREAMS", 0);
  ss = stream_set(senv > 0 ? (int) std::min<long>(senv, 16) : (kmeans ? std::min(o.n_streams, 2) : 1));
  if (!ss) { set_error("bof_flash_gemm: stream creation failed"); return BOF_EHIP; }
  group_ev.assign((size_t) n_groups, std::vector<hipEvent_t>());
  for (auto &v : group_ev)
    for (int s = 0; s < ss->n; s++) {
      hipEvent_t e;
      const int rc0 = res->take_event(&e);
      if (rc0) return rc0;
      v.push_back(e);
    }
  for (size_t pc = 0; pc < slice_end.size(); pc++)
    for (size_t s = 0; s < slice_end[pc].size(); s++) {
      hipEvent_t e;
      const int rc0 = res->take_event(&e);
      if (rc0) return rc0;
      slice_ev[pc].push_back(e);
    }
  const int rc = res->wring.init(std::max(2, o.pinned_slots), chunk + 2 * Mat::kPage);   // slack: widened / page-congruent placement
  if (rc) return rc;
  // raw accumulator panels of the ramp group's chains (beta != 0 only: else the C slot itself carries the sums)
  {
    const size_t nacc = need_acc ? (size_t) (gb[1] - gb[0]) : 0;
    if (res->acc_bytes != mat[2].slot_bytes) { res->drop_acc(0); res->acc_bytes = mat[2].slot_bytes; }
    if (res->acc.size() > nacc) res->drop_acc(nacc);
    res->acc.resize(nacc, nullptr);
    for (char *&p : res->acc)
      if (!p) {
        const hipError_t e = hipMalloc((void **) &p, res->acc_bytes);
        if (e != hipSuccess) {
          (void) hipGetLastError();
          p = nullptr;
          set_error("bof_flash_gemm: no HBM for the accumulator panels of the ramp group");
          return e == hipErrorOutOfMemory ? BOF_ENOMEM : BOF_EHIP;
        }
      }
  }
  if (!res->h2d) BOF_HIP_TRY(copy_stream_create(&res->h2d));
  if (!res->d2h) BOF_HIP_TRY(copy_stream_create(&res->d2h));
  h2d = res->h2d;
  d2h = res->d2h;
  return verify_setup();
}

// BOF_VERIFY: one table per device; per panel the hand-over points it passes and which of them must agree
int PanelRun::verify_setup() {
  if (!verify_wanted(o)) return BOF_OK;
  size_t n = 0;
  for (int x = 0; x < 3; x++) n += mat[x].panels.size();
  // per launch: a consumer-side sum (two with a k-major copy) of every operand panel it reads, one of its C panel,
  // the chain's partial sums behind it and in front of the next one; one spot check
  size_t per_launch = 0;
  for (const Launch &L : launches) per_launch += 2 * (size_t) ((L.l1 - L.l0) + (L.q1 - L.q0) + 1) * 2 + 4;
  const int rc = vf.init(dev, n * Panel::VE_N + 16 + per_launch, launches.size());
  if (rc) return rc;
  for (int x = 0; x < 3; x++) {
    Mat &M = mat[x];
    bool words = (M.f.foffset % 4) == 0;
    for (size_t p = 0; p < M.panels.size() && words; p++)
      for (int c = 0; c < M.n_chunks((int) p, chunk); c++) {
        uint64_t off, len;
        M.chunk_span((int) p, c, chunk, &off, &len);
        words = words && off % 4 == 0 && len % 4 == 0;
      }
    if (!words) continue;       // byte-granular cuts (never with float matrices at float offsets): not summed
    for (size_t p = 0; p < M.panels.size(); p++) {
      Panel &P = M.panels[p];
      if (x < 2 || c_read) {
        P.ve[Panel::VE_HOST_IN] = vf.entry();
        P.ve[Panel::VE_DEV_IN] = vf.entry();
        vf.expect(P.ve[Panel::VE_HOST_IN], P.ve[Panel::VE_DEV_IN], "panel: pinned slot after the file read vs HBM after H2D (mat, panel, device)",
                  x, (int) p, di);
      }
      if (x < 2) {
        P.ve[Panel::VE_DEV_LAST] = vf.entry();
        vf.expect(P.ve[Panel::VE_DEV_IN], P.ve[Panel::VE_DEV_LAST], "panel: HBM after H2D vs HBM after its last use (mat, panel, device)", x,
                  (int) p, di);
        if (M.kmajor_copy) {
          P.ve[Panel::VE_RECT_IN] = vf.entry();
          P.ve[Panel::VE_T_IN] = vf.entry();
          P.ve[Panel::VE_T_LAST] = vf.entry();
          vf.expect(P.ve[Panel::VE_RECT_IN], P.ve[Panel::VE_T_IN], "panel: HBM image vs its k-major copy (mat, panel, device)", x, (int) p, di);
          vf.expect(P.ve[Panel::VE_T_IN], P.ve[Panel::VE_T_LAST], "panel: k-major copy when made vs after its last use (mat, panel, device)", x,
                    (int) p, di);
        }
      } else {
        P.ve[Panel::VE_C_DEV] = vf.entry();
        P.ve[Panel::VE_C_HOST] = vf.entry();
        P.ve[Panel::VE_C_FILE] = vf.entry();
        vf.expect(P.ve[Panel::VE_C_DEV], P.ve[Panel::VE_C_HOST], "C panel: HBM after its last kernel vs pinned slot after D2H (panel, device)",
                  (int) p, di);
        vf.expect(P.ve[Panel::VE_C_HOST], P.ve[Panel::VE_C_FILE], "C panel: pinned slot after D2H vs the file after the write (panel, device)",
                  (int) p, di);
      }
    }
  }
  return BOF_OK;
}

// the HBM image of panel p of operand x has seen its last kernel (st is ordered behind it): sum it once more
hipError_t PanelRun::verify_last_use(int x, int p, hipStream_t st) {
  Mat &M = mat[x];
  Panel &P = M.panels[(size_t) p];
  if (!vf.on || P.v_last_done || x > 1) return hipSuccess;
  P.v_last_done = true;
  hipError_t e = vf.on_device(P.ve[Panel::VE_DEV_LAST], M.panel_ptr(p), 1, (int64_t) (P.bytes / 4), 0, 0, 0, st);
  if (e == hipSuccess && M.kmajor_copy)
    e = vf.on_device(P.ve[Panel::VE_T_LAST], M.tpanel_ptr(p), M.cols, P.nr, M.t_ld(p), 0, M.cols, st);
  return e;
}

// after the call has drained (device idle, writes done): the panels still in HBM, C back from its file, then
// the comparison.  BOF_OK / BOF_EVERIFY / BOF_EIO.
int PanelRun::verify_finish() {
  if (!vf.on) return BOF_OK;
  DeviceScope scope(dev);
  {
    std::lock_guard<std::mutex> lk(vf_mu);
    for (int x = 0; x < 2; x++)
      for (size_t p = 0; p < mat[x].panels.size(); p++)
        if (mat[x].panels[p].state == 2 && mat[x].panel_ptr((int) p) &&
            (mat[x].natural || p + (size_t) mat[x].n_slots >= mat[x].panels.size()))
          BOF_HIP_TRY(verify_last_use(x, (int) p, h2d));
  }
  BOF_HIP_TRY(hipStreamSynchronize(h2d));
  Mat &C = mat[2];
  if (!C.panels.empty() && C.panels[0].ve[Panel::VE_C_FILE] != Verify::kNone) {
    const int fd = file_is_direct(C.f.fd) ? file_buffered_fd(C.f.fd) : C.f.fd;
    std::vector<char> buf(8u << 20);
    for (size_t p = 0; p < C.panels.size() && fd >= 0; p++) {
      const Panel &P = C.panels[p];
      for (uint64_t off = 0; off < P.bytes; off += buf.size()) {
        const size_t len = (size_t) std::min<uint64_t>(buf.size(), P.bytes - off);
        size_t got = 0;
        while (got < len) {
          const ssize_t r = pread(fd, buf.data() + got, len - got, (off_t) (C.file_off((int) p) + off + got));
          if (r <= 0) { set_error("BOF_VERIFY: re-reading C from its file failed"); return BOF_EIO; }
          got += (size_t) r;
        }
        vf.on_host(P.ve[Panel::VE_C_FILE], buf.data(), 1, (int64_t) (len / 4), 0, off / 4);
      }
    }
  }
  return vf.finish(cnt, "bof_flash_gemm (panels)");
}

// the device's launches, in schedule order
void PanelRun::dispatch() {
  (void) hipSetDevice(dev);
  PanelHub &H = *hub;
  Mat &X = mat[xmat], &C = mat[2];
  const int64_t Nk = g.nblk[1];
  const int qdim = dC == 0 ? 2 : 0;            // C's other dimension (its stored columns)
  // per (matrix, stream): the panels whose `ready` (C without a read: whose slot's previous write-back) that
  // stream has already been told to wait for
  std::vector<std::vector<char>> waited((size_t) 3 * (size_t) ss->n);
  std::map<std::pair<int, int64_t>, size_t> chain_post;     // BOF_VERIFY: (C panel, first tile) -> sum behind the chain's last launch
  for (int x = 0; x < 3; x++)
    for (int q = 0; q < ss->n; q++) waited[(size_t) x * (size_t) ss->n + (size_t) q].assign(mat[x].panels.size(), 0);
  // first stored row / column of a launch's part of matrix x along logical dimension d (0 m, 1 k, 2 n)
  auto start_of = [&](const Launch &L, int d) -> int64_t {
    return d == dC ? (int64_t) L.pc * g.blk[dC] + L.d0 : (d == 1 ? L.l0 * g.blk[1] : L.q0 * g.blk[qdim]);
  };
  auto range_of = [&](const Launch &L, int x, int64_t *p0, int64_t *p1) {
    const int rd = mat[x].rdim;
    *p0 = rd == dC ? L.pc : (rd == 1 ? L.l0 : L.q0);
    *p1 = rd == dC ? L.pc + 1 : (rd == 1 ? L.l1 : L.q1);
  };
  size_t t = 0;
  for (int gi = 0; gi < n_groups && !fail && herr == hipSuccess; gi++) {
    TraceRange grange("panel group dispatch");
    for (; t < group_end[(size_t) gi]; t++) {
      const Launch &L = launches[t];
      const bool first = L.l0 == 0, last = L.l1 == Nk;
      // does this launch read the caller's C?  (reference chain: its first k-block scales it; one chain over K: the
      // final launch does; beta == 0: nobody)
      const bool reads_c = c_read && (ref_chain ? first : last);
      // does it store into the C panel's slot?  (a chain with raw accumulator panels only at its end)
      const bool writes_c = !need_acc || ref_chain || last;
      const int cprev = C.natural ? -1 : L.pc - C.n_slots;
      int64_t p0[3], p1[3];
      for (int x = 0; x < 3; x++) range_of(L, x, &p0[x], &p1[x]);
      {
        std::unique_lock<std::mutex> lk(H.mu);
        H.cv.wait(lk, [&] {
          if (H.io_error.load()) return true;
          for (int x = 0; x < 2; x++)
            for (int64_t pp = p0[x]; pp < p1[x]; pp++)
              if (mat[x].panels[(size_t) pp].state != 2) return false;
          if (reads_c) return C.panels[(size_t) L.pc].state == 2;
          if (c_read || !writes_c) return true;               // its C panel comes in through the readers, later
          if (!C.panel_ptr(L.pc)) return false;               // its HBM slot is still being allocated
          return cprev < 0 || C.panels[(size_t) cprev].retired;   // the slot's write-back is on its way
        });
      }
      if (H.io_error.load()) { fail = BOF_EIO; break; }
      // chain -> stream: FIFO = the parent dependency of src/blas/gemm.cpp:122-127
      const int sidx = (int) ((kmeans ? (int64_t) L.pc * Nq + L.q0 : (int64_t) L.pc) % ss->n);
      hipStream_t st = ss->s[sidx];
      for (int x = 0; x < 2 && herr == hipSuccess; x++)
        for (int64_t pp = p0[x]; pp < p1[x] && herr == hipSuccess; pp++) {
          char &w = waited[(size_t) x * (size_t) ss->n + (size_t) sidx][(size_t) pp];
          if (w) continue;
          w = 1;
          herr = hipStreamWaitEvent(st, mat[x].panels[(size_t) pp].ready, 0);
          evt("stream waits for panel ready", x, (int) pp, (uint64_t) (((uint64_t) di << 8) | (uint64_t) sidx));
        }
      if (herr == hipSuccess && (reads_c || (writes_c && !c_read))) {
        char &w = waited[(size_t) 2 * (size_t) ss->n + (size_t) sidx][(size_t) L.pc];
        if (!w) {
          w = 1;
          if (reads_c) {
            herr = hipStreamWaitEvent(st, C.panels[(size_t) L.pc].ready, 0);
            evt("stream waits for panel ready", 2, L.pc, (uint64_t) (((uint64_t) di << 8) | (uint64_t) sidx));
          } else if (cprev >= 0) {
            for (hipEvent_t e : C.panels[(size_t) cprev].retire_ev)
              if (herr == hipSuccess) herr = wait_event_both(st, e);       // (the flusher recorded it behind its D2H copies)
            evt("stream waits for C slot write-back", cprev, L.pc, (uint64_t) (((uint64_t) di << 8) | (uint64_t) sidx));
          }
        }
      }
      if (herr != hipSuccess) break;
      // operands: pointer into the image (a resident operand: the whole matrix; else the panel) + the file's leading
      // dimension, or -- with a k-major copy -- into the copy ([k][rows]), the flag flipped
      const float *po[2];
      int64_t ldo[2];
      char flag[2] = {ta, tb};
      for (int x = 0; x < 2; x++) {
        const Mat &M = mat[x];
        const int64_t R0 = start_of(L, M.rdim);
        const int64_t C0 = start_of(L, M.cdim) + (M.cdim == dC ? M.col_base : 0);
        const int pp = (int) p0[x];
        const int64_t rb = M.whole ? 0 : M.panels[(size_t) pp].r0;      // rows count from the image's first row
        if (M.kmajor_copy) {
          const float *T = (const float *) (M.whole ? *M.twhole_ptr : M.tpanel_ptr(pp));
          po[x] = T + C0 * M.t_ld(pp) + (R0 - rb);
          ldo[x] = M.t_ld(pp);
          flag[x] = flag[x] == 'N' ? 'T' : 'N';
        } else {
          const float *B = (const float *) (M.whole ? *M.whole_ptr : M.panel_ptr(pp));
          po[x] = B + (R0 - rb) * M.ld + C0;
          ldo[x] = M.ld;
        }
      }
      int64_t ext[3];
      ext[dC] = L.d1 > 0 ? L.d1 - L.d0 : C.panels[(size_t) L.pc].nr;
      ext[1] = last ? g.size[1] - L.l0 * g.blk[1] : (L.l1 - L.l0) * g.blk[1];
      ext[qdim] = L.q1 == Nq ? g.size[qdim] - L.q0 * g.blk[qdim] : (L.q1 - L.q0) * g.blk[qdim];
      float *pcp = (float *) C.panel_ptr(L.pc) + L.d0 * C.ld + L.q0 * C.blk_c;
      // what the launch is, in the terms of its kernel entry point
      const bool chain_step = !ref_chain && !(first && last);   // one k-range of a chain that carries raw sums
      float *accp = chain_step && need_acc ? (float *) res->acc[(size_t) (L.pc - gb[0])] + L.q0 * C.blk_c : pcp;
      SpotArgs sa{ord, flag[0], flag[1], ext[0], ext[2], ext[1], alpha, po[0], ldo[0], po[1], ldo[1],
                  ref_chain ? (first ? beta : 1.0f) : beta, chain_step && !last ? accp : pcp, C.ld};
      if (chain_step) {
        if (!first) { sa.ch.acc_in = accp; sa.ch.ld_acc = C.ld; }
        sa.ch.raw_out = !last;
      }
      const int64_t ti = dC == 0 ? L.pc : L.q0, tj = dC == 0 ? L.q0 : L.pc;
      if (has_kv) {
        sa.u1 = kv.c_l2sq + row_base + ti * g.blk[0]; sa.v1 = kv.ones;
        sa.u2 = kv.ones; sa.v2 = kv.p_l2sq + col_base + tj * g.blk[2];
      }
      sa.seed = ((uint64_t) L.pc << 40) ^ ((uint64_t) L.l0 << 20) ^ (uint64_t) L.q0 ^ ((uint64_t) di << 56) ^ ((uint64_t) L.d0 << 8);
      evt("launch", L.pc, (int) (L.l0 * 1024 + L.l1), (uint64_t) (((uint64_t) di << 24) | ((uint64_t) sidx << 16) | (uint64_t) L.q0));
      Verify::Spot spot;
      // the rectangle of C (or of the raw accumulator panel) the launch stores into; the same one it starts from
      const int64_t c_rows = ext[dC], c_cols = ext[qdim];
      if (vf.on) {
        // CONSUMER-side sums, on the compute stream in front of the launch: every operand panel it reads (and the
        // panel's k-major copy), the C panel its beta applies to, the raw sums it continues -- against what the
        // producers summed when they handed the objects over (flash_common.h, "Round 5")
        for (int x = 0; x < 2 && herr == hipSuccess; x++)
          for (int64_t pp = p0[x]; pp < p1[x] && herr == hipSuccess; pp++) {
            const Mat &M = mat[x];
            const Panel &P = M.panels[(size_t) pp];
            if (P.ve[Panel::VE_DEV_IN] == Verify::kNone) continue;
            const size_t e1 = vf.entry();
            vf.expect(P.ve[Panel::VE_DEV_IN], e1, "panel: HBM after H2D vs on the COMPUTE stream in front of a launch (mat, panel, C panel)",
                      x, (int) pp, L.pc);
            herr = vf.on_device(e1, M.panel_ptr((int) pp), 1, (int64_t) (P.bytes / 4), 0, 0, 0, st);
            if (herr == hipSuccess && M.kmajor_copy && P.ve[Panel::VE_T_IN] != Verify::kNone) {
              const size_t e2 = vf.entry();
              vf.expect(P.ve[Panel::VE_T_IN], e2, "panel: k-major copy when made vs on the COMPUTE stream in front of a launch (mat, panel, C panel)",
                        x, (int) pp, L.pc);
              herr = vf.on_device(e2, M.tpanel_ptr((int) pp), M.cols, P.nr, M.t_ld((int) pp), 0, M.cols, st);
            }
          }
        // (a panel multiplied in row slices is whole only in front of its FIRST slice: later ones find the rows the
        //  earlier slices stored)
        if (herr == hipSuccess && reads_c && L.d0 == 0 && C.panels[(size_t) L.pc].ve[Panel::VE_DEV_IN] != Verify::kNone) {
          const size_t e1 = vf.entry();
          vf.expect(C.panels[(size_t) L.pc].ve[Panel::VE_DEV_IN], e1, "C panel: HBM after H2D vs on the COMPUTE stream in front of the launch that reads it (panel, device)",
                    L.pc, di);
          herr = vf.on_device(e1, C.panel_ptr(L.pc), 1, (int64_t) (C.panels[(size_t) L.pc].bytes / 4), 0, 0, 0, st);
        }
        const auto key = std::make_pair(L.pc, L.q0);
        if (herr == hipSuccess && !first) {       // the chain's partial result as the launch before left it
          auto it = chain_post.find(key);
          if (it != chain_post.end()) {
            const size_t e1 = vf.entry();
            vf.expect(it->second, e1, "chain: partial result behind a launch vs in front of the next launch of the chain (C panel, first k-block, first tile)",
                      L.pc, (int) L.l0, (int) L.q0);
            herr = vf.on_device(e1, accp, c_rows, c_cols, C.ld, 0, 0, st);
          }
        }
        // a launch that starts its rectangle from nothing: poison it first (a first k-range that did not store shows)
        if (herr == hipSuccess && first && !reads_c && (chain_step || !c_read))
          herr = launch_from_persistent(dev, [&] {
            return hipMemset2DAsync(chain_step && !last ? (void *) accp : (void *) pcp, (size_t) C.ld * 4, 0xFF, (size_t) c_cols * 4, (size_t) c_rows, st);
          });
        if (herr == hipSuccess)
          herr = vf.spot_before(sa, st, &spot, "launch: 64 sampled outputs recomputed vs stored (C panel, l0 * 1024 + l1, first tile)", L.pc,
                                (int) (L.l0 * 1024 + L.l1), (int) L.q0);
        if (herr != hipSuccess) break;
      }
      herr = ktimer.begin(st);
      if (herr != hipSuccess) break;
      // self-test of the instrumentation ($BOF_VERIFY_INJECT=3): the second launch of the call is dropped
      if (vf.on && t == 1 && env_long("BOF_VERIFY_INJECT", 0) == 3) {
      } else
      if (ref_chain)
        herr = tile_sgemm(ord, flag[0], flag[1], ext[0], ext[2], ext[1], alpha, po[0], ldo[0], po[1], ldo[1], sa.beta, pcp, C.ld,
                          has_kv ? &kv : nullptr, row_base + ti * g.blk[0], col_base + tj * g.blk[2], st);
      else if (!chain_step)
        herr = sgemm(ord, flag[0], flag[1], ext[0], ext[2], ext[1], alpha, po[0], ldo[0], po[1], ldo[1], beta, pcp, C.ld, st);
      else   // one k-range of the chain: raw sums in, raw sums out; the caller's alpha and beta at the end
        herr = sgemm_chain(ord, flag[0], flag[1], ext[0], ext[2], ext[1], alpha, po[0], ldo[0], po[1], ldo[1], beta, sa.c, C.ld, sa.ch, st);
      if (herr == hipSuccess) herr = ktimer.end(st);
      if (herr == hipSuccess && vf.on) {
        herr = vf.spot_after(sa, spot, st);
        if (herr == hipSuccess && !last) {        // what the next launch of the chain must find
          const size_t e1 = vf.entry();
          chain_post[std::make_pair(L.pc, L.q0)] = e1;
          herr = vf.on_device(e1, sa.c, c_rows, c_cols, C.ld, 0, 0, st);
        }
      }
      g_mix[chain_step ? 0 : (L.d1 > 0 ? 2 : 1)]++;
      if (herr == hipSuccess && L.d1 > 0) {       // a row slice: its own event, what the flusher waits for chunk by chunk
        size_t si = 0;
        while (slice_end[(size_t) L.pc][si] != L.d1) si++;
        herr = hipEventRecord(slice_ev[(size_t) L.pc][si], st);
      } else if (herr == hipSuccess && L.fin) {   // the last launch of a ramp panel's chain: the panel may leave
        herr = hipEventRecord(slice_ev[(size_t) L.pc][0], st);
      }
      if (herr != hipSuccess) break;
      const uint64_t n_tasks = L.d0 > 0 ? 0 : (uint64_t) ((L.q1 - L.q0) * (L.l1 - L.l0));     // in the reference's tile tasks
      cnt.tasks += n_tasks;
      H.cnt.tasks += n_tasks;
    }
    if (fail || herr != hipSuccess) break;
    // group finished on the host side: mark where every stream stands, hand its C panels to the
    // flusher and let the streamed operand's panels of this group go
    for (int s = 0; s < ss->n && herr == hipSuccess; s++) herr = hipEventRecord(group_ev[(size_t) gi][(size_t) s], ss->s[s]);
    if (herr != hipSuccess) break;
    const int64_t G0 = gb[(size_t) gi], G1 = gb[(size_t) gi + 1];
    {
      std::lock_guard<std::mutex> lk(H.mu);
      if (!X.natural)
        for (int64_t pc = G0; pc < G1; pc++) {
          Panel &P = X.panels[(size_t) pc];
          P.retire_ev = group_ev[(size_t) gi];
          P.retired = true;
        }
      H.pump_fetches();
    }
    for (int64_t pc = G0; pc < G1; pc++) flush_q.push((int) pc);
    evt("group dispatched", gi, di, (uint64_t) t);
    if (trace_on()) {
      char lbl[64];
      snprintf(lbl, sizeof(lbl), "group %d dispatched", gi);
      trace(lbl);
    }
  }
  if (herr != hipSuccess || fail) H.fail_io(-EIO);
  seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - H.t_begin).count();
}

}  // namespace

void panel_resources_release() {
  std::lock_guard<std::mutex> lk(g_pres_mu);
  for (auto &kv : g_pres) {
    kv.second->drop_all();
    delete kv.second;
  }
  g_pres.clear();
  for (auto &kv : g_rring) {
    kv.second->destroy();
    delete kv.second;
  }
  g_rring.clear();
}

void panel_resources_release_device(int dev) {
  std::lock_guard<std::mutex> lk(g_pres_mu);
  for (auto it = g_pres.begin(); it != g_pres.end();) {
    if (it->first.first == dev) {
      it->second->drop_all();
      delete it->second;
      it = g_pres.erase(it);
    } else {
      ++it;
    }
  }
}

int kmeans_upload(const KmeansHost &kh, KmeansVecs *kv) {
  float *dv = nullptr;
  BOF_HIP_TRY(hipMalloc((void **) &dv, (size_t) (kh.m + kh.n + kh.n_ones) * sizeof(float)));
  hipError_t e = hipMemcpy(dv, kh.c_l2sq, (size_t) kh.m * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv + kh.m, kh.p_l2sq, (size_t) kh.n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv + kh.m + kh.n, kh.ones, (size_t) kh.n_ones * sizeof(float), hipMemcpyHostToDevice);
  // The caller's vectors are PAGEABLE host memory: a synchronous hipMemcpy may return once they are staged, with the
  // transfer still in flight on the null stream -- and the pipelines' compute streams are non-blocking, i.e. not
  // ordered behind the null stream.  Wait for the device before anything is launched.  (Not what the wrong kmeans
  // tiles of rounds 3-4 went away with -- they correlated with the launching thread and per-call event churn,
  // flash_common.h "persistent launcher threads"; cause not proven -- but the same class of ordering hole;
  // $BOF_KMEANS_UPLOAD_SYNC=0 restores the old behaviour.)
  if (e == hipSuccess && env_long("BOF_KMEANS_UPLOAD_SYNC", 1) != 0) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void) hipFree(dv); return hip_fail(e, "flash::kmeans: uploading the norm vectors"); }
  *kv = KmeansVecs{dv, dv + kh.m, dv + kh.m + kh.n};
  return BOF_OK;
}

int flash_gemm_panels(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, float beta,
                      bof_fptr fa, bof_fptr fb, bof_fptr fc, int64_t lda, int64_t ldb, int64_t ldc,
                      const bof_options &o, const std::vector<int> &devs, const KmeansHost *kh) {
  PanelHub H;
  H.t_begin = std::chrono::steady_clock::now();
  evt_mark_call_begin();
  evt("bof_flash_gemm (panels) begin", (int) m, (int) n, (uint64_t) k);
  for (auto &x : g_mix) x = 0;
  const GemmGeometry gfull = gemm_geometry(ord, ta, tb, m, n, k, lda, ldb, ldc, o.gemm_blk);
  if (gfull.nblk[0] * gfull.nblk[2] == 0 || gfull.nblk[1] == 0) return 1;
  int caller_dev = 0;
  BOF_HIP_TRY(hipGetDevice(&caller_dev));
  DeviceScope restore(caller_dev);
  const int dC = gfull.rdim[2];
  const int64_t NpC = gfull.nblk[dC];
  const bof_fptr fp[3] = {fa, fb, fc};

  // ---- C panels dealt to the devices in contiguous ranges; plan of every slab ----------------------
  const int n_use = (int) std::min<int64_t>((int64_t) devs.size(), NpC);
  std::vector<int> used(devs.begin(), devs.begin() + n_use);
  Cleanup guard;
  guard.add([&H] {
    for (auto &Rp : H.runs) {
      PanelRun &R = *Rp;
      DeviceScope ds(R.dev);
      if (R.has_kv) (void) hipFree(const_cast<float *>(R.kv.c_l2sq));
    }
  });
  int64_t p_next = 0;
  for (int d = 0; d < n_use; d++) {
    const int64_t cnt = NpC / n_use + (d < NpC % n_use ? 1 : 0), p0 = p_next;
    p_next += cnt;
    std::unique_ptr<PanelRun> Rp(new PanelRun());
    PanelRun &R = *Rp;
    R.hub = &H; R.di = d; R.dev = used[(size_t) d]; R.o = o;
    R.ktimer.on = o.kernel_timing > 0;
    R.ord = ord; R.ta = ta; R.tb = tb; R.alpha = alpha; R.beta = beta;
    R.kmeans = kh != nullptr;
    // the slab as a problem of its own: rows [p0 * blk, (p0 + cnt) * blk) of the C panel dimension
    // (the last slab runs to the end: a tail-merged last panel stays merged), explicit leading dims
    const int64_t e0 = p0 * gfull.blk[dC];
    const int64_t e1 = p0 + cnt == NpC ? gfull.size[dC] : (p0 + cnt) * gfull.blk[dC];
    const int64_t md = dC == 0 ? e1 - e0 : m, nd = dC == 2 ? e1 - e0 : n;
    R.g = gemm_geometry(ord, ta, tb, md, nd, k, gfull.ld[0], gfull.ld[1], gfull.ld[2], o.gemm_blk);
    if (R.g.nblk[dC] != cnt || (cnt > 1 && R.g.blk[dC] != gfull.blk[dC])) {
      set_error("bof_flash_gemm: internal error: a device slab does not tile like the whole problem");
      return BOF_EINVAL;
    }
    int rep = 0, reps = 0;
    for (int e = 0; e < n_use; e++)
      if (used[(size_t) e] == R.dev) { if (e < d) rep++; reps++; }
    BOF_HIP_TRY(hipSetDevice(R.dev));
    {
      std::lock_guard<std::mutex> lk(g_pres_mu);
      PanelResources *&pr = g_pres[std::make_pair(R.dev, rep)];
      if (!pr) { pr = new PanelResources(); pr->dev = R.dev; }
      R.res = pr;
    }
    const int rc = R.plan(used, reps, n_use > 1 ? gfull.size[dC] : 0, fp, gfull, p0);
    if (rc) return rc;   // +1: some slab is not eligible -> the whole call goes to the tile cache
    H.runs.push_back(std::move(Rp));
  }

  // ---- descriptors: one mode per FILE per call, over every device's requests -----------------------
  for (int x = 0; x < 3; x++) {
    bool aligned = true;
    for (auto &R : H.runs) aligned = aligned && requests_aligned(R->mat[x], R->chunk);
    for (auto &R : H.runs) {
      pick_descriptor(R->mat[x], aligned, o.use_odirect != 0);
      if (R->mat[x].fd < 0) { set_error("bof_flash_gemm: cannot open a buffered descriptor of an unaligned matrix file"); return BOF_EIO; }
    }
  }
  for (int x = 0; x < 2; x++) H.issued[x].assign(H.runs[0]->mat[x].panels.size(), 0);
  {
    uint64_t total = 0;
    bool c_all_resident = true;
    for (auto &R : H.runs) {
      c_all_resident = c_all_resident && R->mat[2].natural;
      for (const auto &xp : R->order) {
        const Mat &M = R->mat[xp.first];
        if (M.shared && R->di != 0) continue;          // read once for all devices
        total += (uint64_t) M.n_chunks(xp.second, R->chunk);
      }
    }
    H.chunks_total.store(total);
    H.writes_after_reads = c_all_resident && env_long("BOF_PANEL_WRITES_AFTER_READS", 0) != 0;
  }
  H.peer_bcast = H.runs.size() > 1 && (o.peer_bcast == 1 || (o.peer_bcast == 0 && env_long("BOF_PEER_BCAST", 0) > 0));
  guard.add([&H] {
    // a rank that gives up tells the peers so (they would wait for the timeout otherwise)
    for (int x = 0; x < 2; x++) {
      if (H.io_error.load()) H.seg[x].ring.fail_all();
      H.seg[x].ring.unmap();
    }
  });
  if (o.share_world > 1) {
    if (o.share_rank < 0 || o.share_rank >= o.share_world || !o.share_name[0] || strlen(o.share_name) > 40) {
      set_error("bof_flash_gemm: share_world > 1 needs share_rank in [0, share_world) and a share_name of 1..40 characters");
      return BOF_EINVAL;
    }
    H.share_world = o.share_world;
    H.share_rank = o.share_rank;
    H.share_timeout_s = (double) env_long("BOF_SHARE_TIMEOUT_S", 120);
    for (int x = 0; x < 2; x++) {
      const Mat &M = H.runs[0]->mat[x];
      if (!M.shared) continue;
      ShareSeg &sg = H.seg[x];
      for (size_t p = 0; p < M.panels.size(); p++) {
        sg.first_chunk.push_back(sg.n_chunks);
        sg.n_chunks += (size_t) M.n_chunks((int) p, H.runs[0]->chunk);
      }
      // ring depth: what the readers of every rank can have in flight, twice over; 2 GiB at the defaults
      const int n_slots = (int) std::min<size_t>(sg.n_chunks, (size_t) std::max<long>(8, env_long("BOF_SHARE_SLOTS", 64)));
      const std::string base = std::string(o.share_name) + "." + "AB"[x];
      if (!sg.ring.map(base, H.runs[0]->chunk + 2 * Mat::kPage, n_slots, std::string(o.share_name))) {
        set_error(std::string("bof_flash_gemm: cannot map the node-shared staging ring ") + base + ": " + strerror(errno));
        return BOF_EIO;
      }
    }
  }
  H.trace("plan");

  // ---- per-device events, streams, write rings; the shared read ring --------------------------------
  for (auto &R : H.runs) {
    BOF_HIP_TRY(hipSetDevice(R->dev));
    t_ordinal_rep = 0;       // which repetition of its ordinal this run is ($BOF_STREAMS_PER_REP: stream sets per repetition)
    for (auto &E : H.runs) {
      if (E.get() == R.get()) break;
      if (E->dev == R->dev) t_ordinal_rep++;
    }
    int rc = R->prepare();
    t_ordinal_rep = 0;
    if (rc) return rc;
    if (kh) {
      rc = kmeans_upload(*kh, &R->kv);
      if (rc) return rc;
      R->has_kv = true;
    }
  }
  BOF_HIP_TRY(hipSetDevice(caller_dev));
  {
    std::lock_guard<std::mutex> lk(g_pres_mu);
    PinnedRing *&rr = g_rring[*std::min_element(used.begin(), used.end())];
    if (!rr) rr = new PinnedRing();
    H.rring = rr;
  }
  // a chunk copied to D devices stays in its slot until the slowest copy is done: two more slots per extra device
  int rc = H.rring->init(std::max(2, o.pinned_slots) + 2 * ((int) H.runs.size() - 1), H.runs[0]->chunk + 2 * Mat::kPage, &used);
  if (rc) return rc;
  H.trace("rings/streams ready");

  const int nd = (int) H.runs.size();
  std::vector<std::thread> readers, writers, flushers, allocators;
  std::vector<std::shared_ptr<LaunchJob>> dispatch_jobs;
  const int n_readers = std::max(1, o.n_io_threads) + (nd - 1);
  const long wenv = o.panel_writers > 0 ? o.panel_writers : env_long("BOF_PANEL_WRITERS", 0);
  const int n_writers = (wenv > 0 ? (int) wenv : std::max(2, std::min(8, o.n_io_threads / 2))) + (nd - 1);
  for (int i = 0; i < n_readers; i++) readers.emplace_back([&H, i] { H.reader_main(i); });
  for (int i = 0; i < n_writers; i++) writers.emplace_back([&H, i] { H.writer_main(i); });
  for (auto &R : H.runs) {
    PanelRun *r = R.get();
    flushers.emplace_back([r] { r->flusher_main(); });
    allocators.emplace_back([r] { r->alloc_main(); });
  }
  {
    std::lock_guard<std::mutex> lk(H.mu);
    H.pump_fetches();
  }

  StallWatch watch("bof_flash_gemm (panels)",
                   [&H] { return H.cnt.rd.load() + H.cnt.wr.load() + H.cnt.h2d.load() + H.cnt.d2h.load() + H.cnt.peer.load() +
                                 H.cnt.tasks.load(); },
                   [&H] { H.fail_io(-ETIMEDOUT); });
  // ---- dispatch: the caller drives the first device, a thread each of the others ---------------------
  for (int d = 1; d < nd; d++) {
    PanelRun *r = H.runs[(size_t) d].get();
    // (a persistent launcher thread per (device, repetition), never a thread made for the call: flash_common.h)
    int rep = 0;
    for (int e = 0; e < d; e++)
      if (H.runs[(size_t) e]->dev == r->dev) rep++;
    dispatch_jobs.push_back(launch_async(r->dev, rep, [r] { r->dispatch(); }));
  }
  H.runs[0]->dispatch();
  for (auto &j : dispatch_jobs) launch_wait(j);

  // ---- drain ------------------------------------------------------------------------------------
  for (auto &th : allocators) th.join();
  H.fetch_q.close();
  for (auto &th : readers) th.join();
  for (auto &R : H.runs) R->flush_q.close();
  for (auto &th : flushers) th.join();
  H.write_q.close();
  for (auto &th : writers) th.join();
  for (auto &R : H.runs) {
    (void) hipSetDevice(R->dev);
    (void) hipDeviceSynchronize();
    R->ktimer.collect(R->cnt);
    H.cnt.klaunch += R->cnt.klaunch.load();
    H.cnt.kns += R->cnt.kns.load();
  }
  (void) hipSetDevice(caller_dev);
  H.trace("drained (writes done)");
  evt("bof_flash_gemm (panels) drained", nd, 0, H.cnt.tasks.load());
  int fail = 0;
  if (!H.io_error.load())
    for (auto &R : H.runs) {
      const bool clean = R->herr == hipSuccess && !R->fail;
      const int vrc = clean ? R->verify_finish() : BOF_OK;
      H.cnt.vchecks += R->cnt.vchecks.load();
      if (vrc && !fail) fail = vrc;
    }
  (void) hipSetDevice(caller_dev);
  evt_dump_env("bof_flash_gemm (panels)");
  for (auto &R : H.runs) {
    if (R->herr != hipSuccess && !fail) fail = hip_fail(R->herr, "bof_flash_gemm (panels) dispatch");
    if (R->fail && !fail) fail = R->fail;
  }
  if (H.io_error.load() && (!fail || fail == BOF_EIO)) {
    const int e = H.io_error.load();
    set_error("bof_flash_gemm: I/O pipeline failed: " + io_error_text(e));
    fail = e == -ENOMEM ? BOF_ENOMEM : BOF_EIO;
  }
  H.cnt.hits = 3 * H.cnt.tasks.load() - std::min<uint64_t>(H.cnt.misses.load(), 3 * H.cnt.tasks.load());
  publish_stats(H.cnt, std::chrono::duration<double>(std::chrono::steady_clock::now() - H.t_begin).count());
  std::vector<bof_flash_stats> per((size_t) nd);
  for (int d = 0; d < nd; d++) {
    const PanelRun &R = *H.runs[(size_t) d];
    bof_flash_stats s{};
    s.bytes_read = R.cnt.rd; s.bytes_written = R.cnt.wr; s.bytes_h2d = R.cnt.h2d; s.bytes_d2h = R.cnt.d2h;
    s.tasks = R.cnt.tasks; s.seconds = R.seconds;
    s.kernel_launches = R.cnt.klaunch; s.kernel_seconds = (double) R.cnt.kns.load() * 1e-9;
    s.verify_checks = R.cnt.vchecks; s.bytes_p2p = R.cnt.p2p;
    per[(size_t) d] = s;
  }
  publish_device_stats(per);
  return fail;
}

}  // namespace bof

extern "C" int bof_share_cleanup(const char *share_name) {
  if (!share_name || !share_name[0]) return BOF_EINVAL;
  for (const char *m : {".A", ".B"}) bof::ShareRing::unlink(std::string(share_name) + m);
  bof::ShareRing::unlink_group(std::string(share_name));
  return BOF_OK;
}

// Diagnostic (no GPU involved): `world` processes call this with the same name / sizes and their own rank; chunk
// c belongs to rank c % world, which fills it with a pattern of (c, position) and produces it; everybody else
// consumes it and checks the pattern.  Returns the number of chunks verified (n_chunks - the own ones) or a
// negative errno.  What tests/test_dist_gloo.py runs with several processes on the CPU box.
extern "C" int64_t bof_share_selftest(const char *share_name, int rank, int world, int64_t n_chunks,
                                      int64_t chunk_bytes, int n_slots, double timeout_s) {
  if (!share_name || world < 2 || rank < 0 || rank >= world || n_chunks <= 0 || chunk_bytes < 8 || n_slots < 1)
    return -EINVAL;
  bof::ShareRing ring;
  if (!ring.map(std::string(share_name) + ".A", (size_t) chunk_bytes, n_slots, std::string(share_name))) return -errno;
  std::atomic<int> stop{0};
  std::vector<uint64_t> buf((size_t) chunk_bytes / 8);
  int64_t verified = 0;
  int rc = 0;
  for (int64_t c = 0; c < n_chunks && !rc; c++) {
    if (c % world == rank) {
      for (size_t i = 0; i < buf.size(); i++) buf[i] = (uint64_t) c * 0x9E3779B97F4A7C15ull + i;
      rc = ring.produce((size_t) c, buf.data(), buf.size() * 8, world, timeout_s, stop);
    } else {
      rc = ring.consume((size_t) c, buf.data(), buf.size() * 8, timeout_s, stop);
      for (size_t i = 0; i < buf.size() && !rc; i++)
        if (buf[i] != (uint64_t) c * 0x9E3779B97F4A7C15ull + i) rc = -EILSEQ;
      if (!rc) verified++;
    }
  }
  if (rc) ring.fail_all();
  ring.unmap();
  return rc ? rc : verified;
}

extern "C" int bof_flash_last_launch_mix(uint64_t out[3]) {
  if (!out) return BOF_EINVAL;
  for (int i = 0; i < 3; i++) out[i] = bof::g_mix[i].load();
  return BOF_OK;
}
