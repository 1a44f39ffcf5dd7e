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
//   * no packing anywhere: a tile task is `pointer into a panel + leading dimension`, exactly
//     what the level-2 tile DAG passes to the kernel (for a k-contiguous operand: into the
//     panel's k-major copy, made once per panel on the H2D stream -- Mat::kmajor_copy);
//   * C panels are written back as they complete, while the next panels compute.
//
// Schedule (same tasks, same k-order per accumulate chain as src/blas/gemm.cpp:83-129, so
// the result is bit-identical to the tile path): the first `group` C panels are processed
// together, l-major, the others one by one (l-major too).  The first group is the ramp: while the
// resident operand streams in, panel l of it unlocks `group` x (tiles per C panel) tile tasks,
// and `group` is chosen so that those take as long as the panel's read (65536^3 from O_DIRECT
// files: a 1 GiB B panel takes 55 ms to read, the 16 tasks of one C panel 15 ms -- with one C
// panel per group the first panel's time was all I/O: 4.9 s, with a ramp of 4 panels 4.4 s).
// Later panels have everything resident but their own X panel, and finish -- and are written
// back -- one at a time, which keeps the tail after the last kernel to one panel.
// Let D be the dimension along which
// C is paneled (m for row-major C, n for column-major).  The operand that does not contain D
// ("Y": B for row-major) is needed whole by every group and stays resident; the other one
// ("X": A) is resident too when it is paneled along k, else its panels stream through a small
// ring, one group ahead.  If that working set does not fit opts->hbm_budget, or C's rows are
// not contiguous in its file (ldc != stored width: writing whole panels would clobber what
// lies between the rows), the call is handed to the tile cache of flash_runtime.cpp.
//
// HBM: one allocation per panel slot, made in first-use order by an allocator thread while the
// first panels are already being read (a cold 65536^3 call spent 0.85 s in hipMalloc before its
// first read otherwise); slots stay with the device's PanelResources between calls.
//
// Threads: n_io_threads readers (file -> pinned slot -> H2D), the caller as dispatcher (tile
// launches on n_streams compute streams), one flusher (HBM -> pinned, chunk by chunk) and
// writers (pinned -> file).  Everything is ordered by hipEvents and two condition variables;
// nothing polls.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "flash_common.h"

namespace bof {
namespace {

struct Panel {
  int64_t r0 = 0, nr = 0;       // stored rows [r0, r0 + nr)
  uint64_t bytes = 0;           // extent: (nr-1)*ld + cols elements
  int state = 0;                // 0 idle, 1 being read, 2 usable            (guarded by mu)
  int remaining = 0;            // chunks whose H2D copy is not enqueued yet  (guarded by mu)
  bool retired = false;         // consumers launched / write-back enqueued   (guarded by mu)
  hipEvent_t ready = nullptr;   // recorded on the H2D stream behind the panel's last copy
  hipEvent_t d2h_done = nullptr;
  std::vector<hipEvent_t> retire_ev;  // what the slot's next occupant must wait for
};

struct Mat {
  bof_fptr f{-1, 0};
  int fd = -1;                  // descriptor every request of this call uses (O_DIRECT or twin)
  bool aio = false;
  int rdim = 0, cdim = 0;
  int64_t rows = 0, cols = 0, ld = 0, blk_r = 0, blk_c = 0;
  std::vector<Panel> panels;
  bool natural = false;         // whole matrix resident at its file offsets
  int n_slots = 0;
  size_t slot_bytes = 0, total_bytes = 0;
  // One HBM allocation per panel slot (resident matrices: one per panel), made in first-use order
  // by a thread of its own while the first panels are already being read: hipMalloc costs 13-36 ms
  // per GiB and used to sit in front of the first read (0.85 s of a cold 65536^3 call).  A null
  // entry = not allocated yet; entries are written under PanelRun::mu.
  std::vector<char *> *slots = nullptr;
  int slot_of(int p) const { return natural ? p : p % n_slots; }
  char *panel_ptr(int p) const { return (*slots)[(size_t) slot_of(p)]; }
  // k-major copy of a k-contiguous operand panel ([rows][k] -> [k][rows], one per slot), made on
  // the H2D stream behind the panel's last copy: the tile tasks then take the LDS-DMA kernel
  // (148.6 instead of 145.7 TFLOP/s at 4096^3) exactly as bof_gemm_resident arranges it for
  // resident operands; same tiles, same k-order, same bits.
  bool kmajor_copy = false;
  size_t tslot_bytes = 0;
  std::vector<char *> *tslots = nullptr;
  char *tpanel_ptr(int p) const { return (*tslots)[(size_t) slot_of(p)]; }
  uint64_t file_off(int p) const { return f.foffset + (uint64_t) panels[(size_t) p].r0 * (uint64_t) ld * 4; }
};

struct ChunkReq { int mat, panel; uint64_t off, bytes; };
struct WriteReq { int wslot; uint64_t file_off, bytes; int panel; bool last; };

struct PanelResources {
  PinnedRing rring, wring;
  std::vector<char *> slot[3];            // kept between calls
  size_t slot_bytes[3] = {0, 0, 0};
  std::vector<char *> tslot[2];           // k-major copies of operand panels (A, B)
  size_t tslot_bytes[2] = {0, 0};
  size_t held_bytes() const {
    size_t tot = 0;
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
};
std::mutex g_pres_mu;
PanelResources *g_pres[64];

int trace_level() {   // BOF_TRACE=1: dispatcher milestones; 2: + every panel read / flush / write
  static const int lvl = getenv("BOF_TRACE") ? std::max(1, atoi(getenv("BOF_TRACE"))) : 0;
  return lvl;
}
bool trace_on() { return trace_level() >= 1; }

struct PanelRun {
  bof_options o;
  GemmGeometry g;
  Mat mat[3];
  int xmat = 0, ymat = 1;       // streamed-or-resident operand / always-resident operand
  bool c_read = false;
  size_t chunk = 32u << 20;
  PanelResources *res = nullptr;
  hipStream_t h2d = nullptr, d2h = nullptr;
  StreamSet *ss = nullptr;
  WorkQueue<ChunkReq> fetch_q;
  WorkQueue<int> flush_q;
  WorkQueue<WriteReq> write_q;
  std::vector<std::pair<int, int>> order;  // (mat, panel) in order of first use
  std::vector<std::pair<int, int>> alloc_order;  // (mat, slot) still to be allocated, in order of first use
  size_t next_fetch = 0;
  std::vector<std::vector<hipEvent_t>> group_ev;  // per group: one event per compute stream
  std::vector<int> group_of;                      // C panel -> group
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> io_error{0};
  Counters cnt;
  int dev = 0;
  std::chrono::steady_clock::time_point t_begin;

  void trace2(const char *what, int x, int p) const {
    if (trace_level() >= 2) {
      char lbl[64];
      snprintf(lbl, sizeof(lbl), "%s %c%d", what, "ABC"[x], p);
      trace(lbl);
    }
  }
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

  // Push the chunk requests of every panel, in first-use order, whose HBM slot is free: the
  // whole-matrix images always are, a ring slot once its previous occupant has retired.
  // Strictly in order, so the readers always work on what is needed soonest.  Caller holds mu.
  void pump_fetches() {
    while (next_fetch < order.size()) {
      const int x = order[next_fetch].first, p = order[next_fetch].second;
      Mat &M = mat[x];
      Panel &P = M.panels[(size_t) p];
      const int prev = M.natural ? -1 : p - M.n_slots;
      if (prev >= 0 && !M.panels[(size_t) prev].retired) break;
      if (!M.panel_ptr(p) || (M.kmajor_copy && !M.tpanel_ptr(p))) break;   // slot still being allocated (alloc_main pumps again)
      const int n_chunks = (int) ((P.bytes + chunk - 1) / chunk);
      P.state = 1;
      P.remaining = n_chunks;
      for (int c = 0; c < n_chunks; c++) {
        const uint64_t off = (uint64_t) c * chunk;
        fetch_q.push(ChunkReq{x, p, off, std::min<uint64_t>(chunk, P.bytes - off)});
      }
      cnt.misses++;
      next_fetch++;
    }
  }

  // HBM slots in first-use order; every new slot may unblock the next fetch / the dispatcher
  void alloc_main() {
    (void) hipSetDevice(dev);
    for (const auto &as : alloc_order) {
      if (io_error.load()) break;
      TraceRange r("panel slot hipMalloc");
      char *p = nullptr, *tp = nullptr;
      Mat &M = mat[as.first];
      hipError_t e = hipSuccess;
      if (!(*M.slots)[(size_t) as.second]) e = hipMalloc((void **) &p, M.slot_bytes);
      if (e == hipSuccess && M.kmajor_copy && !(*M.tslots)[(size_t) as.second]) e = hipMalloc((void **) &tp, M.tslot_bytes);
      if (e != hipSuccess) { fail_io(-1000 - (int) e); break; }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (tp) (*M.tslots)[(size_t) as.second] = tp;   // before the raw slot: a usable raw slot implies its copy's
        if (p) (*M.slots)[(size_t) as.second] = p;
        pump_fetches();
      }
      cv.notify_all();
    }
    trace("HBM panel slots allocated");
  }

  void reader_main() {
    (void) hipSetDevice(dev);
    (void) bind_thread_near_device(dev);
    ChunkReq rq;
    while (fetch_q.pop(rq)) {
      Mat &M = mat[rq.mat];
      Panel &P = M.panels[(size_t) rq.panel];
      const int ps = res->rring.acquire();
      int rc = 0;
      if (!io_error.load()) {
        TraceRange r("panel chunk read");
        rc = file_sread(M.fd, M.file_off(rq.panel) + rq.off, 0, 1, rq.bytes, res->rring.ptr(ps), M.aio);
      }
      if (rc) fail_io(rc);
      hipError_t e = hipSuccess;
      const int prev = M.natural ? -1 : rq.panel - M.n_slots;
      if (prev >= 0)  // WAR: the slot's previous occupant (its events were recorded before it retired)
        for (hipEvent_t w : M.panels[(size_t) prev].retire_ev)
          if (e == hipSuccess) e = hipStreamWaitEvent(h2d, w, 0);
      if (e == hipSuccess && !rc)
        e = hipMemcpyAsync(M.panel_ptr(rq.panel) + rq.off, res->rring.ptr(ps), rq.bytes, hipMemcpyHostToDevice, h2d);
      if (e == hipSuccess) (void) res->rring.mark_busy(ps, h2d);
      res->rring.release(ps);
      cnt.rd += rq.bytes;
      cnt.h2d += rq.bytes;
      {
        // the copy above is enqueued before this decrement, so whoever brings the count to
        // zero records `ready` behind every copy of the panel
        std::lock_guard<std::mutex> lk(mu);
        if (--P.remaining == 0) {
          if (e == hipSuccess && M.kmajor_copy)
            e = transpose_f32((const float *) M.panel_ptr(rq.panel), M.ld, P.nr, M.cols, (float *) M.tpanel_ptr(rq.panel),
                              P.nr, h2d);
          if (e == hipSuccess) e = hipEventRecord(P.ready, h2d);
          P.state = 2;
          trace2("read + H2D queued:", rq.mat, rq.panel);
        }
      }
      if (e != hipSuccess) fail_io(-1000 - (int) e);
      cv.notify_all();
    }
  }

  // HBM -> pinned ring, chunk by chunk, for every finished C panel; then the panel's slot is
  // free for a later one.
  void flusher_main() {
    (void) hipSetDevice(dev);
    (void) bind_thread_near_device(dev);
    int pc;
    while (flush_q.pop(pc)) {
      Mat &C = mat[2];
      Panel &P = C.panels[(size_t) pc];
      hipError_t e = hipSuccess;
      for (hipEvent_t w : group_ev[(size_t) group_of[(size_t) pc]])
        if (e == hipSuccess) e = hipStreamWaitEvent(d2h, w, 0);
      for (uint64_t off = 0; off < P.bytes && e == hipSuccess && !io_error.load(); off += chunk) {
        const uint64_t len = std::min<uint64_t>(chunk, P.bytes - off);
        const int ws = res->wring.acquire();
        e = hipMemcpyAsync(res->wring.ptr(ws), C.panel_ptr(pc) + off, len, hipMemcpyDeviceToHost, d2h);
        if (e == hipSuccess) e = hipEventRecord(res->wring.event(ws), d2h);
        if (e != hipSuccess) { res->wring.release(ws); break; }
        cnt.d2h += len;
        write_q.push(WriteReq{ws, C.file_off(pc) + off, len, pc, off + len >= P.bytes});
      }
      if (e == hipSuccess) e = hipEventRecord(P.d2h_done, d2h);
      if (e != hipSuccess) fail_io(-1000 - (int) e);
      trace2("D2H queued:", 2, pc);
      {
        std::lock_guard<std::mutex> lk(mu);
        P.retire_ev.assign(1, P.d2h_done);
        P.retired = true;
        pump_fetches();
      }
      cv.notify_all();
    }
  }

  void writer_main() {
    (void) hipSetDevice(dev);
    (void) bind_thread_near_device(dev);
    WriteReq rq;
    while (write_q.pop(rq)) {
      hipError_t e = hipEventSynchronize(res->wring.event(rq.wslot));
      if (e != hipSuccess) fail_io(-1000 - (int) e);
      int rc = 0;
      if (!io_error.load()) {
        TraceRange r("panel chunk write");
        rc = file_swrite(mat[2].fd, rq.file_off, 0, 1, rq.bytes, res->wring.ptr(rq.wslot), mat[2].aio);
      }
      if (rc) fail_io(rc);
      cnt.wr += rq.bytes;
      res->wring.release(rq.wslot);
      if (rq.last) trace2("last chunk written:", 2, rq.panel);
    }
  }
};

// One descriptor mode per file per call: O_DIRECT only if EVERY request of the call is sector
// aligned; otherwise every request goes through the buffered twin.  Mixing the two on one file
// lets a direct write and a buffered write of neighbouring regions meet in one page.
void pick_descriptor(Mat &M, bool use_odirect, size_t chunk) {
  const uint64_t A = file_is_direct(M.f.fd) ? file_dio_align(M.f.fd) : 512;
  bool aligned = (M.f.foffset % A) == 0 && (chunk % A) == 0;
  for (const Panel &P : M.panels)
    aligned = aligned && (P.bytes % A) == 0 && (((uint64_t) P.r0 * (uint64_t) M.ld * 4) % A) == 0;
  M.fd = M.f.fd;
  M.aio = false;
  if (file_is_direct(M.f.fd)) {
    if (aligned && use_odirect) M.aio = true;
    else if (aligned) M.aio = false;               // direct descriptor, synchronous requests
    else M.fd = file_buffered_fd(M.f.fd);
  }
}

}  // namespace

void panel_resources_release() {
  std::lock_guard<std::mutex> lk(g_pres_mu);
  for (int d = 0; d < 64; d++) {
    PanelResources *r = g_pres[d];
    if (!r) continue;
    r->rring.destroy();
    r->wring.destroy();
    for (int x = 0; x < 3; x++) r->drop(x, 0);
    for (int x = 0; x < 2; x++) r->drop_t(x, 0);
    delete r;
    g_pres[d] = nullptr;
  }
}

int flash_gemm_panels(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, float beta,
                      bof_fptr fa, bof_fptr fb, bof_fptr fc, int64_t lda, int64_t ldb, int64_t ldc,
                      const bof_options &o, const KmeansVecs *kv) {
  PanelRun R;
  R.t_begin = std::chrono::steady_clock::now();
  R.o = o;
  R.g = gemm_geometry(ord, ta, tb, m, n, k, lda, ldb, ldc, o.gemm_blk);
  const GemmGeometry &g = R.g;
  const int64_t Nm = g.nblk[0], Nk = g.nblk[1], Nn = g.nblk[2];
  if (Nm * Nn == 0 || Nk == 0) return 1;
  R.chunk = (size_t) std::max(1, o.io_chunk_mib) << 20;
  R.c_read = beta != 0.0f;
  BOF_HIP_TRY(hipGetDevice(&R.dev));

  // ---- budget and layout (plan.cpp: pure host logic, also behind bof_flash_gemm_panel_plan) ------
  size_t free_b = 0, total_b = 0;
  BOF_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  {
    std::lock_guard<std::mutex> lk(g_pres_mu);
    if (!g_pres[R.dev & 63]) g_pres[R.dev & 63] = new PanelResources();
    R.res = g_pres[R.dev & 63];
  }
  free_b += R.res->held_bytes();  // what we already hold counts as free
  size_t budget = o.hbm_budget > 0 ? (size_t) o.hbm_budget : (size_t) (free_b * 0.8);
  budget = std::min(budget, (size_t) (free_b * 0.95));
  const int dC = g.rdim[2];                    // 0: C paneled along m, 2: along n
  R.xmat = dC == 0 ? 0 : 1;
  R.ymat = 1 - R.xmat;
  const int64_t NpC = g.nblk[dC];
  const int64_t Nq = dC == 0 ? Nn : Nm;        // C tiles per panel
  bof_panel_plan plan = plan_panels(g, budget, 1);
  if (!plan.eligible) return 1;
  {
    // size of the ramp group (see the header): read time of one panel of the resident operand
    // over the time of the tile tasks one C panel contributes per such panel.  The rates are
    // assumptions (a datacenter NVMe under O_DIRECT; page cache + PCIe otherwise; the fp32 MFMA
    // tile rate) -- BOF_PANEL_GROUP overrides.
    const char *genv = getenv("BOF_PANEL_GROUP");
    int64_t want = genv ? atoll(genv) : 0;
    if (want <= 0) {
      const double t_io = (double) plan.slot_bytes[R.ymat] / (o.use_odirect ? 16e9 : 40e9);
      const double t_task = std::max(2.0 * (double) g.blk[0] * (double) g.blk[1] * (double) g.blk[2] / 140e12, 15e-6);
      want = (int64_t) std::ceil(t_io / (t_task * (double) Nq));
      // at most half of the C panels: a ramp group's panels all complete -- and start their
      // write-back -- together at its end, and on a device-bound problem (cfg2 from O_DIRECT
      // files: 0.70 s with 2-4 of 8 panels, 0.74-0.80 with 5, 0.76 with 1) that burst must not
      // wait for most of the reads
      want = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, 8), NpC / 2));
    }
    for (int64_t G = std::min(want, NpC); G > 1; G--) {
      const bof_panel_plan p2 = plan_panels(g, budget, G);
      if (p2.eligible) { plan = p2; break; }
    }
  }
  const int64_t group = plan.first_group;
  const bof_fptr fp[3] = {fa, fb, fc};
  for (int x = 0; x < 3; x++) {
    Mat &M = R.mat[x];
    M.f = fp[x];
    M.rdim = g.rdim[x]; M.cdim = g.cdim[x];
    M.rows = g.size[M.rdim]; M.cols = g.size[M.cdim]; M.ld = g.ld[x];
    M.blk_r = g.blk[M.rdim]; M.blk_c = g.blk[M.cdim];
    M.panels.resize((size_t) plan.n_panels[x]);
    for (int64_t p = 0; p < plan.n_panels[x]; p++) {
      Panel &P = M.panels[(size_t) p];
      P.r0 = p * M.blk_r;
      P.nr = (p == plan.n_panels[x] - 1) ? M.rows - P.r0 : M.blk_r;
      P.bytes = ((uint64_t) (P.nr - 1) * (uint64_t) M.ld + (uint64_t) M.cols) * 4;
    }
    M.slot_bytes = (size_t) plan.slot_bytes[x];
    M.total_bytes = (size_t) (((uint64_t) (M.rows - 1) * (uint64_t) M.ld + (uint64_t) M.cols) * 4);
    M.natural = plan.resident[x] != 0;
    M.n_slots = (int) plan.n_slots[x];
  }
  Mat &X = R.mat[R.xmat], &C = R.mat[2];

  // ---- task list in execution order, panels in first-use order ------------------------------
  std::vector<bof_gemm_task> tasks;
  tasks.reserve((size_t) (Nm * Nk * Nn));
  R.group_of.assign((size_t) NpC, 0);
  std::vector<std::vector<char>> seen(3);
  for (int x = 0; x < 3; x++) seen[x].assign(R.mat[x].panels.size(), 0);
  int n_groups = 0;
  std::vector<size_t> group_end;               // task index one past each group
  std::vector<int64_t> gb;                     // first C panel of each group, then NpC
  gb.push_back(0);
  for (int64_t pc = group; pc < NpC; pc++) gb.push_back(pc);
  gb.push_back(NpC);
  for (size_t gx = 0; gx + 1 < gb.size(); gx++, n_groups++) {
    const int64_t G0 = gb[gx], G1 = gb[gx + 1];
    for (int64_t l = 0; l < Nk; l++)
      for (int64_t pc = G0; pc < G1; pc++) {
        R.group_of[(size_t) pc] = n_groups;
        for (int64_t q = 0; q < Nq; q++) {
          const int64_t i = dC == 0 ? pc : q, j = dC == 0 ? q : pc;
          bof_gemm_task t;
          gemm_task_at(g, l, i, j, beta, &t);
          tasks.push_back(t);
          const int64_t idx[3] = {i, l, j};
          for (int x = 0; x < 3; x++) {
            const int p = (int) idx[R.mat[x].rdim];
            if (seen[x][(size_t) p]) continue;
            seen[x][(size_t) p] = 1;
            if (x < 2 || R.c_read) R.order.emplace_back(x, p);
          }
        }
      }
    group_end.push_back(tasks.size());
  }

  // ---- descriptors, HBM, rings, streams, events -----------------------------------------------
  for (int x = 0; x < 3; x++) {
    pick_descriptor(R.mat[x], o.use_odirect != 0, R.chunk);
    if (R.mat[x].fd < 0) { set_error("bof_flash_gemm: cannot open a buffered descriptor of an unaligned matrix file"); return BOF_EIO; }
  }
  for (int x = 0; x < 3; x++) {
    // slots kept from an earlier call are reused when they have this call's size; the missing ones
    // are allocated by alloc_main in first-use order while the pipeline already runs
    Mat &M = R.mat[x];
    if (R.res->slot_bytes[x] != M.slot_bytes) {
      R.res->drop(x, 0);
      R.res->slot_bytes[x] = M.slot_bytes;
    }
    const size_t want = (size_t) (M.natural ? (int) M.panels.size() : M.n_slots);
    if (R.res->slot[x].size() > want) R.res->drop(x, want);
    R.res->slot[x].resize(want, nullptr);
    M.slots = &R.res->slot[x];
  }
  {
    // k-major copies (see Mat::kmajor_copy): for an operand stored k-contiguous whose panels'
    // row counts keep the copy's leading dimension a multiple of 4 (vector loads), whose tiles
    // are each used by >= 4 tasks, and whose copies still fit the budget.  BOF_PANEL_KMAJOR=0
    // turns them off, =2 drops the reuse condition (tests).
    const char *kenv = getenv("BOF_PANEL_KMAJOR");
    const int kmode = kenv ? atoi(kenv) : 1;
    size_t extra = 0;
    for (int x = 0; x < 2; x++) {
      Mat &M = R.mat[x];
      const int64_t reuse = x == R.xmat ? Nq : NpC;
      bool ok = kmode > 0 && M.cdim == 1 && M.cols % 4 == 0 && (kmode > 1 || reuse >= 4);
      int64_t max_nr = 0;
      for (const Panel &P : M.panels) {
        ok = ok && P.nr % 4 == 0;
        max_nr = std::max(max_nr, P.nr);
      }
      M.tslot_bytes = round_up((size_t) max_nr * (size_t) M.cols * 4, 2u << 20);
      const size_t cnt = (size_t) (M.natural ? (int) M.panels.size() : M.n_slots);
      if (ok && plan.need_bytes + extra + cnt * M.tslot_bytes > budget) ok = false;
      M.kmajor_copy = ok;
      if (!ok || R.res->tslot_bytes[x] != M.tslot_bytes) {
        R.res->drop_t(x, 0);
        R.res->tslot_bytes[x] = ok ? M.tslot_bytes : 0;
      }
      if (ok) {
        extra += cnt * M.tslot_bytes;
        if (R.res->tslot[x].size() > cnt) R.res->drop_t(x, cnt);
        R.res->tslot[x].resize(cnt, nullptr);
      }
      M.tslots = &R.res->tslot[x];
    }
  }
  {
    std::vector<std::vector<char>> listed(3);
    for (int x = 0; x < 3; x++) listed[x].assign(R.res->slot[x].size(), 0);
    for (const bof_gemm_task &tk : tasks) {
      const int64_t idx[3] = {tk.i, tk.l, tk.j};
      for (int x = 0; x < 3; x++) {
        const int sl = R.mat[x].slot_of((int) idx[R.mat[x].rdim]);
        if (listed[x][(size_t) sl]) continue;
        listed[x][(size_t) sl] = 1;
        if (!(*R.mat[x].slots)[(size_t) sl] || (R.mat[x].kmajor_copy && !(*R.mat[x].tslots)[(size_t) sl]))
          R.alloc_order.emplace_back(x, sl);
      }
    }
  }
  R.trace("plan");
  Cleanup guard;
  guard.add([&R] {
    for (auto &M : R.mat)
      for (auto &P : M.panels) {
        if (P.ready) (void) hipEventDestroy(P.ready);
        if (P.d2h_done) (void) hipEventDestroy(P.d2h_done);
      }
    for (auto &v : R.group_ev)
      for (hipEvent_t e : v) (void) hipEventDestroy(e);
    if (R.h2d) (void) hipStreamDestroy(R.h2d);
    if (R.d2h) (void) hipStreamDestroy(R.d2h);
  });
  for (int x = 0; x < 3; x++)
    for (auto &P : R.mat[x].panels) {
      BOF_HIP_TRY(hipEventCreateWithFlags(&P.ready, hipEventDisableTiming));
      if (x == 2) BOF_HIP_TRY(hipEventCreateWithFlags(&P.d2h_done, hipEventDisableTiming));
    }
  // A 4096^2 tile launch is 256 workgroups = the whole chip, so more than two compute streams only
  // interleave whole-chip kernels of different chains and starve the copy queues: measured on
  // cfg2 files (page cache) 0.70 s with 4 streams, 0.58 s with 2, 0.59 s with 1
  // (profiles/r2/e2e_sweep_*.txt).  BOF_PANEL_STREAMS overrides.
  const char *senv = getenv("BOF_PANEL_STREAMS");
  R.ss = stream_set(senv && atoi(senv) > 0 ? std::min(atoi(senv), 16) : std::min(o.n_streams, 2));
  if (!R.ss) { set_error("bof_flash_gemm: stream creation failed"); return BOF_EHIP; }
  R.group_ev.assign((size_t) n_groups, std::vector<hipEvent_t>());
  for (auto &v : R.group_ev)
    for (int s = 0; s < R.ss->n; s++) {
      hipEvent_t e;
      BOF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      v.push_back(e);
    }
  int rc = R.res->rring.init(std::max(2, o.pinned_slots), R.chunk);
  if (rc) return rc;
  rc = R.res->wring.init(std::max(2, o.pinned_slots), R.chunk);
  if (rc) return rc;
  BOF_HIP_TRY(copy_stream_create(&R.h2d));
  BOF_HIP_TRY(copy_stream_create(&R.d2h));
  R.trace("rings/streams ready");

  std::vector<std::thread> readers, writers;
  const int n_readers = std::max(1, o.n_io_threads);
  const char *wenv = getenv("BOF_PANEL_WRITERS");
  const int n_writers = wenv && atoi(wenv) > 0 ? atoi(wenv) : std::max(2, std::min(8, o.n_io_threads / 2));
  for (int i = 0; i < n_readers; i++) readers.emplace_back([&R] { R.reader_main(); });
  for (int i = 0; i < n_writers; i++) writers.emplace_back([&R] { R.writer_main(); });
  std::thread flusher([&R] { R.flusher_main(); });
  std::thread allocator([&R] { R.alloc_main(); });
  {
    std::lock_guard<std::mutex> lk(R.mu);
    R.pump_fetches();
  }

  // ---- dispatch ---------------------------------------------------------------------------------
  hipError_t herr = hipSuccess;
  int fail = 0;
  // per (matrix, stream): the panel whose events that stream has already been told to wait for
  std::vector<int> waited((size_t) 3 * (size_t) R.ss->n, -1);
  size_t t = 0;
  for (int gi = 0; gi < n_groups && !fail && herr == hipSuccess; gi++) {
    TraceRange grange("panel group dispatch");
    for (; t < group_end[(size_t) gi]; t++) {
      const bof_gemm_task &tk = tasks[t];
      const int64_t idx[3] = {tk.i, tk.l, tk.j};
      int pn[3];
      for (int x = 0; x < 3; x++) pn[x] = (int) idx[R.mat[x].rdim];
      const int cprev = C.natural ? -1 : pn[2] - C.n_slots;
      {
        std::unique_lock<std::mutex> lk(R.mu);
        R.cv.wait(lk, [&] {
          if (R.io_error.load()) return true;
          if (R.mat[0].panels[(size_t) pn[0]].state != 2 || R.mat[1].panels[(size_t) pn[1]].state != 2) return false;
          if (R.c_read) return C.panels[(size_t) pn[2]].state == 2;
          if (!C.panel_ptr(pn[2])) return false;                  // its HBM slot is still being allocated
          return cprev < 0 || C.panels[(size_t) cprev].retired;   // the slot's write-back is on its way
        });
      }
      if (R.io_error.load()) { fail = BOF_EIO; break; }
      const int64_t q = dC == 0 ? tk.j : tk.i;
      const int sidx = (int) (((int64_t) pn[2] * Nq + q) % R.ss->n);   // chain -> stream: FIFO = parent dependency
      hipStream_t st = R.ss->s[sidx];
      for (int x = 0; x < 3 && herr == hipSuccess; x++) {
        int &w = waited[(size_t) x * (size_t) R.ss->n + (size_t) sidx];
        if (w == pn[x]) continue;
        w = pn[x];
        if (x < 2 || R.c_read) herr = hipStreamWaitEvent(st, R.mat[x].panels[(size_t) pn[x]].ready, 0);
        else if (cprev >= 0)
          for (hipEvent_t e : C.panels[(size_t) cprev].retire_ev)
            if (herr == hipSuccess) herr = hipStreamWaitEvent(st, e, 0);
      }
      if (herr != hipSuccess) break;
      // operand tile: pointer into the panel + the file's leading dimension, or -- with a k-major
      // copy -- into the copy ([k][panel rows]: the tile starts at row k0 of it), the flag flipped
      const float *po[2];
      int64_t ldo[2];
      char flag[2] = {ta, tb};
      for (int x = 0; x < 2; x++) {
        const Mat &M = R.mat[x];
        const int64_t k0 = idx[M.cdim] * M.blk_c;
        if (M.kmajor_copy) {
          const int64_t nr = M.panels[(size_t) pn[x]].nr;
          po[x] = (const float *) M.tpanel_ptr(pn[x]) + k0 * nr;
          ldo[x] = nr;
          flag[x] = flag[x] == 'N' ? 'T' : 'N';
        } else {
          po[x] = (const float *) M.panel_ptr(pn[x]) + k0;
          ldo[x] = M.ld;
        }
      }
      float *pcp = (float *) C.panel_ptr(pn[2]) + idx[C.cdim] * C.blk_c;
      herr = tile_sgemm(ord, flag[0], flag[1], tk.M, tk.N, tk.K, alpha, po[0], ldo[0], po[1], ldo[1], tk.beta, pcp, C.ld,
                        kv, tk.i * g.blk[0], tk.j * g.blk[2], st);
      if (herr != hipSuccess) break;
      R.cnt.tasks++;
    }
    if (fail || herr != hipSuccess) break;
    // group finished on the host side: mark where every stream stands, hand its C panels to the
    // flusher and let the streamed operand's panels of this group go
    for (int s = 0; s < R.ss->n && herr == hipSuccess; s++) herr = hipEventRecord(R.group_ev[(size_t) gi][(size_t) s], R.ss->s[s]);
    if (herr != hipSuccess) break;
    const int64_t G0 = gb[(size_t) gi], G1 = gb[(size_t) gi + 1];
    {
      std::lock_guard<std::mutex> lk(R.mu);
      if (!X.natural)
        for (int64_t pc = G0; pc < G1; pc++) {
          Panel &P = X.panels[(size_t) pc];
          P.retire_ev = R.group_ev[(size_t) gi];
          P.retired = true;
        }
      R.pump_fetches();
    }
    for (int64_t pc = G0; pc < G1; pc++) R.flush_q.push((int) pc);
    if (trace_on()) {
      char lbl[64];
      snprintf(lbl, sizeof(lbl), "group %d dispatched", gi);
      R.trace(lbl);
    }
  }

  // ---- drain ------------------------------------------------------------------------------------
  if (herr != hipSuccess || fail) R.fail_io(-EIO);
  allocator.join();
  R.fetch_q.close();
  for (auto &th : readers) th.join();
  R.flush_q.close();
  flusher.join();
  R.write_q.close();
  for (auto &th : writers) th.join();
  (void) hipDeviceSynchronize();
  R.trace("drained (writes done)");
  if (herr != hipSuccess && !fail) fail = hip_fail(herr, "bof_flash_gemm (panels) dispatch");
  if (R.io_error.load() && (!fail || fail == BOF_EIO)) {
    const int e = R.io_error.load();
    set_error("bof_flash_gemm: I/O pipeline failed: " +
              (e > -1000 ? std::string(strerror(-e)) : "HIP error " + std::to_string(-1000 - e)));
    fail = BOF_EIO;
  }
  R.cnt.hits = 3 * R.cnt.tasks.load() - std::min<uint64_t>(R.cnt.misses.load(), 3 * R.cnt.tasks.load());
  publish_stats(R.cnt, std::chrono::duration<double>(std::chrono::steady_clock::now() - R.t_begin).count());
  return fail;
}

}  // namespace bof
