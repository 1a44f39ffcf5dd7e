// flash_runtime.cpp -- level 3: flash::gemm / kmeans on FILE-resident matrices through the
// tile cache (reference src/blas/gemm.cpp:27-202, src/blas/kmeans.cpp:27-198 together with the
// scheduler/cache/io_executor they run on: src/scheduler/{scheduler,cache,io_executor}.cpp),
// the entry points that choose between it and the row-panel pipeline (flash_gemm_panels.cpp),
// the schedule dry run, and what the level-3 calls share (stats, release, region transfers).
// The CSR calls are in flash_csr.cpp.
//
// MI355X-first design (not the reference's):
//   * the "program cache" is HBM: a pool of fixed-size device tile slots; the task
//     list is static, so eviction is Belady-optimal (farthest next use) instead of
//     the reference's hash-map-order eviction, and C accumulators never leave HBM
//     during their k-chain (the reference re-reads/re-writes them once the working
//     set exceeds PROGRAM_BUDGET);
//   * host DRAM only holds a small ring of PINNED staging buffers: reader threads do
//     O_DIRECT AIO strided reads into a slot and immediately enqueue
//     hipMemcpyAsync on a dedicated H2D stream; write-back goes D2H on its own
//     stream into a second ring drained by a writer thread: NVMe->host, host->HBM,
//     compute and HBM->host->NVMe all overlap;
//   * everything is event driven (hipEvents + condition variables): no 50-100 ms
//     scheduler ticks (reference scheduler.cpp:92-93,206-212).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "bof_hip.h"
#include "bof_internal.h"
#include "fileio.h"
#include "flash_common.h"
#include "share_ring.h"

namespace bof {

static bof_flash_stats g_last_stats;
static std::mutex g_stats_mu;

int device_ready() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void) hipGetLastError();
    set_error("no HIP device: the flash path has no CPU fallback");
    return BOF_ENODEV;
  }
  return BOF_OK;
}

static std::vector<bof_flash_stats> g_last_dev_stats;
void publish_device_stats(const std::vector<bof_flash_stats> &per_device) {
  std::lock_guard<std::mutex> lk(g_stats_mu);
  g_last_dev_stats = per_device;
}

void publish_stats(const Counters &c, double seconds) {
  std::lock_guard<std::mutex> lk(g_stats_mu);
  g_last_stats.bytes_read = c.rd; g_last_stats.bytes_written = c.wr;
  g_last_stats.bytes_h2d = c.h2d; g_last_stats.bytes_d2h = c.d2h;
  g_last_stats.tasks = c.tasks; g_last_stats.tile_hits = c.hits;
  g_last_stats.tile_misses = c.misses; g_last_stats.seconds = seconds;
  g_last_stats.bytes_peer = c.peer;
  g_last_stats.kernel_launches = c.klaunch;
  g_last_stats.kernel_seconds = (double) c.kns.load() * 1e-9;
  g_last_stats.bytes_p2p = c.p2p;
  g_last_stats.verify_checks = c.vchecks;
  uint64_t r = 0, w = 0;
  file_io_ops(&r, &w);
  g_last_stats.read_ops = r - c.ops0[0]; g_last_stats.write_ops = w - c.ops0[1];
  g_last_dev_stats.assign(1, g_last_stats);   // a multi-device call replaces this right after
}

// =====================================================================================
// GEMM
// =====================================================================================
namespace {

constexpr int kMaxStreams = 16;

// (bof_options.gemm_chain, default) One chain over the whole K: the C tile's slot carries the chain's RAW accumulators
// from task to task (gemm_f32_mfma.hip, ChainEpi) and the last task scales.  With beta != 0 the caller's C is then an
// operand of the LAST task only: it is fetched as a tile of its own kind (mat 3, "C-in": the file region of the C
// tile, read-only, one use) into any free slot shortly before that task, instead of at the chain's start.
struct Tile {
  int mat = 0;                       // 0 A, 1 B, 2 C, 3 C-in (the caller's C, read by the chain's last task)
  int64_t off = 0, nrows = 0, ncols = 0, ld = 0;  // file region in elements
  std::vector<int> uses;             // task positions (execution order) touching this tile
  size_t next_use = 0;               // index into uses of the first not-yet-launched use
  int slot = -1;
  int state = 0;                     // 0 absent, 1 fetch queued, 2 usable (guarded by mu)
  bool pinned_c = false;             // C accumulator in the middle of its chain
  std::vector<hipEvent_t> launch_waits;  // events the first kernel must wait for (no-fetch alloc)
  int rb = 0, cb = 0;                // block row / block column of the STORED matrix
  // BOF_VERIFY entries of the tile's current stay in HBM (Verify::kNone = not checked)
  size_t ve_host = Verify::kNone, ve_in = Verify::kNone;        // pinned chunk after the read / packed slot after H2D
  size_t ve_chost = Verify::kNone, ve_cfile = Verify::kNone;    // C: pinned chunk after D2H / file after the write
  // a slot taken without a fetch (C, beta == 0): what sat there before, summed once more ahead of the first kernel
  size_t prev_ve_in = Verify::kNone;
  int64_t prev_rows = 0, prev_cols = 0;
};

struct DevSlot {
  char *ptr = nullptr;
  int tile = -1;
  hipEvent_t ready = nullptr;                 // H2D into this slot finished
  hipEvent_t use[kMaxStreams + 1] = {};       // last use per compute stream (+1: D2H stream)
  bool used[kMaxStreams + 1] = {};
};

// Pinned rings and the device slab are expensive to create (pinning 0.5 GB takes ~60 ms,
// about as long as the whole 16384^3 product), so they are kept per device between calls,
// like the reference keeps its program cache for the life of the process.
struct GemmResources {
  PinnedRing rring, wring;
  char *slab = nullptr;
  size_t slab_bytes = 0;
  // The per-slot events (18 a slot) and the two copy streams are kept between calls too: a small call used to
  // create and destroy hundreds of events -- per slab -- and under load (16 processes on one GPU) slot tables were
  // found with the low bytes of an event handle overwritten, as by a late write into an object that had been
  // freed and whose memory the table had taken over (profiles/r4/fuzz_crash.md).  Steady state: no HIP object is
  // created or destroyed by a call; bof_flash_release destroys the pool.
  std::vector<hipEvent_t> ev_pool;
  hipStream_t h2d = nullptr, d2h = nullptr;
  int events(size_t n) {       // grow the pool to at least n events
    while (ev_pool.size() < n) {
      hipEvent_t e = nullptr;
      BOF_HIP_TRY(hipEventCreateWithFlags(&e, pooled_event_flags()));
      ev_pool.push_back(e);
    }
    return BOF_OK;
  }
  void destroy_hip_objects() {
    for (hipEvent_t e : ev_pool) (void) hipEventDestroy(e);
    ev_pool.clear();
    if (h2d) (void) hipStreamDestroy(h2d);
    if (d2h) (void) hipStreamDestroy(d2h);
    h2d = d2h = nullptr;
  }
};
static std::mutex g_res_mu;
static GemmResources *g_res[64];

// I/O unit of the tile cache: a ROW GROUP -- up to kMaxGroup horizontally adjacent tiles of one block
// row of a stored matrix.  Its rows are wide, contiguous file extents (the whole stored row when the
// group spans the matrix: then a chunk of rows is ONE contiguous extent), so the requests are
// group-width x rows instead of the reference's 16 KiB tile rows (one iocb per tile row,
// src/file_handles/flash_file_handle.cpp:444-460: 8.4 GB/s on the box's NVMe against 19 GB/s for
// requests of a MiB and more); a chunk of rows lands in one pinned slot and is scattered into the
// packed tile slots by 2-D copies (53 GB/s, profiles/r2/iobench_*), C row groups take the way back.
// The HBM side is untouched: packed tiles, Belady replacement tile by tile.
constexpr int kMaxGroup = 16;
struct RowGroup {
  int mat = 0;
  std::vector<int> tiles, slots;            // ascending block column
  std::vector<std::vector<hipEvent_t>> waits;   // per tile: WAR events of its slot's previous occupant
  int64_t nrows = 0, width = 0, col0 = 0;   // rows of the block row; elements per group row; first stored column
  int64_t rows_per_chunk = 0;
  int remaining = 0;                        // chunks whose copies are not enqueued yet (guarded by mu)
  // BOF_VERIFY: per tile, the previous occupant of its slot (entry of its after-H2D sum, packed shape)
  std::vector<size_t> prev_ve_in;
  std::vector<int64_t> prev_rows, prev_cols;
  std::vector<char> prev_done;              // its after-last-use sum has been queued (guarded by GemmRun::vf_mu)
};
struct FetchReq { RowGroup *grp; int64_t r0, nr; };
struct WTile { size_t ve_chost; int64_t col, ncols; };       // BOF_VERIFY: a C tile's columns inside a write-back chunk
struct WriteReq { int wslot; uint64_t file_off, stride, nrows, len; int64_t r0; std::shared_ptr<std::vector<WTile>> vt; };

// ---- schedule construction and slot replacement: pure host logic, shared by the real
// ---- pipeline and by bof_flash_gemm_simulate (so the policy is testable without a GPU)
static size_t tile_bytes_of(const Tile &t) { return (size_t) t.nrows * t.ncols * sizeof(float); }

static void build_tiles(const GemmGeometry &g, float beta, bool cin, std::vector<Tile> &tiles, size_t &max_tile) {
  const int64_t Nm = g.nblk[0], Nk = g.nblk[1], Nn = g.nblk[2];
  tiles.assign((size_t) (Nm * Nk + Nk * Nn + Nm * Nn * (cin ? 2 : 1)), Tile());
  max_tile = 0;
  bof_gemm_task t;
  for (int64_t l = 0; l < Nk; l++)
    for (int64_t i = 0; i < Nm; i++)
      for (int64_t j = 0; j < Nn; j++) {
        gemm_task_at(g, l, i, j, beta, &t);
        const int ids[3] = {(int) (i * Nk + l), (int) (Nm * Nk + l * Nn + j),
                            (int) (Nm * Nk + Nk * Nn + i * Nn + j)};
        const int64_t idx[3] = {i, l, j};
        for (int x = 0; x < 3; x++) {
          Tile &T = tiles[ids[x]];
          T.mat = x; T.off = t.off[x]; T.nrows = t.nrows[x]; T.ncols = t.ncols[x];
          T.ld = t.ld_file[x];
          T.rb = (int) idx[g.rdim[x]]; T.cb = (int) idx[g.cdim[x]];
          max_tile = std::max(max_tile, tile_bytes_of(T));
        }
        if (cin && l == 0) {          // the same file region once more, as the last task's read-only operand
          Tile &T = tiles[(size_t) (ids[2] + Nm * Nn)];
          T = tiles[(size_t) ids[2]];
          T.mat = 3;
        }
      }
}
// with C-in tiles a chain reads the caller's C at its END: beta != 0, several k-blocks, one chain over the whole K
static bool cin_wanted(const GemmGeometry &g, float beta, bool ref_chain) { return !ref_chain && beta != 0.0f && g.nblk[1] > 1; }

// C super-blocks of gi x gj accumulate chains sized so that the block's C tiles plus two
// generations of its A/B panels fit the slot budget; inside a block tasks go l-major so all
// its chains advance together.  With everything resident this is the reference's order.
// task_tiles: 4 per task -- A, B, C tile ids, then the C-in tile of a chain's last task (or -1)
static void build_order(const GemmGeometry &g, float beta, bool cin, int64_t n_slots, std::vector<Tile> &tiles,
                        std::vector<bof_gemm_task> &tasks, std::vector<int> &task_tiles,
                        int64_t &gi, int64_t &gj) {
  const int64_t Nm = g.nblk[0], Nk = g.nblk[1], Nn = g.nblk[2];
  gi = Nm; gj = Nn;
  if (Nk * Nn + Nn + 2 > n_slots || Nm == 1) {
    while (gi * gj + 2 * (gi + gj) > n_slots && (gi > 1 || gj > 1)) {
      if (gi >= gj && gi > 1) gi--; else gj--;
    }
  } else {
    // All of B fits beside one row of C tiles and an A panel: sweep C in row blocks with B
    // resident.  A and B are still read exactly once, and the chains of a row block finish
    // (and are written back) while the next rows compute -- with a single whole-C block every
    // write-back lands after the last multiply (measured: 60 % of the cfg2 file run).
    const int64_t fit = (n_slots - Nk * Nn) / (Nn + 2);
    gi = std::max<int64_t>(1, std::min(fit, (Nm + 7) / 8));
  }
  // inside a block and a k step the tiles of one STORED row of C follow each other (row-major C: j inner;
  // column-major C: i inner), so finished C tiles leave as row groups and the operand that runs along
  // that row is requested tile after adjacent tile
  const bool c_rows_along_m = g.rdim[2] == 0;
  for (int64_t I0 = 0; I0 < Nm; I0 += gi)
    for (int64_t J0 = 0; J0 < Nn; J0 += gj)
      for (int64_t l = 0; l < Nk; l++) {
        const int64_t ni = std::min(I0 + gi, Nm) - I0, nj = std::min(J0 + gj, Nn) - J0;
        for (int64_t q = 0; q < ni * nj; q++) {
          const int64_t i = I0 + (c_rows_along_m ? q / nj : q % ni), j = J0 + (c_rows_along_m ? q % nj : q / ni);
          bof_gemm_task t;
          gemm_task_at(g, l, i, j, beta, &t);
          const int pos = (int) tasks.size();
          tasks.push_back(t);
          const int ids[4] = {(int) (i * Nk + l), (int) (Nm * Nk + l * Nn + j),
                              (int) (Nm * Nk + Nk * Nn + i * Nn + j),
                              cin && l == Nk - 1 ? (int) (Nm * Nk + Nk * Nn + Nm * Nn + i * Nn + j) : -1};
          for (int x = 0; x < 4; x++) {
            task_tiles.push_back(ids[x]);
            if (ids[x] >= 0) tiles[ids[x]].uses.push_back(pos);
          }
        }
      }
}

// A slot for a tile that is not resident: a free one, else the resident, idle tile whose
// next use lies farthest in the future (Belady) and beyond `horizon` (tiles needed by tasks
// already committed to are not evictable; C accumulators in mid-chain never are).
// Returns -1 when nothing can be evicted right now.
static int claim_slot(std::vector<Tile> &tiles, std::vector<int> &slot_tile, std::vector<int> &free_slots,
                      int horizon) {
  if (!free_slots.empty()) {
    const int sl = free_slots.back();
    free_slots.pop_back();
    return sl;
  }
  int sl = -1;
  int64_t best = -1;
  for (size_t s = 0; s < slot_tile.size(); s++) {
    const int ot = slot_tile[s];
    if (ot < 0) continue;
    const Tile &o = tiles[ot];
    if (o.pinned_c || o.state != 2) continue;
    const int64_t nu = o.next_use < o.uses.size() ? o.uses[o.next_use] : INT64_MAX;
    if (nu <= horizon) continue;
    if (nu > best) { best = nu; sl = (int) s; }
  }
  if (sl >= 0) {
    Tile &o = tiles[slot_tile[sl]];
    o.slot = -1;
    o.state = 0;
  }
  return sl;
}

// tile id of the tile at stored block (rb, cb) of matrix `mat`, or -1
static int tile_at(const GemmGeometry &g, int mat, int rb, int cb) {
  const int64_t Nm = g.nblk[0], Nk = g.nblk[1], Nn = g.nblk[2];
  int64_t idx[3] = {0, 0, 0};
  const int gm = mat == 3 ? 2 : mat;       // a C-in tile lies where its C tile lies
  if (rb < 0 || cb < 0 || rb >= g.nblk[g.rdim[gm]] || cb >= g.nblk[g.cdim[gm]]) return -1;
  idx[g.rdim[gm]] = rb;
  idx[g.cdim[gm]] = cb;
  if (mat == 0) return (int) (idx[0] * Nk + idx[1]);
  if (mat == 1) return (int) (Nm * Nk + idx[1] * Nn + idx[2]);
  return (int) (Nm * Nk + Nk * Nn + (mat == 3 ? Nm * Nn : 0) + idx[0] * Nn + idx[2]);
}

// The row group fetched together with tile `tid`: its absent, still-needed neighbours in the same block
// row, as many as fit into slots that are FREE or hold DEAD tiles (no use left) beyond a small reserve --
// a prefetch never costs a live tile its slot.  C tiles join only when they have to be read (beta != 0,
// chain not started).  Pure host logic: shared with bof_flash_gemm_simulate.
static void select_row_group(const GemmGeometry &g, const std::vector<Tile> &tiles, const std::vector<int> &slot_tile,
                             const std::vector<int> &free_slots, int tid, int max_group, int64_t reach,
                             std::vector<int> &out) {
  out.assign(1, tid);
  if (max_group <= 1) return;
  // a neighbour is read ahead only if its next use lies within `reach` tasks of this tile's: under a
  // tight budget a tile wanted much later would be the replacement policy's next victim (farthest next
  // use) and be read twice
  const int64_t now = tiles[tid].next_use < tiles[tid].uses.size() ? tiles[tid].uses[tiles[tid].next_use] : 0;
  int64_t spare = (int64_t) free_slots.size() - 4;     // this task's own tiles come first
  for (int ot : slot_tile)
    if (ot >= 0 && tiles[ot].state == 2 && !tiles[ot].pinned_c && tiles[ot].next_use >= tiles[ot].uses.size()) spare++;
  const Tile &T = tiles[tid];
  auto eligible = [&](int id) {
    if (id < 0) return false;
    const Tile &S = tiles[id];
    if (S.slot >= 0 || S.state != 0 || S.next_use >= S.uses.size()) return false;
    if (S.mat == 2 && S.next_use != 0) return false;     // a chain that has started holds its tile already
    return (int64_t) S.uses[S.next_use] - now <= reach;
  };
  for (int step = 1; (int64_t) out.size() <= spare && (int) out.size() < max_group; step++) {   // to the right ...
    const int id = tile_at(g, T.mat, T.rb, T.cb + step);
    if (!eligible(id)) break;
    out.push_back(id);
  }
  for (int step = 1; (int64_t) out.size() <= spare && (int) out.size() < max_group; step++) {   // ... then to the left
    const int id = tile_at(g, T.mat, T.rb, T.cb - step);
    if (!eligible(id)) break;
    out.insert(out.begin(), id);
  }
}

static int64_t group_reach_of(const GemmGeometry &g, int64_t n_slots, int64_t n_tiles, int64_t gi, int64_t gj) {
  if (n_slots >= n_tiles) return INT64_MAX;
  const int64_t Nk = g.nblk[1], Nn = g.nblk[2];
  const bool b_resident = !(Nk * Nn + Nn + 2 > n_slots || g.nblk[0] == 1);     // build_order's branches
  return b_resident ? Nk * gi * gj : 2 * gi * gj;
}

static int tile_group_max() {
  const long v = env_long("BOF_TILE_GROUP", kMaxGroup);      // 1: one tile per request, as the reference reads
  return (int) std::max<long>(1, std::min<long>(v, kMaxGroup));
}

struct GemmRun {
  bof_options o;
  GemmGeometry g;
  char ord, ta, tb;
  float alpha, beta;
  bool ref_chain = false;                    // the reference's per-block rounding (bof_options.gemm_chain = 1, kmeans, alpha == 0)
  bool cin = false;                          // C-in tiles: the caller's C is an operand of every chain's last task
  bof_fptr f[3];
  std::vector<Tile> tiles;
  std::vector<bof_gemm_task> tasks;          // execution order
  std::vector<int> task_tiles;               // 4 per task: A, B, C tile ids, the last task's C-in tile or -1
  std::vector<DevSlot> slots;
  std::vector<int> slot_tile;                // tile id per device slot (-1 = empty)
  std::vector<int> free_slots;
  size_t slot_bytes = 0;
  char *slab = nullptr;
  GemmResources *res = nullptr;
  hipStream_t h2d = nullptr, d2h = nullptr;
  StreamSet *ss = nullptr;
  WorkQueue<FetchReq> fetch_q;
  WorkQueue<WriteReq> write_q;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> io_error{0};
  Counters cnt;
  int dev = 0;
  bool use_aio = true;
  // One descriptor mode per file per call: O_DIRECT + AIO only if EVERY tile region of that
  // matrix is sector aligned, else every request of the call goes through the buffered twin.
  // (A direct write and a buffered write of neighbouring tiles must never meet in one page:
  // the case the reference serialises in io_executor.cpp:28-156.)
  int fd_io[3] = {-1, -1, -1};
  bool aio_io[3] = {false, false, false};

  size_t tile_bytes(const Tile &t) const { return (size_t) t.nrows * t.ncols * sizeof(float); }

  // first error wins; stored under the mutex the waiters hold while they test it, so the
  // wake-up cannot fall between their test and their wait
  void fail_io(int code) {
    {
      std::lock_guard<std::mutex> lk(mu);
      int none = 0;
      io_error.compare_exchange_strong(none, code);
    }
    cv.notify_all();
  }

  // file -> pinned slot (a chunk of the group's rows, group-wide) -> 2-D copies into the packed tile slots
  void reader_main() {
    (void) hipSetDevice(dev);
    (void) bind_thread_near_device(dev);
    FetchReq rq;
    while (fetch_q.pop(rq)) {
      RowGroup &G = *rq.grp;
      const Tile &t0 = tiles[G.tiles[0]];
      const int ps = res->rring.acquire();
      const uint64_t row_bytes = (uint64_t) G.width * 4, bytes = row_bytes * (uint64_t) rq.nr;
      int rc = 0;
      const int fm = G.mat == 3 ? 2 : G.mat;       // a C-in tile is read from the C file
      if (!io_error.load())
        rc = file_sread(fd_io[fm], f[fm].foffset + ((uint64_t) t0.off + (uint64_t) rq.r0 * (uint64_t) t0.ld) * 4,
                        (uint64_t) t0.ld * 4, (uint64_t) rq.nr, row_bytes, res->rring.ptr(ps), aio_io[fm]);
      if (rc) fail_io(rc);
      cnt.rd += bytes;
      hipError_t e = hipSuccess;
      int64_t col = 0;
      for (size_t q = 0; q < G.tiles.size() && e == hipSuccess; q++) {
        const Tile &t = tiles[G.tiles[q]];
        DevSlot &s = slots[G.slots[q]];
        for (hipEvent_t w : G.waits[q])  // WAR: previous occupant's kernels / write-back (every chunk: cheap, order-free)
          if (e == hipSuccess) e = wait_event_both(h2d, w);      // (host-confirmed hand-over: flash_common.h)
        const size_t tw = (size_t) t.ncols * 4;
        std::unique_lock<std::mutex> vlk(vf_mu, std::defer_lock);
        if (vf.on && !rc) {
          vf.on_host(t.ve_host, (const char *) res->rring.ptr(ps) + (size_t) col * 4, rq.nr, t.ncols, G.width,
                     (uint64_t) (rq.r0 * t.ncols));
          // self-test of the instrumentation ($BOF_VERIFY_INJECT=1): damage one word between the sum and the copy
          if (t.mat == 1 && rq.r0 == 0 && q == 0 && env_long("BOF_VERIFY_INJECT", 0) == 1)
            ((uint32_t *) ((char *) res->rring.ptr(ps) + (size_t) col * 4))[1] ^= 0x00400000u;
          vlk.lock();      // the slot's old tile is summed before ANY chunk of the new one lands
          if (!G.prev_done[q]) {
            G.prev_done[q] = 1;
            if (e == hipSuccess) e = verify_old_tile(G.prev_ve_in[q], s.ptr, G.prev_rows[q], G.prev_cols[q], h2d, G.tiles[q]);
            // ... and the slot is poisoned (NaN) before the first byte of the new tile: a read ahead of the fill shows
            if (e == hipSuccess) e = vf.poison(s.ptr, tile_bytes(t), h2d);
            // (both were queued by the launcher thread, the copy below comes from this one: host-confirmed)
            if (e == hipSuccess && host_handover()) e = hipStreamSynchronize(h2d);
          }
        }
        if (e == hipSuccess && !rc)
          e = hipMemcpy2DAsync(s.ptr + (size_t) rq.r0 * tw, tw, (const char *) res->rring.ptr(ps) + (size_t) col * 4,
                               (size_t) row_bytes, tw, (size_t) rq.nr, hipMemcpyHostToDevice, h2d);
        if (vlk.owns_lock()) vlk.unlock();
        col += t.ncols;
      }
      if (e == hipSuccess && res->rring.mark_busy(ps, h2d)) e = hipErrorUnknown;   // or the slot would be refilled under the copy
      // host-confirmed hand-over: this thread saw its own copies of the chunk complete before the chunk counts
      if (host_handover() && e == hipSuccess && !rc) e = hipEventSynchronize(res->rring.event(ps));
      cnt.h2d += bytes;
      res->rring.release(ps);
      bool last = false;
      {
        // every chunk's copies are enqueued before its decrement: whoever reaches zero records the tiles'
        // `ready` events behind all of them
        std::lock_guard<std::mutex> lk(mu);
        if (--G.remaining == 0) {
          last = true;
          for (size_t q = 0; q < G.tiles.size(); q++) {
            const Tile &tq = tiles[G.tiles[q]];
            if (e == hipSuccess) e = vf.on_device(tq.ve_in, slots[G.slots[q]].ptr, tq.nrows, tq.ncols, tq.ncols, 0, 0, h2d);
            if (e == hipSuccess) e = hipEventRecord(slots[G.slots[q]].ready, h2d);
          }
          // a failed copy / record must be visible BEFORE the tiles are: the dispatcher tests io_error right
          // after it sees state 2, and a `ready` that was never recorded would make its wait a no-op
          if (e != hipSuccess || rc) { int none = 0; io_error.compare_exchange_strong(none, rc ? rc : -1000 - (int) e); }
          for (size_t q = 0; q < G.tiles.size(); q++) tiles[G.tiles[q]].state = 2;
        } else if (e != hipSuccess) {
          int none = 0;
          io_error.compare_exchange_strong(none, -1000 - (int) e);
        }
      }
      if (e != hipSuccess) fail_io(-1000 - (int) e);
      if (last) delete &G;
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
      if (vf.on && rq.vt && e == hipSuccess)
        for (const WTile &w : *rq.vt)
          vf.on_host(w.ve_chost, (const char *) res->wring.ptr(rq.wslot) + (size_t) w.col * 4, (int64_t) rq.nrows, w.ncols,
                     (int64_t) (rq.len / 4), (uint64_t) (rq.r0 * w.ncols));
      if (vf.on && rq.vt && rq.r0 == 0 && env_long("BOF_VERIFY_INJECT", 0) == 2)
        ((uint32_t *) res->wring.ptr(rq.wslot))[2] ^= 0x00400000u;            // ($BOF_VERIFY_INJECT=2)
      int rc = 0;
      if (!io_error.load())
        rc = file_swrite(fd_io[2], rq.file_off, rq.stride, rq.nrows, rq.len, res->wring.ptr(rq.wslot), aio_io[2]);
      if (rc) fail_io(rc);
      cnt.wr += rq.nrows * rq.len;
      res->wring.release(rq.wslot);
    }
  }

  // a slot for `tid`, with the events of its previous occupant; -1 when nothing can be evicted right now
  int take_slot(int tid, int horizon, std::vector<hipEvent_t> &waits) {
    const int sl = claim_slot(tiles, slot_tile, free_slots, horizon);
    if (sl < 0) return -1;
    cnt.misses++;
    DevSlot &s = slots[sl];
    // BOF_VERIFY: what the slot held (an A / B tile whose image must still be what its H2D copy delivered)
    tiles[tid].prev_ve_in = Verify::kNone;
    if (vf.on && s.tile >= 0 && tiles[s.tile].mat != 2) {
      tiles[tid].prev_ve_in = tiles[s.tile].ve_in;
      tiles[tid].prev_rows = tiles[s.tile].nrows;
      tiles[tid].prev_cols = tiles[s.tile].ncols;
      tiles[s.tile].ve_in = Verify::kNone;
    }
    for (int q = 0; q <= kMaxStreams; q++)
      if (s.used[q]) { waits.push_back(s.use[q]); s.used[q] = false; }
    s.tile = tid;
    slot_tile[sl] = tid;
    tiles[tid].slot = sl;
    return sl;
  }

  // Find a device slot for `tid`; tiles needed by tasks up to `horizon` are not evictable.  With `fetch`
  // the tile's row group (select_row_group) is read with it.  Returns false when nothing can be evicted
  // right now.  Caller holds mu.
  bool make_resident(int tid, int horizon, bool fetch) {
    Tile &t = tiles[tid];
    if (t.slot >= 0) { cnt.hits++; return true; }
    if (!fetch) {
      std::vector<hipEvent_t> waits;
      if (take_slot(tid, horizon, waits) < 0) return false;
      t.state = 2;
      t.launch_waits = waits;
      return true;
    }
    std::vector<int> want;
    select_row_group(g, tiles, slot_tile, free_slots, tid, group_max, group_reach, want);
    // the tile itself first (if even that fails nothing has changed), then its neighbours outwards; the
    // selection only counts free and dead slots, so their claims do not fail -- if one does, the group ends there
    std::vector<hipEvent_t> w0;
    if (take_slot(tid, horizon, w0) < 0) return false;
    RowGroup *G = new RowGroup();
    G->mat = t.mat;
    const size_t at = (size_t) (std::find(want.begin(), want.end(), tid) - want.begin());
    std::vector<std::pair<int, std::vector<hipEvent_t>>> left, right;     // (tile, waits)
    for (size_t q = at + 1; q < want.size(); q++) {
      std::vector<hipEvent_t> w;
      if (take_slot(want[q], horizon, w) < 0) break;
      right.emplace_back(want[q], w);
    }
    for (size_t q = at; q-- > 0;) {
      std::vector<hipEvent_t> w;
      if (take_slot(want[q], horizon, w) < 0) break;
      left.emplace_back(want[q], w);
    }
    auto add = [&](int id, const std::vector<hipEvent_t> &w) {
      tiles[id].state = 1;
      if (tiles[id].mat == 2) tiles[id].pinned_c = true;    // its chain is about to start: not evictable
      G->tiles.push_back(id); G->slots.push_back(tiles[id].slot); G->waits.push_back(w);
      G->prev_ve_in.push_back(tiles[id].prev_ve_in); G->prev_rows.push_back(tiles[id].prev_rows);
      G->prev_cols.push_back(tiles[id].prev_cols); G->prev_done.push_back(0);
      tiles[id].prev_ve_in = Verify::kNone;
      if (vf.on) {
        tiles[id].ve_host = vf.entry();
        tiles[id].ve_in = vf.entry();
        vf.expect(tiles[id].ve_host, tiles[id].ve_in, "tile: pinned chunk after the file read vs packed slot after H2D (tile id, mat)", id,
                  tiles[id].mat);
      }
    };
    for (size_t q = left.size(); q-- > 0;) add(left[q].first, left[q].second);
    add(tid, w0);
    for (auto &r : right) add(r.first, r.second);
    const Tile &first = tiles[G->tiles.front()], &lastt = tiles[G->tiles.back()];
    G->nrows = t.nrows;
    G->col0 = first.off;
    G->width = (lastt.off - first.off) + lastt.ncols;        // adjacent tiles of one block row: contiguous columns
    G->rows_per_chunk = std::max<int64_t>(1, (int64_t) (res->rring.bytes / ((size_t) G->width * 4)));
    const int64_t n_chunks = (G->nrows + G->rows_per_chunk - 1) / G->rows_per_chunk;
    G->remaining = (int) n_chunks;
    for (int64_t c = 0; c < n_chunks; c++) {
      const int64_t r0 = c * G->rows_per_chunk;
      fetch_q.push(FetchReq{G, r0, std::min(G->rows_per_chunk, G->nrows - r0)});
    }
    return true;
  }

  // ---- write-back of finished C tiles, a row group at a time --------------------------------------------
  std::vector<int> wgroup;              // finished, adjacent C tiles of one block row, not flushed yet
  std::vector<int> wgroup_stream;       // the compute stream each one's last task ran on
  // every tile waiting in wgroup must still own its slot (it is pinned until its write-back is queued); checked
  // before each use because a violation used to surface as a crash inside hipEventRecord on a garbage handle
  bool wgroup_sane(const char *where) {
    for (size_t q = 0; q < wgroup.size(); q++) {
      const int id = wgroup[q];
      const bool ok = id >= 0 && id < (int) tiles.size() && tiles[id].mat == 2 && tiles[id].slot >= 0 &&
                      tiles[id].slot < (int) slots.size() && slot_tile[(size_t) tiles[id].slot] == id;
      if (ok) continue;
      char msg[320];
      const Tile *t = id >= 0 && id < (int) tiles.size() ? &tiles[id] : nullptr;
      snprintf(msg, sizeof(msg),
               "bof_flash_gemm (tile cache): internal error at %s: write-back group entry %zu of %zu is tile %d (mat %d, slot %d of %zu, "
               "slot holds tile %d, state %d, pinned %d, next use %zu of %zu)", where, q, wgroup.size(), id, t ? t->mat : -1,
               t ? t->slot : -2, slots.size(),
               t && t->slot >= 0 && t->slot < (int) slots.size() ? slot_tile[(size_t) t->slot] : -2, t ? t->state : -1,
               t ? (int) t->pinned_c : -1, t ? t->next_use : (size_t) 0, t ? t->uses.size() : (size_t) 0);
      fprintf(stderr, "[bof] %s\n", msg);
      evt("tile cache: write-back group insane", id, t ? t->slot : -2, q);
      evt_dump(stderr, "tile cache: write-back group");
      set_error(msg);
      return false;
    }
    return true;
  }
  hipError_t flush_wgroup() {
    if (wgroup.empty()) return hipSuccess;
    if (!wgroup_sane("flush_wgroup entry")) return hipErrorUnknown;
    const Tile &first = tiles[wgroup.front()], &lastt = tiles[wgroup.back()];
    const int64_t width = (lastt.off - first.off) + lastt.ncols, nrows = first.nrows;
    const int64_t rpc = std::max<int64_t>(1, (int64_t) (res->wring.bytes / ((size_t) width * 4)));
    hipError_t e = hipSuccess;
    for (size_t q = 0; q < wgroup.size() && e == hipSuccess; q++)
      e = hipStreamWaitEvent(d2h, slots[tiles[wgroup[q]].slot].use[wgroup_stream[q]], 0);
    std::shared_ptr<std::vector<WTile>> vt;
    if (vf.on) {
      vt = std::make_shared<std::vector<WTile>>();
      int64_t col = 0;
      for (size_t q = 0; q < wgroup.size() && e == hipSuccess; q++) {
        Tile &t = tiles[wgroup[q]];
        const size_t ve_dev = vf.entry();
        t.ve_chost = vf.entry();
        t.ve_cfile = vf.entry();
        vf.expect(ve_dev, t.ve_chost, "C tile: packed slot after its last kernel vs pinned chunk after D2H (tile id)", wgroup[q]);
        vf.expect(t.ve_chost, t.ve_cfile, "C tile: pinned chunk after D2H vs the file after the write (tile id)", wgroup[q]);
        e = vf.on_device(ve_dev, slots[t.slot].ptr, t.nrows, t.ncols, t.ncols, 0, 0, d2h);
        vt->push_back(WTile{t.ve_chost, col, t.ncols});
        col += t.ncols;
      }
    }
    for (int64_t r0 = 0; r0 < nrows && e == hipSuccess; r0 += rpc) {
      const int64_t nr = std::min(rpc, nrows - r0);
      const int ws = res->wring.acquire();
      int64_t col = 0;
      for (size_t q = 0; q < wgroup.size() && e == hipSuccess; q++) {
        const Tile &t = tiles[wgroup[q]];
        const size_t tw = (size_t) t.ncols * 4;
        e = hipMemcpy2DAsync((char *) res->wring.ptr(ws) + (size_t) col * 4, (size_t) width * 4,
                             slots[t.slot].ptr + (size_t) r0 * tw, tw, tw, (size_t) nr, hipMemcpyDeviceToHost, d2h);
        col += t.ncols;
      }
      if (e == hipSuccess) e = hipEventRecord(res->wring.event(ws), d2h);
      if (e != hipSuccess) { res->wring.release(ws); break; }
      cnt.d2h += (uint64_t) nr * (uint64_t) width * 4;
      write_q.push(WriteReq{ws, f[2].foffset + ((uint64_t) first.off + (uint64_t) r0 * (uint64_t) first.ld) * 4,
                            (uint64_t) first.ld * 4, (uint64_t) nr, (uint64_t) width * 4, r0, vt});
    }
    if (e == hipSuccess && !wgroup_sane("flush_wgroup, behind the copies")) e = hipErrorUnknown;
    for (size_t q = 0; q < wgroup.size() && e == hipSuccess; q++) {
      DevSlot &sc = slots[tiles[wgroup[q]].slot];
      e = hipEventRecord(sc.use[kMaxStreams], d2h);
      sc.used[kMaxStreams] = true;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      for (int id : wgroup) tiles[id].pinned_c = false;      // written back (queued): evictable from now on
    }
    wgroup.clear();
    wgroup_stream.clear();
    return e;
  }
  // a chain has finished on stream `sidx`: its C tile joins the row group being collected
  hipError_t finish_c_tile(int tid, int sidx) {
    hipError_t e = hipSuccess;
    if (!wgroup.empty()) {
      const Tile &b = tiles[wgroup.back()], &t = tiles[tid];
      if (t.rb != b.rb || t.cb != b.cb + 1 || (int) wgroup.size() >= group_max) e = flush_wgroup();
    }
    wgroup.push_back(tid);
    wgroup_stream.push_back(sidx);
    return e;
  }
  int group_max = kMaxGroup;
  int64_t group_reach = INT64_MAX;     // how many tasks ahead a neighbour's next use may lie (select_row_group)

  // ---- BOF_VERIFY (flash_common.h) -------------------------------------------------------------------------
  Verify vf;
  std::mutex vf_mu;        // orders "sum the slot's old tile" before the first copy of the new one
  // the packed tile of `rows x cols` in `ptr` has seen its last kernel (st is ordered behind it): sum it once
  // more and expect the sum it had behind its H2D copy
  hipError_t verify_old_tile(size_t ve_in, char *ptr, int64_t rows, int64_t cols, hipStream_t st, int tile_id) {
    if (!vf.on || ve_in == Verify::kNone) return hipSuccess;
    const size_t e = vf.entry();
    vf.expect(ve_in, e, "tile: packed slot after H2D vs the same slot after the tile's last use (tile id)", tile_id);
    return vf.on_device(e, ptr, rows, cols, cols, 0, 0, st);
  }
  int verify_finish() {
    if (!vf.on) return BOF_OK;
    for (size_t sl = 0; sl < slot_tile.size(); sl++) {      // A / B tiles still resident
      const int id = slot_tile[sl];
      if (id < 0 || tiles[id].mat == 2 || tiles[id].state != 2) continue;
      BOF_HIP_TRY(verify_old_tile(tiles[id].ve_in, slots[sl].ptr, tiles[id].nrows, tiles[id].ncols, h2d, id));
    }
    BOF_HIP_TRY(hipStreamSynchronize(h2d));
    // every C tile back from the file, row by row (the tile's rows lie ld apart)
    const int fd = file_is_direct(f[2].fd) ? file_buffered_fd(f[2].fd) : f[2].fd;
    std::vector<char> row;
    for (size_t id = 0; id < tiles.size() && fd >= 0; id++) {
      const Tile &t = tiles[id];
      if (t.mat != 2 || t.ve_cfile == Verify::kNone) continue;
      row.resize((size_t) t.ncols * 4);
      for (int64_t r = 0; r < t.nrows; r++) {
        size_t got = 0;
        while (got < row.size()) {
          const ssize_t n = pread(fd, row.data() + got, row.size() - got,
                                  (off_t) (f[2].foffset + ((uint64_t) t.off + (uint64_t) r * (uint64_t) t.ld) * 4 + got));
          if (n <= 0) { set_error("BOF_VERIFY: re-reading C from its file failed"); return BOF_EIO; }
          got += (size_t) n;
        }
        vf.on_host(t.ve_cfile, row.data(), 1, t.ncols, 0, (uint64_t) (r * t.ncols));
      }
    }
    return vf.finish(cnt, "bof_flash_gemm (tile cache)");
  }
};

}  // namespace

// The tile-cache pipeline on the CURRENT device (the caller holds the device's call lock).
// out: the call's counters (the caller publishes them).  check_only: stop behind the budget check -- a call
// over several devices asks every slab first, so that BOF_ENOMEM comes before any slab has written C.
static int flash_gemm_tilecache(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha,
                                float beta, bof_fptr fa, bof_fptr fb, bof_fptr fc, int64_t lda,
                                int64_t ldb, int64_t ldc, const bof_options &o, const KmeansHost *kh, Counters *out,
                                bool check_only = false) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc;
  evt_mark_call_begin();
  evt("bof_flash_gemm (tile cache) begin", (int) m, (int) n, (uint64_t) k);
  GemmRun R;
  R.o = o;
  R.ord = ord; R.ta = ta; R.tb = tb; R.alpha = alpha; R.beta = beta;
  R.f[0] = fa; R.f[1] = fb; R.f[2] = fc;
  R.use_aio = R.o.use_odirect != 0;
  R.group_max = tile_group_max();
  BOF_HIP_TRY(hipGetDevice(&R.dev));
  R.g = gemm_geometry(ord, ta, tb, m, n, k, lda, ldb, ldc, R.o.gemm_blk);
  const GemmGeometry &g = R.g;
  const int64_t Nm = g.nblk[0], Nk = g.nblk[1], Nn = g.nblk[2];
  if (Nm * Nn == 0) return BOF_OK;
  KmeansVecs kvd{nullptr, nullptr, nullptr};
  const KmeansVecs *kv = nullptr;
  Cleanup kv_guard;
  if (kh && !check_only) {
    rc = kmeans_upload(*kh, &kvd);
    if (rc) return rc;
    kv = &kvd;
    kv_guard.add([&kvd] { (void) hipFree(const_cast<float *>(kvd.c_l2sq)); });
  }

  // ---- tiles ------------------------------------------------------------------------
  size_t max_tile = 0;
  R.ref_chain = kh != nullptr || R.o.gemm_chain == 1 || alpha == 0.0f;
  R.cin = cin_wanted(g, beta, R.ref_chain);
  build_tiles(g, beta, R.cin, R.tiles, max_tile);
  R.slot_bytes = round_up(max_tile, 4096);
  for (int x = 0; x < 3; x++) {
    const uint64_t A = file_is_direct(R.f[x].fd) ? file_dio_align(R.f[x].fd) : 512;
    bool aligned = (R.f[x].foffset % A) == 0;
    for (const Tile &t : R.tiles)
      if ((t.mat == 3 ? 2 : t.mat) == x)
        aligned = aligned && ((uint64_t) t.off * 4) % A == 0 && ((uint64_t) t.ncols * 4) % A == 0 &&
                  (t.nrows <= 1 || ((uint64_t) t.ld * 4) % A == 0);
    R.fd_io[x] = R.f[x].fd;
    R.aio_io[x] = false;
    if (file_is_direct(R.f[x].fd)) {
      if (aligned) R.aio_io[x] = R.use_aio;
      else R.fd_io[x] = file_buffered_fd(R.f[x].fd);
      if (R.fd_io[x] < 0) { set_error("bof_flash_gemm: cannot open a buffered descriptor of an unaligned matrix file"); return BOF_EIO; }
    }
  }

  // ---- HBM budget -> slot count -> C super-block (gi x gj chains per pass) --------------
  // the panel path's cached slots (up to ~80 % of HBM after a large call) are of no use to this
  // call: give them back before the budget is taken from what is free
  panel_resources_release_device(R.dev);
  size_t free_b = 0, total_b = 0;
  BOF_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  {
    std::lock_guard<std::mutex> lk(g_res_mu);
    if (g_res[R.dev & 63]) free_b += g_res[R.dev & 63]->slab_bytes;   // our own slab is re-used / re-sized below
  }
  size_t budget = R.o.hbm_budget > 0 ? (size_t) R.o.hbm_budget : (size_t) (free_b * 0.8);
  budget = std::min(budget, (size_t) (free_b * 0.95));
  int64_t n_slots = (int64_t) (budget / R.slot_bytes);
  // six slots (two tasks' worth; eight with C-in tiles) or, for problems of fewer tiles than that, all of them
  if (n_slots < std::min<int64_t>(R.cin ? 8 : 6, (int64_t) R.tiles.size())) {
    set_error("bof_flash_gemm: HBM budget below 6 tile slots");
    return BOF_ENOMEM;
  }
  if (check_only) return BOF_OK;
  n_slots = std::min<int64_t>(n_slots, (int64_t) R.tiles.size());
  int64_t gi, gj;
  build_order(g, beta, R.cin, n_slots, R.tiles, R.tasks, R.task_tiles, gi, gj);
  // everything resident: read ahead freely; else at most two k steps of the current C block
  // (with B resident beside a row block of C -- build_order's second branch -- the A tiles of the block's
  //  rows are all used within the block: the whole block is within reach)
  R.group_reach = group_reach_of(g, n_slots, (int64_t) R.tiles.size(), gi, gj);
  const int T = (int) R.tasks.size();

  // ---- resources ----------------------------------------------------------------------
  BOF_TRACE_T("plan done");
  {
    std::lock_guard<std::mutex> lk(g_res_mu);
    if (!g_res[R.dev & 63]) g_res[R.dev & 63] = new GemmResources();
    R.res = g_res[R.dev & 63];
  }
  const size_t slab_need = (size_t) n_slots * R.slot_bytes;
  if (R.res->slab_bytes < slab_need) {  // grow-only device slab, kept between calls
    if (R.res->slab) (void) hipFree(R.res->slab);
    R.res->slab = nullptr;
    R.res->slab_bytes = 0;
    BOF_HIP_TRY(hipMalloc((void **) &R.res->slab, slab_need));
    R.res->slab_bytes = slab_need;
  }
  R.slab = R.res->slab;
  BOF_TRACE_T("device slab allocated");
  R.slots.resize((size_t) n_slots);
  R.slot_tile.assign((size_t) n_slots, -1);
  // events and copy streams come from the device's pool (GemmResources): nothing is created or destroyed here
  rc = R.res->events((size_t) n_slots * (kMaxStreams + 2));
  if (rc) return rc;
  for (int64_t s = 0; s < n_slots; s++) {
    R.slots[s].ptr = R.slab + (size_t) s * R.slot_bytes;
    hipEvent_t *pool = R.res->ev_pool.data() + (size_t) s * (kMaxStreams + 2);
    R.slots[s].ready = pool[0];
    for (int q = 0; q <= kMaxStreams; q++) R.slots[s].use[q] = pool[1 + q];
    R.free_slots.push_back((int) (n_slots - 1 - s));
  }
  // A staging slot must hold at least ONE row of the widest row group (up to group_max adjacent tiles): with short,
  // wide tiles (3 x 128 tiles of a 3 x 2000 problem: a group row is 8 KiB, a tile 1.5 KiB) a slot of one tile's size
  // was overrun by the first chunk -- the segmentation faults of the round-4 fuzz (flush_wgroup -> hipMemcpy2DAsync
  // past the end of the pinned block; the block cache's slack hid it most of the time).
  int64_t max_cols = 1;
  for (const Tile &t : R.tiles) max_cols = std::max(max_cols, t.ncols);
  const size_t ring_bytes = std::max(R.slot_bytes, (size_t) round_up((uint64_t) R.group_max * (uint64_t) max_cols * 4, 4096));
  rc = R.res->rring.init(std::max(2, R.o.pinned_slots), ring_bytes);
  if (rc) return rc;
  // write-back ring: a whole C super-block can be in flight, so the dispatcher does not
  // stall at the end of a block while the writers drain the previous one
  // (buffered writes to one file serialise on the inode lock: more than a few writers only
  // add contention -- 4 writers 0.84 s, 8 writers 1.01 s on the cfg2 file run)
  const int n_writers = std::max(2, std::min(4, R.o.n_io_threads / 2));
  rc = R.res->wring.init((int) std::min<int64_t>(gi * gj, 16) + 2, ring_bytes);
  if (rc) return rc;
  if (!R.res->h2d) BOF_HIP_TRY(copy_stream_create(&R.res->h2d));
  if (!R.res->d2h) BOF_HIP_TRY(copy_stream_create(&R.res->d2h));
  R.h2d = R.res->h2d;
  R.d2h = R.res->d2h;
  if (verify_wanted(R.o)) {
    // (per task: consumer-side sums of its tiles, the chain's partial sums behind it and in front of the next; a spot check)
    rc = R.vf.init(R.dev, 16 * R.tiles.size() + 65536 + 10 * R.tasks.size(), R.tasks.size());
    if (rc) return rc;
  }
  R.ss = stream_set(R.o.n_streams);
  if (!R.ss) { set_error("bof_flash_gemm: stream creation failed"); return BOF_EHIP; }

  // $BOF_CRASH_TRACE=1: a fatal signal during this call also prints the slot table and the write-back group
  g_crash_dump_arg.store(&R);
  g_crash_dump_fn.store([](void *p) {
    GemmRun &G = *(GemmRun *) p;
    fprintf(stderr, "[bof] tile cache at the crash: %zu slots, %zu tiles, d2h %p h2d %p, wgroup:", G.slots.size(), G.tiles.size(),
            (void *) G.d2h, (void *) G.h2d);
    for (size_t q = 0; q < G.wgroup.size(); q++)
      fprintf(stderr, " tile %d (slot %d, stream %d)", G.wgroup[q], G.tiles[(size_t) G.wgroup[q]].slot, G.wgroup_stream[q]);
    fprintf(stderr, "\n");
    for (size_t sl = 0; sl < G.slots.size() && sl < 40; sl++) {
      fprintf(stderr, "[bof]   slot %zu: ptr %p tile %d ready %p use:", sl, (void *) G.slots[sl].ptr, G.slots[sl].tile, (void *) G.slots[sl].ready);
      for (int q = 0; q <= kMaxStreams; q++) fprintf(stderr, " %p%s", (void *) G.slots[sl].use[q], G.slots[sl].used[q] ? "*" : "");
      fprintf(stderr, "\n");
    }
  });
  Cleanup crash_dump_off;
  crash_dump_off.add([&R] {
    void *me = &R;
    if (g_crash_dump_arg.compare_exchange_strong(me, nullptr)) g_crash_dump_fn.store(nullptr);
  });
  BOF_TRACE_T("rings/streams ready");
  std::vector<std::thread> readers, writers;
  for (int i = 0; i < std::max(1, R.o.n_io_threads); i++) readers.emplace_back([&R] { R.reader_main(); });
  for (int i = 0; i < n_writers; i++) writers.emplace_back([&R] { R.writer_main(); });

  StallWatch watch("bof_flash_gemm (tile cache)",
                   [&R] { return R.cnt.rd.load() + R.cnt.wr.load() + R.cnt.h2d.load() + R.cnt.d2h.load() + R.cnt.tasks.load(); },
                   [&R] { R.fail_io(-ETIMEDOUT); });
  // ---- dispatch loop ------------------------------------------------------------------
  const int lookahead = std::max(2, R.o.pinned_slots) * 2;
  int fetch_pos = 0;
  hipError_t herr = hipSuccess;
  KernelTimer ktimer;
  ktimer.on = o.kernel_timing > 0;
  const char *where = "bof_flash_gemm dispatch";   // which step of the loop a HIP error came from
  int fail = 0;
  std::map<int, size_t> chain_post;      // BOF_VERIFY: C tile -> sum behind the last task of its chain so far
  for (int t = 0; t < T && !fail; t++) {
    {
      std::lock_guard<std::mutex> lk(R.mu);
      while (fetch_pos < T && fetch_pos <= t + lookahead) {
        const bof_gemm_task &ft = R.tasks[fetch_pos];
        const int *ids = &R.task_tiles[(size_t) fetch_pos * 4];
        bool ok = R.make_resident(ids[0], fetch_pos, true) &&
                  R.make_resident(ids[1], fetch_pos, true);
        if (ok) {
          Tile &C = R.tiles[ids[2]];
          if (C.slot < 0) {
            // the C tile is READ at the chain's start only when its slot is where beta is applied from: the
            // reference's chain, or a chain of one task; with C-in tiles its slot just carries the raw sums
            ok = R.make_resident(ids[2], fetch_pos, ft.beta != 0.0f && !R.cin);
            if (ok) C.pinned_c = true;
          }
        }
        if (ok && ids[3] >= 0) ok = R.make_resident(ids[3], fetch_pos, true);     // the caller's C, for the chain's last task
        if (!ok) break;
        fetch_pos++;
      }
    }
    if (fetch_pos <= t && !R.wgroup.empty()) {
      // finished C tiles waiting for their row group hold slots: write them back and look again
      herr = R.flush_wgroup();
      if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (row-group write-back)"; break; }
      t--;
      continue;
    }
    if (fetch_pos <= t) {
      set_error("bof_flash_gemm: HBM tile budget too small for one task's working set");
      fail = BOF_ENOMEM;
      break;
    }
    const bof_gemm_task &tk = R.tasks[t];
    const int *ids = &R.task_tiles[(size_t) t * 4];
    if (trace_enabled() && (t == 0 || tk.l != R.tasks[t - 1].l)) {
      char lbl[64];
      snprintf(lbl, sizeof(lbl), "task %d (l=%d) next", t, (int) tk.l);
      BOF_TRACE_T(lbl);
    }
    {
      std::unique_lock<std::mutex> lk(R.mu);
      R.cv.wait(lk, [&] {
        return R.io_error.load() || (R.tiles[ids[0]].state == 2 && R.tiles[ids[1]].state == 2 &&
                                     R.tiles[ids[2]].state == 2 && (ids[3] < 0 || R.tiles[ids[3]].state == 2));
      });
    }
    if (R.io_error.load()) { fail = BOF_EIO; break; }
    const int sidx = (int) ((tk.i * Nn + tk.j) % R.ss->n);
    hipStream_t st = R.ss->s[sidx];
    for (int x = 0; x < 4 && herr == hipSuccess; x++) {
      if (ids[x] < 0) continue;
      Tile &tl = R.tiles[ids[x]];
      DevSlot &s = R.slots[tl.slot];
      if (x != 2 || (tk.l == 0 && tk.beta != 0.0f && !R.cin)) herr = hipStreamWaitEvent(st, s.ready, 0);
      for (hipEvent_t w : tl.launch_waits)
        if (herr == hipSuccess) herr = wait_event_both(st, w);
      tl.launch_waits.clear();
      if (tl.prev_ve_in != Verify::kNone && herr == hipSuccess) {   // slot taken without a fetch: its old tile, once more
        herr = R.verify_old_tile(tl.prev_ve_in, s.ptr, tl.prev_rows, tl.prev_cols, st, ids[x]);
        tl.prev_ve_in = Verify::kNone;
      }
    }
    if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (hipStreamWaitEvent)"; break; }
    DevSlot &sa = R.slots[R.tiles[ids[0]].slot], &sb = R.slots[R.tiles[ids[1]].slot],
            &sc = R.slots[R.tiles[ids[2]].slot];
    // packed tiles: leading dim = stored column count (reference gemm.cpp:117-120)
    herr = ktimer.begin(st);
    if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (timing event)"; break; }
    static const long dbg_sync_each = env_long("BOF_DBG_KM_SYNC_EACH", 0);
    if (dbg_sync_each && kv) (void) hipDeviceSynchronize();
    DevSlot *sin = ids[3] >= 0 ? &R.slots[R.tiles[ids[3]].slot] : nullptr;
    // the reference's task (kmeans; gemm_chain = 1; alpha == 0) or a chain of one: scaled into the C tile's slot; else
    // one k-block of ONE chain over the whole K: the C tile's slot carries the raw sums, the last task scales them and
    // adds beta times the caller's C (its C-in tile)
    const bool chain_step = !(R.ref_chain || Nk == 1);
    SpotArgs spa{ord, ta, tb, tk.M, tk.N, tk.K, alpha, (const float *) sa.ptr, tk.ncols[0], (const float *) sb.ptr, tk.ncols[1],
                 chain_step ? beta : tk.beta, (float *) sc.ptr, tk.ncols[2]};
    if (chain_step) {
      if (tk.l > 0) { spa.ch.acc_in = (const float *) sc.ptr; spa.ch.ld_acc = tk.ncols[2]; }
      spa.ch.raw_out = tk.l < Nk - 1;
      if (sin) { spa.ch.c_in = (const float *) sin->ptr; spa.ch.ld_cin = tk.ncols[2]; }
    }
    if (kv) {
      spa.u1 = kv->c_l2sq + tk.i * g.blk[0]; spa.v1 = kv->ones;
      spa.u2 = kv->ones; spa.v2 = kv->p_l2sq + tk.j * g.blk[2];
    }
    spa.seed = ((uint64_t) t << 20) ^ (uint64_t) R.dev;
    Verify::Spot spot;
    const bool c_fetched = tk.l == 0 && tk.beta != 0.0f && !R.cin;     // the C slot holds the caller's C (read from the file)
    if (R.vf.on) {
      // CONSUMER-side sums on the compute stream in front of the launch (flash_common.h, "Round 5")
      DevSlot *opnd[4] = {&sa, &sb, c_fetched ? &sc : nullptr, sin};
      for (int x = 0; x < 4 && herr == hipSuccess; x++) {
        if (!opnd[x] || ids[x] < 0) continue;
        const Tile &tl = R.tiles[ids[x]];
        if (tl.ve_in == Verify::kNone) continue;
        const size_t e1 = R.vf.entry();
        R.vf.expect(tl.ve_in, e1, "tile: packed slot after H2D vs on the COMPUTE stream in front of a task (tile id, task)", ids[x], t);
        herr = R.vf.on_device(e1, opnd[x]->ptr, tl.nrows, tl.ncols, tl.ncols, 0, 0, st);
      }
      if (herr == hipSuccess && tk.l > 0) {
        auto it = chain_post.find(ids[2]);
        if (it != chain_post.end()) {
          const size_t e1 = R.vf.entry();
          R.vf.expect(it->second, e1, "chain: C tile behind a task vs in front of the next task of its chain (tile id, task)", ids[2], t);
          herr = R.vf.on_device(e1, sc.ptr, tk.nrows[2], tk.ncols[2], tk.ncols[2], 0, 0, st);
        }
      }
      if (herr == hipSuccess && tk.l == 0 && !c_fetched)      // a chain that starts from nothing: poison its slot first
        herr = R.vf.poison(sc.ptr, (size_t) tk.nrows[2] * (size_t) tk.ncols[2] * 4, st);
      if (herr == hipSuccess)
        herr = R.vf.spot_before(spa, st, &spot, "task: 64 sampled outputs recomputed vs stored (task, l, C tile id)", t, (int) tk.l, ids[2]);
      if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (BOF_VERIFY in front of the launch)"; break; }
    }
    // self-test of the instrumentation ($BOF_VERIFY_INJECT=3): the second task of the call is dropped
    if (R.vf.on && t == 1 && env_long("BOF_VERIFY_INJECT", 0) == 3) {
    } else
    if (!chain_step)
      herr = tile_sgemm(ord, ta, tb, tk.M, tk.N, tk.K, alpha, (const float *) sa.ptr, tk.ncols[0],
                        (const float *) sb.ptr, tk.ncols[1], tk.beta, (float *) sc.ptr, tk.ncols[2], kv,
                        tk.i * g.blk[0], tk.j * g.blk[2], st);
    else
      herr = sgemm_chain(ord, ta, tb, tk.M, tk.N, tk.K, alpha, (const float *) sa.ptr, tk.ncols[0], (const float *) sb.ptr,
                         tk.ncols[1], beta, (float *) sc.ptr, tk.ncols[2], spa.ch, st);
    if (herr == hipSuccess) herr = ktimer.end(st);
    if (herr == hipSuccess && R.vf.on) {
      herr = R.vf.spot_after(spa, spot, st);
      if (herr == hipSuccess && tk.l < Nk - 1) {
        const size_t e1 = R.vf.entry();
        chain_post[ids[2]] = e1;
        herr = R.vf.on_device(e1, sc.ptr, tk.nrows[2], tk.ncols[2], tk.ncols[2], 0, 0, st);
      }
    }
    if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (tile kernel launch)"; break; }
    R.cnt.tasks++;
    DevSlot *used[4] = {&sa, &sb, &sc, sin};
    for (int x = 0; x < 4 && herr == hipSuccess; x++) {
      if (!used[x]) continue;
      herr = hipEventRecord(used[x]->use[sidx], st);
      used[x]->used[sidx] = true;
    }
    if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (hipEventRecord)"; break; }
    {
      std::lock_guard<std::mutex> lk(R.mu);
      for (int x = 0; x < 4; x++)
        if (ids[x] >= 0) R.tiles[ids[x]].next_use++;
    }
    if (tk.l == Nk - 1) {  // chain finished: the C tile joins its row group; written back, it becomes evictable
      herr = R.finish_c_tile(ids[2], sidx);
      if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (C tile hand-over)"; break; }
      // the last chain of this block row in the current super-block: nothing more will join
      const bool row_done = t + 1 >= T || R.tasks[t + 1].l != Nk - 1 ||
                            R.tiles[R.task_tiles[(size_t) (t + 1) * 4 + 2]].rb != R.tiles[ids[2]].rb;
      if (row_done) herr = R.flush_wgroup();
      if (herr != hipSuccess) { where = "bof_flash_gemm dispatch (row-group write-back)"; break; }
    }
  }
  if (herr == hipSuccess && !fail) {
    herr = R.flush_wgroup();
    if (herr != hipSuccess) where = "bof_flash_gemm dispatch (row-group write-back)";
  }

  // ---- drain ---------------------------------------------------------------------------
  BOF_TRACE_T("all tasks dispatched");
  R.fetch_q.close();
  for (auto &th : readers) th.join();
  R.write_q.close();
  for (auto &th : writers) th.join();
  (void) hipDeviceSynchronize();
  ktimer.collect(R.cnt);
  BOF_TRACE_T("drained (writes done)");
  evt("bof_flash_gemm (tile cache) drained", R.dev, 0, R.cnt.tasks.load());
  if (herr == hipSuccess && !fail && !R.io_error.load()) fail = R.verify_finish();
  evt_dump_env("bof_flash_gemm (tile cache)");
  if (herr != hipSuccess && !fail) fail = hip_fail(herr, where);
  if (R.io_error.load() && (!fail || fail == BOF_EIO)) {
    const int e = R.io_error.load();
    set_error("bof_flash_gemm: I/O pipeline failed: " + io_error_text(e));
    fail = BOF_EIO;
  }
  // the pinned rings and the device slab stay cached for the next call (bof_flash_release)
  BOF_TRACE_T("resources released");
  if (out) {
    out->rd += R.cnt.rd.load(); out->wr += R.cnt.wr.load(); out->h2d += R.cnt.h2d.load(); out->d2h += R.cnt.d2h.load();
    out->tasks += R.cnt.tasks.load(); out->hits += R.cnt.hits.load(); out->misses += R.cnt.misses.load();
    out->klaunch += R.cnt.klaunch.load(); out->kns += R.cnt.kns.load(); out->vchecks += R.cnt.vchecks.load();
  }
  return fail;
}

static bof_flash_stats stats_of(const Counters &c, double seconds) {
  bof_flash_stats s{};
  s.bytes_read = c.rd; s.bytes_written = c.wr; s.bytes_h2d = c.h2d; s.bytes_d2h = c.d2h;
  s.tasks = c.tasks; s.tile_hits = c.hits; s.tile_misses = c.misses; s.seconds = seconds;
  s.kernel_launches = c.klaunch; s.kernel_seconds = (double) c.kns.load() * 1e-9;
  s.verify_checks = c.vchecks;
  return s;
}

// flash::gemm / flash::kmeans on files: the device list, the per-device call locks, then the row-panel
// pipeline (all devices in one hub) or -- small budgets, C rows with gaps -- the tile cache, where
// every device runs the single-device pipeline on its slab of C (no operand is shared there: it is
// the budget-limited path).
static int flash_gemm_impl(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha,
                           float beta, bof_fptr fa, bof_fptr fb, bof_fptr fc, int64_t lda,
                           int64_t ldb, int64_t ldc, const bof_options *opts, const KmeansHost *kh = nullptr) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = device_ready();
  if (rc) return rc;
  const bof_options o = resolved(opts);
  std::vector<int> devs;
  rc = resolve_devices(o, devs);
  if (rc) return rc;
  DeviceCallLock call_lock(devs);
  TraceRange range("bof_flash_gemm");
  dev_cache_release();        // the CSR pipelines' cached HBM blocks (up to 8 GiB a device) are free memory for the budgets below
  if (o.io_request_kib > 0) (void) bof_file_set_request_bytes((uint64_t) o.io_request_kib << 10);
  file_set_engine(o.io_engine);
  const GemmGeometry g = gemm_geometry(ord, ta, tb, m, n, k, lda, ldb, ldc, o.gemm_blk);
  auto nothing_to_do = [&t_begin] {      // the call's statistics are those of a call that moved nothing
    publish_stats(Counters(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    return BOF_OK;
  };
  if (g.nblk[0] * g.nblk[2] == 0) return nothing_to_do();
  // k == 0: the reference's tiler has NUM_B[1] == 0 k-blocks, builds an empty task array, links nothing, hands the
  // scheduler nothing and returns 0 -- the C file is never opened for I/O (src/blas/gemm.cpp:69-75 block counts,
  // :83-129 the task loops, :176-200 add_task / flush).  Same here: no byte of C moves, not even by beta.
  if (g.nblk[1] == 0) return nothing_to_do();

  // ---- large working budgets: whole row panels in file layout, big sequential requests ----
  // One process per GPU with a shared operand (share_world > 1): a rank that cannot do its part -- an error, or a
  // call the panel path cannot take (eligibility depends on THIS rank's free HBM, gemm_path, ldc) -- must not leave
  // its peers waiting for panels it will never publish, nor fall back silently to reading B itself: it raises the
  // ring group's failure word (share_ring.h), which ends every peer's wait at its next check, and reports an error.
  auto tell_peers = [&o] {
    if (o.share_world > 1 && o.share_name[0]) ShareRing::mark_group_failed(std::string(o.share_name));
  };
  if (o.gemm_path != 1) {
    rc = flash_gemm_panels(ord, ta, tb, m, n, k, alpha, beta, fa, fb, fc, lda, ldb, ldc, o, devs, kh);
    if (rc < 0) tell_peers();
    if (rc <= 0) return rc;
    if (o.gemm_path == 2) {
      tell_peers();
      set_error("bof_flash_gemm: gemm_path = 2 (panels) but the call is not eligible: C rows must be "
                "contiguous in the file (ldc = stored width) and B, two A panels and three C panels must "
                "fit hbm_budget");
      return BOF_ENOMEM;
    }
  }

  if (o.share_world > 1) {
    tell_peers();
    set_error("bof_flash_gemm: share_world > 1 (an operand read once per node) needs the row-panel path, and this rank's "
              "call is not eligible for it (gemm_path = 1, C rows not contiguous in the file, or B + the panel rings do "
              "not fit this rank's hbm_budget); the other ranks of the group have been told and fail too");
    return BOF_EINVAL;
  }

  // ---- tile cache ---------------------------------------------------------------------------
  rc = BOF_OK;
  const int dC = g.rdim[2];
  const int64_t NpC = g.nblk[dC];
  const int n_use = (int) std::min<int64_t>((int64_t) devs.size(), NpC);
  struct Slab {
    int dev;
    int64_t m, n;
    bof_fptr f[3];
    KmeansHost kh;
    Counters cnt;
    int rc = 0;
    std::string err;
    double seconds = 0;
  };
  std::vector<std::unique_ptr<Slab>> slabs;
  int64_t p_next = 0;
  for (int d = 0; d < n_use; d++) {
    const int64_t cnt = NpC / n_use + (d < NpC % n_use ? 1 : 0), p0 = p_next;
    p_next += cnt;
    const int64_t e0 = p0 * g.blk[dC], e1 = p0 + cnt == NpC ? g.size[dC] : (p0 + cnt) * g.blk[dC];
    std::unique_ptr<Slab> S(new Slab());
    S->dev = devs[(size_t) d];
    S->m = dC == 0 ? e1 - e0 : m;
    S->n = dC == 2 ? e1 - e0 : n;
    S->f[0] = fa; S->f[1] = fb; S->f[2] = fc;
    S->f[2].foffset += (uint64_t) e0 * (uint64_t) g.ld[2] * 4;
    const int xm = dC == 0 ? 0 : 1;   // the operand that carries the C panel dimension
    S->f[xm].foffset += g.rdim[xm] == dC ? (uint64_t) e0 * (uint64_t) g.ld[xm] * 4 : (uint64_t) e0 * 4;
    if (kh) {
      S->kh = *kh;
      S->kh.m = S->m; S->kh.n = S->n;
      if (dC == 0) S->kh.c_l2sq += e0; else S->kh.p_l2sq += e0;
    }
    slabs.push_back(std::move(S));
  }
  auto run_slab = [&](Slab &S) {
    DeviceScope ds(S.dev);
    S.rc = flash_gemm_tilecache(ord, ta, tb, S.m, S.n, k, alpha, beta, S.f[0], S.f[1], S.f[2], g.ld[0], g.ld[1],
                                g.ld[2], o, kh ? &S.kh : nullptr, &S.cnt);
    if (S.rc) S.err = bof_last_error();
    S.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  };
  if (slabs.size() == 1) {
    run_slab(*slabs[0]);
  } else {
    for (auto &S : slabs) {     // every slab's budget first: BOF_ENOMEM must not leave C half written
      DeviceScope ds(S->dev);
      rc = flash_gemm_tilecache(ord, ta, tb, S->m, S->n, k, alpha, beta, S->f[0], S->f[1], S->f[2], g.ld[0], g.ld[1],
                                g.ld[2], o, nullptr, nullptr, true);
      if (rc) return rc;
    }
    // one thread per distinct ordinal: slabs that share a device (an ordinal listed twice) run one after
    // the other -- they share that device's tile slab and rings
    std::vector<int> ords;
    for (auto &S : slabs)
      if (std::find(ords.begin(), ords.end(), S->dev) == ords.end()) ords.push_back(S->dev);
    // Slabs that share an ordinal run one after the other on that ordinal's launcher: the calling thread for the
    // first ordinal, a PERSISTENT launcher thread for every other one (never a thread made for the call:
    // flash_common.h, "persistent launcher threads").  $BOF_DBG_SLAB_THREAD=1 restores a fresh thread per call --
    // the configuration that produced the wrong tiles -- for the experiment of profiles/r4.
    if (env_long("BOF_DBG_SLAB_THREAD", 0) == 1) {
      std::vector<std::thread> th;
      for (int od : ords)
        th.emplace_back([&, od] {
          for (auto &S : slabs)
            if (S->dev == od) run_slab(*S);
        });
      for (auto &t : th) t.join();
    } else {
      std::vector<std::shared_ptr<LaunchJob>> jobs;
      for (size_t q = 1; q < ords.size(); q++) {
        const int od = ords[q];
        jobs.push_back(launch_async(od, 0, [&, od] {
          for (auto &S : slabs)
            if (S->dev == od) run_slab(*S);
        }));
      }
      for (auto &S : slabs)
        if (S->dev == ords[0]) run_slab(*S);
      for (auto &j : jobs) launch_wait(j);
    }
  }
  Counters total;
  std::vector<bof_flash_stats> per;
  for (auto &S : slabs) {
    total.rd += S->cnt.rd.load(); total.wr += S->cnt.wr.load(); total.h2d += S->cnt.h2d.load(); total.d2h += S->cnt.d2h.load();
    total.tasks += S->cnt.tasks.load(); total.hits += S->cnt.hits.load(); total.misses += S->cnt.misses.load();
    total.klaunch += S->cnt.klaunch.load(); total.kns += S->cnt.kns.load(); total.vchecks += S->cnt.vchecks.load();
    per.push_back(stats_of(S->cnt, S->seconds));
    if (S->rc && !rc) { rc = S->rc; set_error(S->err); }
  }
  total.ops0[0] = slabs[0]->cnt.ops0[0]; total.ops0[1] = slabs[0]->cnt.ops0[1];
  publish_stats(total, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  if (per.size() > 1) publish_device_stats(per);
  return rc;
}

}  // namespace bof

using namespace bof;

extern "C" {

int bof_flash_gemm(char ord, char ta, char tb, uint64_t m, uint64_t n, uint64_t k, float alpha,
                   float beta, bof_fptr a, bof_fptr b, bof_fptr c, uint64_t lda, uint64_t ldb,
                   uint64_t ldc, const bof_options *opts) {
  if (!(ord == 'R' || ord == 'C') || !(ta == 'N' || ta == 'T') || !(tb == 'N' || tb == 'T') ||
      a.fd < 0 || b.fd < 0 || c.fd < 0) {
    set_error("bof_flash_gemm: bad argument");
    return BOF_EINVAL;
  }
  return flash_gemm_impl(ord, ta, tb, (int64_t) m, (int64_t) n, (int64_t) k, alpha, beta, a, b, c,
                         (int64_t) lda, (int64_t) ldb, (int64_t) ldc, opts);
}

// flash::kmeans (reference src/blas/kmeans.cpp:27-198): the gemm tiler with KMeansTask tasks.
// The norm vectors live in host memory as in the reference; they are uploaded once and every
// tile task adds its slices in its store.
int bof_flash_kmeans(char ord, char ta, char tb, uint64_t m, uint64_t n, uint64_t k, float alpha,
                     float beta, bof_fptr a, bof_fptr b, bof_fptr c, uint64_t lda, uint64_t ldb,
                     uint64_t ldc, const float *c_l2sq, const float *p_l2sq, const float *ones,
                     const bof_options *opts) {
  if (!(ord == 'R' || ord == 'C') || !(ta == 'N' || ta == 'T') || !(tb == 'N' || tb == 'T') ||
      a.fd < 0 || b.fd < 0 || c.fd < 0 || !c_l2sq || !p_l2sq || !ones) {
    set_error("bof_flash_kmeans: bad argument");
    return BOF_EINVAL;
  }
  if (m == 0 || n == 0 || k == 0) return BOF_OK;   // nothing to do (the reference divides by zero: kmeans.cpp:52, 76-77)
  const bof_options o = resolved(opts);
  const GemmGeometry g = gemm_geometry(ord, ta, tb, (int64_t) m, (int64_t) n, (int64_t) k, (int64_t) lda,
                                       (int64_t) ldb, (int64_t) ldc, o.gemm_blk);
  // `ones` is indexed by tile-local row and column (kmeans_task.h:74-81 passes it un-offset)
  int64_t n_ones = 0;
  for (int d = 0; d < 3; d += 2) {
    const int64_t last = g.size[d] - (g.nblk[d] - 1) * g.blk[d];
    n_ones = std::max(n_ones, std::max(last, std::min(g.size[d], g.blk[d])));
  }
  // every device of the call gets its own copy of the three vectors (flash_gemm_panels / the tile cache)
  const KmeansHost kh{c_l2sq, p_l2sq, ones, (int64_t) m, (int64_t) n, n_ones};
  return flash_gemm_impl(ord, ta, tb, (int64_t) m, (int64_t) n, (int64_t) k, alpha, beta, a, b, c,
                         (int64_t) lda, (int64_t) ldb, (int64_t) ldc, opts, &kh);
}

int bof_flash_gemm_simulate(char ord, char ta, char tb, uint64_t m, uint64_t n, uint64_t k, float beta,
                            uint64_t lda, uint64_t ldb, uint64_t ldc, int64_t blk, int64_t n_slots,
                            int32_t lookahead, bof_flash_stats *out) {
  if (!(ord == 'R' || ord == 'C') || !(ta == 'N' || ta == 'T') || !(tb == 'N' || tb == 'T') ||
      blk <= 0 || n_slots < 6 || !out) {
    set_error("bof_flash_gemm_simulate: bad argument (n_slots >= 6)");
    return BOF_EINVAL;
  }
  const GemmGeometry g = gemm_geometry(ord, ta, tb, (int64_t) m, (int64_t) n, (int64_t) k,
                                       (int64_t) lda, (int64_t) ldb, (int64_t) ldc, blk);
  memset(out, 0, sizeof(*out));
  if (g.nblk[0] * g.nblk[1] * g.nblk[2] == 0) return BOF_OK;
  std::vector<Tile> tiles;
  std::vector<bof_gemm_task> tasks;
  std::vector<int> task_tiles;
  size_t max_tile = 0;
  const bool cin = cin_wanted(g, beta, false);      // the default arithmetic (bof_options.gemm_chain = 0)
  build_tiles(g, beta, cin, tiles, max_tile);
  n_slots = std::min<int64_t>(n_slots, (int64_t) tiles.size());
  int64_t gi, gj;
  build_order(g, beta, cin, n_slots, tiles, tasks, task_tiles, gi, gj);
  const int64_t group_reach = group_reach_of(g, n_slots, (int64_t) tiles.size(), gi, gj);
  std::vector<int> slot_tile((size_t) n_slots, -1), free_slots;
  for (int64_t s = 0; s < n_slots; s++) free_slots.push_back((int) (n_slots - 1 - s));
  const int T = (int) tasks.size();
  int fetch_pos = 0;
  auto take = [&](int tid, int horizon, bool fetch) {
    const int sl = claim_slot(tiles, slot_tile, free_slots, horizon);
    if (sl < 0) return false;
    out->tile_misses++;
    slot_tile[sl] = tid;
    tiles[tid].slot = sl;
    tiles[tid].state = 2;  // I/O completes instantly in the simulation
    if (fetch) out->bytes_read += tile_bytes_of(tiles[tid]);
    return true;
  };
  const int group_max = tile_group_max();
  auto resident = [&](int tid, int horizon, bool fetch) {   // GemmRun::make_resident without the I/O
    Tile &t = tiles[tid];
    if (t.slot >= 0) { out->tile_hits++; return true; }
    if (!fetch) return take(tid, horizon, false);
    std::vector<int> want;
    select_row_group(g, tiles, slot_tile, free_slots, tid, group_max, group_reach, want);
    if (!take(tid, horizon, true)) return false;
    const size_t at = (size_t) (std::find(want.begin(), want.end(), tid) - want.begin());
    for (size_t q = at + 1; q < want.size(); q++) {
      if (!take(want[q], horizon, true)) break;
      if (tiles[want[q]].mat == 2) tiles[want[q]].pinned_c = true;
    }
    for (size_t q = at; q-- > 0;) {
      if (!take(want[q], horizon, true)) break;
      if (tiles[want[q]].mat == 2) tiles[want[q]].pinned_c = true;
    }
    return true;
  };
  for (int t = 0; t < T; t++) {
    while (fetch_pos < T && fetch_pos <= t + std::max(lookahead, 0)) {
      const int *ids = &task_tiles[(size_t) fetch_pos * 4];
      bool ok = resident(ids[0], fetch_pos, true) && resident(ids[1], fetch_pos, true);
      if (ok && tiles[ids[2]].slot < 0) {
        ok = resident(ids[2], fetch_pos, tasks[fetch_pos].beta != 0.0f && !cin);
        if (ok) tiles[ids[2]].pinned_c = true;
      }
      if (ok && ids[3] >= 0) ok = resident(ids[3], fetch_pos, true);
      if (!ok) break;
      fetch_pos++;
    }
    if (fetch_pos <= t) { set_error("simulate: tile budget too small"); return BOF_ENOMEM; }
    const int *ids = &task_tiles[(size_t) t * 4];
    for (int x = 0; x < 4; x++)
      if (ids[x] >= 0) tiles[ids[x]].next_use++;
    out->tasks++;
    if (tasks[t].l == g.nblk[1] - 1) {
      out->bytes_written += tile_bytes_of(tiles[ids[2]]);
      tiles[ids[2]].pinned_c = false;
    }
  }
  out->bytes_h2d = out->bytes_read;
  out->bytes_d2h = out->bytes_written;
  return BOF_OK;
}

static int region_transfer(bof_fptr f, uint64_t bytes, void *dptr, bool to_device, const bof_options *opts,
                           void *after) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = device_ready();
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> call_lock(device_call_mutex());
  if (f.fd < 0 || (!dptr && bytes)) { set_error("bof_file_to_device / bof_device_to_file: bad argument"); return BOF_EINVAL; }
  const bof_options o = resolved(opts);
  hipStream_t st = nullptr;
  hipEvent_t ev = nullptr;
  BOF_HIP_TRY(pooled_stream(&st, true));
  Cleanup guard;
  guard.add([&] {
    if (ev) pooled_event_return(ev);
    pooled_stream_return(st);
  });
  // the private copy stream is non-blocking: order it behind whatever the caller has queued
  // on `after` for this buffer (an allocator's fill kernel, the kernels that produced it)
  BOF_HIP_TRY(pooled_event(&ev));
  BOF_HIP_TRY(hipEventRecord(ev, (hipStream_t) after));
  BOF_HIP_TRY(hipStreamWaitEvent(st, ev, 0));
  Counters cnt;
  rc = stream_file(f, bytes, (char *) dptr, to_device, st, o.use_odirect != 0, o.n_io_threads, cnt);
  const hipError_t e = hipStreamSynchronize(st);
  if (!rc && e != hipSuccess) rc = hip_fail(e, "region transfer");
  publish_stats(cnt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  return rc;
}
int bof_file_to_device(bof_fptr f, uint64_t bytes, void *dptr, const bof_options *opts, void *stream) {
  return region_transfer(f, bytes, dptr, true, opts, stream);
}
int bof_device_to_file(bof_fptr f, uint64_t bytes, const void *dptr, const bof_options *opts, void *stream) {
  return region_transfer(f, bytes, const_cast<void *>(dptr), false, opts, stream);
}

int bof_flash_release(void) {
  // waits for level-2 / level-3 calls that are running on any device (their slabs, rings and scratch go)
  std::vector<int> all;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) { (void) hipGetLastError(); count = 0; }
  for (int d = 0; d < std::min(count, 64); d++) all.push_back(d);
  DeviceCallLock quiesce(all);
  scratch_release_all();
  panel_resources_release();
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (int d = 0; d < 64; d++) {
    GemmResources *r = g_res[d];
    if (!r) continue;
    {
      DeviceScope ds(d);
      r->rring.destroy();
      r->wring.destroy();
      r->destroy_hip_objects();
      if (r->slab) (void) hipFree(r->slab);
    }
    delete r;
    g_res[d] = nullptr;
  }
  uring_release_buffers();
  dev_cache_release();
  hip_pools_release();
  pinned_cache_release();
  file_unmap_all();
  return BOF_OK;
}

int bof_flash_last_device_stats(bof_flash_stats *out, int cap) {
  std::lock_guard<std::mutex> lk(g_stats_mu);
  const int n = (int) g_last_dev_stats.size();
  for (int i = 0; out && i < std::min(n, cap); i++) out[i] = g_last_dev_stats[(size_t) i];
  return n;
}

int bof_flash_last_stats(bof_flash_stats *out) {
  if (!out) return BOF_EINVAL;
  std::lock_guard<std::mutex> lk(g_stats_mu);
  *out = g_last_stats;
  return BOF_OK;
}

}  // extern "C"
