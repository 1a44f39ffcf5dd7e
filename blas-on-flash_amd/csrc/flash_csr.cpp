// flash_csr.cpp -- level 3 for the CSR calls: flash::csrmm ('N' and 'T', file or host B / C),
// flash::csrgemv and flash::csrcsc on FILE-resident matrices (reference src/blas/csrmm.cpp:64-126,
// 203-266, 355-422, src/blas/csrgemv.cpp:14-97, src/blas/csrcsc.cpp:32-159 with the scheduler /
// cache / io_executor they run on).  Every row block is used exactly once, so the "program cache"
// is a ring of block contexts: reader threads fill a context (file -> pinned -> HBM), the
// dispatcher launches the block's kernel, a pool of retire threads writes C back; B (csrmm) and
// x, y (csrgemv) stay resident in HBM for the call.  The transposition builds A^T in HBM
// (csrcsc_kernels.hip) or, beyond the budget, row block by row block through temporary files.
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
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "bof_hip.h"
#include "bof_internal.h"
#include "fileio.h"
#include "flash_common.h"

namespace bof {

// =====================================================================================
// CSRMM / CSRGEMV: every row block is used exactly once -> a ring of block contexts
// =====================================================================================
namespace {

struct CsrCtx {
  char *d_idx = nullptr, *d_val = nullptr, *d_c = nullptr, *d_c_rm = nullptr;
  char *h_idx = nullptr, *h_val = nullptr, *h_c = nullptr;
  // the block's slice of the row offsets (rows + 1 entries) travels with the block instead of the whole array
  // going to HBM up front (400 MB at the cfg5 size: a staged, synchronous copy out of pageable memory in front
  // of the first read), and csrgemv 'N' results leave block by block the same way (y slices are disjoint)
  char *h_ia = nullptr, *d_ia = nullptr, *h_y = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
  int64_t owner = -1;   // block id this context is reserved for (guarded by mu)
  int state = 0;        // 0 being filled, 1 loaded (H2D enqueued)
};

// HBM -> pageable host memory through pinned chunks: a plain hipMemcpy into pageable memory runs
// at ~5 GB/s on this stack (40 ms for the 200 MB result vector of a cfg5-size csrgemv); here up
// to n_thr threads each pull 8 MiB chunks into a pinned slot and copy them out.
int device_to_pageable(void *dst, const void *src, uint64_t bytes, int n_thr) {
  const uint64_t chunk = 8ull << 20;
  const int64_t nc = (int64_t) ((bytes + chunk - 1) / chunk);
  if (nc <= 2) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
  n_thr = (int) std::max<int64_t>(1, std::min<int64_t>(n_thr, nc));
  std::atomic<int64_t> next{0};
  std::atomic<int> fail{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  auto worker = [&, dev] {
    void *pin = nullptr;
    hipStream_t st = nullptr;
    if (hipSetDevice(dev) != hipSuccess || pooled_stream(&st, true) != hipSuccess) { fail.store(-1); return; }
    if (pinned_alloc(&pin, chunk) != BOF_OK) { fail.store(-1); pooled_stream_return(st); return; }
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= nc || fail.load()) break;
      const uint64_t o = (uint64_t) i * chunk, len = std::min<uint64_t>(chunk, bytes - o);
      if (hipMemcpyAsync(pin, (const char *) src + o, len, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) { fail.store(-1); break; }
      memcpy((char *) dst + o, pin, len);
    }
    pinned_free(pin);
    pooled_stream_return(st);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_thr; t++) th.emplace_back(worker);
  worker();
  for (auto &x : th) x.join();
  return fail.load();
}

// pageable host memory -> HBM through pinned chunks on up to n_thr threads; the copies are queued on streams of
// their own and `after` (the caller's stream) is ordered behind all of them when this returns.
int pageable_to_device(void *dst, const void *src, uint64_t bytes, int n_thr, hipStream_t after) {
  const uint64_t chunk = 8ull << 20;
  const int64_t nc = (int64_t) ((bytes + chunk - 1) / chunk);
  if (nc <= 2) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, after) == hipSuccess ? 0 : -1;
  n_thr = (int) std::max<int64_t>(1, std::min<int64_t>(n_thr, nc));
  std::atomic<int64_t> next{0};
  std::atomic<int> fail{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  auto worker = [&, dev] {
    void *pin[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    if (hipSetDevice(dev) != hipSuccess || pooled_stream(&st, true) != hipSuccess) { fail.store(-1); return; }
    if (pinned_alloc(&pin[0], chunk) != BOF_OK || pinned_alloc(&pin[1], chunk) != BOF_OK) fail.store(-1);
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int q = 0; q < 2 && !fail.load(); q++)
      if (pooled_event(&ev[q]) != hipSuccess) fail.store(-1);
    bool busy[2] = {false, false};
    for (int q = 0; !fail.load(); q ^= 1) {
      const int64_t i = next.fetch_add(1);
      if (i >= nc) break;
      const uint64_t o = (uint64_t) i * chunk, len = std::min<uint64_t>(chunk, bytes - o);
      if (busy[q] && hipEventSynchronize(ev[q]) != hipSuccess) { fail.store(-1); break; }
      memcpy(pin[q], (const char *) src + o, len);
      if (hipMemcpyAsync((char *) dst + o, pin[q], len, hipMemcpyHostToDevice, st) != hipSuccess ||
          hipEventRecord(ev[q], st) != hipSuccess) { fail.store(-1); break; }
      busy[q] = true;
    }
    if (st) (void) hipStreamSynchronize(st);
    for (int q = 0; q < 2; q++) {
      if (ev[q]) pooled_event_return(ev[q]);
      if (pin[q]) pinned_free(pin[q]);
    }
    if (st) pooled_stream_return(st);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_thr; t++) th.emplace_back(worker);
  worker();
  for (auto &x : th) x.join();
  (void) after;      // every copy has completed (the workers synchronised their streams): nothing left to order
  return fail.load();
}

// host arrays that are about to be overwritten whole: no value-initialisation pass
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  template <class U> void construct(U *p) noexcept { ::new ((void *) p) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *) p) U(std::forward<A>(a)...); }
};
using HostI64 = std::vector<int64_t, NoInitAlloc<int64_t>>;

struct CsrRun {
  bof_options o;
  bool is_mm = true;
  char ord_b = 'R', trans = 'N';
  int64_t m = 0, n = 0, k = 0;
  int64_t c_ld = 0;   // rows of the WHOLE column-major C (= m unless this call is one device's row shard)
  float alpha = 1.f, beta = 0.f;
  bof_fptr fa, fja, fb, fc;
  const float *host_b = nullptr;  // csrmm overload with B and C in host memory
  float *host_c = nullptr;
  // A already in HBM (the transposed matrix of csrmm 'T'): 0-based arrays, nothing to read
  const float *res_val = nullptr;
  const int64_t *res_col = nullptr;
  float *host_y = nullptr;       // csrgemv 'N': the caller's result vector, filled block by block (null: one copy at the end)
  const int64_t *ia = nullptr;   // host offsets of the rows of this call (m + 1 entries, absolute)
  HostI64 ia_store;              // ... when this call read them itself
  int64_t *ia_pinned = nullptr;  // ... into a block of the pinned cache
  ~CsrRun() { if (ia_pinned) pinned_free(ia_pinned); }
  std::vector<int64_t> st, sz;
  std::vector<CsrCtx> ctx;
  int depth = 3;
  std::atomic<int64_t> next_blk{0};
  hipStream_t h2d = nullptr, d2h = nullptr;
  WorkQueue<int64_t> done_q;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> io_error{0};
  Counters cnt;
  int dev = 0;
  bool use_aio = true;

  void fail_io(int code) {  // see GemmRun::fail_io
    {
      std::lock_guard<std::mutex> lk(mu);
      int none = 0;
      io_error.compare_exchange_strong(none, code);
    }
    cv.notify_all();
  }

  // the C file is read and written by several threads at once: ONE descriptor mode for every
  // request of the call, as in flash::gemm -- a direct and a buffered write must never meet in one page:
  //  * every block's region sector-aligned: O_DIRECT as it is;
  //  * row-major C with unaligned blocks (k = 100: 400-byte rows): O_DIRECT is KEPT (c_widen) -- a block's whole
  //    pages go out with O_DIRECT straight from the pinned buffer (the D2H copy lands at the block's file offset
  //    modulo the page), the partial first / last page through the buffered twin, where the page two neighbouring
  //    blocks share is merged by the page cache whichever device, thread or process writes it first
  //    (file_write_split); a block's old contents (beta != 0) are read as the sector-aligned superset
  //    (file_read_widened).  The reference: sector RMW with neighbour write ordering,
  //    src/file_handles/flash_file_handle.cpp:558-716, src/scheduler/io_executor.cpp:28-156;
  //  * column-major C with unaligned column pieces, $BOF_UNALIGNED_DIRECT=0, or a device whose O_DIRECT
  //    granularity exceeds a page: the buffered twin for every request.
  int fd_c = -1;
  bool aio_c = false;
  bool c_widen = false;
  static constexpr uint64_t kPage = 4096;
  uint64_t c_off(int64_t b) const { return fc.foffset + (uint64_t) st[b] * (uint64_t) k * 4; }     // row-major C
  uint64_t c_wdelta(int64_t b) const { return c_widen ? c_off(b) % kPage : 0; }     // where the block sits in h_c for its write
  std::atomic<uint64_t> cnt_buffered_c{0};   // bytes of C that went through the buffered twin although the caller's descriptor is O_DIRECT
  uint64_t fsize_ja = 0, fsize_a = 0;
  uint64_t sector = 512;  // the reference widens to SECTOR_LEN = 512; a 4Kn device reports more
  // sector-widened segment of a block (reference csrmm_task.h:156-172), clamped to the
  // end of the file (the last sector of a file is usually partial)
  void seg(int64_t b, int esz, const bof_fptr &f, uint64_t &start, uint64_t &len, uint64_t &delta) const {
    const uint64_t z = (uint64_t) ia[st[b]], nnz = (uint64_t) (ia[st[b] + sz[b]] - ia[st[b]]);
    const uint64_t b0 = f.foffset + z * esz, b1 = b0 + nnz * esz;
    const uint64_t fsize = esz == 8 ? fsize_ja : fsize_a;
    start = b0 / sector * sector;
    uint64_t end = round_up(b1, sector);
    if (fsize && end > fsize) end = std::max(b1, std::min(end, fsize));
    len = nnz ? end - start : 0;
    delta = b0 - start;
  }
  size_t c_bytes(int64_t b) const { return (size_t) sz[b] * k * sizeof(float); }

  void reader_main() {
    (void) hipSetDevice(dev);
    (void) bind_thread_near_device(dev);
    const int64_t nb = (int64_t) st.size();
    for (;;) {
      const int64_t b = next_blk.fetch_add(1);
      if (b >= nb) break;
      CsrCtx &c = ctx[b % depth];
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return c.owner == b || io_error.load(); });
      }
      if (io_error.load()) {
        { std::lock_guard<std::mutex> lk(mu); c.state = 1; }
        cv.notify_all();
        continue;
      }
      uint64_t s0 = 0, l0 = 0, d0 = 0, s1 = 0, l1 = 0, d1 = 0;
      int rc = 0;
      if (!res_val) {
        seg(b, 8, fja, s0, l0, d0);
        seg(b, 4, fa, s1, l1, d1);
        rc = file_sread(fja.fd, s0, 0, 1, l0, c.h_idx, use_aio);
        if (!rc) rc = file_sread(fa.fd, s1, 0, 1, l1, c.h_val, use_aio);
        cnt.rd += l0 + l1;
      }
      hipError_t e = hipSuccess;
      if (!rc && c.h_ia) {
        const size_t ib = (size_t) (sz[b] + 1) * 8;
        memcpy(c.h_ia, ia + st[b], ib);
        e = hipMemcpyAsync(c.d_ia, c.h_ia, ib, hipMemcpyHostToDevice, h2d);
        cnt.h2d += ib;
      }
      if (!rc && !res_val && e == hipSuccess) {
        e = hipMemcpyAsync(c.d_idx, c.h_idx, l0, hipMemcpyHostToDevice, h2d);
        if (e == hipSuccess) e = hipMemcpyAsync(c.d_val, c.h_val, l1, hipMemcpyHostToDevice, h2d);
        cnt.h2d += l0 + l1;
      }
      if (!rc && e == hipSuccess && is_mm && beta != 0.f) {
        uint64_t cdelta = 0;      // (widened read: where the block's first byte landed in h_c)
        // C block: 'R' contiguous rows, 'C' strided columns of the block (packed [k][r])
        if (host_c) {
          if (ord_b == 'R') memcpy(c.h_c, host_c + (size_t) st[b] * k, c_bytes(b));
          else
            for (int64_t j = 0; j < k; j++)
              memcpy(c.h_c + (size_t) j * sz[b] * 4, host_c + (size_t) j * c_ld + st[b], (size_t) sz[b] * 4);
        } else if (ord_b == 'R' && c_widen)
          rc = file_read_widened(fd_c, c_off(b), c_bytes(b), c.h_c, &cdelta, aio_c);
        else if (ord_b == 'R')
          rc = file_sread(fd_c, fc.foffset + (uint64_t) st[b] * k * 4, 0, 1, c_bytes(b), c.h_c, aio_c);
        else
          rc = file_sread(fd_c, fc.foffset + (uint64_t) st[b] * 4, (uint64_t) c_ld * 4, (uint64_t) k,
                          (uint64_t) sz[b] * 4, c.h_c, aio_c);
        cnt.rd += c_bytes(b);
        if (!rc) e = hipMemcpyAsync(c.d_c, c.h_c + cdelta, c_bytes(b), hipMemcpyHostToDevice, h2d);
        cnt.h2d += c_bytes(b);
      }
      if (!rc && e == hipSuccess) e = hipEventRecord(c.ready, h2d);
      // host-confirmed hand-over (flash_common.h): this thread saw its own copies of the block complete before the
      // dispatcher (another thread, another stream) is shown the block
      if (!rc && e == hipSuccess && host_handover()) e = hipEventSynchronize(c.ready);
      // the pinned buffers are reused only after this block retires (owner hand-over),
      // which is after its kernels, which wait for these copies
      if (rc) fail_io(rc);
      if (e != hipSuccess) fail_io(-1000 - (int) e);
      { std::lock_guard<std::mutex> lk(mu); c.state = 1; }
      cv.notify_all();
    }
  }

  // retires a block: waits for its last GPU op, writes C (csrmm), hands its context to block b + depth
  // (run by a small pool: blocks in flight always sit in different contexts)
  void retire_main() {
    (void) hipSetDevice(dev);
    (void) bind_thread_near_device(dev);
    int64_t b;
    while (done_q.pop(b)) {
      CsrCtx &c = ctx[b % depth];
      hipError_t e = hipEventSynchronize(c.done);
      if (e != hipSuccess) fail_io(-1000 - (int) e);
      if (!is_mm && host_y && c.h_y && !io_error.load()) {
        memcpy(host_y + st[b], c.h_y, (size_t) sz[b] * 4);
      } else if (is_mm && !io_error.load() && host_c) {
        if (ord_b == 'R') memcpy(host_c + (size_t) st[b] * k, c.h_c, c_bytes(b));
        else
          for (int64_t j = 0; j < k; j++)
            memcpy(host_c + (size_t) j * c_ld + st[b], c.h_c + (size_t) j * sz[b] * 4, (size_t) sz[b] * 4);
      } else if (is_mm && !io_error.load()) {
        int rc;
        if (ord_b == 'R' && c_widen)
          rc = file_write_split(fd_c, c_off(b), c_bytes(b), c.h_c + c_wdelta(b), aio_c);
        else if (ord_b == 'R')
          rc = file_swrite(fd_c, fc.foffset + (uint64_t) st[b] * k * 4, 0, 1, c_bytes(b), c.h_c, aio_c);
        else
          rc = file_swrite(fd_c, fc.foffset + (uint64_t) st[b] * 4, (uint64_t) c_ld * 4, (uint64_t) k,
                           (uint64_t) sz[b] * 4, c.h_c, aio_c);
        if (rc) fail_io(rc);
        cnt.wr += c_bytes(b);
        if (fd_c != fc.fd) cnt_buffered_c += c_bytes(b);
      }
      { std::lock_guard<std::mutex> lk(mu); c.owner = b + depth; c.state = 0; }
      cv.notify_all();
    }
  }
};

// File -> host array, in 16 MiB pieces taken by up to n_thr threads (one thread reads the page
// cache at ~5 GB/s and first touches the destination's pages alone: the 400 MB of offsets of the
// cfg5-size matrix took 85 ms of a 280 ms csrgemv call).
int read_host(const bof_fptr &f, uint64_t bytes, void *dst, bool use_aio, int n_thr = 8) {
  const uint64_t piece = 16ull << 20;
  const int64_t np = (int64_t) ((bytes + piece - 1) / piece);
  n_thr = (int) std::max<int64_t>(1, std::min<int64_t>(n_thr, np));
  if (n_thr == 1) return file_sread(f.fd, f.foffset, 0, 1, bytes, dst, use_aio);
  std::atomic<int64_t> next{0};
  std::atomic<int> fail{0};
  auto worker = [&] {
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= np || fail.load()) break;
      const uint64_t o = (uint64_t) i * piece, len = std::min<uint64_t>(piece, bytes - o);
      const int io = file_sread(f.fd, f.foffset + o, 0, 1, len, (char *) dst + o, use_aio);
      if (io) { int z = 0; fail.compare_exchange_strong(z, io); }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_thr; t++) th.emplace_back(worker);
  worker();
  for (auto &x : th) x.join();
  return fail.load();
}


}  // namespace

// Whole array <-> file with up to n_thr workers; each owns a 2-slot pinned ring and takes
// 32 MiB chunks off a shared counter.  to_device: file -> pinned -> HBM; else the reverse.
int stream_file(const bof_fptr &f, uint64_t bytes, char *dptr, bool to_device, hipStream_t st,
                bool use_aio, int n_thr, Counters &cnt) {
  if (bytes == 0) return BOF_OK;
  const size_t chunk = (size_t) std::min<uint64_t>(32ull << 20, round_up(bytes, 4096));
  const int64_t nchunks = (int64_t) ((bytes + chunk - 1) / chunk);
  n_thr = (int) std::max<int64_t>(1, std::min<int64_t>(n_thr, nchunks));
  std::atomic<int64_t> next{0};
  std::atomic<int> fail{0};
  int dev = 0;
  BOF_HIP_TRY(hipGetDevice(&dev));
  auto worker = [&, dev] {
    (void) hipSetDevice(dev);
    PinnedRing ring;
    if (ring.init(2, chunk)) { fail.store(-1000); return; }
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= nchunks || fail.load()) break;
      const uint64_t o = (uint64_t) i * chunk, len = std::min<uint64_t>(chunk, bytes - o);
      const int sl = ring.acquire();
      int io = 0;
      hipError_t e = hipSuccess;
      if (to_device) {
        io = file_sread(f.fd, f.foffset + o, 0, 1, len, ring.ptr(sl), use_aio);
        if (!io) e = hipMemcpyAsync(dptr + o, ring.ptr(sl), len, hipMemcpyHostToDevice, st);
        if (!io && e == hipSuccess && ring.mark_busy(sl, st)) e = hipErrorUnknown;
        cnt.rd += len; cnt.h2d += len;
      } else {
        e = hipMemcpyAsync(ring.ptr(sl), dptr + o, len, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && ring.mark_busy(sl, st)) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipEventSynchronize(ring.event(sl));
        if (e == hipSuccess) io = file_swrite(f.fd, f.foffset + o, 0, 1, len, ring.ptr(sl), use_aio);
        cnt.d2h += len; cnt.wr += len;
      }
      ring.release(sl);
      if (io) fail.store(io);
      if (e != hipSuccess) fail.store(-1000 - (int) e);
    }
    ring.destroy();
  };
  std::vector<std::thread> th;
  for (int i = 1; i < n_thr; i++) th.emplace_back(worker);
  worker();
  for (auto &t : th) t.join();
  const int fl = fail.load();
  if (fl) {
    set_error(std::string(to_device ? "loading" : "storing") + " an array failed: " +
              (fl > -1000 ? std::string(strerror(-fl)) : "HIP error " + std::to_string(-1000 - fl)));
    return fl > -1000 ? BOF_EIO : BOF_EHIP;
  }
  return BOF_OK;
}

namespace {

// A^T of a file-resident CSR matrix, built in HBM scratch (SCR_TR_*), offsets also on the host
struct ResidentCsr {
  const float *val = nullptr;
  const int64_t *col = nullptr, *ia_dev = nullptr;
  HostI64 ia_host;
  int64_t nnz = 0;
};

}  // namespace

// Reads CSR(a, ia, ja) (m x n) whole into HBM and transposes it there (csrcsc_kernels.hip).
// The reference does this out of core with per-row-block mkl_csrcsc + a column-block merge
// through temporary files (src/blas/csrcsc.cpp:32-159) because its program cache is 8 GiB of
// DRAM; with 288 GB of HBM the matrices of the BASELINE family (12 GB) fit whole, so the
// transposition is one device-side sort.  Callers that can go out of core (flash::csrcsc) check
// the budget first; here a working set beyond free HBM is refused with BOF_ENOMEM.
static int flash_transpose_to_hbm(int64_t m, int64_t n, bof_fptr fa, bof_fptr fia, bof_fptr fja,
                                  const bof_options &o, Counters &cnt, ResidentCsr &out) {
  const bool use_aio = o.use_odirect != 0;
  std::vector<int64_t> ia((size_t) m + 1, 0);
  if (m > 0) {
    const int io = read_host(fia, (uint64_t) (m + 1) * 8, ia.data(), use_aio);
    if (io) { set_error(std::string("reading ia failed: ") + strerror(-io)); return BOF_EIO; }
    cnt.rd += (uint64_t) (m + 1) * 8;
  }
  const int64_t z = ia[0], nnz = ia[(size_t) m] - z;
  if (nnz < 0) { set_error("csrcsc: offsets are not ascending"); return BOF_EINVAL; }
  out.nnz = nnz;
  const size_t in_bytes = (size_t) nnz * 12 + (size_t) (m + 1) * 8;
  const size_t out_bytes = (size_t) nnz * 12 + (size_t) (n + 1) * 8 + csrcsc_workspace_bytes(n, nnz);
  size_t free_b = 0, total_b = 0;
  dev_cache_release();        // cached call-lifetime blocks of earlier CSR calls are free memory for this check
  BOF_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  if (in_bytes + out_bytes > free_b) {  // scratch that is already allocated only helps
    set_error("csrcsc: the matrix needs " + std::to_string((in_bytes + out_bytes) >> 20) +
              " MiB of HBM, " + std::to_string(free_b >> 20) + " MiB are free (out-of-core transposition "
              "is not implemented)");
    return BOF_ENOMEM;
  }
  void *vt = nullptr, *ct = nullptr, *pt = nullptr;
  int rc = scratch_get(SCR_TR_VAL, (size_t) std::max<int64_t>(nnz, 1) * 4, &vt);
  if (!rc) rc = scratch_get(SCR_TR_COL, (size_t) std::max<int64_t>(nnz, 1) * 8, &ct);
  if (!rc) rc = scratch_get(SCR_TR_PTR, (size_t) (n + 1) * 8, &pt);
  if (rc) return rc;
  char *d_val = nullptr, *d_col = nullptr, *d_ia = nullptr;
  hipStream_t st = nullptr;
  Cleanup guard;
  guard.add([&] {
    (void) hipFree(d_val); (void) hipFree(d_col); (void) hipFree(d_ia);
    if (st) pooled_stream_return(st);
  });
  BOF_HIP_TRY(pooled_stream(&st, false));
  BOF_HIP_TRY(hipMalloc((void **) &d_val, (size_t) std::max<int64_t>(nnz, 1) * 4));
  BOF_HIP_TRY(hipMalloc((void **) &d_col, (size_t) std::max<int64_t>(nnz, 1) * 8));
  BOF_HIP_TRY(hipMalloc((void **) &d_ia, (size_t) (m + 1) * 8));
  BOF_HIP_TRY(hipMemcpyAsync(d_ia, ia.data(), (size_t) (m + 1) * 8, hipMemcpyHostToDevice, st));
  cnt.h2d += (uint64_t) (m + 1) * 8;
  bof_fptr fv = fa, fc = fja;
  fv.foffset += (uint64_t) z * 4;
  fc.foffset += (uint64_t) z * 8;
  rc = stream_file(fv, (uint64_t) nnz * 4, d_val, true, st, use_aio, o.n_io_threads, cnt);
  if (!rc) rc = stream_file(fc, (uint64_t) nnz * 8, d_col, true, st, use_aio, o.n_io_threads, cnt);
  if (rc) return rc;
  void *ws = nullptr;
  if (m > 0 && nnz > 0) {
    rc = scratch_get(SCR_CSRCSC, csrcsc_workspace_bytes(n, nnz), &ws);
    if (rc) return rc;
  }
  BOF_HIP_TRY(scsrcsc(m, n, nnz, (const float *) d_val, (const int64_t *) d_ia, (const int64_t *) d_col,
                      (float *) vt, (int64_t *) pt, (int64_t *) ct, ws, st));
  out.ia_host.resize((size_t) n + 1);
  BOF_HIP_TRY(hipMemcpyAsync(out.ia_host.data(), pt, (size_t) (n + 1) * 8, hipMemcpyDeviceToHost, st));
  BOF_HIP_TRY(hipStreamSynchronize(st));
  cnt.d2h += (uint64_t) (n + 1) * 8;
  out.val = (const float *) vt; out.col = (const int64_t *) ct; out.ia_dev = (const int64_t *) pt;
  return BOF_OK;
}

// An unnamed temporary file next to `near_fd` (falls back to $TMPDIR, /tmp): the reference's
// flash_malloc'ed block files (src/blas/csrcsc.cpp:61-66).
static int temp_file_near(int near_fd) {
  char link[64], path[4096];
  snprintf(link, sizeof(link), "/proc/self/fd/%d", near_fd);
  std::vector<std::string> dirs;
  const ssize_t len = readlink(link, path, sizeof(path) - 1);
  if (len > 0) {
    path[len] = 0;
    std::string d(path);
    const size_t slash = d.rfind('/');
    if (slash != std::string::npos) dirs.push_back(slash ? d.substr(0, slash) : "/");
  }
  if (getenv("TMPDIR")) dirs.push_back(getenv("TMPDIR"));
  dirs.push_back("/tmp");
  for (const auto &d : dirs) {
    const int fd = open(d.c_str(), O_TMPFILE | O_RDWR, 0600);
    if (fd >= 0) return fd;
  }
  return -1;
}

// Out-of-core transposition for matrices whose working set exceeds the HBM budget -- the
// reference's scheme (src/blas/csrcsc.cpp:32-159) with the GPU doing both halves:
//   phase A: row blocks sized to the budget are transposed in HBM (bof::scsrcsc) and their
//            (values, block-local row ids) written to temporary files, offsets kept on the host;
//   phase B: column blocks sized to the budget gather their runs from every row block's file
//            segment and are merged in block order (csc_merge_kernel) into the output files.
static int flash_csrcsc_blocked(int64_t m, int64_t n, const std::vector<int64_t> &ia, bof_fptr fja, bof_fptr fa,
                                bof_fptr fia_tr, bof_fptr fja_tr, bof_fptr fa_tr, const bof_options &o,
                                size_t budget, Counters &cnt) {
  const bool use_aio = o.use_odirect != 0;
  const int64_t z = ia[0], nnz = ia[(size_t) m] - z;
  // per-non-zero HBM cost of a block: input 12 B + output 12 B + sort workspace
  const size_t fixed = (size_t) (n + 1) * 8 * 2 + (1 << 20);
  const size_t per_nnz = 24 + (csrcsc_workspace_bytes(n, 1 << 24) >> 24) + 1;
  if (budget <= fixed + per_nnz * 4096) {
    set_error("csrcsc: HBM budget too small for the out-of-core transposition");
    return BOF_ENOMEM;
  }
  const int64_t blk_nnz = (int64_t) ((budget - fixed) / per_nnz);
  // ---- row blocks ----------------------------------------------------------------------
  std::vector<int64_t> rb;  // block boundaries (rows)
  rb.push_back(0);
  while (rb.back() < m) {
    const int64_t r0 = rb.back();
    int64_t r1 = std::upper_bound(ia.begin() + r0 + 1, ia.begin() + m + 1, ia[(size_t) r0] + blk_nnz) - ia.begin() - 1;
    if (r1 <= r0) {
      set_error("csrcsc: one row of the matrix exceeds the HBM budget");
      return BOF_ENOMEM;
    }
    rb.push_back(std::min(r1, m));
  }
  const int nb = (int) rb.size() - 1;
  const int tfd_val = temp_file_near(fa_tr.fd), tfd_col = temp_file_near(fja_tr.fd);
  hipStream_t st = nullptr;
  char *d_val = nullptr, *d_col = nullptr, *d_ia = nullptr, *d_vt = nullptr, *d_ct = nullptr, *d_pt = nullptr;
  char *d_aux = nullptr;
  Cleanup guard;
  guard.add([&] {
    if (tfd_val >= 0) close(tfd_val);
    if (tfd_col >= 0) close(tfd_col);
    (void) hipFree(d_val); (void) hipFree(d_col); (void) hipFree(d_ia); (void) hipFree(d_vt); (void) hipFree(d_ct);
    (void) hipFree(d_pt); (void) hipFree(d_aux);
    if (st) pooled_stream_return(st);
  });
  if (tfd_val < 0 || tfd_col < 0) { set_error("csrcsc: cannot create temporary files"); return BOF_EIO; }
  BOF_HIP_TRY(pooled_stream(&st, false));
  int64_t max_b = 0, max_rows = 0;
  for (int b = 0; b < nb; b++) {
    max_b = std::max(max_b, ia[(size_t) rb[b + 1]] - ia[(size_t) rb[b]]);
    max_rows = std::max(max_rows, rb[b + 1] - rb[b]);
  }
  const size_t cap = (size_t) std::max<int64_t>(max_b, 1);
  BOF_HIP_TRY(hipMalloc((void **) &d_val, cap * 4));
  BOF_HIP_TRY(hipMalloc((void **) &d_col, cap * 8));
  BOF_HIP_TRY(hipMalloc((void **) &d_vt, cap * 4));
  BOF_HIP_TRY(hipMalloc((void **) &d_ct, cap * 8));
  BOF_HIP_TRY(hipMalloc((void **) &d_ia, (size_t) (max_rows + 1) * 8));
  BOF_HIP_TRY(hipMalloc((void **) &d_pt, (size_t) (n + 1) * 8));
  void *ws = nullptr;
  int rc = scratch_get(SCR_CSRCSC, csrcsc_workspace_bytes(n, (int64_t) cap), &ws);
  if (rc) return rc;
  std::vector<std::vector<int64_t>> bptr((size_t) nb, std::vector<int64_t>((size_t) n + 1));
  std::vector<int64_t> ia_tr((size_t) n + 1, 0);
  for (int b = 0; b < nb; b++) {
    const int64_t r0 = rb[b], r1 = rb[b + 1], zb = ia[(size_t) r0] - z, nb_nnz = ia[(size_t) r1] - ia[(size_t) r0];
    BOF_HIP_TRY(hipMemcpyAsync(d_ia, ia.data() + r0, (size_t) (r1 - r0 + 1) * 8, hipMemcpyHostToDevice, st));
    bof_fptr fv = fa, fc = fja;
    fv.foffset += (uint64_t) (z + zb) * 4;
    fc.foffset += (uint64_t) (z + zb) * 8;
    rc = stream_file(fv, (uint64_t) nb_nnz * 4, d_val, true, st, use_aio, o.n_io_threads, cnt);
    if (!rc) rc = stream_file(fc, (uint64_t) nb_nnz * 8, d_col, true, st, use_aio, o.n_io_threads, cnt);
    if (rc) return rc;
    BOF_HIP_TRY(scsrcsc(r1 - r0, n, nb_nnz, (const float *) d_val, (const int64_t *) d_ia, (const int64_t *) d_col,
                        (float *) d_vt, (int64_t *) d_pt, (int64_t *) d_ct, ws, st));
    BOF_HIP_TRY(hipMemcpyAsync(bptr[(size_t) b].data(), d_pt, (size_t) (n + 1) * 8, hipMemcpyDeviceToHost, st));
    BOF_HIP_TRY(hipStreamSynchronize(st));
    bof_fptr tv{tfd_val, (uint64_t) zb * 4}, tc{tfd_col, (uint64_t) zb * 8};
    rc = stream_file(tv, (uint64_t) nb_nnz * 4, d_vt, false, st, false, o.n_io_threads, cnt);
    if (!rc) rc = stream_file(tc, (uint64_t) nb_nnz * 8, d_ct, false, st, false, o.n_io_threads, cnt);
    if (rc) return rc;
    for (int64_t c = 0; c < n; c++) ia_tr[(size_t) c + 1] += bptr[(size_t) b][(size_t) c + 1] - bptr[(size_t) b][(size_t) c];
    cnt.tasks++;
  }
  for (int64_t c = 0; c < n; c++) ia_tr[(size_t) c + 1] += ia_tr[(size_t) c];
  if (ia_tr[(size_t) n] != nnz) { set_error("csrcsc: block transposes lost entries"); return BOF_EHIP; }
  // ---- column blocks: gather the runs of every row block, merge, write ----------------------
  (void) hipFree(d_ia); d_ia = nullptr;
  (void) hipFree(d_pt); d_pt = nullptr;
  int64_t c0 = 0;
  while (c0 < n) {
    // largest c1 with nnz(c0..c1) <= cap and auxiliary arrays within reason
    int64_t c1 = std::upper_bound(ia_tr.begin() + c0 + 1, ia_tr.begin() + n + 1, ia_tr[(size_t) c0] + (int64_t) cap) -
                 ia_tr.begin() - 1;
    c1 = std::min<int64_t>(std::max(c1, c0 + 1), n);
    c1 = std::min<int64_t>(c1, c0 + std::max<int64_t>(1, (int64_t) (64 << 20) / (nb + 1)));
    const int64_t cw = c1 - c0, out_nnz = ia_tr[(size_t) c1] - ia_tr[(size_t) c0];
    if (out_nnz > (int64_t) cap) { set_error("csrcsc: one column of the matrix exceeds the HBM budget"); return BOF_ENOMEM; }
    // auxiliary arrays: nb x (cw+1) block offsets (relative to the block's segment), nb bases,
    // nb first rows, cw+1 output offsets
    std::vector<int64_t> aux((size_t) nb * (size_t) (cw + 1) + 2 * (size_t) nb + (size_t) (cw + 1));
    int64_t *h_bp = aux.data(), *h_base = h_bp + (size_t) nb * (size_t) (cw + 1), *h_r0 = h_base + nb,
            *h_out = h_r0 + nb;
    int64_t fill = 0;
    for (int b = 0; b < nb; b++) {
      const std::vector<int64_t> &bp = bptr[(size_t) b];
      const int64_t s = bp[(size_t) c0], e = bp[(size_t) c1], zb = ia[(size_t) rb[b]] - z;
      for (int64_t c = 0; c <= cw; c++) h_bp[(size_t) b * (size_t) (cw + 1) + (size_t) c] = bp[(size_t) (c0 + c)] - s;
      h_base[b] = fill;
      h_r0[b] = rb[b];
      if (e > s) {
        bof_fptr tv{tfd_val, (uint64_t) (zb + s) * 4}, tc{tfd_col, (uint64_t) (zb + s) * 8};
        rc = stream_file(tv, (uint64_t) (e - s) * 4, d_val + (size_t) fill * 4, true, st, false, o.n_io_threads, cnt);
        if (!rc) rc = stream_file(tc, (uint64_t) (e - s) * 8, d_col + (size_t) fill * 8, true, st, false, o.n_io_threads, cnt);
        if (rc) return rc;
      }
      fill += e - s;
    }
    for (int64_t c = 0; c <= cw; c++) h_out[c] = ia_tr[(size_t) (c0 + c)] - ia_tr[(size_t) c0];
    (void) hipFree(d_aux); d_aux = nullptr;
    BOF_HIP_TRY(hipMalloc((void **) &d_aux, aux.size() * 8));
    BOF_HIP_TRY(hipMemcpyAsync(d_aux, aux.data(), aux.size() * 8, hipMemcpyHostToDevice, st));
    const int64_t *g = (const int64_t *) d_aux;
    BOF_HIP_TRY(csc_merge(nb, cw, g, g + (size_t) nb * (size_t) (cw + 1), g + (size_t) nb * (size_t) (cw + 1) + nb,
                          g + (size_t) nb * (size_t) (cw + 1) + 2 * (size_t) nb, (const float *) d_val,
                          (const int64_t *) d_col, (float *) d_vt, (int64_t *) d_ct, st));
    BOF_HIP_TRY(hipStreamSynchronize(st));
    bof_fptr ov = fa_tr, oc = fja_tr;
    ov.foffset += (uint64_t) ia_tr[(size_t) c0] * 4;
    oc.foffset += (uint64_t) ia_tr[(size_t) c0] * 8;
    rc = stream_file(ov, (uint64_t) out_nnz * 4, d_vt, false, st, use_aio, o.n_io_threads, cnt);
    if (!rc) rc = stream_file(oc, (uint64_t) out_nnz * 8, d_ct, false, st, use_aio, o.n_io_threads, cnt);
    if (rc) return rc;
    cnt.tasks++;
    c0 = c1;
  }
  const int io = file_swrite(fia_tr.fd, fia_tr.foffset, 0, 1, (uint64_t) (n + 1) * 8, ia_tr.data(), use_aio);
  if (io) { set_error(std::string("writing ia_tr failed: ") + strerror(-io)); return BOF_EIO; }
  cnt.wr += (uint64_t) (n + 1) * 8;
  return BOF_OK;
}

static int flash_csrcsc_impl(int64_t m, int64_t n, bof_fptr fia, bof_fptr fja, bof_fptr fa,
                             bof_fptr fia_tr, bof_fptr fja_tr, bof_fptr fa_tr, const bof_options *opts) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = device_ready();
  if (rc) return rc;
  bof_options o = resolved(opts);
  std::vector<int> devs;
  rc = resolve_devices(o, devs);
  if (rc) return rc;
  devs.resize(1);   // the transposition is one device's sort: the first device of the list
  DeviceCallLock call_lock(devs);
  DeviceScope on_dev(devs[0]);
  o.n_devices = 1; o.devices[0] = devs[0];
  file_set_engine(o.io_engine);
  Counters cnt;
  {
    // does the whole matrix fit?  (input + output + sort workspace against the budget)
    std::vector<int64_t> ia((size_t) m + 1, 0);
    if (m > 0) {
      const int io = read_host(fia, (uint64_t) (m + 1) * 8, ia.data(), o.use_odirect != 0);
      if (io) { set_error(std::string("reading ia failed: ") + strerror(-io)); return BOF_EIO; }
    }
    const int64_t nnz = ia[(size_t) m] - ia[0];
    size_t free_b = 0, total_b = 0;
    dev_cache_release();      // cached call-lifetime blocks of earlier CSR calls are free memory for this budget
    BOF_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = o.hbm_budget > 0 ? std::min((size_t) o.hbm_budget, (size_t) (free_b * 0.9)) : (size_t) (free_b * 0.9);
    const size_t need = (size_t) std::max<int64_t>(nnz, 0) * 24 + (size_t) (m + n + 2) * 8 + csrcsc_workspace_bytes(n, std::max<int64_t>(nnz, 0));
    if (nnz > 0 && need > budget) {
      cnt.rd += (uint64_t) (m + 1) * 8;
      rc = flash_csrcsc_blocked(m, n, ia, fja, fa, fia_tr, fja_tr, fa_tr, o, budget, cnt);
      publish_stats(cnt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
      return rc;
    }
  }
  ResidentCsr T;
  rc = flash_transpose_to_hbm(m, n, fa, fia, fja, o, cnt, T);
  if (rc) return rc;
  hipStream_t st = nullptr;
  BOF_HIP_TRY(pooled_stream(&st, false));
  Cleanup guard;
  guard.add([&] { pooled_stream_return(st); });
  const bool use_aio = o.use_odirect != 0;
  rc = stream_file(fa_tr, (uint64_t) T.nnz * 4, (char *) T.val, false, st, use_aio, o.n_io_threads, cnt);
  if (!rc) rc = stream_file(fja_tr, (uint64_t) T.nnz * 8, (char *) T.col, false, st, use_aio, o.n_io_threads, cnt);
  if (!rc) {
    const int io = file_swrite(fia_tr.fd, fia_tr.foffset, 0, 1, (uint64_t) (n + 1) * 8, T.ia_host.data(), use_aio);
    if (io) { set_error(std::string("writing ia_tr failed: ") + strerror(-io)); rc = BOF_EIO; }
    cnt.wr += (uint64_t) (n + 1) * 8;
  }
  cnt.tasks++;
  publish_stats(cnt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  return rc;
}

// What a multi-device call hands each device's pipeline (all null / absent in a single-device call).
// Can every row block's piece(s) of the C file go through O_DIRECT?  The answer must be ONE per file per call, over
// the blocks of every device: the blocks are cut by the nnz budget, so some are sector-aligned and some are not, and
// a pipeline that wrote its aligned blocks with O_DIRECT beside one that wrote its unaligned blocks through the page
// cache met it in a shared page -- the dirty page later went over the direct write (a lost update in the file; found
// by tools/mock_stress.sh in round 5, regression: tests/native/host_pipeline.cpp "csrmix").  Callers that cut one C
// file over PROCESSES keep the cuts page-aligned instead (bof_dist.csr_row_shard).
static bool c_blocks_aligned(const bof_fptr &fc, char ord_b, int64_t k, int64_t c_ld, const int64_t *st, const int64_t *sz,
                             int64_t nb) {
  const uint64_t A = file_dio_align(fc.fd);
  for (int64_t b = 0; b < nb; b++) {
    const bool ok = ord_b == 'R' ? (fc.foffset + (uint64_t) st[b] * (uint64_t) k * 4) % A == 0 && ((uint64_t) sz[b] * (uint64_t) k * 4) % A == 0
                                 : (fc.foffset + (uint64_t) st[b] * 4) % A == 0 && ((uint64_t) sz[b] * 4) % A == 0 &&
                                       ((uint64_t) c_ld * 4) % A == 0;
    if (!ok) return false;
  }
  return true;
}

// The C file's mode for one call (CsrRun::fd_c): 1 O_DIRECT as it is, 2 O_DIRECT with widened reads / page-split
// writes (row-major C), 0 the buffered twin.
static int c_file_mode(const bof_fptr &fc, char ord_b, int64_t k, int64_t c_ld, const int64_t *st, const int64_t *sz, int64_t nb,
                       bool use_odirect) {
  if (c_blocks_aligned(fc, ord_b, k, c_ld, st, sz, nb)) return 1;
  if (ord_b == 'R' && use_odirect && file_dio_align(fc.fd) <= 4096 && env_long("BOF_UNALIGNED_DIRECT", 1) != 0) return 2;
  return 0;
}

// What the last flash::csrmm call did with its C FILE (bof_flash_last_c_file: tests and drivers can see that an
// unaligned C stayed on O_DIRECT): the mode (c_file_mode; -1 = C was not an O_DIRECT file) and the bytes of whole row
// blocks that went through the buffered twin instead (mode 0 only; all pipelines of a multi-device call added up).
static std::atomic<int> g_last_c_mode{-1};
static std::atomic<uint64_t> g_last_c_twin_bytes{0};

struct CsrExtra {
  const int64_t *ia = nullptr;     // the offsets of this device's rows, already on the host
  char *shared_op = nullptr;       // B (csrmm) / x (csrgemv 'N') already on its way into THIS device's HBM ...
  hipEvent_t shared_ready = nullptr;  // ... complete when this event (of this device) has fired;
  std::shared_future<int> *shared_fed = nullptr;   // ... which is RECORDED once this future is set (its value: the
                                                   // feed's return code).  Waiting on an event that has not been
                                                   // recorded yet is a no-op, so the host waits for the future first.
  float *partial_y = nullptr;      // csrgemv 'T': zeroed full-length vector in this device's HBM that takes
                                   // the partial sums and STAYS there (the caller reduces the partials)
  int64_t c_ld = 0;                // column-major C: rows of the whole matrix (0: this call's m)
  int c_direct = -1;               // C file, decided by the caller over the row blocks of EVERY device of the call
                                   // (c_file_mode: 1 O_DIRECT, 2 O_DIRECT widened / page-split, 0 the buffered twin;
                                   // -1: this pipeline decides on its own blocks)
  Counters *out = nullptr;         // counters are added here instead of being published
};

// One device's pipeline of csrmm (is_mm) / csrgemv on the CURRENT device (the caller holds the call lock).
// For csrgemv: hb = input vector (host), hc = output vector (host).
static int flash_csr_device(bool is_mm, char trans, int64_t m, int64_t n, int64_t k, float alpha,
                            float beta, bof_fptr fa, bof_fptr fia, bof_fptr fja, char ord_b,
                            bof_fptr fb, bof_fptr fc, const float *hb, float *hc,
                            const bof_options &ro, const ResidentCsr *res = nullptr,
                            Counters *carry = nullptr, const CsrExtra *ex = nullptr) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = BOF_OK;
  TraceRange range(is_mm ? "bof_flash_csrmm" : "bof_flash_csrgemv");
  CsrRun R;
  if (carry) {  // bytes moved by the transposition that produced `res`
    R.cnt.rd += carry->rd.load(); R.cnt.h2d += carry->h2d.load(); R.cnt.d2h += carry->d2h.load();
  }
  if (res) { R.res_val = res->val; R.res_col = res->col; }
  R.o = ro;
  R.is_mm = is_mm; R.trans = trans; R.ord_b = ord_b;
  R.m = m; R.n = n; R.k = k; R.alpha = alpha; R.beta = beta;
  R.c_ld = ex && ex->c_ld > 0 ? ex->c_ld : m;
  R.fa = fa; R.fja = fja; R.fb = fb; R.fc = fc;
  if (is_mm && fb.fd < 0) { R.host_b = hb; R.host_c = hc; }
  if (!is_mm && trans == 'N' && hc && !(ex && ex->partial_y)) R.host_y = hc;
  R.use_aio = R.o.use_odirect != 0;
  BOF_HIP_TRY(hipGetDevice(&R.dev));
  if (m == 0) return BOF_OK;

  // offsets are read to the host first, as the reference does (csrmm.cpp:69-71)
  if (res) {
    R.ia = res->ia_host.data();
  } else if (ex && ex->ia) {
    R.ia = ex->ia;
  } else {
    // into a block of the pinned cache: its pages exist already (a fresh std::vector is faulted in page by page while
    // it is filled: the 400 MB of offsets of the cfg5-size matrix took 33 ms from the page cache)
    rc = pinned_alloc((void **) &R.ia_pinned, (size_t) (m + 1) * 8);
    if (rc) return rc;
    int io = read_host(fia, (uint64_t) (m + 1) * 8, R.ia_pinned, R.use_aio);
    if (io) { pinned_free(R.ia_pinned); R.ia_pinned = nullptr; set_error(std::string("reading ia failed: ") + strerror(-io)); return BOF_EIO; }
    R.cnt.rd += (uint64_t) (m + 1) * 8;
    R.ia = R.ia_pinned;
  }
  BOF_TRACE_T("csr: offsets on the host");
  const int64_t nb = bof_csr_blocks(R.ia, m, 128, R.o.csrmm_rblk, R.o.max_nnzs, nullptr, nullptr, 0);
  R.st.resize((size_t) nb); R.sz.resize((size_t) nb);
  bof_csr_blocks(R.ia, m, 128, R.o.csrmm_rblk, R.o.max_nnzs, R.st.data(), R.sz.data(), nb);

  if (!res) {
    struct stat sb;
    if (fstat(fja.fd, &sb) == 0) R.fsize_ja = (uint64_t) sb.st_size;
    if (fstat(fa.fd, &sb) == 0) R.fsize_a = (uint64_t) sb.st_size;
    if (file_is_direct(fja.fd)) R.sector = std::max(R.sector, file_dio_align(fja.fd));
    if (file_is_direct(fa.fd)) R.sector = std::max(R.sector, file_dio_align(fa.fd));
  }
  size_t max_idx = 0, max_val = 0, max_c = 0;
  int64_t max_rows = 0;
  for (int64_t b = 0; b < nb; b++) max_rows = std::max(max_rows, R.sz[b]);
  for (int64_t b = 0; b < nb; b++) {
    uint64_t s, l, d;
    if (!res) {
      R.seg(b, 8, fja, s, l, d); max_idx = std::max<size_t>(max_idx, l);
      R.seg(b, 4, fa, s, l, d);  max_val = std::max<size_t>(max_val, l);
    }
    if (is_mm) max_c = std::max(max_c, R.c_bytes(b));
  }
  if (is_mm && fc.fd >= 0) {
    R.fd_c = fc.fd;
    if (file_is_direct(fc.fd)) {
      // one descriptor mode per FILE per call (see c_blocks_aligned): with several devices the caller has decided
      const int mode = ex && ex->c_direct >= 0 ? ex->c_direct
                                               : c_file_mode(fc, ord_b, k, R.c_ld, R.st.data(), R.sz.data(), nb, R.o.use_odirect != 0);
      g_last_c_mode = mode;
      if (mode == 1) R.aio_c = R.use_aio;
      else if (mode == 2) { R.aio_c = R.use_aio; R.c_widen = true; }
      else R.fd_c = file_buffered_fd(fc.fd);
      if (R.fd_c < 0) { set_error("flash csrmm: cannot open a buffered descriptor of the C file"); return BOF_EIO; }
    }
  }
  max_idx = std::max<size_t>(max_idx, R.sector); max_val = std::max<size_t>(max_val, R.sector);
  max_c = std::max<size_t>(max_c, 512);

  int64_t *d_ia = nullptr;
  char *d_b = nullptr, *d_x = nullptr, *d_y = nullptr;
  char *own_b = nullptr, *own_x = nullptr, *own_y = nullptr;   // what this call allocated (and frees)
  hipEvent_t resident_ev = nullptr;
  Cleanup guard;
  guard.add([&] {
    for (auto &c : R.ctx) {
      dev_cache_free(c.d_idx); dev_cache_free(c.d_val); dev_cache_free(c.d_c); dev_cache_free(c.d_c_rm);
      pinned_free(c.h_idx);
      pinned_free(c.h_val);
      pinned_free(c.h_c);
      pinned_free(c.h_ia);
      pinned_free(c.h_y);
      dev_cache_free(c.d_ia);
      if (c.ready) pooled_event_return(c.ready);
      if (c.done) pooled_event_return(c.done);
    }
    if (resident_ev) pooled_event_return(resident_ev);
    dev_cache_free(own_b); dev_cache_free(own_x); dev_cache_free(own_y);
    if (R.h2d) pooled_stream_return(R.h2d);
    if (R.d2h) pooled_stream_return(R.d2h);
  });
  BOF_HIP_TRY(pooled_stream(&R.h2d, true));
  BOF_HIP_TRY(pooled_stream(&R.d2h, true));
  if (res) d_ia = const_cast<int64_t *>(res->ia_dev);     // else: per block, with the block (CsrCtx::d_ia)
  const int64_t xlen = trans == 'N' ? n : m, ylen = trans == 'N' ? m : n;
  const bool ext_op = ex && ex->shared_op;        // B / x comes from the multi-device caller
  const bool ext_y = ex && ex->partial_y;         // the partial result stays in the caller's vector
  if (is_mm) {
    if (!ext_op) { rc = dev_cache_alloc((void **) &d_b, (size_t) n * k * 4); if (rc) return rc; }
  } else {
    if (!ext_op) { rc = dev_cache_alloc((void **) &d_x, (size_t) xlen * 4); if (rc) return rc; }
    if (!ext_y) { rc = dev_cache_alloc((void **) &d_y, (size_t) ylen * 4); if (rc) return rc; }
  }
  own_b = d_b; own_x = d_x; own_y = d_y;
  if (ext_op) (is_mm ? d_b : d_x) = ex->shared_op;
  if (ext_y) d_y = (char *) ex->partial_y;
  BOF_HIP_TRY(pooled_event(&resident_ev));
  // B (csrmm) / x (csrgemv) go to HBM on a thread of their own while the block contexts are set up
  // and the first row blocks are already being read; the compute streams wait for `resident_ev`
  // before the first kernel (36 ms of B at cfg3, 25 ms of x at cfg5 size used to sit in front of
  // the first block read).
  auto upload_resident = [&]() -> int {
    (void) hipSetDevice(R.dev);
    if (ext_op) {   // on its way already (multi-device call): order this device's streams behind it
      if (ex->shared_fed) {
        const int fed = ex->shared_fed->get();      // the feeder has queued every copy and recorded the event
        if (fed) return fed;
      }
      BOF_HIP_TRY(wait_event_both(R.h2d, ex->shared_ready));      // (recorded by the feeder thread)
      if (!is_mm && trans == 'T' && !ext_y) BOF_HIP_TRY(hipMemsetAsync(d_y, 0, (size_t) ylen * 4, R.h2d));
      BOF_HIP_TRY(hipEventRecord(resident_ev, R.h2d));
      return BOF_OK;
    }
    if (is_mm) {
      // B stays resident for the whole call (one shared read, like the reference's
      // single "use_full" cache key, csrmm_task.h:175-183)
      if (R.host_b) {
        BOF_HIP_TRY(hipMemcpyAsync(d_b, R.host_b, (size_t) n * k * 4, hipMemcpyHostToDevice, R.h2d));
        R.cnt.h2d += (uint64_t) n * k * 4;
      } else {
        const int rc2 = stream_file(fb, (uint64_t) n * k * 4, d_b, true, R.h2d, R.use_aio, R.o.n_io_threads, R.cnt);
        if (rc2) return rc2;
      }
      if (ord_b == 'C') {  // column-major B (n x k, ld = n) -> row-major copy used by the kernel
        void *tmp = nullptr;
        const int rc2 = scratch_get(SCR_B_RM, (size_t) n * k * 4, &tmp);
        if (rc2) return rc2;
        BOF_HIP_TRY(hipMemcpyAsync(tmp, d_b, (size_t) n * k * 4, hipMemcpyDeviceToDevice, R.h2d));
        // (this is a thread created for the call: the kernel goes through the device's persistent launcher)
        BOF_HIP_TRY(launch_from_persistent(R.dev, [&] { return transpose_f32((const float *) tmp, n, k, n, (float *) d_b, k, R.h2d); }));
      }
    } else {
      // x lives in pageable host memory (include/flash_blas.h:55-57): through pinned chunks on several threads
      // (a plain copy out of pageable memory is staged by the runtime on the calling thread: 200 MB in 25-40 ms)
      if (pageable_to_device(d_x, hb, (uint64_t) xlen * 4, std::max(2, R.o.n_io_threads / 2), R.h2d)) {
        set_error("flash csr: copying the host vector to the device through pinned chunks failed (pinned block, copy stream or HIP copy)");
        return BOF_EHIP;
      }
      if (trans == 'T' && !ext_y) BOF_HIP_TRY(hipMemsetAsync(d_y, 0, (size_t) ylen * 4, R.h2d));
      R.cnt.h2d += (uint64_t) xlen * 4;
    }
    BOF_HIP_TRY(hipEventRecord(resident_ev, R.h2d));
    BOF_TRACE_T("csr: B / x resident (queued)");
    return BOF_OK;
  };
  int resident_rc = BOF_OK;
  std::thread resident_thread([&] { resident_rc = upload_resident(); });
  // (joined before the first kernel launch and on every early return below)
  struct Joiner {
    std::thread &t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } resident_joiner{resident_thread};

  // one context = one row block in flight (index + value segments, C block); the device delivers
  // its sequential rate only with several large requests queued, so as many blocks are in flight
  // as there are staging slots, each read by its own thread, and retired (C written back) by a
  // small pool instead of one thread
  // Row-block contexts in flight.  csrmm: TWICE the reader threads (round 6) -- a context is busy from its block's
  // first read to the end of its C write-back, and with as many contexts as readers a reader waits for a context about
  // a quarter of the time: cfg3 from files, three interleaved rounds on one lease (profiles/r6/cfg3_sweep.txt): 8
  // readers x 16 contexts 0.99-1.03 s O_DIRECT / 0.236-0.248 s page cache, every call; x 8 contexts 1.00-1.18 /
  // 0.25-0.31; x 12 and 6 x 12 in between.  Bounded by 3 GiB of pinned staging (the pinned-block cache keeps 4 GiB;
  // 24 contexts of cfg3's 171 MB went past it and paid for re-pinning: 0.33 s from the page cache).
  int64_t want_depth = std::max(2, R.o.pinned_slots);
  if (is_mm) {
    const int64_t per_ctx = (int64_t) (max_idx + max_val + max_c) + (max_rows + 1) * 8;
    const int64_t cap = std::max<int64_t>(want_depth, (int64_t) (3ull << 30) / std::max<int64_t>(per_ctx, 1));
    want_depth = std::min<int64_t>(std::max<int64_t>(want_depth, 2 * (int64_t) std::max(1, R.o.n_io_threads)), cap);
  }
  if (const long forced = env_long("BOF_CSR_CONTEXTS", 0))      // (experiments: N contexts; negative: the rule of rounds 1-5)
    want_depth = forced < 0 ? std::max(2, R.o.pinned_slots) : std::max<long>(2, forced);
  R.depth = (int) std::min<int64_t>(want_depth, nb);
  R.ctx.resize((size_t) R.depth);
  for (int i = 0; i < R.depth; i++) {
    CsrCtx &c = R.ctx[i];
    if (!res) {
      rc = dev_cache_alloc((void **) &c.d_idx, max_idx);
      if (!rc) rc = dev_cache_alloc((void **) &c.d_val, max_val);
      if (rc) return rc;
      rc = pinned_alloc((void **) &c.h_idx, max_idx);
      if (!rc) rc = pinned_alloc((void **) &c.h_val, max_val);
      if (rc) return rc;
    }
    if (!res) {
      rc = dev_cache_alloc((void **) &c.d_ia, (size_t) (max_rows + 1) * 8);
      if (rc) return rc;
      rc = pinned_alloc((void **) &c.h_ia, (size_t) (max_rows + 1) * 8);
      if (rc) return rc;
    }
    if (!is_mm && trans == 'N' && !ext_y && hc) {
      rc = pinned_alloc((void **) &c.h_y, (size_t) std::max<int64_t>(max_rows, 1) * 4);
      if (rc) return rc;
    }
    if (is_mm) {
      rc = dev_cache_alloc((void **) &c.d_c, max_c);
      if (!rc && ord_b == 'C') rc = dev_cache_alloc((void **) &c.d_c_rm, max_c);
      if (rc) return rc;
      rc = pinned_alloc((void **) &c.h_c, max_c + 2 * CsrRun::kPage);      // slack: widened read / page-congruent placement
      if (rc) return rc;
    }
    BOF_HIP_TRY(pooled_event(&c.ready));
    BOF_HIP_TRY(pooled_event(&c.done));
    c.owner = i;
  }
  StreamSet *ss = stream_set(R.o.n_streams);
  if (!ss) { set_error("flash csr: stream creation failed"); return BOF_EHIP; }

  BOF_TRACE_T("csr: block contexts ready");
  std::vector<std::thread> readers;
  for (int i = 0; i < std::max(1, std::min<int>(R.o.n_io_threads, R.depth)); i++)
    readers.emplace_back([&R] { R.reader_main(); });
  std::vector<std::thread> retirers;
  for (int i = 0; i < std::max(1, std::min(4, R.depth / 2)); i++) retirers.emplace_back([&R] { R.retire_main(); });

  StallWatch watch("flash csr pipeline",
                   [&R] { return R.cnt.rd.load() + R.cnt.wr.load() + R.cnt.h2d.load() + R.cnt.d2h.load() + R.cnt.tasks.load(); },
                   [&R] { R.fail_io(-ETIMEDOUT); });
  hipError_t herr = hipSuccess;
  KernelTimer ktimer;
  ktimer.on = R.o.kernel_timing > 0;
  int fail = 0;
  // bof_options.verify / $BOF_VERIFY: a LAUNCH RECEIPT per csrmm / csrgemv launch (round 6, profiles/r6/incident_csrmm: in one
  // launch in ~10^7, with several processes on the GPU, the workgroups of one XCD ran under their predecessors' IDs --
  // one set of rows updated twice, another not at all).  Every workgroup counts itself in seen[blockIdx.x], a checker
  // behind the launch on the same stream compares every entry with 1 and clears it; one region of `seen` per compute
  // stream.  A miss fails the call with BOF_EVERIFY when it has drained.
  unsigned *d_seen = nullptr, *d_flag = nullptr;
  int64_t seen_stride = 0;
  uint64_t receipts = 0;
  if (verify_wanted(R.o)) {
    int64_t rmax = 0;
    for (int64_t b = 0; b < nb; b++) rmax = std::max(rmax, R.sz[b]);
    seen_stride = is_mm ? scsrmm_receipt_entries('R', rmax) : scsrgemv_receipt_entries(rmax);
    const size_t words = (size_t) seen_stride * (size_t) ss->n + 1;
    herr = hipMalloc((void **) &d_seen, words * sizeof(unsigned));
    // (cleared ON A COMPUTE STREAM and waited for: hipMemset of device memory returns before the fill has run and the
    //  null stream does not order the non-blocking compute streams behind it -- under load the first launches counted
    //  into words the fill then cleared: 8-process fuzz of the second session, profiles/r6/session2/fuzz_csr_receipts)
    if (herr == hipSuccess) herr = hipMemsetAsync(d_seen, 0, words * sizeof(unsigned), ss->s[0]);
    if (herr == hipSuccess) herr = hipStreamSynchronize(ss->s[0]);
    if (herr == hipSuccess) d_flag = d_seen + words - 1;
  }
  // one csrmm launch + its receipt check ($BOF_VERIFY_INJECT=4, self-test: the first launch of the call runs twice)
  auto csrmm_launch = [&](int64_t b, int64_t r, int64_t w, const float *val, const int64_t *col, const int64_t *bia,
                          const float *bp, float *cp, hipStream_t st) -> hipError_t {
    unsigned *seen = d_flag ? d_seen + (size_t) seen_stride * (size_t) (b % ss->n) : nullptr;
    hipError_t e = scsrmm('R', r, w, n, alpha, val, col, bia, bp, k, beta, cp, k, st, seen);
    if (e == hipSuccess && seen && receipts == 0 && env_long("BOF_VERIFY_INJECT", 0) == 4)
      e = scsrmm('R', r, w, n, alpha, val, col, bia, bp, k, beta, cp, k, st, seen);
    if (e == hipSuccess && seen) {
      e = csr_receipt_check(seen, scsrmm_receipt_entries('R', r), d_flag, st);
      receipts++;
    }
    return e;
  };
  resident_thread.join();
  if (resident_rc) {
    set_error("flash csr: bringing the resident operand (B / x) into HBM failed");
    fail = resident_rc;
  }
  if (!fail && host_handover()) herr = hipEventSynchronize(resident_ev);       // (recorded by the upload thread)
  for (int i = 0; i < ss->n && !fail && herr == hipSuccess; i++) herr = hipStreamWaitEvent(ss->s[i], resident_ev, 0);
  for (int64_t b = 0; b < nb && !fail && herr == hipSuccess; b++) {
    CsrCtx &c = R.ctx[b % R.depth];
    {
      std::unique_lock<std::mutex> lk(R.mu);
      R.cv.wait(lk, [&] { return (c.owner == b && c.state == 1) || R.io_error.load(); });
    }
    if (R.io_error.load()) { fail = BOF_EIO; break; }
    hipStream_t st = ss->s[b % ss->n];
    herr = hipStreamWaitEvent(st, c.ready, 0);
    if (herr != hipSuccess) break;
    const int64_t s = R.st[b], r = R.sz[b];
    const int64_t *bia = c.d_ia ? (const int64_t *) c.d_ia : d_ia + s;     // the block's offsets (any base)
    const int64_t *col;
    const float *val;
    if (res) {
      col = res->col + R.ia[(size_t) s];
      val = res->val + R.ia[(size_t) s];
    } else {
      uint64_t s0, l0, d0, s1, l1, d1;
      R.seg(b, 8, fja, s0, l0, d0);
      R.seg(b, 4, fa, s1, l1, d1);
      col = (const int64_t *) (c.d_idx + d0);  // un-shift the sector widening
      val = (const float *) (c.d_val + d1);
    }
    herr = ktimer.begin(st);
    if (herr != hipSuccess) break;
    if (is_mm) {
      if (ord_b == 'C' && beta != 0.f)  // C block arrived packed column-major [k][r]
        herr = transpose_f32((const float *) c.d_c, r, k, r, (float *) c.d_c_rm, k, st);
      for (int64_t j0 = 0; j0 < k && herr == hipSuccess; j0 += R.o.csrmm_cblk) {
        const int64_t w = std::min(k - j0, R.o.csrmm_cblk);
        // ('C': same row-major kernel on the transposed block; d_b is row-major here)
        herr = csrmm_launch(b, r, w, val, col, bia, (const float *) d_b + j0, (float *) (ord_b == 'R' ? c.d_c : c.d_c_rm) + j0, st);
      }
      if (ord_b == 'C' && herr == hipSuccess)  // [r][k] -> packed column-major block [k][r]
        herr = transpose_f32((const float *) c.d_c_rm, k, r, k, (float *) c.d_c, r, st);
      if (herr == hipSuccess) herr = ktimer.end(st);
      if (herr != hipSuccess) break;
      // C block -> pinned buffer on the D2H stream, after the kernels
      herr = hipEventRecord(c.done, st);
      if (herr == hipSuccess) herr = hipStreamWaitEvent(R.d2h, c.done, 0);
      if (herr == hipSuccess)
        herr = hipMemcpyAsync(c.h_c + R.c_wdelta(b), c.d_c, R.c_bytes(b), hipMemcpyDeviceToHost, R.d2h);
      if (herr == hipSuccess) herr = hipEventRecord(c.done, R.d2h);
      R.cnt.d2h += R.c_bytes(b);
    } else {
      unsigned *seen = d_flag ? d_seen + (size_t) seen_stride * (size_t) (b % ss->n) : nullptr;
      if (trans == 'N')
        herr = scsrgemv('N', r, n, val, bia, col, (const float *) d_x, (float *) d_y + s, st, seen);
      else
        herr = scsrgemv('T', r, n, val, bia, col, (const float *) d_x + s, (float *) d_y, st, seen);
      // ($BOF_VERIFY_INJECT=4, self-test: the call's first launch runs twice -- 'T' adds its products twice, 'N' stores
      //  the same y again: the receipt is what tells)
      if (herr == hipSuccess && seen && receipts == 0 && env_long("BOF_VERIFY_INJECT", 0) == 4)
        herr = scsrgemv(trans == 'N' ? 'N' : 'T', r, n, val, bia, col, (const float *) d_x + (trans == 'N' ? 0 : s),
                        (float *) d_y + (trans == 'N' ? s : 0), st, seen);
      if (herr == hipSuccess && seen) {
        herr = csr_receipt_check(seen, scsrgemv_receipt_entries(r), d_flag, st);
        receipts++;
      }
      if (herr == hipSuccess) herr = ktimer.end(st);
      if (herr == hipSuccess) herr = hipEventRecord(c.done, st);
      if (herr == hipSuccess && c.h_y) {      // 'N': this block's slice of y leaves now (the retire thread copies it out)
        herr = hipStreamWaitEvent(R.d2h, c.done, 0);
        if (herr == hipSuccess) herr = hipMemcpyAsync(c.h_y, (const float *) d_y + s, (size_t) r * 4, hipMemcpyDeviceToHost, R.d2h);
        if (herr == hipSuccess) herr = hipEventRecord(c.done, R.d2h);
        R.cnt.d2h += (uint64_t) r * 4;
      }
    }
    if (herr != hipSuccess) break;
    R.cnt.tasks++;
    R.done_q.push(b);
  }
  BOF_TRACE_T("csr: all blocks dispatched");
  if (herr != hipSuccess || fail) R.fail_io(-EIO);  // releases readers parked on a context hand-over
  for (auto &th : readers) th.join();
  R.done_q.close();
  for (auto &th : retirers) th.join();
  (void) hipDeviceSynchronize();
  ktimer.collect(R.cnt);
  BOF_TRACE_T("csr: drained (C written)");
  if (d_seen) {
    unsigned missed = 0;
    if (herr == hipSuccess && !fail) herr = hipMemcpy(&missed, d_flag, sizeof(missed), hipMemcpyDeviceToHost);
    (void) hipFree(d_seen);
    R.cnt.vchecks += receipts;
    if (herr == hipSuccess && !fail && missed) {
      evt("verify mismatch", (int) missed, 0, receipts);
      set_error(std::string(is_mm ? "flash csrmm" : "flash csrgemv") + ": BOF_VERIFY: " + std::to_string(missed) + " workgroup receipts of the call's " + std::to_string(receipts) +
                (is_mm ? " csrmm" : " csrgemv") + " launches are not 1 (a workgroup ran twice or not at all: profiles/r6/incident_csrmm) -- the result is not to be trusted");
      fail = BOF_EVERIFY;
    }
  }
  if (!is_mm && !ext_y && !fail && herr == hipSuccess && !R.host_y) {
    if (device_to_pageable(hc, d_y, (uint64_t) ylen * 4, R.o.n_io_threads)) herr = hipErrorUnknown;
    R.cnt.d2h += (uint64_t) ylen * 4;
    BOF_TRACE_T("csr: y on the host");
  }
  if (herr != hipSuccess && !fail) fail = hip_fail(herr, "flash csr dispatch");
  if (R.io_error.load() && (!fail || fail == BOF_EIO)) {
    const int e = R.io_error.load();
    set_error("flash csr: I/O pipeline failed: " + io_error_text(e));
    fail = BOF_EIO;
  }
  if (is_mm) g_last_c_twin_bytes += R.cnt_buffered_c.load();
  if (ex && ex->out) {
    Counters &o = *ex->out;
    o.rd += R.cnt.rd.load(); o.wr += R.cnt.wr.load(); o.h2d += R.cnt.h2d.load(); o.d2h += R.cnt.d2h.load();
    o.tasks += R.cnt.tasks.load();
    o.klaunch += R.cnt.klaunch.load(); o.kns += R.cnt.kns.load();
    o.vchecks += R.cnt.vchecks.load();
  } else {
    publish_stats(R.cnt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  }
  return fail;
}

// One source (a file region, or a host array when f is null) -> the same bytes in the HBM of several
// devices: every chunk is read ONCE into a pinned slot and copied from there to each device on that
// device's stream, i.e. over that device's own PCIe link.
static int fan_out(const bof_fptr *f, const char *host, uint64_t bytes, const std::vector<int> &devs,
                   const std::vector<char *> &dst, const std::vector<hipStream_t> &st, bool use_aio, int n_thr,
                   Counters &cnt) {
  if (bytes == 0) return BOF_OK;
  const size_t chunk = (size_t) std::min<uint64_t>(32ull << 20, round_up(bytes, 4096));
  const int64_t nchunks = (int64_t) ((bytes + chunk - 1) / chunk);
  n_thr = (int) std::max<int64_t>(1, std::min<int64_t>(n_thr, nchunks));
  std::atomic<int64_t> next{0};
  std::atomic<int> fail{0};
  auto worker = [&] {
    (void) hipSetDevice(devs[0]);
    (void) bind_thread_near_device(devs[0]);
    PinnedRing ring;
    if (ring.init(2, chunk, &devs)) { fail.store(-1000); return; }
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= nchunks || fail.load()) break;
      const uint64_t o = (uint64_t) i * chunk, len = std::min<uint64_t>(chunk, bytes - o);
      const int sl = ring.acquire();
      int io = 0;
      if (f) {
        io = file_sread(f->fd, f->foffset + o, 0, 1, len, ring.ptr(sl), use_aio);
        cnt.rd += len;
      } else {
        memcpy(ring.ptr(sl), host + o, len);
      }
      hipError_t e = hipSuccess;
      for (size_t d = 0; d < devs.size() && !io && e == hipSuccess; d++) {
        e = hipSetDevice(devs[d]);
        if (e == hipSuccess) e = hipMemcpyAsync(dst[d] + o, ring.ptr(sl), len, hipMemcpyHostToDevice, st[d]);
        if (e == hipSuccess && ring.mark_busy(sl, st[d], (int) d)) e = hipErrorUnknown;
        cnt.h2d += len;
      }
      ring.release(sl);
      if (io) fail.store(io);
      if (e != hipSuccess) fail.store(-1000 - (int) e);
    }
    ring.destroy();
  };
  std::vector<std::thread> th;
  for (int i = 1; i < n_thr; i++) th.emplace_back(worker);
  worker();
  for (auto &t : th) t.join();
  const int fl = fail.load();
  if (fl) {
    set_error("bringing the shared operand into the devices' HBM failed: " +
              (fl > -1000 ? std::string(strerror(-fl)) : "HIP error " + std::to_string(-1000 - fl)));
    return fl > -1000 ? BOF_EIO : BOF_EHIP;
  }
  return BOF_OK;
}

// flash::csrmm 'N' / flash::csrgemv on files: the device list, the call locks, then either the
// single-device pipeline or -- several devices in this process -- the row blocks dealt to the
// devices in contiguous, nnz-balanced ranges, each device running that pipeline on its
// rows.  What every device needs (B; x of csrgemv 'N') is read / staged ONCE and copied to all of
// them (fan_out); C rows and y slices are disjoint.  csrgemv 'T' leaves a full-length partial on
// every device: device d sums segment d of all partials straight out of its peers' HBM
// (reduce-scatter shape over xGMI; the reference's mutex-guarded vector add,
// include/tasks/csrgemv_task.h:169-176) and sends that segment to the host itself, so the
// result leaves over D PCIe links and no all-gather is needed.
static int flash_csr_impl(bool is_mm, char trans, int64_t m, int64_t n, int64_t k, float alpha,
                          float beta, bof_fptr fa, bof_fptr fia, bof_fptr fja, char ord_b,
                          bof_fptr fb, bof_fptr fc, const float *hb, float *hc,
                          const bof_options *opts, const ResidentCsr *res = nullptr,
                          Counters *carry = nullptr) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = device_ready();
  if (rc) return rc;
  const bof_options o = resolved(opts);
  std::vector<int> devs;
  rc = resolve_devices(o, devs);
  if (rc) return rc;
  if (res) devs.resize(1);   // a matrix that already sits in one device's HBM (csrmm 'T') is used there
  DeviceCallLock call_lock(devs);
  if (is_mm) { g_last_c_mode = -1; g_last_c_twin_bytes = 0; }      // (bof_flash_last_c_file: this call's C file)
  if (o.io_request_kib > 0) (void) bof_file_set_request_bytes((uint64_t) o.io_request_kib << 10);
  file_set_engine(o.io_engine);
  if (devs.size() == 1 || m == 0) {
    DeviceScope ds(devs[0]);
    return flash_csr_device(is_mm, trans, m, n, k, alpha, beta, fa, fia, fja, ord_b, fb, fc, hb, hc, o, res, carry);
  }

  // ---- offsets on the host, the reference's row blocks, contiguous block ranges per device -------
  const bool use_aio = o.use_odirect != 0;
  Counters total;
  HostI64 ia((size_t) m + 1);
  {
    const int io = read_host(fia, (uint64_t) (m + 1) * 8, ia.data(), use_aio);
    if (io) { set_error(std::string("reading ia failed: ") + strerror(-io)); return BOF_EIO; }
    total.rd += (uint64_t) (m + 1) * 8;
  }
  const int64_t nb = bof_csr_blocks(ia.data(), m, 128, o.csrmm_rblk, o.max_nnzs, nullptr, nullptr, 0);
  std::vector<int64_t> bst((size_t) nb), bsz((size_t) nb);
  bof_csr_blocks(ia.data(), m, 128, o.csrmm_rblk, o.max_nnzs, bst.data(), bsz.data(), nb);
  const int D = (int) std::min<int64_t>((int64_t) devs.size(), nb);
  std::vector<int> used(devs.begin(), devs.begin() + D);
  std::vector<int64_t> cut((size_t) D + 1, 0);   // block index where each device's range starts
  cut[(size_t) D] = nb;
  {
    const int64_t z = ia[0], nnz = ia[(size_t) m] - z;
    for (int d = 1; d < D; d++) {
      const int64_t target = z + nnz / D * d + nnz % D * d / D;
      int64_t b = std::lower_bound(bst.begin(), bst.end(), target,
                                   [&](int64_t row, int64_t t) { return ia[(size_t) row] < t; }) - bst.begin();
      b = std::max(b, cut[(size_t) d - 1] + 1);          // at least one block each ...
      cut[(size_t) d] = std::min<int64_t>(b, nb - (D - d));   // ... for the later devices too
    }
  }
  struct Shard {
    int dev = 0;
    int64_t row0 = 0, rows = 0;
    char *op = nullptr;          // B / x of this device
    float *partial = nullptr;    // csrgemv 'T'
    hipStream_t st = nullptr;
    hipEvent_t ready = nullptr;
    Counters cnt;
    int rc = 0;
    std::string err;
    double seconds = 0;
  };
  std::vector<Shard> sh((size_t) D);
  // csrgemv 'T': peer access between every pair of distinct devices (xGMI), enabled BEFORE the partial
  // vectors are allocated; without it the partials cross through staging copies (hipMemcpyPeer)
  bool peers = true;
  if (!is_mm && trans == 'T')
    for (int a = 0; a < D && peers; a++)
      for (int b = 0; b < D && peers; b++) {
        if (used[(size_t) a] == used[(size_t) b]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, used[(size_t) a], used[(size_t) b]) != hipSuccess || !can) { peers = false; break; }
        DeviceScope ds(used[(size_t) a]);
        const hipError_t e = hipDeviceEnablePeerAccess(used[(size_t) b], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peers = false;
        (void) hipGetLastError();
      }
  const bool share_op = is_mm || trans == 'N';           // csrgemv 'T' uses a private slice of x per device
  const uint64_t op_bytes = is_mm ? (uint64_t) n * k * 4 : (uint64_t) n * 4;
  Cleanup guard;
  guard.add([&] {
    for (Shard &S : sh) {
      DeviceScope ds(S.dev);
      (void) hipFree(S.op);
      (void) hipFree(S.partial);
      if (S.ready) pooled_event_return(S.ready);
      if (S.st) pooled_stream_return(S.st);
    }
  });
  for (int d = 0; d < D; d++) {
    Shard &S = sh[(size_t) d];
    S.dev = used[(size_t) d];
    S.row0 = bst[(size_t) cut[(size_t) d]];
    S.rows = (cut[(size_t) d + 1] == nb ? m : bst[(size_t) cut[(size_t) d + 1]]) - S.row0;
    DeviceScope ds(S.dev);
    BOF_HIP_TRY(pooled_stream(&S.st, true));
    BOF_HIP_TRY(pooled_event(&S.ready));
    if (share_op) BOF_HIP_TRY(hipMalloc((void **) &S.op, std::max<uint64_t>(op_bytes, 4)));
    if (!is_mm && trans == 'T') {
      BOF_HIP_TRY(hipMalloc((void **) &S.partial, (size_t) std::max<int64_t>(n, 1) * 4));
      BOF_HIP_TRY(hipMemsetAsync(S.partial, 0, (size_t) n * 4, S.st));
      BOF_HIP_TRY(hipStreamSynchronize(S.st));   // zero before the device's first kernel adds into it
    }
  }

  // ---- the shared operand goes to every device while the pipelines start ------------------------------
  int feed_rc = BOF_OK;
  std::string feed_err;
  std::promise<int> fed_promise;
  std::shared_future<int> fed = fed_promise.get_future().share();
  // (on a persistent launcher thread of its own -- it launches the transposes of a column-major B)
  auto feeder = launch_async(sh[0].dev, 1 << 20, [&] {
    struct SetOnExit {     // the pipelines wait for this future before they wait for the device-side events
      std::promise<int> &p; int &rc;
      ~SetOnExit() { p.set_value(rc); }
    } set_on_exit{fed_promise, feed_rc};
    if (!share_op) return;
    std::vector<char *> dst;
    std::vector<hipStream_t> sts;
    for (Shard &S : sh) { dst.push_back(S.op); sts.push_back(S.st); }
    const bool from_file = is_mm && fb.fd >= 0;
    feed_rc = fan_out(from_file ? &fb : nullptr, (const char *) hb, op_bytes, used, dst, sts, use_aio, o.n_io_threads, total);
    std::map<int, hipEvent_t> last_on_dev;    // shards that share an ordinal share its scratch slot: one after the other
    for (Shard &S : sh) {
      DeviceScope ds(S.dev);
      hipError_t e = hipSuccess;
      if (!feed_rc && is_mm && ord_b == 'C') {  // column-major B (n x k, ld = n) -> the row-major image the kernel reads
        void *tmp = nullptr;
        if (scratch_get(SCR_B_RM, (size_t) op_bytes, &tmp)) e = hipErrorOutOfMemory;
        auto prev = last_on_dev.find(S.dev);
        if (e == hipSuccess && prev != last_on_dev.end()) e = hipStreamWaitEvent(S.st, prev->second, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(tmp, S.op, (size_t) op_bytes, hipMemcpyDeviceToDevice, S.st);
        if (e == hipSuccess) e = transpose_f32((const float *) tmp, n, k, n, (float *) S.op, k, S.st);
      }
      if (e == hipSuccess) e = hipEventRecord(S.ready, S.st);   // recorded even after a failed feed: nobody may wait forever
      if (e == hipSuccess) last_on_dev[S.dev] = S.ready;
      if (e != hipSuccess && !feed_rc) feed_rc = hip_fail(e, "shared operand");
    }
    if (feed_rc) feed_err = bof_last_error();
  });

  // the C file's descriptor mode, over the row blocks of all devices (c_blocks_aligned)
  const int c_direct = is_mm && fc.fd >= 0 && file_is_direct(fc.fd)
                           ? c_file_mode(fc, ord_b, k, m, bst.data(), bsz.data(), nb, o.use_odirect != 0) : -1;
  // ---- one pipeline per device on its rows ------------------------------------------------------------
  auto run_shard = [&](Shard &S) {
    DeviceScope ds(S.dev);
    struct RepScope {        // which repetition of its ordinal this shard is ($BOF_STREAMS_PER_REP)
      explicit RepScope(int r) { t_ordinal_rep = r; }
      ~RepScope() { t_ordinal_rep = 0; }
    } rep_scope((int) std::count_if(sh.begin(), sh.begin() + (&S - sh.data()), [&](const Shard &E) { return E.dev == S.dev; }));
    CsrExtra ex;
    ex.ia = ia.data() + S.row0;
    ex.out = &S.cnt;
    if (share_op) { ex.shared_op = S.op; ex.shared_ready = S.ready; ex.shared_fed = &fed; }
    bof_fptr c_f = fc;
    const float *b_h = hb;
    float *c_h = hc;
    if (is_mm) {
      ex.c_ld = m;
      ex.c_direct = c_direct;
      const uint64_t c_off = ord_b == 'R' ? (uint64_t) S.row0 * (uint64_t) k : (uint64_t) S.row0;
      if (fc.fd >= 0) c_f.foffset += c_off * 4;
      if (c_h) c_h += c_off;
    } else if (trans == 'N') {
      c_h += S.row0;               // y rows of this device
    } else {
      b_h += S.row0;               // x rows of this device; the partial stays in HBM
      ex.partial_y = S.partial;
    }
    S.rc = flash_csr_device(is_mm, trans, S.rows, n, k, alpha, beta, fa, fia, fja, ord_b, fb, c_f, b_h, c_h, o, nullptr,
                            nullptr, &ex);
    if (S.rc) S.err = bof_last_error();
    S.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  };
  // the calling thread drives the first shard, a PERSISTENT launcher thread each of the others (never a thread made
  // for the call: flash_common.h, "persistent launcher threads")
  auto rep_of = [&sh](int d) {
    int rep = 0;
    for (int e = 0; e < d; e++)
      if (sh[(size_t) e].dev == sh[(size_t) d].dev) rep++;
    return rep;
  };
  {
    std::vector<std::shared_ptr<LaunchJob>> jobs;
    for (int d = 1; d < D; d++) jobs.push_back(launch_async(sh[(size_t) d].dev, rep_of(d), [&, d] { run_shard(sh[(size_t) d]); }));
    run_shard(sh[0]);
    for (auto &j : jobs) launch_wait(j);
  }
  launch_wait(feeder);
  if (feed_rc) { rc = feed_rc; set_error(feed_err); }
  for (Shard &S : sh)
    if (S.rc && !rc) { rc = S.rc; set_error(S.err); }

  // ---- csrgemv 'T': reduce-scatter of the partials between the devices, every segment home over its own link ----
  if (!rc && !is_mm && trans == 'T' && n > 0) {
    for (Shard &S : sh) {   // every partial complete before anybody reads it
      DeviceScope ds(S.dev);
      (void) hipDeviceSynchronize();
    }
    std::vector<int> seg_rc((size_t) D, 0);
    auto reduce_segment = [&](int d) {
      Shard &S = sh[(size_t) d];
      DeviceScope ds(S.dev);
      const int64_t s0 = n / D * d + std::min<int64_t>(d, n % D), len = n / D + (d < n % D ? 1 : 0);
      if (len == 0) return;
      hipError_t e = hipSuccess;
      float *tmp = nullptr;
      if (peers) {
        const float *srcs[BOF_MAX_DEVICES];
        for (int e2 = 0; e2 < D; e2++) srcs[e2] = sh[(size_t) e2].partial + s0;
        e = sum_partials(S.partial + s0, srcs, D, len, S.st);
      } else {
        e = hipMalloc((void **) &tmp, (size_t) len * 4);
        for (int e2 = 0; e2 < D && e == hipSuccess; e2++) {
          if (e2 == d) continue;
          e = hipMemcpyPeerAsync(tmp, S.dev, sh[(size_t) e2].partial + s0, sh[(size_t) e2].dev, (size_t) len * 4, S.st);
          const float *two[2] = {S.partial + s0, tmp};
          if (e == hipSuccess) e = sum_partials(S.partial + s0, two, 2, len, S.st);
        }
      }
      if (e == hipSuccess) e = hipStreamSynchronize(S.st);
      if (e == hipSuccess && device_to_pageable(hc + s0, S.partial + s0, (uint64_t) len * 4, std::max(1, o.n_io_threads / D)))
        e = hipErrorUnknown;
      (void) hipFree(tmp);
      S.cnt.d2h += (uint64_t) len * 4;
      if (e != hipSuccess) seg_rc[(size_t) d] = hip_fail(e, "csrgemv 'T': reducing the partial sums");
    };
    // NOTE on summation order: segment sums run e = 0 .. D-1, a fixed order; the partials themselves
    // are sums of fp32 atomics (as in the single-device call), exact on integer data
    std::vector<std::shared_ptr<LaunchJob>> jobs;
    for (int d = 1; d < D; d++) jobs.push_back(launch_async(sh[(size_t) d].dev, rep_of(d), [&, d] { reduce_segment(d); }));
    reduce_segment(0);
    for (auto &j : jobs) launch_wait(j);
    for (int d = 0; d < D; d++)
      if (seg_rc[(size_t) d] && !rc) { rc = seg_rc[(size_t) d]; set_error("csrgemv 'T': reducing the partial sums across the devices failed"); }
  }

  std::vector<bof_flash_stats> per;
  for (Shard &S : sh) {
    total.rd += S.cnt.rd.load(); total.wr += S.cnt.wr.load(); total.h2d += S.cnt.h2d.load(); total.d2h += S.cnt.d2h.load();
    total.tasks += S.cnt.tasks.load();
    total.klaunch += S.cnt.klaunch.load(); total.kns += S.cnt.kns.load();
    total.vchecks += S.cnt.vchecks.load();
    bof_flash_stats ps{};
    ps.bytes_read = S.cnt.rd; ps.bytes_written = S.cnt.wr; ps.bytes_h2d = S.cnt.h2d; ps.bytes_d2h = S.cnt.d2h;
    ps.tasks = S.cnt.tasks; ps.seconds = S.seconds;
    ps.kernel_launches = S.cnt.klaunch; ps.kernel_seconds = (double) S.cnt.kns.load() * 1e-9;
    ps.verify_checks = S.cnt.vchecks;
    per.push_back(ps);
  }
  publish_stats(total, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  publish_device_stats(per);
  return rc;
}

// csrmm with trans_a = 'T': C[n x k] = alpha * A^T * B[m x k] + beta * C.  A^T is built in HBM
// and the row-block pipeline of the 'N' case runs over it with nothing left to read for A
// (the reference writes A^T to temporary files first, src/blas/csrmm.cpp:355-422).
static int flash_csrmm_trans(uint64_t m, uint64_t n, uint64_t k, float alpha, float beta, bof_fptr a,
                             bof_fptr ia, bof_fptr ja, char ord_b, bof_fptr b, bof_fptr c,
                             const float *hb, float *hc, const bof_options *opts) {
  int rc = device_ready();
  if (rc) return rc;
  if (n == 0) return BOF_OK;
  bof_options o = resolved(opts);
  std::vector<int> devs;
  rc = resolve_devices(o, devs);
  if (rc) return rc;
  devs.resize(1);   // A^T is built in ONE device's HBM and used there: the first device of the list
  DeviceCallLock call_lock(devs);
  DeviceScope on_dev(devs[0]);
  o.n_devices = 1; o.devices[0] = devs[0];
  opts = &o;
  file_set_engine(o.io_engine);
  Counters cnt;
  bof_fptr none{-1, 0};
  {
    // A^T + the sort workspace must fit beside B and the block contexts; if they do not, A^T goes
    // to temporary files through the out-of-core transposition and the ordinary file pipeline
    // of the 'N' case runs on those (what the reference intends, src/blas/csrmm.cpp:355-386)
    std::vector<int64_t> iav((size_t) m + 1, 0);
    if (m > 0) {
      const int io = read_host(ia, (uint64_t) (m + 1) * 8, iav.data(), o.use_odirect != 0);
      if (io) { set_error(std::string("reading ia failed: ") + strerror(-io)); return BOF_EIO; }
    }
    const int64_t nnz = iav[(size_t) m] - iav[0];
    size_t free_b = 0, total_b = 0;
    dev_cache_release();      // cached call-lifetime blocks of earlier CSR calls are free memory for this budget
    BOF_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = o.hbm_budget > 0 ? std::min((size_t) o.hbm_budget, (size_t) (free_b * 0.9)) : (size_t) (free_b * 0.9);
    const size_t need = (size_t) std::max<int64_t>(nnz, 0) * 24 + (size_t) (m + n + 2) * 8 +
                        csrcsc_workspace_bytes((int64_t) n, std::max<int64_t>(nnz, 0)) + (size_t) m * k * 4;
    if (nnz > 0 && need > budget) {
      const int anchor = c.fd >= 0 ? c.fd : a.fd;
      const int t_ia = temp_file_near(anchor), t_ja = temp_file_near(anchor), t_a = temp_file_near(anchor);
      Cleanup guard;
      guard.add([&] {
        for (int fd : {t_ia, t_ja, t_a})
          if (fd >= 0) { file_forget(fd); close(fd); }
      });
      if (t_ia < 0 || t_ja < 0 || t_a < 0) { set_error("csrmm 'T': cannot create temporary files"); return BOF_EIO; }
      bof_options plain = o;
      plain.use_odirect = 0;   // the temporaries are buffered descriptors
      rc = flash_csrcsc_blocked((int64_t) m, (int64_t) n, iav, ja, a, bof_fptr{t_ia, 0}, bof_fptr{t_ja, 0},
                                bof_fptr{t_a, 0}, plain, budget, cnt);
      if (rc) return rc;
      return flash_csr_impl(true, 'N', (int64_t) n, (int64_t) m, (int64_t) k, alpha, beta, bof_fptr{t_a, 0},
                            bof_fptr{t_ia, 0}, bof_fptr{t_ja, 0}, ord_b, b, c, hb, hc, opts, nullptr, &cnt);
    }
  }
  ResidentCsr T;
  rc = flash_transpose_to_hbm((int64_t) m, (int64_t) n, a, ia, ja, o, cnt, T);
  if (rc) return rc;
  return flash_csr_impl(true, 'N', (int64_t) n, (int64_t) m, (int64_t) k, alpha, beta, none, none, none,
                        ord_b, b, c, hb, hc, opts, &T, &cnt);
}

}  // namespace bof

using namespace bof;

extern "C" {

int bof_flash_csrmm(char trans_a, uint64_t m, uint64_t n, uint64_t k, float alpha, float beta,
                    bof_fptr a, bof_fptr ia, bof_fptr ja, char ord_b, bof_fptr b, bof_fptr c,
                    const bof_options *opts) {
  if (trans_a != 'N' && trans_a != 'T') {  // reference csrmm.cpp:446-449
    set_error("bof_flash_csrmm: unrecognized value for param: trans_a");
    return BOF_EINVAL;
  }
  if (ord_b != 'R' && ord_b != 'C') {      // reference csrmm.cpp:433-436, 442-445
    set_error("bof_flash_csrmm: unrecognized value for param: ord_b");
    return BOF_EINVAL;
  }
  if (n > (uint64_t) INT32_MAX || (trans_a == 'T' && m > (uint64_t) INT32_MAX) || a.fd < 0 ||
      ia.fd < 0 || ja.fd < 0 || b.fd < 0 || c.fd < 0) {
    set_error("bof_flash_csrmm: bad argument");
    return BOF_EINVAL;
  }
  if (k == 0) return BOF_OK;
  if (trans_a == 'T') return flash_csrmm_trans(m, n, k, alpha, beta, a, ia, ja, ord_b, b, c, nullptr, nullptr, opts);
  return flash_csr_impl(true, 'N', (int64_t) m, (int64_t) n, (int64_t) k, alpha, beta, a, ia, ja,
                        ord_b, b, c, nullptr, nullptr, opts);
}

int bof_flash_csrcsc(uint64_t m, uint64_t n, bof_fptr ia, bof_fptr ja, bof_fptr a, bof_fptr ia_tr,
                     bof_fptr ja_tr, bof_fptr a_tr, const bof_options *opts) {
  if (m > (uint64_t) INT32_MAX || n > (uint64_t) INT32_MAX || ia.fd < 0 || ja.fd < 0 || a.fd < 0 ||
      ia_tr.fd < 0 || ja_tr.fd < 0 || a_tr.fd < 0) {
    set_error("bof_flash_csrcsc: bad argument (m, n must fit 31 bits)");
    return BOF_EINVAL;
  }
  return flash_csrcsc_impl((int64_t) m, (int64_t) n, ia, ja, a, ia_tr, ja_tr, a_tr, opts);
}

int bof_flash_csrmm_inmem(char trans_a, uint64_t m, uint64_t n, uint64_t k, float alpha, float beta,
                          bof_fptr a, bof_fptr ia, bof_fptr ja, char ord_b, const float *b, float *c,
                          const bof_options *opts) {
  if ((trans_a != 'N' && trans_a != 'T') || (ord_b != 'R' && ord_b != 'C') || !b || !c || a.fd < 0 ||
      ia.fd < 0 || ja.fd < 0 || n > (uint64_t) INT32_MAX || (trans_a == 'T' && m > (uint64_t) INT32_MAX)) {
    set_error("bof_flash_csrmm_inmem: bad argument");
    return BOF_EINVAL;
  }
  if (k == 0) return BOF_OK;
  bof_fptr none{-1, 0};
  if (trans_a == 'T') return flash_csrmm_trans(m, n, k, alpha, beta, a, ia, ja, ord_b, none, none, b, c, opts);
  return flash_csr_impl(true, 'N', (int64_t) m, (int64_t) n, (int64_t) k, alpha, beta, a, ia, ja,
                        ord_b, none, none, b, c, opts);
}

int bof_flash_csrgemv(char trans_a, uint64_t m, uint64_t n, bof_fptr a, bof_fptr ia, bof_fptr ja,
                      const float *b, float *c, const bof_options *opts) {
  if ((trans_a != 'N' && trans_a != 'T') || !b || !c || a.fd < 0 || ia.fd < 0 || ja.fd < 0 ||
      n > (uint64_t) INT32_MAX) {
    set_error("bof_flash_csrgemv: bad argument");
    return BOF_EINVAL;
  }
  bof_fptr none{-1, 0};
  return flash_csr_impl(false, trans_a, (int64_t) m, (int64_t) n, 1, 1.f, 0.f, a, ia, ja, 'R', none,
                        none, b, c, opts);
}

}  // extern "C"

extern "C" int bof_flash_last_c_file(uint64_t *twin_bytes) {
  if (twin_bytes) *twin_bytes = bof::g_last_c_twin_bytes.load();
  return bof::g_last_c_mode.load();
}
