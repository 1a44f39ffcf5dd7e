// fileio.cpp -- strided file <-> packed host buffer transfers for the tile reader.
//
// Host-side equivalent of the reference's FlashFileHandle::read/write/sread/swrite
// (src/file_handles/flash_file_handle.cpp:247-716): a region is
// {offset, stride, n_strides, len_per_stride} in bytes (StrideInfo,
// include/file_handles/file_handle.h:19-34) and the memory side is packed.
//
//  * sector-aligned requests on an O_DIRECT descriptor go through Linux kernel
//    AIO (raw io_setup/io_submit/io_getevents syscalls -- libaio is only a thin
//    wrapper and is not required), one iocb per stride, contiguous requests cut
//    into 32 MiB pieces, one AIO context per calling thread;
//  * anything unaligned (CSR index/value segments, leading dimensions that are
//    not multiples of 128 floats) goes through a buffered descriptor of the same
//    file, where the kernel's page cache does the read-modify-write that the
//    reference does by hand with bounce buffers and write-overlap ordering
//    (flash_file_handle.cpp:462-716, io_executor.cpp:28-156).  Linux keeps
//    O_DIRECT and buffered I/O on one file coherent at syscall granularity.
#include <errno.h>
#include <fcntl.h>
#include <linux/aio_abi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "bof_internal.h"
#include "fileio.h"

namespace bof {

static constexpr uint64_t kSector = 512;
static constexpr uint64_t kMaxChunk = 32ull << 20;  // reference MAX_CHUNK_SIZE
// A contiguous O_DIRECT transfer is cut into requests of this size, all submitted together:
// several medium requests in flight beat one huge one on every device measured so far
// (profiles/r2/iobench_*.json).  BOF_IO_REQUEST_KIB / bof_file_set_request_bytes override.
static std::atomic<uint64_t> g_req_bytes{0};
static uint64_t request_bytes() {
  uint64_t v = g_req_bytes.load(std::memory_order_relaxed);
  if (v == 0) {
    const char *e = getenv("BOF_IO_REQUEST_KIB");
    v = e && atoll(e) > 0 ? (uint64_t) atoll(e) << 10 : (4ull << 20);
    v = std::min(std::max<uint64_t>(v / kSector * kSector, kSector), kMaxChunk);
    g_req_bytes.store(v);
  }
  return v;
}
// Engine for aligned O_DIRECT requests: 1 kernel AIO, 2 io_uring.  Level-3 calls set it from
// bof_options.io_engine at their start (0 there = $BOF_IO_ENGINE, read at that call, else AIO).
static std::atomic<int> g_engine{0};
void file_set_engine(int engine) {
  if (engine != 1 && engine != 2) {
    const char *e = getenv("BOF_IO_ENGINE");
    engine = e && !strcmp(e, "uring") ? 2 : 1;
  }
  g_engine.store(engine, std::memory_order_relaxed);
}
static constexpr unsigned kAioEvents = 1024;
static constexpr int kIoRetries = 5;                // reference submit_and_reap retries

static inline bool al(uint64_t v, uint64_t a = kSector) { return (v % a) == 0; }

// Alignment O_DIRECT needs on this file: statx(STATX_DIOALIGN) where the kernel and the file
// system report it (4Kn NVMe namespaces need 4096), else the reference's SECTOR_LEN = 512.
// The struct is read by offset: the image's headers predate the field (Linux 6.1).
uint64_t file_dio_align(int fd) {
  alignas(8) unsigned char stx[256];
  memset(stx, 0, sizeof(stx));
  const unsigned kStatxDioAlign = 0x2000u;
#ifdef SYS_statx
  if (syscall(SYS_statx, fd, "", 0x1000 /* AT_EMPTY_PATH */, kStatxDioAlign, stx) == 0) {
    uint32_t mask, mem_align, off_align;
    memcpy(&mask, stx + 0x00, 4);
    memcpy(&mem_align, stx + 0x98, 4);
    memcpy(&off_align, stx + 0x9c, 4);
    if ((mask & kStatxDioAlign) && off_align >= kSector && (off_align & (off_align - 1)) == 0 && off_align <= 65536)
      return std::max<uint64_t>(off_align, mem_align <= 4096 ? kSector : mem_align);
  }
#endif
  return kSector;
}

// requests handed to the kernel (iocbs + pread/pwrite calls), process-wide
static std::atomic<uint64_t> g_rd_ops{0}, g_wr_ops{0};
void file_io_ops(uint64_t *reads, uint64_t *writes) {
  *reads = g_rd_ops.load();
  *writes = g_wr_ops.load();
}

// ---- per-thread AIO context -------------------------------------------------------
struct AioCtx {
  aio_context_t ctx = 0;
  bool ok = false;
  AioCtx() { ok = (syscall(SYS_io_setup, kAioEvents, &ctx) == 0); }
  ~AioCtx() { if (ok) syscall(SYS_io_destroy, ctx); }
};
// Contexts are leased, not owned, by threads.  io_setup / io_destroy are expensive on a large
// machine (io_destroy waits for an RCU grace period: the twelve I/O threads of one level-3 call
// spent 0.5 s of a 1.4 s cfg2 run just exiting, profiles/r2/e2e_panels_trace_level2_aio.txt), and
// the pipelines start fresh reader / writer threads per call.  A thread takes a context from the
// pool at its first request and hands it back when it exits; aio_run never returns with
// requests in flight, so a pooled context is always idle.
struct AioPool {
  std::mutex mu;
  std::vector<AioCtx *> idle;
  AioCtx *get() {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!idle.empty()) { AioCtx *c = idle.back(); idle.pop_back(); return c; }
    }
    return new AioCtx();
  }
  void put(AioCtx *c) {
    std::lock_guard<std::mutex> lk(mu);
    idle.push_back(c);
  }
  ~AioPool() {
    for (AioCtx *c : idle) delete c;
  }
};
static AioPool &aio_pool() {
  static AioPool p;
  return p;
}
struct AioLease {
  AioCtx *c;
  AioLease() : c(aio_pool().get()) {}
  ~AioLease() { aio_pool().put(c); }
};
static AioCtx &tls_ctx() {
  static thread_local AioLease l;
  return *l.c;
}

static int aio_run(std::vector<struct iocb> &cbs) {
  AioCtx &c = tls_ctx();
  if (!c.ok) return -ENOSYS;
  std::vector<struct iocb *> ptrs(cbs.size());
  for (size_t i = 0; i < cbs.size(); i++) ptrs[i] = &cbs[i];
  std::vector<struct io_event> evs(kAioEvents);
  size_t submitted = 0, done = 0;
  int retries = 0;
  int first_err = 0;  // once set nothing new is submitted, but everything in flight is reaped:
                      // the context is per thread and outlives this call, and the kernel may
                      // still be writing into the caller's buffer (and into `cbs`)
  while (done < submitted || (!first_err && submitted < cbs.size())) {
    const size_t inflight = submitted - done;
    if (!first_err && submitted < cbs.size() && inflight < kAioEvents) {
      const long want = (long) std::min<size_t>(cbs.size() - submitted, kAioEvents - inflight);
      long r = syscall(SYS_io_submit, c.ctx, want, ptrs.data() + submitted);
      if (r < 0) {
        if ((errno == EAGAIN || errno == EINTR) && ++retries <= kIoRetries) continue;
        if (inflight == 0 || errno != EAGAIN) first_err = -errno;
      } else {
        submitted += (size_t) r;
      }
    }
    if (submitted > done) {
      long r = syscall(SYS_io_getevents, c.ctx, 1L, (long) std::min<size_t>(submitted - done, kAioEvents),
                       evs.data(), nullptr);
      if (r < 0) {
        if (errno == EINTR) continue;
        // the context itself is unusable: replace it so that stale completions can never be
        // delivered to a later call (io_destroy cancels / waits for what is still in flight)
        const int e = -errno;
        syscall(SYS_io_destroy, c.ctx);
        c.ctx = 0;
        c.ok = (syscall(SYS_io_setup, kAioEvents, &c.ctx) == 0);
        return first_err ? first_err : e;
      }
      for (long i = 0; i < r; i++) {
        const struct iocb *cb = reinterpret_cast<const struct iocb *>(evs[i].obj);
        if ((int64_t) evs[i].res < 0) { if (!first_err) first_err = (int) evs[i].res; }
        else if ((uint64_t) evs[i].res != cb->aio_nbytes) { if (!first_err) first_err = -EIO; }  // short transfer
      }
      done += (size_t) r;
    }
  }
  return first_err;
}

// ---- buffered twin of an O_DIRECT descriptor ---------------------------------------
static std::mutex g_twin_mu;
static std::unordered_map<int, int> g_twin;
static int buffered_twin(int fd) {
  std::lock_guard<std::mutex> lk(g_twin_mu);
  auto it = g_twin.find(fd);
  if (it != g_twin.end()) {
    // descriptor numbers get recycled: the cached twin must still name the same file
    struct stat a, b;
    if (it->second >= 0 && fstat(fd, &a) == 0 && fstat(it->second, &b) == 0 &&
        a.st_dev == b.st_dev && a.st_ino == b.st_ino)
      return it->second;
    if (it->second >= 0) ::close(it->second);
    g_twin.erase(it);
  }
  const std::string path = "/proc/self/fd/" + std::to_string(fd);
  int t = ::open(path.c_str(), O_RDWR);
  if (t < 0) t = ::open(path.c_str(), O_RDONLY);
  g_twin[fd] = t;
  return t;
}
int file_buffered_fd(int fd) { return file_is_direct(fd) ? buffered_twin(fd) : fd; }
void unmap_for_forget(int fd);
void file_forget(int fd) {
  std::lock_guard<std::mutex> lk(g_twin_mu);
  unmap_for_forget(fd);
  auto it = g_twin.find(fd);
  if (it != g_twin.end()) {
    if (it->second >= 0) {
      unmap_for_forget(it->second);
      ::close(it->second);
    }
    g_twin.erase(it);
  }
}
bool file_is_direct(int fd) {
  const int fl = fcntl(fd, F_GETFL);
  return fl >= 0 && (fl & O_DIRECT);
}

// ---- large buffered writes into cached pages: through a shared mapping ----------------------
// pwrite() into one file serialises on its inode lock: 13 GB/s whatever the number of writer
// threads, while stores through a MAP_SHARED mapping reach 125-185 GB/s with 8-16 threads
// (profiles/r2/iobench_*.json) -- that lock, not PCIe, bounded cfg3 from page-cache-resident
// files (the 5.12 GB of C: 0.39 s of a 0.44 s call).  A store into a page that is NOT in the page
// cache would first fault it in from the device, so the mapping is used only for ranges that
// mincore() reports resident (a file that was just written or read: the buffered case), in files
// without holes;
// anything else takes pwrite.  The mapping is per descriptor, checked against the file's identity
// on every use (a descriptor number may have been reused) and only used while the file keeps the
// size it was mapped with; dropped by file_forget.
// BOF_MMAP_WRITES=0 turns it off.
namespace {
struct MapEntry {
  char *base = nullptr;
  uint64_t size = 0;
  dev_t dev = 0;
  ino_t ino = 0;
  bool no_holes = false;   // every block of the file is allocated (checked until it is)
  int users = 0;           // stores in flight through the mapping (guarded by g_map_mu): nobody unmaps under them
};

// A store into a hole has its block allocated at fault / write-back time, where a full disk means
// SIGBUS or lost data instead of pwrite's ENOSPC: the mapping is only used for files without
// holes.  Asked through a private descriptor (lseek moves the file position of the one it is given).
bool file_has_no_holes(int fd, uint64_t size) {
  char path[64];
  snprintf(path, sizeof(path), "/proc/self/fd/%d", fd);
  const int probe = ::open(path, O_RDONLY | O_CLOEXEC);
  if (probe < 0) return false;
  const off_t hole = ::lseek(probe, 0, SEEK_HOLE);
  ::close(probe);
  return hole >= 0 && (uint64_t) hole >= size;
}
std::mutex g_map_mu;
std::condition_variable g_map_cv;    // a mapping's last user has left
std::unordered_map<int, MapEntry> g_map;
std::atomic<uint64_t> g_mapped_bytes{0};

// Drops the mapping of fd once no store is in flight through it (file_forget and
// bof_flash_release may run while another thread is inside mapped_write's memcpy).
void unmap_locked(int fd, std::unique_lock<std::mutex> &lk) {
  auto it = g_map.find(fd);
  if (it == g_map.end()) return;
  g_map_cv.wait(lk, [&] {
    it = g_map.find(fd);
    return it == g_map.end() || it->second.users == 0;
  });
  if (it == g_map.end()) return;
  if (it->second.base) ::munmap(it->second.base, it->second.size);
  g_map.erase(it);
}

bool mapped_write(int fd, const char *buf, uint64_t len, uint64_t off) {
  const bool on = !getenv("BOF_MMAP_WRITES") || atoi(getenv("BOF_MMAP_WRITES")) != 0;
  if (!on || len < (1u << 20)) return false;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || off + len > (uint64_t) sb.st_size) return false;
  char *base = nullptr;
  {
    std::unique_lock<std::mutex> lk(g_map_mu);
    auto it = g_map.find(fd);
    if (it != g_map.end() && (it->second.dev != sb.st_dev || it->second.ino != sb.st_ino)) {
      unmap_locked(fd, lk);   // the descriptor number now names another file
      it = g_map.end();
    }
    // same file, other size (it grew under pwrite): other threads may be storing through the
    // mapping right now, so it is left alone and this request takes pwrite
    if (it != g_map.end() && it->second.size != (uint64_t) sb.st_size) return false;
    if (it != g_map.end() && it->second.base && !it->second.no_holes) {
      it->second.no_holes = file_has_no_holes(fd, it->second.size);
      if (!it->second.no_holes) return false;
    }
    if (it == g_map.end()) {
      MapEntry e;
      e.dev = sb.st_dev; e.ino = sb.st_ino; e.size = (uint64_t) sb.st_size;
      void *p = ::mmap(nullptr, e.size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      e.base = p == MAP_FAILED ? nullptr : (char *) p;   // nullptr: remembered, not retried (O_WRONLY, no mmap)
      if (e.base) e.no_holes = file_has_no_holes(fd, e.size);
      it = g_map.emplace(fd, e).first;
      if (e.base && !e.no_holes) return false;
    }
    base = it->second.base;
    if (!base) return false;
    it->second.users++;      // from here to the end of the copy the mapping cannot go away
  }
  bool done = false;
  const uint64_t pg = 4096, a0 = off / pg * pg, a1 = (off + len + pg - 1) / pg * pg;
  std::vector<unsigned char> vec((size_t) ((a1 - a0) / pg));
  if (::mincore(base + a0, (size_t) (a1 - a0), vec.data()) == 0) {
    done = true;
    for (unsigned char v : vec)
      if (!(v & 1)) { done = false; break; }
  }
  if (done) {
    memcpy(base + off, buf, (size_t) len);
    g_mapped_bytes += len;
  }
  {
    std::lock_guard<std::mutex> lk(g_map_mu);
    auto it = g_map.find(fd);
    if (it != g_map.end()) it->second.users--;
  }
  g_map_cv.notify_all();
  return done;
}
}  // namespace
uint64_t file_mapped_write_bytes() { return g_mapped_bytes.load(); }
void unmap_for_forget(int fd) {
  std::unique_lock<std::mutex> lk(g_map_mu);
  unmap_locked(fd, lk);
}
// every write mapping goes (bof_flash_release), each once its stores in flight are done
void file_unmap_all() {
  std::unique_lock<std::mutex> lk(g_map_mu);
  while (!g_map.empty()) unmap_locked(g_map.begin()->first, lk);
}

static int rw_full(int fd, bool wr, char *buf, uint64_t len, uint64_t off) {
  if (wr && mapped_write(fd, buf, len, off)) { g_wr_ops++; return 0; }
  uint64_t done = 0;
  int retries = 0;
  while (done < len) {
    ssize_t r = wr ? ::pwrite(fd, buf + done, len - done, (off_t) (off + done))
                   : ::pread(fd, buf + done, len - done, (off_t) (off + done));
    (wr ? g_wr_ops : g_rd_ops)++;
    if (r < 0) {
      if ((errno == EINTR || errno == EAGAIN) && ++retries <= kIoRetries) continue;
      return -errno;
    }
    if (r == 0) return -EIO;  // EOF inside the region
    done += (uint64_t) r;
  }
  return 0;
}

static int strided_io(int fd, bool wr, uint64_t offset, uint64_t stride, uint64_t n_strides,
                      uint64_t len, void *buf, bool use_aio) {
  if (n_strides == 0 || len == 0) return 0;
  if (n_strides > 1 && stride == len) {  // dense region: one contiguous transfer
    len *= n_strides;
    n_strides = 1;
  }
  const bool direct = file_is_direct(fd);
  const uint64_t A = direct ? file_dio_align(fd) : kSector;
  const bool aligned = al(offset, A) && al(len, A) && (n_strides == 1 || al(stride, A)) &&
                       al(reinterpret_cast<uintptr_t>(buf), A);
  if (direct && !aligned) {
    fd = buffered_twin(fd);
    if (fd < 0) return -EBADF;
  }
  char *p = static_cast<char *>(buf);
  if (g_engine.load(std::memory_order_relaxed) == 0) file_set_engine(0);   // first use outside a level-3 call
  const bool use_uring = g_engine.load(std::memory_order_relaxed) == 2;
  if (direct && aligned && use_aio && use_uring) {
    std::vector<IoPiece> pieces;
    const uint64_t piece = request_bytes();
    for (uint64_t s = 0; s < n_strides; s++)
      for (uint64_t o = 0; o < len; o += piece)
        pieces.push_back(IoPiece{fd, wr, p + s * len + o, std::min(piece, len - o), offset + s * stride + o});
    const int rc = uring_run(pieces);
    if (rc != -ENOSYS) {
      (wr ? g_wr_ops : g_rd_ops) += pieces.size();
      return rc;
    }
  }
  if (direct && aligned && use_aio && tls_ctx().ok) {
    std::vector<struct iocb> cbs;
    const uint64_t piece = request_bytes();
    cbs.reserve(n_strides == 1 ? (size_t) (len / piece + 1) : (size_t) n_strides);
    for (uint64_t s = 0; s < n_strides; s++) {
      for (uint64_t o = 0; o < len; o += piece) {
        struct iocb cb;
        memset(&cb, 0, sizeof(cb));
        cb.aio_fildes = (uint32_t) fd;
        cb.aio_lio_opcode = wr ? IOCB_CMD_PWRITE : IOCB_CMD_PREAD;
        cb.aio_buf = reinterpret_cast<uint64_t>(p + s * len + o);
        cb.aio_nbytes = std::min(piece, len - o);
        cb.aio_offset = (int64_t) (offset + s * stride + o);
        cbs.push_back(cb);
      }
    }
    const int rc = aio_run(cbs);
    if (rc != -ENOSYS) {
      (wr ? g_wr_ops : g_rd_ops) += cbs.size();
      return rc;
    }
  }
  for (uint64_t s = 0; s < n_strides; s++) {
    const int rc = rw_full(fd, wr, p + s * len, len, offset + s * stride);
    if (rc) return rc;
  }
  return 0;
}

// ---- unaligned regions of an O_DIRECT file without giving up O_DIRECT ---------------------------------
// The reference keeps O_DIRECT for leading dimensions that are not multiples of a sector: it reads the
// sector-aligned superset into a bounce buffer and, for writes, read-modify-writes the first and last
// sector, ordering overlapping writes of neighbouring tiles (src/file_handles/flash_file_handle.cpp:462-506,
// 558-716; src/scheduler/io_executor.cpp:28-156).  Here:
//  * read: the aligned superset goes straight into the caller's (pinned) buffer, *delta tells where the
//    first requested byte landed; what lies beyond the last whole sector of the FILE (an unaligned file
//    size) comes through the buffered twin;
//  * write: the whole PAGES of the region are written with O_DIRECT from the caller's buffer, the partial
//    first / last page through the buffered twin (then pushed to the device).  A page is therefore never
//    written by both paths -- the lost-update case the reference's ordering exists for -- and two
//    neighbours that share an edge page both go through the page cache, which merges byte ranges.
//    The caller's buffer must hold the region at an address congruent to its file offset modulo the page.
int file_read_widened(int fd, uint64_t off, uint64_t len, void *buf, uint64_t *delta, bool use_aio) {
  const uint64_t A = file_dio_align(fd);
  const uint64_t lo = off / A * A, end = off + len;
  *delta = off - lo;
  if (len == 0) return 0;
  struct stat sb;
  if (fstat(fd, &sb) != 0) return -errno;
  if ((uint64_t) sb.st_size < end) return -EIO;                     // the region itself ends beyond the file
  const uint64_t hi_direct = std::min((end + A - 1) / A * A, (uint64_t) sb.st_size / A * A);
  char *p = static_cast<char *>(buf);
  if (hi_direct > lo) {
    const int rc = strided_io(fd, false, lo, 0, 1, hi_direct - lo, p, use_aio);
    if (rc) return rc;
  }
  if (hi_direct < end) {                                             // tail inside the file's last, partial sector
    const int twin = buffered_twin(fd);
    if (twin < 0) return -EBADF;
    const uint64_t from = std::max(hi_direct, off);
    return rw_full(twin, false, p + (from - lo), end - from, from);
  }
  return 0;
}

int file_write_split(int fd, uint64_t off, uint64_t len, const void *buf, bool use_aio) {
  if (len == 0) return 0;
  const uint64_t P = std::max<uint64_t>(4096, file_dio_align(fd));
  if (reinterpret_cast<uintptr_t>(buf) % P != off % P) return -EINVAL;
  const uint64_t end = off + len;
  const uint64_t f0 = std::min((off + P - 1) / P * P, end), f1 = std::max(end / P * P, f0);
  const char *p = static_cast<const char *>(buf);
  int rc = 0;
  if (f1 > f0) rc = strided_io(fd, true, f0, 0, 1, f1 - f0, const_cast<char *>(p + (f0 - off)), use_aio);
  if (rc || (f0 == off && f1 == end)) return rc;
  const int twin = buffered_twin(fd);
  if (twin < 0) return -EBADF;
  const uint64_t edge[2][2] = {{off, f0}, {f1, end}};
  for (const auto &e : edge) {
    if (e[1] <= e[0]) continue;
    rc = rw_full(twin, true, const_cast<char *>(p + (e[0] - off)), e[1] - e[0], e[0]);
    if (rc) return rc;
    // O_DIRECT semantics for the edge as well: on its way to the device before the call returns
    if (sync_file_range(twin, (off64_t) e[0], (off64_t) (e[1] - e[0]),
                        SYNC_FILE_RANGE_WAIT_BEFORE | SYNC_FILE_RANGE_WRITE | SYNC_FILE_RANGE_WAIT_AFTER) != 0 &&
        errno != ENOSYS && errno != EINVAL && errno != ESPIPE)
      return -errno;
  }
  return 0;
}

int file_sread(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides, uint64_t len,
               void *buf, bool use_aio) {
  return strided_io(fd, false, offset, stride, n_strides, len, buf, use_aio);
}
int file_swrite(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides, uint64_t len,
                const void *buf, bool use_aio) {
  return strided_io(fd, true, offset, stride, n_strides, len, const_cast<void *>(buf), use_aio);
}

}  // namespace bof

extern "C" int bof_file_set_request_bytes(uint64_t bytes) {
  if (bytes < 512 || bytes % 512) { bof::set_error("bof_file_set_request_bytes: need a multiple of 512"); return BOF_EINVAL; }
  bof::g_req_bytes.store(std::min<uint64_t>(bytes, bof::kMaxChunk));
  return BOF_OK;
}
extern "C" int bof_file_forget(int fd) {
  bof::file_forget(fd);
  return BOF_OK;
}
extern "C" int bof_file_sread(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides,
                              uint64_t len_per_stride, void *buf, int use_aio) {
  const int rc = bof::file_sread(fd, offset, stride, n_strides, len_per_stride, buf, use_aio != 0);
  if (rc) { bof::set_error(std::string("bof_file_sread: ") + strerror(-rc)); return BOF_EIO; }
  return BOF_OK;
}
extern "C" int bof_file_swrite(int fd, uint64_t offset, uint64_t stride, uint64_t n_strides,
                               uint64_t len_per_stride, const void *buf, int use_aio) {
  const int rc = bof::file_swrite(fd, offset, stride, n_strides, len_per_stride, buf, use_aio != 0);
  if (rc) { bof::set_error(std::string("bof_file_swrite: ") + strerror(-rc)); return BOF_EIO; }
  return BOF_OK;
}
