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
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
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
void file_forget(int fd) {
  std::lock_guard<std::mutex> lk(g_twin_mu);
  auto it = g_twin.find(fd);
  if (it != g_twin.end()) {
    if (it->second >= 0) ::close(it->second);
    g_twin.erase(it);
  }
}
bool file_is_direct(int fd) {
  const int fl = fcntl(fd, F_GETFL);
  return fl >= 0 && (fl & O_DIRECT);
}

static int rw_full(int fd, bool wr, char *buf, uint64_t len, uint64_t off) {
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
  static const bool use_uring = getenv("BOF_IO_ENGINE") && !strcmp(getenv("BOF_IO_ENGINE"), "uring");
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
