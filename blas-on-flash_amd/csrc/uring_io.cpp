// uring_io.cpp -- io_uring engine of the file reader/writer, raw syscalls only
// (liburing is not required): one ring per I/O thread, the pinned staging slots registered as
// FIXED buffers so that the kernel neither pins nor maps their pages per request
// (IORING_OP_READ_FIXED / WRITE_FIXED); requests that lie outside every registered buffer use
// the plain IORING_OP_READ / WRITE.  Selected with BOF_IO_ENGINE=uring behind the same
// file_sread / file_swrite interface as the kernel-AIO engine of fileio.cpp, which stays the
// default and the fallback (uring_run returns -ENOSYS when rings cannot be created).
#include <errno.h>
#include <linux/io_uring.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "fileio.h"

namespace bof {

namespace {

// ---- fixed-buffer table (the pinned ring slots), process-wide -----------------------------
struct BufTable {
  std::mutex mu;
  std::vector<struct iovec> bufs;
  std::atomic<uint64_t> gen{1};
} g_tab;

struct Uring {
  int fd = -1;
  bool ok = false, fixed = false;
  uint64_t reg_gen = 0;
  std::vector<struct iovec> reg;  // what this ring has registered
  unsigned entries = 0;
  unsigned *sq_head = nullptr, *sq_tail = nullptr, *sq_mask = nullptr, *sq_array = nullptr;
  unsigned *cq_head = nullptr, *cq_tail = nullptr, *cq_mask = nullptr;
  struct io_uring_sqe *sqes = nullptr;
  struct io_uring_cqe *cqes = nullptr;
  void *sq_ptr = nullptr, *cq_ptr = nullptr;
  size_t sq_sz = 0, cq_sz = 0, sqes_sz = 0;

  Uring() {
    struct io_uring_params p;
    memset(&p, 0, sizeof(p));
    fd = (int) syscall(__NR_io_uring_setup, 128, &p);
    if (fd < 0) return;
    entries = p.sq_entries;
    sq_sz = p.sq_off.array + p.sq_entries * sizeof(unsigned);
    cq_sz = p.cq_off.cqes + p.cq_entries * sizeof(struct io_uring_cqe);
    const bool single = (p.features & IORING_FEAT_SINGLE_MMAP) != 0;
    if (single) sq_sz = cq_sz = std::max(sq_sz, cq_sz);
    sq_ptr = mmap(nullptr, sq_sz, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, IORING_OFF_SQ_RING);
    if (sq_ptr == MAP_FAILED) { sq_ptr = nullptr; return; }
    cq_ptr = single ? sq_ptr
                    : mmap(nullptr, cq_sz, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, IORING_OFF_CQ_RING);
    if (cq_ptr == MAP_FAILED) { cq_ptr = nullptr; return; }
    sqes_sz = p.sq_entries * sizeof(struct io_uring_sqe);
    sqes = (struct io_uring_sqe *) mmap(nullptr, sqes_sz, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd,
                                        IORING_OFF_SQES);
    if (sqes == MAP_FAILED) { sqes = nullptr; return; }
    char *s = (char *) sq_ptr, *c = (char *) cq_ptr;
    sq_head = (unsigned *) (s + p.sq_off.head);
    sq_tail = (unsigned *) (s + p.sq_off.tail);
    sq_mask = (unsigned *) (s + p.sq_off.ring_mask);
    sq_array = (unsigned *) (s + p.sq_off.array);
    cq_head = (unsigned *) (c + p.cq_off.head);
    cq_tail = (unsigned *) (c + p.cq_off.tail);
    cq_mask = (unsigned *) (c + p.cq_off.ring_mask);
    cqes = (struct io_uring_cqe *) (c + p.cq_off.cqes);
    ok = true;
  }
  ~Uring() {
    if (sqes) munmap(sqes, sqes_sz);
    if (cq_ptr && cq_ptr != sq_ptr) munmap(cq_ptr, cq_sz);
    if (sq_ptr) munmap(sq_ptr, sq_sz);
    if (fd >= 0) close(fd);
  }

  // bring this ring's registration in line with the process-wide table
  void sync_buffers() {
    const uint64_t g = g_tab.gen.load();
    if (g == reg_gen) return;
    std::vector<struct iovec> want;
    {
      std::lock_guard<std::mutex> lk(g_tab.mu);
      want = g_tab.bufs;
    }
    if (fixed) (void) syscall(__NR_io_uring_register, fd, IORING_UNREGISTER_BUFFERS, nullptr, 0);
    fixed = false;
    reg.clear();
    if (!want.empty() &&
        syscall(__NR_io_uring_register, fd, IORING_REGISTER_BUFFERS, want.data(), (unsigned) want.size()) == 0) {
      fixed = true;
      reg = want;
    }
    reg_gen = g;
  }
  int fixed_index(const void *p, uint64_t len) const {
    if (!fixed) return -1;
    for (size_t i = 0; i < reg.size(); i++) {
      const char *b = (const char *) reg[i].iov_base;
      if ((const char *) p >= b && (const char *) p + len <= b + reg[i].iov_len) return (int) i;
    }
    return -1;
  }
};

// Rings are leased from a pool for the same reason as the AIO contexts of fileio.cpp: the
// pipelines start fresh I/O threads per call, and a ring is expensive to build (three mappings
// plus the registration -- i.e. long-term pinning -- of every staging slot) and to tear down.
// A pooled ring keeps its buffer registration across calls; uring_run never returns with
// requests in flight, so a pooled ring is always idle.
struct RingPool {
  std::mutex mu;
  std::vector<Uring *> idle;
  Uring *get() {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!idle.empty()) { Uring *r = idle.back(); idle.pop_back(); return r; }
    }
    return new Uring();
  }
  void put(Uring *r) {
    std::lock_guard<std::mutex> lk(mu);
    idle.push_back(r);
  }
  ~RingPool() {
    for (Uring *r : idle) delete r;
  }
};
RingPool &ring_pool() {
  static RingPool p;
  return p;
}
struct RingLease {
  Uring *r;
  RingLease() : r(ring_pool().get()) {}
  ~RingLease() { ring_pool().put(r); }
};
Uring &tls_ring() {
  static thread_local RingLease l;
  return *l.r;
}

std::atomic<uint64_t> g_fixed_ops{0}, g_plain_ops{0};

}  // namespace

void file_buffers_add(void *ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_tab.mu);
  struct iovec v;
  v.iov_base = ptr;
  v.iov_len = bytes;
  g_tab.bufs.push_back(v);
  g_tab.gen++;
}
void file_buffers_remove(void *ptr) {
  std::lock_guard<std::mutex> lk(g_tab.mu);
  for (size_t i = 0; i < g_tab.bufs.size(); i++)
    if (g_tab.bufs[i].iov_base == ptr) {
      g_tab.bufs.erase(g_tab.bufs.begin() + (long) i);
      g_tab.gen++;
      return;
    }
}
// drop the buffer registration (= the long-term page pins) of every idle ring: called when the
// staging memory itself is being given back (bof_flash_release)
void uring_release_buffers() {
  RingPool &p = ring_pool();
  std::lock_guard<std::mutex> lk(p.mu);
  for (Uring *r : p.idle)
    if (r->ok && r->fixed) {
      (void) syscall(__NR_io_uring_register, r->fd, IORING_UNREGISTER_BUFFERS, nullptr, 0);
      r->fixed = false;
      r->reg.clear();
      r->reg_gen = 0;
    }
}
void uring_op_counts(uint64_t *fixed, uint64_t *plain) {
  *fixed = g_fixed_ops.load();
  *plain = g_plain_ops.load();
}

int uring_run(const std::vector<IoPiece> &pieces) {
  Uring &r = tls_ring();
  if (!r.ok) return -ENOSYS;
  r.sync_buffers();
  // placed: SQEs written into the ring; consumed: taken by the kernel; done: completions reaped
  size_t placed = 0, consumed = 0, done = 0;
  int first_err = 0, soft_retries = 0;
  // as with the AIO engine: after an error nothing new goes out, but everything in flight is
  // reaped before returning (the kernel owns the caller's buffer until then)
  while (done < placed || (!first_err && placed < pieces.size())) {
    if (!first_err) {
      unsigned tail = *r.sq_tail;
      while (placed < pieces.size() && placed - done < r.entries) {
        const IoPiece &p = pieces[placed];
        const unsigned idx = tail & *r.sq_mask;
        struct io_uring_sqe *sqe = &r.sqes[idx];
        memset(sqe, 0, sizeof(*sqe));
        const int fi = r.fixed_index(p.buf, p.len);
        sqe->opcode = (uint8_t) (fi >= 0 ? (p.wr ? IORING_OP_WRITE_FIXED : IORING_OP_READ_FIXED)
                                         : (p.wr ? IORING_OP_WRITE : IORING_OP_READ));
        sqe->fd = p.fd;
        sqe->off = p.off;
        sqe->addr = (uint64_t) (uintptr_t) p.buf;
        sqe->len = (uint32_t) p.len;
        if (fi >= 0) sqe->buf_index = (uint16_t) fi;
        sqe->user_data = placed;
        (fi >= 0 ? g_fixed_ops : g_plain_ops)++;
        r.sq_array[idx] = idx;
        tail++;
        placed++;
      }
      __atomic_store_n(r.sq_tail, tail, __ATOMIC_RELEASE);
    }
    const unsigned to_submit = (unsigned) (placed - consumed);
    if (to_submit == 0 && consumed == done) break;
    auto reap = [&] {
      unsigned head = *r.cq_head;
      const unsigned ctail = __atomic_load_n(r.cq_tail, __ATOMIC_ACQUIRE);
      while (head != ctail) {
        const struct io_uring_cqe &c = r.cqes[head & *r.cq_mask];
        const IoPiece &p = pieces[(size_t) c.user_data];
        if (c.res < 0) { if (!first_err) first_err = c.res; }
        else if ((uint64_t) c.res != p.len) { if (!first_err) first_err = -EIO; }   // short transfer
        head++;
        done++;
      }
      __atomic_store_n(r.cq_head, head, __ATOMIC_RELEASE);
    };
    const long rc = syscall(__NR_io_uring_enter, r.fd, to_submit, 1u, IORING_ENTER_GETEVENTS, nullptr, 0);
    if (rc < 0) {
      if (errno == EINTR) continue;
      const int e = -errno;
      if ((e == -EAGAIN || e == -EBUSY) && consumed > done) {
        // completion queue pressure: reap what is there and try again
        (void) syscall(__NR_io_uring_enter, r.fd, 0u, 1u, IORING_ENTER_GETEVENTS, nullptr, 0);
      } else if ((e == -EAGAIN || e == -EBUSY) && ++soft_retries <= 5) {
        usleep(200u << soft_retries);     // transient (memory pressure, io-wq limits): nothing in flight, try again
        continue;
      } else {
        // The ring cannot take this submission.  Closing an io_uring descriptor only QUEUES its
        // teardown, so what the kernel already holds is reaped first: the caller's buffer (a pinned
        // ring slot that will be reused) must be quiescent when we return.
        while (done < consumed) {
          const long w = syscall(__NR_io_uring_enter, r.fd, 0u, (unsigned) std::min<size_t>(consumed - done, r.entries),
                                 IORING_ENTER_GETEVENTS, nullptr, 0);
          if (w < 0 && errno != EINTR && errno != EAGAIN && errno != EBUSY) break;
          reap();
        }
        const bool quiescent = done == consumed;
        // SQEs that were placed but never consumed must not survive into a later call: new ring
        r.~Uring();
        new (&r) Uring();
        if (first_err) return first_err;
        // nothing was taken by the kernel at all: let the AIO engine do the whole transfer
        if (quiescent && consumed == 0 && (e == -EAGAIN || e == -EBUSY || e == -ENOMEM)) return -ENOSYS;
        return e;
      }
    } else {
      consumed += (size_t) rc;
    }
    reap();
    // (after an error, SQEs the kernel has not taken yet stay valid requests: the next enter
    //  consumes them and they are reaped like the others -- they cannot be recalled)
  }
  return first_err;
}

}  // namespace bof
