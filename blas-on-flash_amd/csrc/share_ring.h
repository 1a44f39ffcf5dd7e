// share_ring.h -- the node-shared staging ring of a share_world > 1 level-3 call (one process per GPU):
// pure host code (POSIX shared memory + futex), used by flash_gemm_panels.cpp and exercised without a GPU
// through bof_share_selftest (tests/test_dist_gloo.py).
//
// A shared operand on its way between the ranks: a RING of `n_slots` chunk-sized slots in POSIX shared memory
// (the pages are allocated on the first lap and reused: a whole image of B in tmpfs cost a page fault + a
// zeroed page per 4 KiB, 16 GiB published at ~4 GB/s) with two words per slot: `ready` = 1 + the index of the
// chunk it holds (0: none yet, ~0: a rank failed) and `consumed` = how many peers have copied that chunk out.
// Chunk c goes to slot c % n_slots; its owner waits until the slot's previous occupant (chunk c - n_slots) has
// been taken by all world - 1 peers, copies the chunk in and publishes; a peer waits for `ready` == c + 1,
// copies the chunk out and adds itself to `consumed`.  Both waits are futex waits.  All ranks handle the
// shared chunks in the same order and their readers take requests in queue order, so whoever the earliest
// unfinished chunk waits for has already passed everything that chunk's slot depends on: no cycle of waits.
// Failure: besides the per-slot `ready` = ~0 there is ONE ring-group-wide word in a segment of its own
// (`<share_name>.fail`, shared by the rings of all operands of the call).  A rank that gives up -- or that cannot
// take part at all (its call is not eligible for the panel path and never maps a ring) -- sets it; every wait of
// every rank tests it, so nobody sits out the timeout and a later publish cannot erase the news.
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>

namespace bof {

struct ShareSlot { uint32_t ready, consumed; };
constexpr uint32_t kShareFailed = 0xFFFFFFFFu;

inline void *shm_map(const std::string &name, size_t bytes) {
  const int fd = ::shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) return nullptr;
  void *p = MAP_FAILED;
  if (::ftruncate(fd, (off_t) bytes) == 0) p = ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  ::close(fd);
  return p == MAP_FAILED ? nullptr : p;
}
inline void word_publish(uint32_t *f, uint32_t v) {
  __atomic_store_n(f, v, __ATOMIC_RELEASE);
  ::syscall(SYS_futex, f, FUTEX_WAKE, INT32_MAX, nullptr, nullptr, 0);
}
inline void word_add_publish(uint32_t *f) {
  __atomic_fetch_add(f, 1u, __ATOMIC_ACQ_REL);
  ::syscall(SYS_futex, f, FUTEX_WAKE, INT32_MAX, nullptr, nullptr, 0);
}
// waits until done(*f); returns false after `timeout_s` / once `stop` is set
template <class Pred>
inline bool word_wait(uint32_t *f, Pred done, double timeout_s, const std::atomic<int> &stop) {
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
  for (;;) {
    const uint32_t v = __atomic_load_n(f, __ATOMIC_ACQUIRE);
    if (done(v)) return true;
    if (stop.load() || std::chrono::steady_clock::now() >= t_end) return false;
    struct timespec ts = {0, 200 * 1000 * 1000};   // re-check the stop flag / deadline five times a second
    ::syscall(SYS_futex, f, FUTEX_WAIT, v, &ts, nullptr, 0);
  }
}

struct ShareRing {
  char *base = nullptr;
  size_t slot_bytes = 0;
  int n_slots = 0;
  ShareSlot *st = nullptr;
  uint32_t *fail_word = nullptr;     // the ring group's "a rank has failed" word (`<group>.fail`), or null

  // maps (creating if need be) `name`.data / `name`.flags and the group's failure word; false + errno on failure
  bool map(const std::string &name, size_t slot_bytes_, int n_slots_, const std::string &group = std::string()) {
    slot_bytes = slot_bytes_;
    n_slots = n_slots_;
    base = (char *) shm_map(name + ".data", slot_bytes * (size_t) n_slots);
    st = (ShareSlot *) shm_map(name + ".flags", sizeof(ShareSlot) * (size_t) n_slots);
    if (!group.empty()) fail_word = (uint32_t *) shm_map(group + ".fail", 64);
    return base && st && (group.empty() || fail_word);
  }
  void unmap() {
    if (base) ::munmap(base, slot_bytes * (size_t) n_slots);
    if (st) ::munmap(st, sizeof(ShareSlot) * (size_t) n_slots);
    if (fail_word) ::munmap(fail_word, 64);
    base = nullptr; st = nullptr; fail_word = nullptr;
  }
  static void unlink(const std::string &name) {
    (void) ::shm_unlink((name + ".data").c_str());
    (void) ::shm_unlink((name + ".flags").c_str());
  }
  static void unlink_group(const std::string &group) { (void) ::shm_unlink((group + ".fail").c_str()); }
  // a rank that cannot take part in the call at all: the peers' waits end at their next check
  static void mark_group_failed(const std::string &group) {
    uint32_t *w = (uint32_t *) shm_map(group + ".fail", 64);
    if (!w) return;
    word_publish(w, 1u);
    ::munmap(w, 64);
  }
  bool group_failed() const { return fail_word && __atomic_load_n(fail_word, __ATOMIC_ACQUIRE) != 0; }
  // the owner of chunk `ci` puts it into the ring (src == nullptr: tells the peers that it will not come);
  // 0, -ETIMEDOUT, or -ECANCELED when `stop` was raised
  int produce(size_t ci, const void *src, size_t bytes, int world, double timeout_s, const std::atomic<int> &stop) {
    ShareSlot *sl = st + ci % (size_t) n_slots;
    if (!src) { word_publish(&sl->ready, kShareFailed); return 0; }
    if (group_failed()) { word_publish(&sl->ready, kShareFailed); return -EIO; }
    if (ci >= (size_t) n_slots) {   // the slot's previous occupant must have been taken by every peer
      const uint32_t prev = (uint32_t) (ci - (size_t) n_slots) + 1, peers = (uint32_t) world - 1;
      bool dead = false;
      const bool ok = word_wait(&sl->consumed, [&](uint32_t v) {
        const uint32_t r = __atomic_load_n(&sl->ready, __ATOMIC_ACQUIRE);
        if (r == kShareFailed || group_failed()) { dead = true; return true; }     // a peer gave up: fail fast
        return r == prev && v >= peers; }, timeout_s, stop);
      if (!ok || dead) {
        word_publish(&sl->ready, kShareFailed);
        return dead ? -EIO : stop.load() ? -ECANCELED : -ETIMEDOUT;
      }
    }
    __atomic_store_n(&sl->consumed, 0u, __ATOMIC_RELAXED);
    memcpy(base + (ci % (size_t) n_slots) * slot_bytes, src, bytes);
    word_publish(&sl->ready, (uint32_t) ci + 1);
    return 0;
  }
  // a peer takes chunk `ci` out of the ring: 0, -EIO (its owner failed), -ETIMEDOUT, -ECANCELED
  int consume(size_t ci, void *dst, size_t bytes, double timeout_s, const std::atomic<int> &stop) {
    ShareSlot *sl = st + ci % (size_t) n_slots;
    const uint32_t want = (uint32_t) ci + 1;
    uint32_t seen = 0;
    const bool ok = word_wait(&sl->ready, [&](uint32_t v) { seen = v; return v == want || v == kShareFailed || group_failed(); },
                              timeout_s, stop);
    if (!ok) return stop.load() ? -ECANCELED : -ETIMEDOUT;
    if (seen != want) return -EIO;
    memcpy(dst, base + (ci % (size_t) n_slots) * slot_bytes, bytes);
    word_add_publish(&sl->consumed);
    return 0;
  }
  void fail_all() {   // a rank that gives up: nobody waits for its chunks, and the news cannot be overwritten
    if (fail_word) word_publish(fail_word, 1u);
    for (int q = 0; st && q < n_slots; q++) word_publish(&st[q].ready, kShareFailed);
  }
};

}  // namespace bof
