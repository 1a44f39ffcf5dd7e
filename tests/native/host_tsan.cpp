// Host-only harness run under ThreadSanitizer by tests/test_host_sanitizers.py: the pieces of the level-3
// runtime whose correctness is an ordering argument rather than a computation --
//   * ShareRing (share_ring.h): the node-shared staging ring of the one-process-per-GPU GEMM.  Here its
//     "ranks" are threads on ONE mapping, so that ThreadSanitizer sees the slot memory and the futex words as
//     the same addresses: the release / acquire chain producer -> consumers -> next producer must cover the
//     memcpy into and out of every slot;
//   * fileio.cpp's shared mapping for large buffered writes against file_forget / file_unmap_all running at the
//     same time (the in-flight count of round 2's review), many writers on one descriptor;
//   * WorkQueue and StallWatch (flash_common.h).
// No GPU involved; stand-ins below for the device half of the library.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "flash_common.h"
#include "share_ring.h"

namespace bof {
void set_error(const std::string &msg) { fprintf(stderr, "set_error: %s\n", msg.c_str()); }
int hip_fail(hipError_t, const char *) { return BOF_EHIP; }
long env_long(const char *name, long dflt) {
  const char *v = getenv(name);
  return v && *v ? atol(v) : dflt;
}
}  // namespace bof

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); exit(1); } \
  } while (0)

static void share_ring() {
  const std::string name = "/bof_tsan_" + std::to_string(getpid());
  bof::ShareRing::unlink(name);
  bof::ShareRing ring;
  const size_t chunk = 64 << 10;
  const int world = 4, n_slots = 3;
  const size_t n_chunks = 120;
  CHECK(ring.map(name, chunk, n_slots));
  std::atomic<int> stop{0};
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (int rank = 0; rank < world; rank++)
    th.emplace_back([&, rank] {
      std::vector<uint64_t> buf(chunk / 8);
      for (size_t c = 0; c < n_chunks; c++) {
        int rc;
        if ((int) (c % world) == rank) {
          for (size_t i = 0; i < buf.size(); i++) buf[i] = c * 1000003ull + i;
          rc = ring.produce(c, buf.data(), chunk, world, 30.0, stop);
        } else {
          rc = ring.consume(c, buf.data(), chunk, 30.0, stop);
          for (size_t i = 0; i < buf.size() && !rc; i += 97)
            if (buf[i] != c * 1000003ull + i) rc = -EILSEQ;
        }
        if (rc) { bad.store(rc); ring.fail_all(); return; }
      }
    });
  for (auto &t : th) t.join();
  CHECK(bad.load() == 0);
  // a rank that gives up: everybody else comes back with an error instead of waiting for the time-out
  std::thread quitter([&] { ring.fail_all(); });
  std::vector<uint64_t> buf(chunk / 8);
  const int rc = ring.consume(n_chunks + 1, buf.data(), chunk, 30.0, stop);
  quitter.join();
  CHECK(rc == -EIO);
  ring.unmap();
  bof::ShareRing::unlink(name);
}

static void mapped_writes(const char *dir) {
  const std::string path = std::string(dir) + "/tsan.bin";
  const uint64_t block = 2u << 20, n_blocks = 24;
  {
    std::vector<char> zeros(block, 0);
    FILE *f = fopen(path.c_str(), "wb");
    CHECK(f);
    for (uint64_t i = 0; i < n_blocks; i++) CHECK(fwrite(zeros.data(), 1, block, f) == block);   // cached pages, no holes
    fclose(f);
  }
  const int fd = open(path.c_str(), O_RDWR);
  CHECK(fd >= 0);
  std::atomic<bool> done{false};
  std::vector<std::thread> th;
  std::atomic<int> bad{0};
  for (int t = 0; t < 6; t++)
    th.emplace_back([&, t] {
      std::vector<char> src(block);
      for (int round = 0; round < 6; round++)
        for (uint64_t b = (uint64_t) t; b < n_blocks; b += 6) {
          memset(src.data(), (int) (1 + (b + (uint64_t) round) % 200), block);
          if (bof::file_swrite(fd, b * block, 0, 1, block, src.data(), false)) bad.store(1);
        }
    });
  std::thread forgetter([&] {   // what bof_file_forget / bof_flash_release do while stores are in flight
    for (int i = 0; !done.load(); i++) {
      if (i & 1) bof::file_forget(fd); else bof::file_unmap_all();
      usleep(300);
    }
  });
  for (auto &t : th) t.join();
  done.store(true);
  forgetter.join();
  CHECK(bad.load() == 0);
  std::vector<char> chk(block);
  for (uint64_t b = 0; b < n_blocks; b++) {
    CHECK(bof::file_sread(fd, b * block, 0, 1, block, chk.data(), false) == 0);
    const char want = (char) (1 + (b + 5) % 200);
    CHECK(chk[0] == want && chk[block - 1] == want && chk[block / 2] == want);
  }
  bof::file_forget(fd);
  close(fd);
  unlink(path.c_str());
}

// O_DIRECT requests from many threads at once: the per-thread / pooled kernel-AIO contexts (or io_uring rings
// with BOF_IO_ENGINE=uring) and the request counters
static void direct_io(const char *dir) {
  const std::string path = std::string(dir) + "/tsan_direct.bin";
  const uint64_t band = 1u << 20, n_thr = 8, rounds = 6;
  {
    std::vector<char> img(band);
    FILE *f = fopen(path.c_str(), "wb");
    CHECK(f);
    for (uint64_t t = 0; t < n_thr; t++) {
      memset(img.data(), (int) t, band);
      CHECK(fwrite(img.data(), 1, band, f) == band);
    }
    fclose(f);
  }
  int fd = open(path.c_str(), O_RDWR | O_DIRECT);
  const bool direct = fd >= 0;
  if (!direct) fd = open(path.c_str(), O_RDWR);
  CHECK(fd >= 0);
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (uint64_t t = 0; t < n_thr; t++)
    th.emplace_back([&, t] {
      void *buf = nullptr;
      if (posix_memalign(&buf, 4096, band)) { bad.store(1); return; }
      unsigned char *p = (unsigned char *) buf;
      for (uint64_t r = 0; r < rounds; r++) {
        // the band as 16 strided rows of 64 KiB, then as one request
        if (bof::file_sread(fd, t * band, 64 << 10, 16, 64 << 10, buf, direct)) bad.store(2);
        if (p[0] != (unsigned char) (t + r) || p[band - 1] != (unsigned char) (t + r)) bad.store(3);
        memset(buf, (int) (t + r + 1), band);
        if (bof::file_swrite(fd, t * band, 0, 1, band, buf, direct)) bad.store(4);
      }
      free(buf);
    });
  for (auto &t : th) t.join();
  CHECK(bad.load() == 0);
  uint64_t rd = 0, wr = 0;
  bof::file_io_ops(&rd, &wr);
  CHECK(rd > 0 && wr > 0);
  bof::file_forget(fd);
  close(fd);
  unlink(path.c_str());
}

static void queue_and_watch() {
  bof::WorkQueue<int> q;
  std::atomic<long> sum{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 4; t++)
    th.emplace_back([&] {
      int v;
      while (q.pop(v)) sum += v;
    });
  long want = 0;
  for (int i = 1; i <= 2000; i++) { q.push(i); want += i; }
  q.close();
  for (auto &t : th) t.join();
  CHECK(sum.load() == want);

  // a watch over a counter that moves is silent; over one that stands still it fires once, then is destroyed
  setenv("BOF_STALL_TIMEOUT_S", "1", 1);
  std::atomic<uint64_t> progress{0};
  std::atomic<int> fired{0};
  {
    bof::StallWatch w("tsan harness (moving)", [&] { return progress.load(); }, [&] { fired++; });
    for (int i = 0; i < 30; i++) { progress++; usleep(50 * 1000); }
  }
  CHECK(fired.load() == 0);
  {
    bof::StallWatch w("tsan harness (standing still: this line is expected)", [&] { return progress.load(); }, [&] { fired++; });
    for (int i = 0; i < 60 && !fired.load(); i++) usleep(50 * 1000);
  }
  CHECK(fired.load() == 1);
  setenv("BOF_STALL_TIMEOUT_S", "0", 1);
  { bof::StallWatch off("off", [&] { return progress.load(); }, [&] { fired++; }); usleep(20 * 1000); }
  CHECK(fired.load() == 1);
}

int main(int argc, char **argv) {
  CHECK(argc > 1);
  share_ring();
  mapped_writes(argv[1]);
  direct_io(argv[1]);
  queue_and_watch();
  printf("host_tsan ok (mapped-write bytes %llu)\n", (unsigned long long) bof::file_mapped_write_bytes());
  return 0;
}
