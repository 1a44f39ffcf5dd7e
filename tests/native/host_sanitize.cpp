// Host-only harness run under AddressSanitizer + UBSan by tests/test_host_sanitizers.py:
// the tilers (plan.cpp) and the strided file reader/writer (fileio.cpp) compiled with g++,
// exercised over aligned / unaligned / strided / multi-threaded patterns.  No GPU involved
// (GPU sanitizers are not available on the pool; the host runtime pieces are checked here).
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "bof_hip.h"
#include "bof_internal.h"
#include "fileio.h"

// stand-ins for the device-side half of the library (c_api.hip and the kernels): the code under
// test here never reaches them
namespace bof {
void set_error(const std::string &msg) { fprintf(stderr, "set_error: %s\n", msg.c_str()); }
int hip_fail(hipError_t, const char *) { return BOF_EHIP; }
StreamSet *stream_set(int) { return nullptr; }
thread_local int t_ordinal_rep = 0;
int scratch_get(int, size_t, void **) { return BOF_ENODEV; }
void scratch_release_all() {}
bof_options resolved(const bof_options *o) { return o ? *o : bof_options{}; }
size_t csrcsc_workspace_bytes(int64_t, int64_t) { return 0; }
hipError_t sgemm(char, char, char, int64_t, int64_t, int64_t, float, const float *, int64_t, const float *,
                 int64_t, float, float *, int64_t, hipStream_t) { return hipErrorUnknown; }
hipError_t sgemm_spot_capture(const SpotArgs &, float *, hipStream_t) { return hipErrorUnknown; }
hipError_t sgemm_spot_check(const SpotArgs &, const float *, unsigned long long *, unsigned long long *, hipStream_t) {
  return hipErrorUnknown;
}
hipError_t sgemm_chain(char, char, char, int64_t, int64_t, int64_t, float, const float *, int64_t, const float *,
                       int64_t, float, float *, int64_t, const GemmChain &, hipStream_t) { return hipErrorUnknown; }
hipError_t sgemm_rank1x2(char, char, char, int64_t, int64_t, int64_t, float, const float *, int64_t, const float *,
                         int64_t, float, float *, int64_t, const float *, const float *, const float *, const float *,
                         hipStream_t) { return hipErrorUnknown; }
hipError_t scsrmm(char, int64_t, int64_t, int64_t, float, const float *, const int64_t *, const int64_t *,
                  const float *, int64_t, float, float *, int64_t, hipStream_t, unsigned *) { return hipErrorUnknown; }
int64_t scsrmm_receipt_entries(char, int64_t) { return 0; }
hipError_t csr_receipt_check(unsigned *, int64_t, unsigned *, hipStream_t) { return hipErrorUnknown; }
hipError_t scsrgemv(char, int64_t, int64_t, const float *, const int64_t *, const int64_t *, const float *,
                    float *, hipStream_t, unsigned *) { return hipErrorUnknown; }
int64_t scsrgemv_receipt_entries(int64_t) { return 0; }
hipError_t transpose_f32(const float *, int64_t, int64_t, int64_t, float *, int64_t, hipStream_t) {
  return hipErrorUnknown;
}
hipError_t csc_merge(int, int64_t, const int64_t *, const int64_t *, const int64_t *, const int64_t *, const float *,
                     const int64_t *, float *, int64_t *, hipStream_t) { return hipErrorUnknown; }
hipError_t scsrcsc(int64_t, int64_t, int64_t, const float *, const int64_t *, const int64_t *, float *,
                   int64_t *, int64_t *, void *, hipStream_t) { return hipErrorUnknown; }
hipError_t sum_partials(float *, const float *const *, int, int64_t, hipStream_t) { return hipErrorUnknown; }
hipError_t verify_sum(const void *, int64_t, int64_t, int64_t, uint64_t, int64_t, unsigned long long *, hipStream_t) { return hipErrorUnknown; }
}  // namespace bof
extern "C" const char *bof_last_error(void) { return ""; }

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); exit(1); } \
  } while (0)

static void plans() {
  const char ords[] = {'R', 'C'}, tr[] = {'N', 'T'};
  const int64_t shapes[][4] = {{640, 500, 600, 256}, {4096, 4096, 4096, 1024}, {1, 1, 1, 128},
                               {300, 200, 500, 128}, {32768, 32768, 32768, 4096}, {4200, 4100, 4099, 4096}};
  for (auto &sh : shapes)
    for (char o : ords)
      for (char a : tr)
        for (char b : tr) {
          int64_t nblk[3];
          const int64_t nt = bof_gemm_plan(o, a, b, sh[0], sh[1], sh[2], 0.5f, 0, 0, 0, sh[3], nullptr, 0, nblk);
          CHECK(nt == nblk[0] * nblk[1] * nblk[2]);
          std::vector<bof_gemm_task> tasks((size_t) nt);
          CHECK(bof_gemm_plan(o, a, b, sh[0], sh[1], sh[2], 0.5f, 0, 0, 0, sh[3], tasks.data(), nt, nblk) == nt);
          int64_t flops = 0;
          for (auto &t : tasks) {
            CHECK(t.M > 0 && t.N > 0 && t.K > 0);
            CHECK(t.parent < nt);
            flops += t.M * t.N * t.K;
          }
          CHECK(flops == sh[0] * sh[1] * sh[2]);  // the tiles cover the iteration space exactly once
        }
  // CSR blocking: ragged offsets, empty rows, a row heavier than the budget
  std::vector<int64_t> ia(5001);
  ia[0] = 7;
  for (int i = 0; i < 5000; i++) ia[i + 1] = ia[i] + (i % 97 == 0 ? 0 : (i == 1234 ? 50000 : i % 13));
  const int64_t nb = bof_csr_blocks(ia.data(), 5000, 128, 1000, 3000, nullptr, nullptr, 0);
  std::vector<int64_t> st((size_t) nb), sz((size_t) nb);
  CHECK(bof_csr_blocks(ia.data(), 5000, 128, 1000, 3000, st.data(), sz.data(), nb) == nb);
  int64_t next = 0;
  for (int64_t b = 0; b < nb; b++) { CHECK(st[b] == next && sz[b] > 0); next += sz[b]; }
  CHECK(next == 5000);
  CHECK(bof_csr_blocks(ia.data(), 0, 128, 1000, 3000, nullptr, nullptr, 0) == 0);
}

static void files(const char *dir) {
  const std::string path = std::string(dir) + "/san.bin";
  const uint64_t rows = 300, ld = 1024, cols = 640;  // floats; ld*4 = 4096 B rows (sector aligned)
  std::vector<float> img(rows * ld);
  for (size_t i = 0; i < img.size(); i++) img[i] = (float) (i % 1000);
  for (int direct = 0; direct < 2; direct++) {
    FILE *f = fopen(path.c_str(), "wb");   // fresh image: the loop body writes into the file
    CHECK(f && fwrite(img.data(), 4, img.size(), f) == img.size());
    fclose(f);
    int fd = open(path.c_str(), O_RDWR | (direct ? O_DIRECT : 0));
    if (fd < 0 && direct) continue;  // filesystem without O_DIRECT
    CHECK(fd >= 0);
    void *buf = nullptr;
    CHECK(posix_memalign(&buf, 4096, rows * cols * 4 + 4096) == 0);
    float *p = (float *) buf;
    // with BOF_IO_ENGINE=uring the aligned O_DIRECT requests below go through io_uring; this buffer
    // is registered as a FIXED buffer (what PinnedRing does with its slots), the per-thread ones
    // further down are not and take the plain READ opcode
    bof::file_buffers_add(buf, rows * cols * 4 + 4096);
    // aligned strided tile (AIO path when direct)
    CHECK(bof::file_sread(fd, 128 * 4, ld * 4, rows, cols * 4, p, true) == 0);
    for (uint64_t r = 0; r < rows; r += 37) CHECK(p[r * cols + 5] == img[r * ld + 128 + 5]);
    // unaligned offset / length / destination (buffered twin when direct)
    CHECK(bof::file_sread(fd, 3 * 4, ld * 4, 17, 101 * 4, p + 1, true) == 0);
    CHECK(p[1 + 16 * 101 + 100] == img[16 * ld + 3 + 100]);
    // write a tile back shifted by one column, read it again
    for (uint64_t i = 0; i < rows * cols; i++) p[i] = -(float) i;
    CHECK(bof::file_swrite(fd, 128 * 4, ld * 4, rows, cols * 4, p, true) == 0);
    std::vector<float> chk(cols);
    CHECK(bof::file_sread(fd, (5 * ld + 128) * 4, 0, 1, cols * 4, chk.data(), true) == 0);
    CHECK(chk[7] == -(float) (5 * cols + 7));
    CHECK(bof::file_swrite(fd, 1 * 4, ld * 4, 9, 33 * 4, p + 3, true) == 0);  // unaligned write
    // reads past the end of the file fail cleanly
    CHECK(bof::file_sread(fd, rows * ld * 4 - 512, 0, 1, 4096, p, true) != 0);
    // 8 threads, each with its own AIO context, reading disjoint row bands
    std::vector<std::thread> th;
    std::vector<int> rc(8, -1);
    for (int t = 0; t < 8; t++)
      th.emplace_back([&, t] {
        void *tb = nullptr;
        if (posix_memalign(&tb, 4096, 32 * cols * 4)) return;
        rc[t] = bof::file_sread(fd, (uint64_t) t * 32 * ld * 4, ld * 4, 32, cols * 4, tb, true);
        free(tb);
      });
    for (auto &x : th) x.join();
    for (int t = 0; t < 8; t++) CHECK(rc[t] == 0);
    // a large contiguous transfer cut into many requests (more than the ring holds at once)
    {
      const uint64_t big = 24u << 20;
      void *bb = nullptr;
      CHECK(posix_memalign(&bb, 4096, big) == 0);
      unsigned char *q = (unsigned char *) bb;
      for (uint64_t i = 0; i < big; i++) q[i] = (unsigned char) (i * 2654435761u >> 24);
      const std::string p2 = std::string(dir) + "/san_big.bin";
      int f2 = open(p2.c_str(), O_RDWR | O_CREAT | O_TRUNC | (direct ? O_DIRECT : 0), 0600);
      CHECK(f2 >= 0);
      CHECK(ftruncate(f2, (off_t) big) == 0);
      CHECK(bof_file_set_request_bytes(64 << 10) == BOF_OK);       // 384 requests
      CHECK(bof::file_swrite(f2, 0, 0, 1, big, bb, true) == 0);
      memset(bb, 0, big);
      CHECK(bof::file_sread(f2, 0, 0, 1, big, bb, true) == 0);
      for (uint64_t i = 0; i < big; i += 4099) CHECK(q[i] == (unsigned char) (i * 2654435761u >> 24));
      CHECK(bof_file_set_request_bytes(4 << 20) == BOF_OK);
      if (!direct) {
        // the pages are in the page cache now: large buffered writes go through the shared mapping
        // (fileio.cpp mapped_write), 8 threads at once on disjoint 3 MiB bands
        const uint64_t before = bof::file_mapped_write_bytes();
        for (uint64_t i = 0; i < big; i++) q[i] = (unsigned char) (255 - (i * 40503u >> 8));
        std::vector<std::thread> wt;
        std::vector<int> wrc(8, -1);
        const uint64_t band = big / 8;
        for (int t = 0; t < 8; t++)
          wt.emplace_back([&, t] { wrc[t] = bof::file_swrite(f2, t * band, 0, 1, band, q + t * band, true); });
        for (auto &x : wt) x.join();
        for (int t = 0; t < 8; t++) CHECK(wrc[t] == 0);
        CHECK(bof::file_mapped_write_bytes() == before + big);
        std::vector<unsigned char> back(big);
        CHECK(pread(f2, back.data(), big, 0) == (ssize_t) big);
        CHECK(memcmp(back.data(), q, big) == 0);
        // small writes and writes past the mapped size keep taking pwrite
        CHECK(bof::file_swrite(f2, 4096, 0, 1, 8192, q, true) == 0);
        CHECK(ftruncate(f2, (off_t) (big + (2u << 20))) == 0);
        CHECK(bof::file_swrite(f2, big, 0, 1, 2u << 20, q, true) == 0);
        CHECK(bof::file_mapped_write_bytes() == before + big);
        CHECK(pread(f2, back.data(), 2u << 20, (off_t) big) == (ssize_t) (2u << 20));
        CHECK(memcmp(back.data(), q, 2u << 20) == 0);
      }
      bof::file_forget(f2);
      close(f2);
      unlink(p2.c_str());
      free(bb);
    }
    bof::file_buffers_remove(buf);
    bof::file_forget(fd);
    close(fd);
    free(buf);
  }
  unlink(path.c_str());
  uint64_t fixed = 0, plain = 0;
  bof::uring_op_counts(&fixed, &plain);
  printf("io_uring requests: %llu fixed-buffer, %llu plain\n", (unsigned long long) fixed, (unsigned long long) plain);
}

// the level-3 schedule (tile list, task order, Belady slot replacement) as a dry run
static void schedules() {
  const char ords[] = {'R', 'C'}, tr[] = {'N', 'T'};
  const int64_t cases[][5] = {{640, 500, 600, 256, 6},    {640, 500, 600, 256, 9},   {640, 500, 600, 256, 64},
                              {2048, 2048, 2048, 256, 40}, {4096, 1024, 512, 512, 7}, {65536, 65536, 65536, 4096, 128}};
  for (auto &c : cases)
    for (char o : ords)
      for (char a : tr)
        for (char b : tr)
          for (float beta : {0.f, 2.f}) {
            bof_flash_stats st;
            memset(&st, 0, sizeof(st));
            const int rc = bof_flash_gemm_simulate(o, a, b, (uint64_t) c[0], (uint64_t) c[1], (uint64_t) c[2], beta,
                                                   0, 0, 0, c[3], c[4], 16, &st);
            CHECK(rc == BOF_OK);
            const uint64_t cbytes = (uint64_t) c[0] * c[1] * 4;
            CHECK(st.bytes_written == cbytes);               // every C tile is written exactly once
            CHECK(st.bytes_read >= (uint64_t) (c[0] * c[2] + c[2] * c[1]) * 4 + (beta != 0.f ? cbytes : 0));
          }
  bof_flash_stats st;
  CHECK(bof_flash_gemm_simulate('R', 'N', 'N', 640, 500, 600, 0.f, 0, 0, 0, 256, 3, 16, &st) != BOF_OK);  // < 6 slots
}

int main(int argc, char **argv) {
  plans();
  schedules();
  files(argc > 1 ? argv[1] : "/tmp");
  printf("host_sanitize ok\n");
  return 0;
}
