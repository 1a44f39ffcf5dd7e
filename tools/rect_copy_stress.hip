// rect_copy_stress.hip -- diagnostic: does hipMemcpy2DAsync keep stream order the way the tile cache relies on it?
//
// The tile cache's row-group I/O (flash_runtime.cpp) is the library's only user of 2-D copies: a wide chunk of a
// stored block row lands in a pinned slot and is scattered into the packed tile slots with one
// hipMemcpy2DAsync(H2D) per tile on the copy stream; an event recorded behind them is what the compute stream
// waits for.  The write-back gathers finished tiles with hipMemcpy2DAsync(D2H) into a pinned slot behind an
// event wait on the compute stream.  This program repeats exactly that pattern with self-checking data:
//   pinned slot (generation g) --2-D H2D x3--> tile slots --event--> kernel: tile += 1 --event--> 2-D D2H x3 --> pinned
// and verifies every element on the host.  Any element that still carries an older generation means a copy or a
// kernel ran out of order.  Build: hipcc --offload-arch=gfx950 -O2 tools/rect_copy_stress.hip -o rect_copy_stress
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); }  \
  } while (0)

__global__ void bump(float *t, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) t[i] += 1.0f;
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 60;
  const int n_tiles = 3, ring = 2, n_groups = 6;
  const int rows = 256;
  int widths[n_tiles] = {256, 256, 194};          // the shapes of the one mismatch seen (k = 706 in three blocks)
  int wide = 0;
  for (int w : widths) wide += w;
  hipStream_t h2d, d2h, comp[3];
  int least, greatest;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&h2d, hipStreamNonBlocking, greatest));
  CK(hipStreamCreateWithPriority(&d2h, hipStreamNonBlocking, greatest));
  for (auto &s : comp) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *pin_in[ring], *pin_out[ring];
  hipEvent_t in_free[ring], out_done[ring];
  for (int i = 0; i < ring; i++) {
    CK(hipHostMalloc((void **) &pin_in[i], (size_t) rows * wide * 4, hipHostMallocPortable));
    CK(hipHostMalloc((void **) &pin_out[i], (size_t) rows * wide * 4, hipHostMallocPortable));
    CK(hipEventCreateWithFlags(&in_free[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&out_done[i], hipEventDisableTiming));
  }
  // tile slots: n_groups groups of n_tiles packed tiles, reused round robin (like a slab under a budget)
  float *slot[n_groups][n_tiles];
  hipEvent_t ready[n_groups][n_tiles], used[n_groups][n_tiles], flushed[n_groups];
  for (int g = 0; g < n_groups; g++) {
    for (int t = 0; t < n_tiles; t++) {
      CK(hipMalloc((void **) &slot[g][t], (size_t) rows * widths[t] * 4));
      CK(hipEventCreateWithFlags(&ready[g][t], hipEventDisableTiming));
      CK(hipEventCreateWithFlags(&used[g][t], hipEventDisableTiming));
    }
    CK(hipEventCreateWithFlags(&flushed[g], hipEventDisableTiming));
  }
  std::vector<char> in_busy(ring, 0), out_busy(ring, 0), slot_used(n_groups, 0);
  std::vector<long> out_gen(ring, -1);
  long gen = 0, bad = 0, checked = 0;
  const auto check_out = [&](int o) {
    if (out_gen[o] < 0) return;
    CK(hipEventSynchronize(out_done[o]));
    const float want = (float) (out_gen[o] % 100000) + 1.0f;
    for (long i = 0; i < (long) rows * wide; i++)
      if (pin_out[o][i] != want) {
        if (bad < 10) fprintf(stderr, "generation %ld: element %ld = %g, want %g\n", out_gen[o], i, pin_out[o][i], want);
        bad++;
      }
    checked++;
    out_gen[o] = -1;
  };
  const double t_end = (double) clock() / CLOCKS_PER_SEC + seconds;
  struct timespec ts0;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  for (;; gen++) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    if ((ts.tv_sec - ts0.tv_sec) + 1e-9 * (ts.tv_nsec - ts0.tv_nsec) > seconds) break;
    (void) t_end;
    const int i = (int) (gen % ring), g = (int) (gen % n_groups), o = (int) (gen % ring);
    // the pinned input slot is free once the copies that read it last have run
    if (in_busy[i]) CK(hipEventSynchronize(in_free[i]));
    const float v = (float) (gen % 100000);
    for (long q = 0; q < (long) rows * wide; q++) pin_in[i][q] = v;
    // WAR on the tile slots: their previous occupants' write-back
    if (slot_used[g]) CK(hipStreamWaitEvent(h2d, flushed[g], 0));
    int col = 0;
    for (int t = 0; t < n_tiles; t++) {
      CK(hipMemcpy2DAsync(slot[g][t], (size_t) widths[t] * 4, pin_in[i] + col, (size_t) wide * 4, (size_t) widths[t] * 4, rows,
                          hipMemcpyHostToDevice, h2d));
      col += widths[t];
    }
    CK(hipEventRecord(in_free[i], h2d));
    in_busy[i] = 1;
    for (int t = 0; t < n_tiles; t++) CK(hipEventRecord(ready[g][t], h2d));
    // one "task" per tile on its chain's stream
    for (int t = 0; t < n_tiles; t++) {
      CK(hipStreamWaitEvent(comp[t], ready[g][t], 0));
      const int n = rows * widths[t];
      bump<<<(n + 255) / 256, 256, 0, comp[t]>>>(slot[g][t], n);
      CK(hipEventRecord(used[g][t], comp[t]));
    }
    // write-back of the row group
    check_out(o);
    for (int t = 0; t < n_tiles; t++) CK(hipStreamWaitEvent(d2h, used[g][t], 0));
    col = 0;
    for (int t = 0; t < n_tiles; t++) {
      CK(hipMemcpy2DAsync(pin_out[o] + col, (size_t) wide * 4, slot[g][t], (size_t) widths[t] * 4, (size_t) widths[t] * 4, rows,
                          hipMemcpyDeviceToHost, d2h));
      col += widths[t];
    }
    CK(hipEventRecord(out_done[o], d2h));
    CK(hipEventRecord(flushed[g], d2h));
    slot_used[g] = 1;
    out_gen[o] = gen;
  }
  for (int o = 0; o < ring; o++) check_out(o);
  CK(hipDeviceSynchronize());
  printf("rect_copy_stress: %ld generations, %ld checked, %ld bad elements\n", gen, checked, bad);
  return bad ? 1 : 0;
}
