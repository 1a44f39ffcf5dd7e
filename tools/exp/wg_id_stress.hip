// wg_id_stress.hip -- does every workgroup of a launch run exactly once, under the load the fuzz puts on the GPU?
//
// Round 6, profiles/r6/incident_csrmm/: three wrong C files in ~234 000 drawn csrmm cases (eight processes on one GPU),
// all of one shape -- in ONE launch of the (unchanged, 5-round-old) one-wave-per-row kernel the rows of some
// workgroups were updated TWICE (in place: C = alpha A B + beta C, so twice shows) and the rows of their NEIGHBOURS
// (blockIdx + 1) not at all, periodically with the number of XCDs: workgroups 8j of the launch did the work of
// workgroups 8j + 7 (case 1), workgroups 8j + 5 / 8j + 7 that of 8j + 4 / 8j + 6 (case 2).  As if a workgroup had been
// handed its neighbour's ID.  This program asks the hardware directly: every launch, every workgroup adds 1 to ITS
// counter (indexed by blockIdx.x) after a few microseconds of dependent arithmetic; a checker kernel then counts the
// counters that are not 1 and keeps the first few (launch, index, value).  Run several copies at once:
//   for p in 1..8: tools/exp/wg_id_stress SECONDS &
// Build: hipcc --offload-arch=gfx950 -O2 tools/exp/wg_id_stress.hip -o tools/exp/wg_id_stress
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#define CK(x)                                                                                    \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); }   \
  } while (0)

// like the csrmm kernel: 4 waves per workgroup, one "row" per wave, in place: row value v -> 3 v + 1
__global__ void __launch_bounds__(256) work(unsigned *cnt, float *rows, int n_wg, int spin) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long) blockIdx.x * 4 + wave;
  float *p = rows + row * 64 + lane;
  float v = *p;
  float x = v;
  for (int i = 0; i < spin; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);      // a few microseconds of dependent work
  *p = __builtin_fmaf(3.0f, v, 1.0f) + (x - x);
  if (threadIdx.x == 0) atomicAdd(&cnt[blockIdx.x], 1u);
  (void) n_wg;
}

// counters != 1 or rows != 3 * before + 1 -> anomalies[0] += 1, first 8 kept as (launch, index, counter value, row value bits)
__global__ void check(unsigned *cnt, float *rows, float before, int n_wg, unsigned launch, unsigned long long *anom) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_wg) return;
  const unsigned c = cnt[w];
  const float want = 3.0f * before + 1.0f;
  bool bad = c != 1u;
  float seen = want;
  for (int r = 0; r < 4 && !bad; r++)
    for (int l = 0; l < 64; l += 21) {
      seen = rows[((long) w * 4 + r) * 64 + l];
      if (seen != want) { bad = true; break; }
    }
  if (bad) {
    const unsigned long long k = atomicAdd(&anom[0], 1ull);
    if (k < 8) {
      anom[1 + 4 * k] = launch;
      anom[2 + 4 * k] = (unsigned long long) w;
      anom[3 + 4 * k] = c;
      anom[4 + 4 * k] = (unsigned long long) __float_as_uint(seen);
    }
  }
  cnt[w] = 0;
}

__global__ void fill(float *rows, long n, float v) {
  const long i = (long) blockIdx.x * 256 + threadIdx.x;
  if (i < n) rows[i] = v;
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 30;
  const int spin = argc > 2 ? atoi(argv[2]) : 2000;
  CK(hipSetDevice(0));
  const int sizes[] = {79, 32, 256, 1000, 241};       // workgroups per launch (the two incidents: 79 and 32)
  const int max_wg = 1000;
  unsigned *cnt;
  float *rows;
  unsigned long long *anom;
  CK(hipMalloc((void **) &cnt, max_wg * 4));
  CK(hipMalloc((void **) &rows, (size_t) max_wg * 4 * 64 * 4));
  CK(hipMalloc((void **) &anom, 8 * 40));
  CK(hipMemset(cnt, 0, max_wg * 4));
  CK(hipMemset(anom, 0, 8 * 40));
  hipStream_t st[2];
  for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long launches = 0;
  unsigned li = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int rep = 0; rep < 200; rep++, li++) {
      const int n_wg = sizes[li % 5];
      hipStream_t s = st[li & 1];
      const float before = (float) (li % 7);
      hipLaunchKernelGGL(fill, dim3((n_wg * 256 + 255) / 256), dim3(256), 0, s, rows, (long) n_wg * 256, before);
      hipLaunchKernelGGL(work, dim3(n_wg), dim3(256), 0, s, cnt, rows, n_wg, spin);
      hipLaunchKernelGGL(check, dim3((n_wg + 255) / 256), dim3(256), 0, s, cnt, rows, before, n_wg, li, anom);
      CK(hipStreamSynchronize(s));      // (one buffer set: launches of the two streams alternate, never overlap)
      launches++;
    }
  }
  CK(hipDeviceSynchronize());
  unsigned long long h[40];
  CK(hipMemcpy(h, anom, sizeof(h), hipMemcpyDeviceToHost));
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("{\"seconds\":%.0f,\"launches\":%llu,\"spin\":%d,\"workgroups_with_a_wrong_count_or_row\":%llu,\"first\":[", dt, launches, spin, h[0]);
  for (unsigned long long k = 0; k < (h[0] < 8 ? h[0] : 8); k++)
    printf("%s{\"launch\":%llu,\"workgroup\":%llu,\"counter\":%llu,\"row_bits\":%llu}", k ? "," : "", h[1 + 4 * k], h[2 + 4 * k], h[3 + 4 * k], h[4 + 4 * k]);
  printf("]}\n");
  return h[0] ? 1 : 0;
}
