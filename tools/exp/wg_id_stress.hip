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

__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xF;
}
// like the csrmm kernel: 4 waves per workgroup, one "row" per wave, in place: row value v -> 3 v + 1.
// where[w]: the XCC the workgroup with blockIdx.x == w ran on (+ 16 x its execution count so far)
__global__ void __launch_bounds__(256) work(unsigned *cnt, unsigned *where, float *rows, int n_wg, int spin) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long) blockIdx.x * 4 + wave;
  float *p = rows + row * 64 + lane;
  float v = *p;
  float x = v;
  for (int i = 0; i < spin; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);      // a few microseconds of dependent work
  *p = __builtin_fmaf(3.0f, v, 1.0f) + (x - x);
  if (threadIdx.x == 0) {
    atomicAdd(&cnt[blockIdx.x], 1u);
    atomicAdd(&where[blockIdx.x], 16u + xcc_id());
  }
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

__global__ void fill(float *rows, long n, float v, unsigned *cnt_fill, unsigned *where_fill) {
  const long i = (long) blockIdx.x * 256 + threadIdx.x;
  if (i < n) rows[i] = v;
  if (threadIdx.x == 0) {
    atomicAdd(&cnt_fill[blockIdx.x], 1u);
    atomicAdd(&where_fill[blockIdx.x], 16u + xcc_id());
  }
}
// first anomaly of the process: keep the whole launch's counters (both kernels) for the host to print
__global__ void snapshot(const unsigned *cnt_fill, const unsigned *where_fill, const unsigned *where, const unsigned long long *anom,
                         unsigned *snap, int n_wg, unsigned launch) {
  if (anom[0] == 0 || snap[0] != 0) return;      // (one thread: <<<1, 1>>>)
  snap[0] = 1; snap[1] = launch; snap[2] = (unsigned) n_wg;
  for (int w = 0; w < n_wg; w++) { snap[4 + 3 * w] = cnt_fill[w]; snap[5 + 3 * w] = where_fill[w]; snap[6 + 3 * w] = where[w]; }
}
__global__ void clear3(unsigned *a, unsigned *b, unsigned *c, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { a[i] = 0; b[i] = 0; c[i] = 0; }
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 30;
  const int spin = argc > 2 ? atoi(argv[2]) : 2000;
  CK(hipSetDevice(0));
  const int sizes[] = {79, 32, 256, 1000, 241};       // workgroups per launch (the two incidents: 79 and 32)
  const int max_wg = 1000;
  unsigned *cnt, *where, *cnt_fill, *where_fill, *snap;
  float *rows;
  unsigned long long *anom;
  CK(hipMalloc((void **) &cnt, max_wg * 4));
  CK(hipMalloc((void **) &where, max_wg * 4));
  CK(hipMalloc((void **) &cnt_fill, max_wg * 4));
  CK(hipMalloc((void **) &where_fill, max_wg * 4));
  CK(hipMalloc((void **) &snap, (4 + 3 * max_wg) * 4));
  CK(hipMemset(where, 0, max_wg * 4));
  CK(hipMemset(cnt_fill, 0, max_wg * 4));
  CK(hipMemset(where_fill, 0, max_wg * 4));
  CK(hipMemset(snap, 0, (4 + 3 * max_wg) * 4));
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
      hipLaunchKernelGGL(fill, dim3((n_wg * 256 + 255) / 256), dim3(256), 0, s, rows, (long) n_wg * 256, before, cnt_fill, where_fill);
      hipLaunchKernelGGL(work, dim3(n_wg), dim3(256), 0, s, cnt, where, rows, n_wg, spin);
      hipLaunchKernelGGL(check, dim3((n_wg + 255) / 256), dim3(256), 0, s, cnt, rows, before, n_wg, li, anom);
      hipLaunchKernelGGL(snapshot, dim3(1), dim3(1), 0, s, cnt_fill, where_fill, where, anom, snap, n_wg, li);
      hipLaunchKernelGGL(clear3, dim3((n_wg + 255) / 256), dim3(256), 0, s, cnt_fill, where_fill, where, n_wg);
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
  if (h[0]) {      // the first anomalous launch, workgroup by workgroup: fill's execution count and XCC, work's XCC
    static unsigned sn[4 + 3 * 1000];
    CK(hipMemcpy(sn, snap, sizeof(sn), hipMemcpyDeviceToHost));
    printf("first anomalous launch %u, %u workgroups; per workgroup w: fill ran F times (on XCCs whose ids sum to f), work on XCC x -- only the workgroups that are not (1, w %% 8, w %% 8):\n", sn[1], sn[2]);
    for (unsigned w = 0; w < sn[2]; w++) {
      const unsigned cf = sn[4 + 3 * w], wf = sn[5 + 3 * w], ww = sn[6 + 3 * w];
      if (cf != 1 || (wf >> 4) != 1 || (ww >> 4) != 1 || (wf & 15) != (ww & 15))
        printf("  w=%u (w%%8=%u): fill count %u, fill where-sum 0x%x, work where-sum 0x%x\n", w, w % 8, cf, wf, ww);
    }
  }
  return h[0] ? 1 : 0;
}
