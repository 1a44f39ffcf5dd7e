// d2h_engine.hip -- does a pinned D2H hipMemcpyAsync run on an SDMA engine or as a blit kernel on the CUs?
// A kernel that holds every CU for ~20 ms is started on one stream; 32 MiB D2H (and, for comparison, H2D)
// copies are issued on another (high-priority) stream right behind it.  A copy that completes in ~0.6 ms while
// the kernel is still running went through SDMA; one that completes only after the kernel needed CUs.
// Run under different environments (HSA_ENABLE_SDMA, GPU_FORCE_BLIT_COPY_SIZE, HSA_REV_COPY_DIR, ...):
//   hipcc --offload-arch=gfx950 -O2 tools/exp/d2h_engine.hip -o tools/exp/d2h_engine
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) hog(float *out, long iters) {
  // 1024 threads x 64 KB of LDS: one workgroup per CU, nothing else fits beside it
  __shared__ float pad[16384];
  float v = threadIdx.x;
  for (long i = 0; i < iters; i++) v = v * 1.0000001f + 0.5f;
  pad[threadIdx.x] = v;
  __syncthreads();
  if (v == 12345.f) out[blockIdx.x] = pad[0];
}

int main(int argc, char **argv) {
  const size_t sz = 32u << 20;
  const int n = argc > 1 ? atoi(argv[1]) : 16;
  const bool portable = argc > 2 && atoi(argv[2]);      // hipHostMallocPortable, as the library's staging rings
  const bool wait_ev = argc > 3 && atoi(argv[3]);       // the copy stream first waits for an event of the kernel stream
  char *d, *h;
  float *o;
  CK(hipMalloc(&d, sz * 2));
  CK(hipMalloc(&o, 4096));
  CK(hipHostMalloc(&h, sz * 2, portable ? hipHostMallocPortable : hipHostMallocDefault));
  hipStream_t sk, sc;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi));
  hipEvent_t k0, k1, c0, c1;
  CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
  // calibrate the hog to ~20 ms
  long iters = 200000;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(k0, sk));
    hipLaunchKernelGGL(hog, dim3(512), dim3(1024), 0, sk, o, iters);
    CK(hipEventRecord(k1, sk));
    CK(hipStreamSynchronize(sk));
    float ms; CK(hipEventElapsedTime(&ms, k0, k1));
    if (rep < 2) iters = (long) (iters * 20.0 / ms);
    else printf("hog kernel: %.2f ms (512 workgroups x 1024 threads = every wave slot of the chip)\n", ms);
  }
  for (int dir = 0; dir < 2; dir++) {
    // alone
    CK(hipEventRecord(c0, sc));
    for (int i = 0; i < n; i++)
      CK(dir == 0 ? hipMemcpyAsync(h + (i & 1) * sz, d + (i & 1) * sz, sz, hipMemcpyDeviceToHost, sc)
                  : hipMemcpyAsync(d + (i & 1) * sz, h + (i & 1) * sz, sz, hipMemcpyHostToDevice, sc));
    CK(hipEventRecord(c1, sc));
    CK(hipStreamSynchronize(sc));
    float alone; CK(hipEventElapsedTime(&alone, c0, c1));
    // behind a chip-filling kernel: two hogs back to back (~40 ms), copies issued right after the launch
    CK(hipEventRecord(k0, sk));
    hipLaunchKernelGGL(hog, dim3(512), dim3(1024), 0, sk, o, iters);
    hipLaunchKernelGGL(hog, dim3(512), dim3(1024), 0, sk, o, iters);
    CK(hipEventRecord(k1, sk));
    if (wait_ev) {   // (an event recorded BEFORE the hogs: already complete, but the queue now carries a barrier)
      CK(hipStreamWaitEvent(sc, k0, 0));
    }
    auto t0 = std::chrono::steady_clock::now();
    CK(dir == 0 ? hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, sc) : hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, sc));
    CK(hipStreamSynchronize(sc));
    const double first = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(sk));
    float kms; CK(hipEventElapsedTime(&kms, k0, k1));
    if (dir == 0) printf("[portable=%d wait_event=%d] ", (int) portable, (int) wait_ev);
    printf("%s: %d x 32 MiB alone %.2f ms (%.1f GB/s); one copy issued behind two hog kernels returned after %.2f ms "
           "(kernels: %.1f ms) -> %s\n", dir == 0 ? "D2H" : "H2D", n, alone, n * sz / alone / 1e6, first, kms,
           first < 0.5 * kms ? "DMA engine (did not wait for CUs)" : "needed the CUs (blit kernel)");
  }
  return 0;
}
