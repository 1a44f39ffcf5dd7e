// vmm_probe.hip -- can a resident operand be ONE virtual range whose physical backing arrives panel by panel?
// hipMemAddressReserve + per-panel hipMemCreate / hipMemMap / hipMemSetAccess, timed against one big hipMalloc and
// against per-panel hipMallocs; a kernel then reads across the panel boundaries.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(float *p, size_t n, float v) {
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v + (float) (i & 1023);
}
__global__ void sum(const float *p, size_t n, double *out) {
  double s = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) s += p[i];
  atomicAdd(out, s);
}
int main(int argc, char **argv) {
  const size_t panel = (size_t) (argc > 1 ? atoll(argv[1]) : 1024) << 20;   // MiB per panel
  const int n_panels = argc > 2 ? atoi(argv[2]) : 16;
  CK(hipSetDevice(0));
  CK(hipFree(0));
  int vmm = 0;
  CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0));
  printf("VMM supported: %d\n", vmm);
  double t0 = now();
  void *big = nullptr;
  CK(hipMalloc(&big, panel * n_panels));
  printf("one hipMalloc of %zu MiB: %.1f ms\n", (panel * n_panels) >> 20, (now() - t0) * 1e3);
  t0 = now();
  CK(hipFree(big));
  printf("hipFree: %.1f ms\n", (now() - t0) * 1e3);
  std::vector<void *> parts;
  t0 = now();
  for (int i = 0; i < n_panels; i++) { void *p; CK(hipMalloc(&p, panel)); parts.push_back(p); }
  printf("%d hipMallocs of %zu MiB: %.1f ms (%.1f ms each)\n", n_panels, panel >> 20, (now() - t0) * 1e3, (now() - t0) * 1e3 / n_panels);
  for (void *p : parts) CK(hipFree(p));
  if (!vmm) return 0;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity: %zu KiB\n", gran >> 10);
  void *base = nullptr;
  t0 = now();
  CK(hipMemAddressReserve(&base, panel * n_panels, 0, nullptr, 0));
  printf("reserve: %.2f ms\n", (now() - t0) * 1e3);
  std::vector<hipMemGenericAllocationHandle_t> hs((size_t) n_panels);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  double *d_out;
  CK(hipMalloc((void **) &d_out, 8));
  for (int i = 0; i < n_panels; i++) {
    t0 = now();
    CK(hipMemCreate(&hs[(size_t) i], panel, &prop, 0));
    double t1 = now();
    CK(hipMemMap((char *) base + (size_t) i * panel, panel, 0, hs[(size_t) i], 0));
    double t2 = now();
    CK(hipMemSetAccess((char *) base + (size_t) i * panel, panel, &acc, 1));
    double t3 = now();
    if (i < 4 || i == n_panels - 1) printf("panel %d: create %.2f map %.2f access %.2f ms\n", i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    // a kernel on the part mapped so far while later panels are still unmapped
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (float *) ((char *) base + (size_t) i * panel), panel / 4, (float) i);
  }
  CK(hipDeviceSynchronize());
  CK(hipMemset(d_out, 0, 8));
  t0 = now();
  hipLaunchKernelGGL(sum, dim3(2048), dim3(256), 0, 0, (const float *) base, panel * n_panels / 4, d_out);
  CK(hipDeviceSynchronize());
  double dt = now() - t0, h = 0;
  CK(hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost));
  double want = 0;
  for (int i = 0; i < n_panels; i++) want += (double) (panel / 4) * i + (double) (panel / 4 / 1024) * (1023.0 * 1024 / 2);
  printf("sum across %d panels: %.6g (want %.6g) in %.2f ms = %.0f GB/s\n", n_panels, h, want, dt * 1e3, panel * n_panels / dt / 1e9);
  // H2D into the mapped range from pinned memory, and a peer-style copy
  void *hbuf;
  CK(hipHostMalloc(&hbuf, 64 << 20, 0));
  t0 = now();
  for (int r = 0; r < 8; r++) CK(hipMemcpyAsync((char *) base + panel - (32 << 20) + (size_t) r * (64 << 20), hbuf, 64 << 20, hipMemcpyHostToDevice, 0));
  CK(hipDeviceSynchronize());
  printf("H2D into the range (crossing a panel boundary): %.1f GB/s\n", 8.0 * (64 << 20) / (now() - t0) / 1e9);
  t0 = now();
  for (int i = 0; i < n_panels; i++) {
    CK(hipMemUnmap((char *) base + (size_t) i * panel, panel));
    CK(hipMemRelease(hs[(size_t) i]));
  }
  CK(hipMemAddressFree(base, panel * n_panels));
  printf("unmap + release + free: %.1f ms\n", (now() - t0) * 1e3);
  return 0;
}
