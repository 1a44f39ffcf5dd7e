// Dev tool: sustained fp32 MFMA rate and clock of this device (bare loop, operands in
// registers), to separate "clock under load" from "pipe idle" when reading the GEMM
// kernel's roofline fraction.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) peak(float *out, int iters, unsigned long long *clk) {
  f32x16 acc[4];
  for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int a = 0; a < 4; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main(int argc, char **argv) {
  int waves_per_simd = argc > 1 ? atoi(argv[1]) : 1;
  int blocks = 256 * waves_per_simd, iters = 20000;
  float *out; unsigned long long *clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    peak<<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double) blocks * 4 * iters * 32 * 4096.0;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("blocks %d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz\n", blocks, ms, flops / ms / 1e9,
           (double) h[0] / (double) h[1] * 0.1);
  }
  return 0;
}
