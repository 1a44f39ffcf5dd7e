// LDS read-modify-write rates on gfx950: what bounds gemv_t_accumulate_kernel and radix_hist_kernel.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int W = 8192, ITERS = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int spread) {
  __shared__ float ys[W];
  for (int i = threadIdx.x; i < W; i += 256) ys[i] = 0.f;
  __syncthreads();
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < ITERS; it++) {
    s = s * 1664525u + 1013904223u;
    uint32_t a;
    if (spread == 0) a = (threadIdx.x + it * 256) & (W - 1);          // lane-consecutive: conflict-free
    else if (spread == 1) a = (s >> 8) & (W - 1);                     // random in 8192
    else a = (s >> 8) & 255;                                          // random in 256 (histogram-like)
    const float v = (float) (s & 7);
    if (MODE == 0) atomicAdd(&ys[a], v);                               // ds_add_f32
    else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t *>(ys) + a, s & 7u);  // ds_add_u32
    else if (MODE == 2) ys[a] = v;                                     // ds_write_b32
    else if (MODE == 3) ys[a] += v;                                    // read + add + write (racy: timing only)
    else if (MODE == 4) acc += ys[a];                                  // ds_read_b32
    else if (MODE == 5) {                                              // CAS loop (ds_cmpst_rtn_b32)
      uint32_t *p = reinterpret_cast<uint32_t *>(ys) + a;
      uint32_t old = *p, assumed;
      do {
        assumed = old;
        old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + v));
      } while (old != assumed);
    } else if (MODE == 6) atomicAdd(reinterpret_cast<double *>(ys) + (a >> 1), (double) v);  // ds_add_f64
    else if (MODE == 7) acc += atomicAdd(&ys[a], v);                   // ds_add_rtn_f32
  }
  __syncthreads();
  float t = acc;
  for (int i = threadIdx.x; i < W; i += 256) t += ys[i];
  if (t == 123.456f) out[0] = t;
}

template <int MODE>
void run(const char *name, int spread, float *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 5 * 4;
  k<MODE><<<grid, 256>>>(d, spread);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(d, spread);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double) grid * 256 * ITERS;
  printf("%-28s spread=%d: %.3f ms  %.1f G lane-ops/s  (%.2f per CU per ns)\n", name, spread, ms, ops / ms / 1e6,
         ops / ms / 1e6 / 256);
}

int main() {
  float *d;
  hipMalloc(&d, 4);
  for (int sp = 0; sp < 3; sp++) {
    run<0>("ds_add_f32", sp, d);
    run<1>("ds_add_u32", sp, d);
    run<2>("ds_write_b32", sp, d);
    run<3>("read+add+write (racy)", sp, d);
    run<4>("ds_read_b32", sp, d);
    run<5>("CAS loop", sp, d);
    run<6>("ds_add_f64", sp, d);
    run<7>("ds_add_rtn_f32", sp, d);
  }
  return 0;
}
