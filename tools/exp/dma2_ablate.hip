// dma2_ablate.hip -- dev tool: where does sgemm_tile256_dma2_kernel lose its last 4-5 %?
// Launches the product kernel's ablated instantiations (ABL bits: 1 no barrier, 2 no vmcnt wait, 4 no DMA pieces,
// 8 no fragment reads -- wrong results by construction, timing only) and the candidate with progress counters in LDS
// instead of s_barrier (SYNC = 1: must be bit-equal to the barrier kernel), on the panel path's two launch shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iblas-on-flash_amd/csrc tools/exp/dma2_ablate.hip -o tools/exp/dma2_ablate
#include "../../blas-on-flash_amd/csrc/gemm_f32_mfma.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace bof;
// the host launcher's HIP object pools live in flash_support.cpp; this tool launches the kernels itself
namespace bof {
hipError_t pooled_event(hipEvent_t *e, bool timing) { return hipEventCreateWithFlags(e, timing ? 0 : hipEventDisableTiming); }
void pooled_event_return(hipEvent_t e) { (void) hipEventDestroy(e); }
hipError_t pooled_stream(hipStream_t *s, bool) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
void pooled_stream_return(hipStream_t s) { if (s) (void) hipStreamDestroy(s); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void fill(float *p, size_t n, uint32_t seed) {
  size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t) gridDim.x * 256;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t) i * 2654435761u + seed;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    p[i] = (float) (h & 0xFFFF) / 32768.0f - 1.0f;
  }
}
__global__ void diff_count(const float *a, const float *b, size_t n, unsigned long long *out) {
  size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t) gridDim.x * 256;
  unsigned long long bad = 0;
  for (; i < n; i += stride) bad += __float_as_uint(a[i]) != __float_as_uint(b[i]);
  if (bad) atomicAdd(out, bad);
}

template <int ABL, int SYNC, int IL = 0>
static double run(const char *name, const float *A, const float *B, float *C, int M, int N, int K, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int tm = M / 256, tn = N / 256;
  double best = 1e30, sum = 0;
  for (int r = 0; r < reps + 1; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((sgemm_tile256_dma2_kernel<NoEpi, ABL, SYNC, IL>), dim3(tm * tn), dim3(256), 0, 0, A, (int64_t) M, B, (int64_t) N, C,
                       (int64_t) N, M, N, K, 1.0f, 0.0f, tm, tn, NoEpi{});
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r == 0) continue;      // warm-up
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double tf = 2.0 * M * N * K / 1e9;
  printf("{\"variant\":\"%s\",\"M\":%d,\"N\":%d,\"K\":%d,\"ms_best\":%.4f,\"ms_mean\":%.4f,\"tflops_best\":%.2f,\"frac_best\":%.4f,\"frac_mean\":%.4f}\n",
         name, M, N, K, best, sum / reps, tf / best, tf / best / 157.3, tf / (sum / reps) / 157.3);
  fflush(stdout);
  return best;
}

int main(int argc, char **argv) {
  const int M = 4096, N = 32768;
  const int Kbig = argc > 1 ? atoi(argv[1]) : 32768;
  float *A, *B, *C, *C2;
  CK(hipMalloc(&A, (size_t) Kbig * M * 4));      // 'T': stored [K][M]
  CK(hipMalloc(&B, (size_t) Kbig * N * 4));      // 'N': stored [K][N]
  CK(hipMalloc(&C, (size_t) M * N * 4));
  CK(hipMalloc(&C2, (size_t) M * N * 4));
  fill<<<4096, 256>>>(A, (size_t) Kbig * M, 1u);
  fill<<<4096, 256>>>(B, (size_t) Kbig * N, 2u);
  CK(hipDeviceSynchronize());
  unsigned long long *bad;
  CK(hipMalloc(&bad, 8));
  for (int K : {4096, Kbig}) {
    const int reps = K <= 4096 ? 20 : 5;
    run<0, 0>("base (barrier)", A, B, C, M, N, K, reps);
    run<0, 1>("LDS progress counters instead of the barrier", A, B, C2, M, N, K, reps);
    CK(hipMemset(bad, 0, 8));
    diff_count<<<2048, 256>>>(C, C2, (size_t) M * N, bad);
    unsigned long long hb = 0;
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("{\"check\":\"counters vs barrier, K=%d\",\"words_that_differ\":%llu}\n", K, hb);
    for (int pass = 0; pass < 2; pass++) {      // the candidates with their side work spread over the MFMA gaps (IL = 1)
      float *Cx = pass ? C2 : C2;
      if (pass == 0) run<0, 0, 1>("barrier, side work interleaved", A, B, Cx, M, N, K, reps);
      else run<0, 1, 1>("LDS counters, side work interleaved", A, B, Cx, M, N, K, reps);
      CK(hipMemset(bad, 0, 8));
      diff_count<<<2048, 256>>>(C, C2, (size_t) M * N, bad);
      CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
      printf("{\"check\":\"%s vs barrier, K=%d\",\"words_that_differ\":%llu}\n", pass ? "counters + interleave" : "barrier + interleave", K, hb);
    }
    run<4, 0, 1>("ablate: interleaved, no DMA pieces", A, B, C2, M, N, K, reps);
    run<8, 0, 1>("ablate: interleaved, no fragment reads", A, B, C2, M, N, K, reps);
    run<1, 0>("ablate: no barrier", A, B, C2, M, N, K, reps);
    run<2, 0>("ablate: no vmcnt wait", A, B, C2, M, N, K, reps);
    run<3, 0>("ablate: no barrier, no vmcnt wait", A, B, C2, M, N, K, reps);
    run<4, 0>("ablate: no DMA pieces", A, B, C2, M, N, K, reps);
    run<7, 0>("ablate: no DMA, no barrier, no vmcnt", A, B, C2, M, N, K, reps);
    run<8, 0>("ablate: no fragment reads", A, B, C2, M, N, K, reps);
    run<15, 0>("ablate: MFMAs only", A, B, C2, M, N, K, reps);
    run<0, 0>("base again", A, B, C, M, N, K, reps);
  }
  return 0;
}
