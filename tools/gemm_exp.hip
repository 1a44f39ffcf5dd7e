// Dev tool: ablation of the 256x256 double-buffered GEMM kernel (timing only; outputs of
// the ablated variants are wrong by construction).  EXP bits:
//   1 skip LDS staging writes in the loop     2 skip global loads in the loop
//   4 skip the per-slab barrier               8 operands from registers (no LDS reads)
//  16 skip MFMAs
#include "../blas-on-flash_amd/csrc/gemm_f32_mfma.hip"
#include <cstdio>
#include <vector>
using namespace bof;

// one K-slab of a KMAJOR operand (32 k-rows x 256 floats) straight into LDS: each k-row is 1 KiB
// = one wave-wide global_load_lds_dwordx4; wave w moves rows 4w..4w+3
__device__ __forceinline__ void dma_kmajor(const float *g, int64_t ld, int x0, int k0, float *sdst, int wave, int lane) {
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int krow = wave * 4 + u;
    const float *src = g + (int64_t) (k0 + krow) * ld + x0 + lane * 4;
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *) (sdst + krow * 256), 16, 0, 0);
  }
}

__device__ unsigned long long g_stamps[8 * 8];
#define STAMP(idx)                                                                         \
  if (EXP & 32768) {                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    unsigned long long _t;                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    tsum[idx] += _t - tprev;                                                               \
    tprev = _t;                                                                            \
  }
template <int EXP, int AMODE, int BMODE>
__global__ void __launch_bounds__(512, 2)
expk(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
     float *__restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NTHR = 512;
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int LDS_A = (AMODE == XMAJOR) ? BM * XLD : BK * BM;
  constexpr int LDS_B = (BMODE == XMAJOR) ? BN * XLD : BK * BN;
  constexpr int LDS_BUF = LDS_A + LDS_B;
  __shared__ __attribute__((aligned(16))) float lds[2 * LDS_BUF];
  int bid = blockIdx.x;
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  f32x16 acc[MT][NT];
  for (int a = 0; a < MT; a++) for (int b = 0; b < NT; b++) for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  const int nkt = K / BK;
  auto ra = g2r<AMODE, BM, NTHR, false>(A, lda, m0, 0, M, K, t);
  auto rb = g2r<BMODE, BN, NTHR, false>(B, ldb, n0, 0, N, K, t);
  r2s<AMODE, BM, NTHR>(lds, ra, t);
  r2s<BMODE, BN, NTHR>(lds + LDS_A, rb, t);
  r2s<AMODE, BM, NTHR>(lds + LDS_BUF, ra, t);
  r2s<BMODE, BN, NTHR>(lds + LDS_BUF + LDS_A, rb, t);
  ra = g2r<AMODE, BM, NTHR, false>(A, lda, m0, BK, M, K, t);
  rb = g2r<BMODE, BN, NTHR, false>(B, ldb, n0, BK, N, K, t);
  __syncthreads();
  if (EXP & 1024) { if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1); }
  if (EXP & 2048) { if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_sleep(32); }
  if (EXP & 32768) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
  for (int kt = 0; kt < nkt; kt++) {
    const float *sA = lds + ((kt & 1) ? LDS_BUF : 0), *sB = sA + LDS_A;
    STAMP(0)
    float *nA2 = lds + ((kt & 1) ? 0 : LDS_BUF), *nB2 = nA2 + LDS_A;
    if ((EXP & 4096) && BMODE == KMAJOR) {
      if (kt + 1 < nkt) {
        float *nA = lds + ((kt & 1) ? 0 : LDS_BUF);
        r2s<AMODE, BM, NTHR>(nA, ra, t);
        dma_kmajor(B, ldb, n0, (kt + 1) * BK, nA + LDS_A, wave, lane);
      }
      if (kt + 2 < nkt) ra = g2r<AMODE, BM, NTHR, false>(A, lda, m0, (kt + 2) * BK, M, K, t);
    } else {
    if (kt + 1 < nkt && !(EXP & 1) && !(EXP & 32) && !(EXP & 256)) {
      float *nA = lds + ((kt & 1) ? 0 : LDS_BUF);
      r2s<AMODE, BM, NTHR>(nA, ra, t);
      r2s<BMODE, BN, NTHR>(nA + LDS_A, rb, t);
    }
    if (kt + 2 < nkt && !(EXP & 2) && !(EXP & 32) && !(EXP & 256)) {
      ra = g2r<AMODE, BM, NTHR, false>(A, lda, m0, (kt + 2) * BK, M, K, t);
      rb = g2r<BMODE, BN, NTHR, false>(B, ldb, n0, (kt + 2) * BK, N, K, t);
    }
    }
    STAMP(1)
    if (EXP & 8192) {
      // operand fragments double-buffered in registers: group q+1 is read from LDS before the
      // MFMAs of group q are issued
      f32x4 a[2][MT], b[2][NT];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) a[0][mt] = s2op<AMODE, BM>(sA, wm * WTM + mt * 32 + i, 0, h);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) b[0][nt] = s2op<BMODE, BN>(sB, wn * WTN + nt * 32 + i, 0, h);
#pragma unroll
      for (int q = 0; q < BK / 8; q++) {
        if (q + 1 < BK / 8) {
#pragma unroll
          for (int mt = 0; mt < MT; mt++) a[(q + 1) & 1][mt] = s2op<AMODE, BM>(sA, wm * WTM + mt * 32 + i, q + 1, h);
#pragma unroll
          for (int nt = 0; nt < NT; nt++) b[(q + 1) & 1][nt] = s2op<BMODE, BN>(sB, wn * WTN + nt * 32 + i, q + 1, h);
        }
        if (EXP & 16384) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mt][c], b[q & 1][nt][c], acc[mt][nt], 0, 0, 0);
        if (EXP & 16384) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int q = 0; q < BK / 8; q++) {
      f32x4 a[MT], b[NT];
      if (EXP & 8) {
        for (int mt = 0; mt < MT; mt++) a[mt] = ra.v[mt & 3] + (float) q;
        for (int nt = 0; nt < NT; nt++) b[nt] = rb.v[nt & 3] + (float) q;
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[mt] = s2op<AMODE, BM>(sA, wm * WTM + mt * 32 + i, q, h);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b[nt] = s2op<BMODE, BN>(sB, wn * WTN + nt * 32 + i, q, h);
      }
      if (EXP & 256) {
        constexpr int LATEQ = (EXP & 512) ? 2 : 3;
        if (q == LATEQ) {
          __builtin_amdgcn_sched_barrier(0);
          if (kt + 1 < nkt) {
            r2s<AMODE, BM, NTHR>(nA2, ra, t);
            r2s<BMODE, BN, NTHR>(nB2, rb, t);
          }
          if (kt + 2 < nkt) {
            ra = g2r<AMODE, BM, NTHR, false>(A, lda, m0, (kt + 2) * BK, M, K, t);
            rb = g2r<BMODE, BN, NTHR, false>(B, ldb, n0, (kt + 2) * BK, N, K, t);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!(EXP & 16)) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
          for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][c], b[nt][c], acc[mt][nt], 0, 0, 0);
          if (EXP & 32) {
            if (q == ((EXP & 64) ? 1 : 0) && kt + 1 < nkt) {
              r2s1<AMODE, BM>(nA2, ra.v[c], t + c * NTHR);
              r2s1<BMODE, BN>(nB2, rb.v[c], t + c * NTHR);
            }
            if (q == ((EXP & 64) ? 2 : 1) && kt + 2 < nkt) {
              ra.v[c] = g2r1<AMODE, BM, false>(A, lda, m0, (kt + 2) * BK, M, K, t + c * NTHR);
              rb.v[c] = g2r1<BMODE, BN, false>(B, ldb, n0, (kt + 2) * BK, N, K, t + c * NTHR);
            }
            if (EXP & 128) __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        for (int mt = 0; mt < MT; mt++) for (int nt = 0; nt < NT; nt++) acc[mt][nt][0] += a[mt][0] * b[nt][1];
      }
    }
    }
    STAMP(2)
    if (!(EXP & 4)) __syncthreads();
    STAMP(3)
  }
  if ((EXP & 32768) && blockIdx.x == 7 && lane == 0) {
    for (int x = 0; x < 4; x++) g_stamps[wave * 8 + x] = tsum[x];
  }
  float *ctile = C + (int64_t) m0 * ldc + n0;
  const int lane_off = (wm * WTM + 4 * h) * (int) ldc + wn * WTN + i;
  for (int mt = 0; mt < MT; mt++) for (int nt = 0; nt < NT; nt++) for (int r = 0; r < 16; r++)
    (ctile + ((int64_t) (mt * 32 + (r & 3) + 8 * (r >> 2)) * ldc + nt * 32))[lane_off] = acc[mt][nt][r];
}

template <int EXP, int AM, int BMo>
float run(const float *A, const float *B, float *C, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    for (int it = 0; it < 5; it++)
      hipLaunchKernelGGL((expk<EXP, AM, BMo>), dim3(256), dim3(512), 0, 0, A, n, B, n, C, n, n, n, n, 16, 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms / 5 < best ? ms / 5 : best;
  }
  return best;
}
template <int EXP> void dump_stamps(const float *A, const float *B, float *C, int n) {
  hipLaunchKernelGGL((expk<EXP, 0, 1>), dim3(256), dim3(512), 0, 0, A, n, B, n, C, n, n, n, n, 16, 16);
  hipDeviceSynchronize();
  unsigned long long h[64];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
  for (int w = 0; w < 8; w++)
    printf("EXP %d wave %d per-slab cycles: barrier->top %.0f  staging %.0f  compute %.0f  barrier-wait %.0f\n", EXP, w,
           h[w * 8 + 0] / 128.0, h[w * 8 + 1] / 128.0, h[w * 8 + 2] / 128.0, h[w * 8 + 3] / 128.0);
}
template <int AM, int BMo>
float run_prod(const float *A, const float *B, float *C, int n, int waves) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    for (int it = 0; it < 5; it++) {
      if (waves == 4)
        hipLaunchKernelGGL((sgemm_tile_kernel<256, 256, 2, 2, true, AM, BMo, false>), dim3(256), dim3(256), 0, 0, A, (int64_t) n, B, (int64_t) n, C, (int64_t) n, n, n, n, 1.0f, 0.0f, 16, 16);
      else
        hipLaunchKernelGGL((sgemm_tile_kernel<256, 256, 2, 4, true, AM, BMo, false>), dim3(256), dim3(512), 0, 0, A, (int64_t) n, B, (int64_t) n, C, (int64_t) n, n, n, n, 1.0f, 0.0f, 16, 16);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms / 5 < best ? ms / 5 : best;
  }
  return best;
}
template <int ABL>
float run_1w(const float *A, const float *B, float *C, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    for (int it = 0; it < 5; it++)
      hipLaunchKernelGGL((sgemm_tile256_1w_kernel<0, 1, ABL>), dim3(256), dim3(256), 0, 0, A, (int64_t) n, B, (int64_t) n, C, (int64_t) n, n, n, n, 1.0f, 0.0f, 16, 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms / 5 < best ? ms / 5 : best;
  }
  return best;
}
#define R1(E) { float t = run_1w<E>(A, B, C, n); printf("1w ABL %2d: %.4f ms  %.1f TF\n", E, t, 2.0 * n * n * n / t / 1e9); }
#define R(E) printf("EXP %2d  NN %.4f ms (%.1f TF)   NT %.4f ms   TN %.4f ms\n", E, run<E, 0, 1>(A, B, C, n), 2.0 * n * n * n / run<E, 0, 1>(A, B, C, n) / 1e9, run<E, 0, 0>(A, B, C, n), run<E, 1, 1>(A, B, C, n));
int main() {
  const int n = 4096;
  float *A, *B, *C;
  hipMalloc(&A, (size_t) n * n * 4); hipMalloc(&B, (size_t) n * n * 4); hipMalloc(&C, (size_t) n * n * 4);
  std::vector<float> h((size_t) n * n);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float) ((i * 2654435761u) >> 8 & 0xffff) / 32768.0f - 1.0f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  R1(0) R1(32)
  return 0;
}
