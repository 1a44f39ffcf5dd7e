// Dev tool: ablation of the 256x256 double-buffered GEMM kernel (timing only; outputs of
// the ablated variants are wrong by construction).  EXP bits:
//   1 skip LDS staging writes in the loop     2 skip global loads in the loop
//   4 skip the per-slab barrier               8 operands from registers (no LDS reads)
//  16 skip MFMAs
#include "../blas-on-flash_amd/csrc/gemm_f32_mfma.hip"
#include <cstdio>
#include <vector>
using namespace bof;

template <int EXP, int AMODE, int BMODE>
__global__ void __launch_bounds__(512, 2)
expk(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
     float *__restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
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
  for (int kt = 0; kt < nkt; kt++) {
    const float *sA = lds + ((kt & 1) ? LDS_BUF : 0), *sB = sA + LDS_A;
    float *nA2 = lds + ((kt & 1) ? 0 : LDS_BUF), *nB2 = nA2 + LDS_A;
    if (kt + 1 < nkt && !(EXP & 1) && !(EXP & 32) && !(EXP & 256)) {
      float *nA = lds + ((kt & 1) ? 0 : LDS_BUF);
      r2s<AMODE, BM, NTHR>(nA, ra, t);
      r2s<BMODE, BN, NTHR>(nA + LDS_A, rb, t);
    }
    if (kt + 2 < nkt && !(EXP & 2) && !(EXP & 32) && !(EXP & 256)) {
      ra = g2r<AMODE, BM, NTHR, false>(A, lda, m0, (kt + 2) * BK, M, K, t);
      rb = g2r<BMODE, BN, NTHR, false>(B, ldb, n0, (kt + 2) * BK, N, K, t);
    }
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
    if (!(EXP & 4)) __syncthreads();
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
#define R(E) printf("EXP %2d  NN %.4f ms (%.1f TF)   NT %.4f ms   TN %.4f ms\n", E, run<E, 0, 1>(A, B, C, n), 2.0 * n * n * n / run<E, 0, 1>(A, B, C, n) / 1e9, run<E, 0, 0>(A, B, C, n), run<E, 1, 1>(A, B, C, n));
int main() {
  const int n = 4096;
  float *A, *B, *C;
  hipMalloc(&A, (size_t) n * n * 4); hipMalloc(&B, (size_t) n * n * 4); hipMalloc(&C, (size_t) n * n * 4);
  std::vector<float> h((size_t) n * n);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float) ((i * 2654435761u) >> 8 & 0xffff) / 32768.0f - 1.0f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  R(0) R(1024) R(2048) R(3072)
  return 0;
}
