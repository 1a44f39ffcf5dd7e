// gemm_f32_mfma.hip -- fp32 tile GEMM for gfx950 on v_mfma_f32_32x32x2_f32.
//
// Replaces the cblas_sgemm call inside GemmTask::execute
// (reference include/tasks/gemm_task.h:67-93): C = alpha*op(A)*op(B) + beta*C on
// one tile (M,N,K <= GEMM_BLK_SIZE+127), all eight (order, transA, transB)
// combinations.  Column-major is run as the row-major product of the swapped
// operands, so the device code only sees row-major C and two operand storage
// modes:
//   XMAJOR  element (x,k) at g[x*ld + k]   (k contiguous: A 'N', B 'T')
//   KMAJOR  element (x,k) at g[k*ld + x]   (x contiguous: A 'T', B 'N')
//
// Kernel shape: 256 threads = 4 waves (2x2), block tile 128x128x32, wave tile
// 64x64 = 2x2 MFMA 32x32 accumulators (64 VGPR).  Global->register->LDS staging
// with the next K-slab's global loads issued before the current slab's MFMAs
// (async-STAGE split); single LDS buffer (36 KB) so 3-4 blocks stay resident
// per CU and hide each other's barriers.
//
// Numerics: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain.  The LDS images
// are arranged so MFMA step s consumes k = 2s (lanes 0-31) and 2s+1 (lanes
// 32-63): every output element is therefore EXACTLY
//   acc = fmaf(a[k], b[k], acc) for k = 0..K-1, acc0 = 0
//   c   = beta == 0 ? alpha*acc : fmaf(alpha, acc, beta*c)
// which is what oracle/bof_oracle.c::orc_sgemm computes -> bit-exact parity.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bof {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, NTHR = 256;
constexpr int XLD = BK + 4;  // XMAJOR LDS row stride (floats): conflict-free b128 reads
enum { XMAJOR = 0, KMAJOR = 1 };
constexpr int LDS_FLOATS_OP = BM * XLD;  // >= BK*BM, one operand slab

// ---- global -> registers ---------------------------------------------------
// One K-slab of one operand = 1024 float4; thread t owns float4 #(t + 256*p),
// p = 0..3.  Kept as four named vector registers (an array passed by reference
// ends up in scratch and forces an early vmcnt wait).
struct Stage { f32x4 v0, v1, v2, v3; };

template <int MODE, bool GUARD>
__device__ __forceinline__ f32x4 g2r1(const float *__restrict__ g, int64_t ld, int x0, int k0,
                                      int X, int K, int f) {
  int x, k;
  if (MODE == XMAJOR) { x = x0 + (f >> 3); k = k0 + 4 * (f & 7); }
  else                { k = k0 + (f >> 5); x = x0 + 4 * (f & 31); }
  const float *src = (MODE == XMAJOR) ? g + (int64_t) x * ld + k : g + (int64_t) k * ld + x;
  if (!GUARD) return *reinterpret_cast<const f32x4 *>(src);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (MODE == XMAJOR) {
    if (x < X) {
      if (k + 0 < K) v[0] = src[0];
      if (k + 1 < K) v[1] = src[1];
      if (k + 2 < K) v[2] = src[2];
      if (k + 3 < K) v[3] = src[3];
    }
  } else {
    if (k < K) {
      if (x + 0 < X) v[0] = src[0];
      if (x + 1 < X) v[1] = src[1];
      if (x + 2 < X) v[2] = src[2];
      if (x + 3 < X) v[3] = src[3];
    }
  }
  return v;
}
template <int MODE, bool GUARD>
__device__ __forceinline__ Stage g2r(const float *__restrict__ g, int64_t ld, int x0, int k0,
                                     int X, int K, int t) {
  Stage s;
  s.v0 = g2r1<MODE, GUARD>(g, ld, x0, k0, X, K, t);
  s.v1 = g2r1<MODE, GUARD>(g, ld, x0, k0, X, K, t + NTHR);
  s.v2 = g2r1<MODE, GUARD>(g, ld, x0, k0, X, K, t + 2 * NTHR);
  s.v3 = g2r1<MODE, GUARD>(g, ld, x0, k0, X, K, t + 3 * NTHR);
  return s;
}

// ---- registers -> LDS --------------------------------------------------------
// XMAJOR image: row x holds, per group of 8 k's, [k0 k2 k4 k6 | k1 k3 k5 k7] so a
// lane of half h reads its four operands k = 8q+2c+h (c=0..3) with one
// ds_read_b128.  KMAJOR image: plain [k][x].
template <int MODE>
__device__ __forceinline__ void r2s1(float *__restrict__ s, const f32x4 v, int f) {
  if (MODE == XMAJOR) {
    const int row = f >> 3, kq = f & 7;
    float *dst = s + row * XLD + 8 * (kq >> 1) + 2 * (kq & 1);
    *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[2]);
    *reinterpret_cast<float2 *>(dst + 4) = make_float2(v[1], v[3]);
  } else {
    const int krow = f >> 5, xq = f & 31;
    *reinterpret_cast<f32x4 *>(s + krow * BM + 4 * xq) = v;
  }
}
template <int MODE>
__device__ __forceinline__ void r2s(float *__restrict__ s, const Stage &r, int t) {
  r2s1<MODE>(s, r.v0, t);
  r2s1<MODE>(s, r.v1, t + NTHR);
  r2s1<MODE>(s, r.v2, t + 2 * NTHR);
  r2s1<MODE>(s, r.v3, t + 3 * NTHR);
}

// ---- LDS -> MFMA operands ----------------------------------------------------
// returns the 4 operands (c = 0..3) of k-group q for sub-tile row/col `x`
template <int MODE>
__device__ __forceinline__ f32x4 s2op(const float *__restrict__ s, int x, int q, int h) {
  if (MODE == XMAJOR) {
    return *reinterpret_cast<const f32x4 *>(s + x * XLD + 8 * q + 4 * h);
  } else {
    const float *p = s + (8 * q + h) * BM + x;
    f32x4 v;
    v[0] = p[0]; v[1] = p[2 * BM]; v[2] = p[4 * BM]; v[3] = p[6 * BM];
    return v;
  }
}

template <int AMODE, int BMODE, bool GUARD>
__global__ void __launch_bounds__(NTHR)
sgemm_tile_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                  int64_t ldb, float *__restrict__ C, int64_t ldc, int M, int N, int K,
                  float alpha, float beta, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float lds[2 * LDS_FLOATS_OP];
  float *sA = lds, *sB = lds + LDS_FLOATS_OP;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch),
  // so give each XCD a contiguous run of tiles, walked in groups of 8 tile-rows
  // so concurrently resident blocks share A row-panels / B column-panels in L2.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int nkt = (K + BK - 1) / BK;
  Stage ra = g2r<AMODE, GUARD>(A, lda, m0, 0, M, K, t);
  Stage rb = g2r<BMODE, GUARD>(B, ldb, n0, 0, N, K, t);
  r2s<AMODE>(sA, ra, t);
  r2s<BMODE>(sB, rb, t);
  __syncthreads();

  for (int kt = 0; kt < nkt; kt++) {
    if (kt + 1 < nkt) {  // issue the next slab's global loads under this slab's MFMAs
      ra = g2r<AMODE, GUARD>(A, lda, m0, (kt + 1) * BK, M, K, t);
      rb = g2r<BMODE, GUARD>(B, ldb, n0, (kt + 1) * BK, N, K, t);
    }
#pragma unroll
    for (int q = 0; q < BK / 8; q++) {
      const f32x4 a0 = s2op<AMODE>(sA, wm * 64 + i, q, h);
      const f32x4 a1 = s2op<AMODE>(sA, wm * 64 + 32 + i, q, h);
      const f32x4 b0 = s2op<BMODE>(sB, wn * 64 + i, q, h);
      const f32x4 b1 = s2op<BMODE>(sB, wn * 64 + 32 + i, q, h);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      r2s<AMODE>(sA, ra, t);
      r2s<BMODE>(sB, rb, t);
      __syncthreads();
    }
  }

  // epilogue: 32x32 accumulator map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Addresses are (uniform tile pointer + uniform element offset)[32-bit lane offset] so
  // they stay in SGPRs; for beta != 0 a sub-tile's 16 C values are fetched before its
  // stores so the loads pipeline.
  float *ctile = C + (int64_t) m0 * ldc + n0;
  const int lrow = wm * 64 + 4 * h, lcol = wn * 64 + i;
  const int lane_off = lrow * (int) ldc + lcol;
#pragma unroll
  for (int mt = 0; mt < 2; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      f32x16 old;
      if (beta != 0.f) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int dr = mt * 32 + (r & 3) + 8 * (r >> 2), dc = nt * 32;
          const float *src = ctile + ((int64_t) dr * ldc + dc);
          old[r] = (!GUARD || (m0 + lrow + dr < M && n0 + lcol + dc < N)) ? src[lane_off] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int dr = mt * 32 + (r & 3) + 8 * (r >> 2), dc = nt * 32;
        float *dst = ctile + ((int64_t) dr * ldc + dc);
        if (!GUARD || (m0 + lrow + dr < M && n0 + lcol + dc < N))
          dst[lane_off] = (beta == 0.f) ? alpha * acc[mt][nt][r]
                                        : __builtin_fmaf(alpha, acc[mt][nt][r], beta * old[r]);
      }
    }
}

template <int AMODE, int BMODE>
static hipError_t launch_modes(const float *A, int64_t lda, const float *B, int64_t ldb,
                               float *C, int64_t ldc, int M, int N, int K, float alpha,
                               float beta, hipStream_t st) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const bool fast = (M % BM == 0) && (N % BN == 0) && (K % BK == 0) && (K > 0) && (lda % 4 == 0) &&
                    (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  dim3 grid(tiles_m * tiles_n), block(NTHR);
  if (fast)
    hipLaunchKernelGGL((sgemm_tile_kernel<AMODE, BMODE, false>), grid, block, 0, st, A, lda, B,
                       ldb, C, ldc, M, N, K, alpha, beta, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((sgemm_tile_kernel<AMODE, BMODE, true>), grid, block, 0, st, A, lda, B,
                       ldb, C, ldc, M, N, K, alpha, beta, tiles_m, tiles_n);
  return hipGetLastError();
}

// Row-major core: C[M x N] = alpha*op(A)*op(B) + beta*C.
static hipError_t sgemm_rm(bool ta, bool tb, int M, int N, int K, float alpha, const float *A,
                           int64_t lda, const float *B, int64_t ldb, float beta, float *C,
                           int64_t ldc, hipStream_t st) {
  // A: 'N' stored [M][K] -> XMAJOR; 'T' stored [K][M] -> KMAJOR
  // B: 'N' stored [K][N] -> KMAJOR; 'T' stored [N][K] -> XMAJOR
  if (!ta && !tb) return launch_modes<XMAJOR, KMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st);
  if (!ta && tb)  return launch_modes<XMAJOR, XMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st);
  if (ta && !tb)  return launch_modes<KMAJOR, KMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st);
  return launch_modes<KMAJOR, XMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st);
}

// cblas_sgemm argument meaning.  Column-major: C^T = op(B)^T * op(A)^T.
hipError_t sgemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha,
                 const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                 int64_t ldc, hipStream_t st) {
  if (m == 0 || n == 0) return hipSuccess;
  if (ord == 'C')
    return sgemm_rm(tb == 'T', ta == 'T', (int) n, (int) m, (int) k, alpha, b, ldb, a, lda, beta,
                    c, ldc, st);
  return sgemm_rm(ta == 'T', tb == 'T', (int) m, (int) n, (int) k, alpha, a, lda, b, ldb, beta, c,
                  ldc, st);
}

}  // namespace bof
