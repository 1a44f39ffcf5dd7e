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
// Numerics: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain.  The LDS images
// are arranged so MFMA step s consumes k = 2s (lanes 0-31) and 2s+1 (lanes
// 32-63): every output element is therefore EXACTLY
//   acc = fmaf(a[k], b[k], acc) for k = 0..K-1, acc0 = 0
//   c   = beta == 0 ? alpha*acc : fmaf(alpha, acc, beta*c)
// which is what oracle/bof_oracle.c::orc_sgemm computes -> bit-exact parity.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "bof_internal.h"

namespace bof {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int XLD = BK + 4;  // XMAJOR LDS row stride (floats): conflict-free b128 reads
enum { XMAJOR = 0, KMAJOR = 1 };

// ---- epilogue policies -------------------------------------------------------
// NoEpi: plain sgemm (what every GemmTask runs).  Rank1x2: the two K = 1 products that
// KMeansTask::execute adds to its tile after the main product (reference
// include/tasks/kmeans_task.h:73-81), fused into the store: each is rounded as the
// k-ordered chain above rounds a K = 1 sgemm with alpha = beta = 1,
//   c = c + round(u1[row]*v1[col]);  c = c + round(u2[row]*v2[col])
// (row, col: element of the row-major C this launch produces), so the fused tile equals the
// three-call sequence bit for bit while C crosses HBM once instead of five times.
//
// ChainEpi: one k-range of an ACCUMULATE CHAIN (flash::gemm's k-blocks, src/blas/gemm.cpp:122-127: task (l, i, j) runs
// behind (l-1, i, j) and adds to the same C tile).  The reference rounds once per block (C = alpha*A_l*B_l + 1*C);
// here the chain carries the RAW accumulators from launch to launch instead -- the fp32 partial sums leave the
// registers unscaled (raw_out), the next k-range starts from them (acc_in) and only the last launch applies
//   c = beta == 0 ? alpha*acc : fmaf(alpha, acc, beta*c_in)
// -- so a chain cut at ANY k positions produces exactly the bits of ONE launch over the whole K (fp32 values
// survive the round trip through HBM unchanged, and an MFMA accumulator started from a value continues the same
// k-ordered fmaf chain): the result of flash::gemm no longer depends on the tile size, the HBM budget or the
// schedule, and equals what drivers/in_mem_gemm.cpp:63-70 computes with one call.
struct NoEpi {
  static constexpr bool active = false;
  static constexpr bool chain = false;
};
struct Rank1x2 {
  static constexpr bool active = true;
  static constexpr bool chain = false;
  const float *u1, *v1, *u2, *v2;
  __host__ __device__ Rank1x2 shifted(int64_t dr, int64_t dc) const { return Rank1x2{u1 + dr, v1 + dc, u2 + dr, v2 + dc}; }
};
struct ChainEpi {
  static constexpr bool active = false;
  static constexpr bool chain = true;
  const float *acc_in;   // raw partial sums of the k-ranges before this one, laid out like C with ld_acc; nullptr: start at 0
  int64_t ld_acc;
  const float *c_in;     // final store: the matrix the caller's beta applies to, laid out like C with ld_cin; nullptr: C itself
  int64_t ld_cin;
  int raw_out;           // 1: store the accumulators unscaled (the chain goes on); 0: final store
  __host__ __device__ ChainEpi shifted(int64_t dr, int64_t dc) const {
    return ChainEpi{acc_in ? acc_in + dr * ld_acc + dc : nullptr, ld_acc, c_in ? c_in + dr * ld_cin + dc : nullptr, ld_cin, raw_out};
  }
};
inline NoEpi epi_shift(const NoEpi &e, int64_t, int64_t) { return e; }
inline Rank1x2 epi_shift(const Rank1x2 &e, int64_t dr, int64_t dc) { return e.shifted(dr, dc); }
inline ChainEpi epi_shift(const ChainEpi &e, int64_t dr, int64_t dc) { return e.shifted(dr, dc); }

// ---- global -> registers ---------------------------------------------------
// One K-slab of one operand of extent BX is BX*8 float4; thread t owns float4
// #(t + NTHR*p), p = 0..NP-1, kept in registers across the MFMA phase.
template <int NP> struct Stage { f32x4 v[NP]; };

template <int MODE, int BX, bool GUARD>
__device__ __forceinline__ f32x4 g2r1(const float *__restrict__ g, int64_t ld, int x0, int k0,
                                      int X, int K, int f) {
  int x, k;
  if (MODE == XMAJOR) { x = x0 + (f >> 3); k = k0 + 4 * (f & 7); }
  else                { k = k0 + f / (BX / 4); x = x0 + 4 * (f % (BX / 4)); }
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
template <int MODE, int BX, int NTHR, bool GUARD>
__device__ __forceinline__ Stage<BX * 8 / NTHR> g2r(const float *__restrict__ g, int64_t ld,
                                                    int x0, int k0, int X, int K, int t) {
  Stage<BX * 8 / NTHR> s;
#pragma unroll
  for (int p = 0; p < BX * 8 / NTHR; p++)
    s.v[p] = g2r1<MODE, BX, GUARD>(g, ld, x0, k0, X, K, t + p * NTHR);
  return s;
}

// ---- registers -> LDS --------------------------------------------------------
// XMAJOR image: row x holds, per group of 8 k's, [k0 k2 k4 k6 | k1 k3 k5 k7] so a
// lane of half h reads its four operands k = 8q+2c+h (c=0..3) with one
// ds_read_b128.  KMAJOR image: plain [k][x].
template <int MODE, int BX>
__device__ __forceinline__ void r2s1(float *__restrict__ s, const f32x4 v, int f) {
  if (MODE == XMAJOR) {
    const int row = f >> 3, kq = f & 7;
    float *dst = s + row * XLD + 8 * (kq >> 1) + 2 * (kq & 1);
    *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[2]);
    *reinterpret_cast<float2 *>(dst + 4) = make_float2(v[1], v[3]);
  } else {
    const int krow = f / (BX / 4), xq = f % (BX / 4);
    *reinterpret_cast<f32x4 *>(s + krow * BX + 4 * xq) = v;
  }
}
template <int MODE, int BX, int NTHR>
__device__ __forceinline__ void r2s(float *__restrict__ s, const Stage<BX * 8 / NTHR> &r, int t) {
#pragma unroll
  for (int p = 0; p < BX * 8 / NTHR; p++) r2s1<MODE, BX>(s, r.v[p], t + p * NTHR);
}

// ---- LDS -> MFMA operands ----------------------------------------------------
// returns the 4 operands (c = 0..3) of k-group q for sub-tile row/col `x`
template <int MODE, int BX>
__device__ __forceinline__ f32x4 s2op(const float *__restrict__ s, int x, int q, int h) {
  if (MODE == XMAJOR) {
    return *reinterpret_cast<const f32x4 *>(s + x * XLD + 8 * q + 4 * h);
  } else {
    const float *p = s + (8 * q + h) * BX + x;
    f32x4 v;
    v[0] = p[0]; v[1] = p[2 * BX]; v[2] = p[4 * BX]; v[3] = p[6 * BX];
    return v;
  }
}

// Block tile BM x BN x 32 computed by WM x WN waves; wave tile (BM/WM) x (BN/WN) =
// MT x NT accumulators of 32x32.  Two shapes are instantiated:
//   <128,128, 2x2 waves, single LDS buffer>: 64 accumulator VGPRs, 3 blocks/CU, two
//        barriers per slab -- ragged / small problems (GUARD variant) and tiles that
//        are not multiples of 256;
//   <256,256, 2x4 waves, double LDS buffer>: wave tile 128x64 (128 accumulator VGPRs),
//        144 KB LDS = one block per CU; a 4096^2 C tile is exactly 256 blocks, so the
//        whole launch is co-resident (no tail round); one barrier per slab: while slab
//        kt is multiplied out of buffer kt&1, slab kt+1 (fetched one slab earlier) is
//        written to the other buffer and slab kt+2's global loads are in flight.
template <int BM, int BN, int WM, int WN, bool DBUF, int AMODE, int BMODE, bool GUARD, class EP = NoEpi>
__global__ void __launch_bounds__(64 * WM * WN, 3)
sgemm_tile_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                  int64_t ldb, float *__restrict__ C, int64_t ldc, int M, int N, int K,
                  float alpha, float beta, int tiles_m, int tiles_n, EP ep) {
  constexpr int NTHR = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int LDS_A = (AMODE == XMAJOR) ? BM * XLD : BK * BM;
  constexpr int LDS_B = (BMODE == XMAJOR) ? BN * XLD : BK * BN;
  constexpr int LDS_BUF = LDS_A + LDS_B;
  __shared__ __attribute__((aligned(16))) float lds[(DBUF ? 2 : 1) * LDS_BUF];

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch),
  // so give each XCD a contiguous run of tiles, walked in groups of tile-rows
  // so concurrently resident blocks share A row-panels / B column-panels in L2.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = (BM == 256) ? 4 : 8;
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
  const int wm = wave / WN, wn = wave % WN;

  f32x16 acc[MT][NT];
  bool acc_from_mem = false;
  if constexpr (EP::chain) acc_from_mem = ep.acc_in != nullptr;
  if (!acc_from_mem) {
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  } else if constexpr (EP::chain) {   // the chain's partial sums so far (same element map as the store below)
    const float *atile = ep.acc_in + (int64_t) m0 * ep.ld_acc + n0;
    const int a_lane = (wm * WTM + 4 * h) * (int) ep.ld_acc + wn * WTN + i;
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int dr = a * 32 + (r & 3) + 8 * (r >> 2), dc = b * 32;
          const bool ok = !GUARD || (m0 + wm * WTM + 4 * h + dr < M && n0 + wn * WTN + i + dc < N);
          acc[a][b][r] = ok ? (atile + ((int64_t) dr * ep.ld_acc + dc))[a_lane] : 0.f;
        }
  }

  const int nkt = (K + BK - 1) / BK;
  auto ra = g2r<AMODE, BM, NTHR, GUARD>(A, lda, m0, 0, M, K, t);
  auto rb = g2r<BMODE, BN, NTHR, GUARD>(B, ldb, n0, 0, N, K, t);
  r2s<AMODE, BM, NTHR>(lds, ra, t);
  r2s<BMODE, BN, NTHR>(lds + LDS_A, rb, t);
  if (DBUF && nkt > 1) {
    ra = g2r<AMODE, BM, NTHR, GUARD>(A, lda, m0, BK, M, K, t);
    rb = g2r<BMODE, BN, NTHR, GUARD>(B, ldb, n0, BK, N, K, t);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; kt++) {
    const float *sA = lds + ((DBUF && (kt & 1)) ? LDS_BUF : 0);
    const float *sB = sA + LDS_A;
    if (DBUF) {
      if (kt + 1 < nkt) {  // slab kt+1 (loaded during slab kt-1) -> the other buffer
        float *nA = lds + ((kt & 1) ? 0 : LDS_BUF);
        r2s<AMODE, BM, NTHR>(nA, ra, t);
        r2s<BMODE, BN, NTHR>(nA + LDS_A, rb, t);
      }
      if (kt + 2 < nkt) {  // slab kt+2's global loads fly under this slab's MFMAs
        ra = g2r<AMODE, BM, NTHR, GUARD>(A, lda, m0, (kt + 2) * BK, M, K, t);
        rb = g2r<BMODE, BN, NTHR, GUARD>(B, ldb, n0, (kt + 2) * BK, N, K, t);
      }
    } else if (kt + 1 < nkt) {  // issue the next slab's global loads under this slab's MFMAs
      ra = g2r<AMODE, BM, NTHR, GUARD>(A, lda, m0, (kt + 1) * BK, M, K, t);
      rb = g2r<BMODE, BN, NTHR, GUARD>(B, ldb, n0, (kt + 1) * BK, N, K, t);
    }
#pragma unroll
    for (int q = 0; q < BK / 8; q++) {
      f32x4 a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) a[mt] = s2op<AMODE, BM>(sA, wm * WTM + mt * 32 + i, q, h);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) b[nt] = s2op<BMODE, BN>(sB, wn * WTN + nt * 32 + i, q, h);
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][c], b[nt][c], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
    if (!DBUF && kt + 1 < nkt) {
      r2s<AMODE, BM, NTHR>(lds, ra, t);
      r2s<BMODE, BN, NTHR>(lds + LDS_A, rb, t);
      __syncthreads();
    }
  }

  // epilogue: 32x32 accumulator map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Addresses are (uniform tile pointer + uniform element offset)[32-bit lane offset] so
  // they stay in SGPRs; for beta != 0 a sub-tile's 16 C values are fetched before its
  // stores so the loads pipeline.
  float *ctile = C + (int64_t) m0 * ldc + n0;
  const int lrow = wm * WTM + 4 * h, lcol = wn * WTN + i;
  const int lane_off = lrow * (int) ldc + lcol;
  // ChainEpi: the final store may take the caller's C from another matrix; a raw store takes none
  const float *itile = ctile;
  int64_t ldi = ldc;
  int lane_in = lane_off;
  bool raw = false;
  if constexpr (EP::chain) {
    raw = ep.raw_out != 0;
    if (ep.c_in) {
      itile = ep.c_in + (int64_t) m0 * ep.ld_cin + n0;
      ldi = ep.ld_cin;
      lane_in = lrow * (int) ep.ld_cin + lcol;
    }
  }
  // Rank1x2: this lane's column factors once, the 16 row factors of an accumulator row block once
  // per mt (fetched per element they doubled the instruction count of the store loop, and the
  // stores to C keep the compiler from hoisting them)
  float v1c[NT], v2c[NT];
  if constexpr (EP::active) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int col = n0 + lcol + nt * 32;
      const bool ok = !GUARD || col < N;
      v1c[nt] = ok ? ep.v1[col] : 0.f;
      v2c[nt] = ok ? ep.v2[col] : 0.f;
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    float u1r[16], u2r[16];
    if constexpr (EP::active) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + lrow + mt * 32 + (r & 3) + 8 * (r >> 2);
        const bool ok = !GUARD || row < M;
        u1r[r] = ok ? ep.u1[row] : 0.f;
        u2r[r] = ok ? ep.u2[row] : 0.f;
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      f32x16 old;
      if (beta != 0.f && !raw) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int dr = mt * 32 + (r & 3) + 8 * (r >> 2), dc = nt * 32;
          const float *src = itile + ((int64_t) dr * ldi + dc);
          old[r] = (!GUARD || (m0 + lrow + dr < M && n0 + lcol + dc < N)) ? src[lane_in] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int dr = mt * 32 + (r & 3) + 8 * (r >> 2), dc = nt * 32;
        float *dst = ctile + ((int64_t) dr * ldc + dc);
        if (!GUARD || (m0 + lrow + dr < M && n0 + lcol + dc < N)) {
          float t = raw ? acc[mt][nt][r]
                        : (beta == 0.f) ? alpha * acc[mt][nt][r] : __builtin_fmaf(alpha, acc[mt][nt][r], beta * old[r]);
          if constexpr (EP::active) {
            t = __fadd_rn(t, __fmul_rn(u1r[r], v1c[nt]));
            t = __fadd_rn(t, __fmul_rn(u2r[r], v2c[nt]));
          }
          dst[lane_off] = t;
        }
      }
    }
  }
}

// Epilogue shared by the one-wave-per-SIMD 256 x 256 kernels: the wave's 128 x 128 tile (4 x 4
// accumulators of 32 x 32; lane (i, h) holds rows 8q + 4h' ... of column i, the MFMA output
// layout) goes out as c = beta == 0 ? alpha*acc : fmaf(alpha, acc, beta*c).
template <class EP>
__device__ __forceinline__ void store_wave_tile_128(float *__restrict__ C, int64_t ldc, int m0, int n0, int wm,
                                                    int wn, int h, int i, const f32x16 (&acc)[4][4], float alpha,
                                                    float beta, const EP &ep) {
  float *ctile = C + (int64_t) m0 * ldc + n0;
  const int lane_off = (wm * 128 + 4 * h) * (int) ldc + wn * 128 + i;
  // ChainEpi: the final store may take the caller's C from another matrix; a raw store takes none
  const float *itile = ctile;
  int64_t ldi = ldc;
  int lane_in = lane_off;
  bool raw = false;
  if constexpr (EP::chain) {
    raw = ep.raw_out != 0;
    if (ep.c_in) {
      itile = ep.c_in + (int64_t) m0 * ep.ld_cin + n0;
      ldi = ep.ld_cin;
      lane_in = (wm * 128 + 4 * h) * (int) ep.ld_cin + wn * 128 + i;
    }
  }
  float v1c[4], v2c[4];
  if constexpr (EP::active) {
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      v1c[nt] = ep.v1[n0 + wn * 128 + i + nt * 32];
      v2c[nt] = ep.v2[n0 + wn * 128 + i + nt * 32];
    }
  }
#pragma unroll
  for (int mt = 0; mt < 4; mt++) {
    float u1r[16], u2r[16];
    if constexpr (EP::active) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        u1r[r] = ep.u1[m0 + wm * 128 + 4 * h + mt * 32 + (r & 3) + 8 * (r >> 2)];
        u2r[r] = ep.u2[m0 + wm * 128 + 4 * h + mt * 32 + (r & 3) + 8 * (r >> 2)];
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      f32x16 old;
      if (beta != 0.f && !raw) {
#pragma unroll
        for (int r = 0; r < 16; r++)
          old[r] = (itile + ((int64_t) (mt * 32 + (r & 3) + 8 * (r >> 2)) * ldi + nt * 32))[lane_in];
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float *dst = ctile + ((int64_t) (mt * 32 + (r & 3) + 8 * (r >> 2)) * ldc + nt * 32);
        float t = raw ? acc[mt][nt][r]
                      : (beta == 0.f) ? alpha * acc[mt][nt][r] : __builtin_fmaf(alpha, acc[mt][nt][r], beta * old[r]);
        if constexpr (EP::active) {
          t = __fadd_rn(t, __fmul_rn(u1r[r], v1c[nt]));
          t = __fadd_rn(t, __fmul_rn(u2r[r], v2c[nt]));
        }
        dst[lane_off] = t;
      }
    }
  }
}

// The accumulators of a wave's 128 x 128 tile at the start of a launch: zero, or -- ChainEpi with acc_in -- the
// chain's raw partial sums, fetched with the element map of the store above.
template <class EP>
__device__ __forceinline__ void init_wave_tile_128(f32x16 (&acc)[4][4], const EP &ep, int m0, int n0, int wm, int wn,
                                                   int h, int i) {
  bool from_mem = false;
  if constexpr (EP::chain) from_mem = ep.acc_in != nullptr;
  if (!from_mem) {
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    return;
  }
  if constexpr (EP::chain) {
    const float *atile = ep.acc_in + (int64_t) m0 * ep.ld_acc + n0;
    const int lane_off = (wm * 128 + 4 * h) * (int) ep.ld_acc + wn * 128 + i;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          acc[a][b][r] = (atile + ((int64_t) (a * 32 + (r & 3) + 8 * (r >> 2)) * ep.ld_acc + b * 32))[lane_off];
  }
}

// ---------------------------------------------------------------------------------------
// 256 x 256 x 32 block tile, ONE wave per SIMD (4 waves, wave tile 128 x 128 = 16
// accumulators = 256 AGPRs), double-buffered LDS, one barrier per slab.
//
// Why not two waves per SIMD: measured with s_memtime stamps (tools/gemm_exp.hip) on the
// 8-wave shape, the matrix pipe is arbitrated by age, so the older wave of each SIMD runs
// its 128 MFMAs nearly back-to-back, then idles ~8k cycles at the barrier while the younger
// wave runs alone with every LDS/staging stall exposed (13 % of the pipe idle).  With one
// wave per SIMD nothing is arbitrated; instead the wave hides its own latencies: the side
// work of a slab (operand reads for the NEXT k-group, LDS staging writes of slab kt+1,
// global loads of slab kt+2) is cut into 16 slots, one after each batch of 16 MFMAs, so the
// in-order issue stream never leaves the matrix pipe waiting on it.
//
// Variant 2 (the first version computed its addresses per access): every address in the
// slab loop is "per-thread base register + compile-time constant" (LDS: `offset:` immediates;
// global: uniform SGPR base + 32-bit per-thread byte offset), so a side slot contains memory
// instructions only, no address VALU.  A VMEM/LDS instruction costs ~40-60 issue cycles; with
// its 3-6 address VALUs it overflows the 64-cycle shadow of an fp32 MFMA and the overflow is
// matrix-pipe idle time.
struct Bases1w {
  const float *a_rd, *b_rd;  // operand-read bases in the slab being multiplied
  float *a_wr, *b_wr;        // staging-write bases in the other buffer
};

template <int MODE>
__device__ __forceinline__ f32x4 rd_op(const float *__restrict__ base, int sub, int q) {
  if (MODE == XMAJOR) {
    return *reinterpret_cast<const f32x4 *>(base + sub * 32 * XLD + 8 * q);
  } else {
    const float *p = base + (8 * q) * 256 + sub * 32;
    f32x4 v;
    v[0] = p[0]; v[1] = p[2 * 256]; v[2] = p[4 * 256]; v[3] = p[6 * 256];
    return v;
  }
}
template <int MODE>
__device__ __forceinline__ void wr_stage(float *__restrict__ base, const f32x4 v, int p) {
  if (MODE == XMAJOR) {
    float *d = base + p * 32 * XLD;
    *reinterpret_cast<float2 *>(d) = make_float2(v[0], v[2]);
    *reinterpret_cast<float2 *>(d + 4) = make_float2(v[1], v[3]);
  } else {
    *reinterpret_cast<f32x4 *>(base + p * 4 * 256) = v;
  }
}
// uniform tile/slab origin (SGPRs) + this thread's constant byte offset
template <int MODE>
__device__ __forceinline__ f32x4 ld_stage(const float *__restrict__ origin, int64_t ld, int k0, int p,
                                          unsigned goff) {
  const float *u = (MODE == XMAJOR) ? origin + (int64_t) (p * 32) * ld + k0
                                    : origin + (int64_t) (k0 + 4 * p) * ld;
  return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(u) + goff);
}

// K-tail version (only the last, partial slab is fetched with it): elements at k >= K read as 0,
// which leaves every fmaf chain unchanged.
template <int MODE>
__device__ __forceinline__ f32x4 ld_stage_tail(const float *__restrict__ origin, int64_t ld, int k0, int p,
                                               unsigned goff, int K, int t) {
  const float *u = (MODE == XMAJOR) ? origin + (int64_t) (p * 32) * ld + k0
                                    : origin + (int64_t) (k0 + 4 * p) * ld;
  const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(u) + goff);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (MODE == XMAJOR) {
    const int k = k0 + 4 * (t & 7);
    if (k + 0 < K) v[0] = src[0];
    if (k + 1 < K) v[1] = src[1];
    if (k + 2 < K) v[2] = src[2];
    if (k + 3 < K) v[3] = src[3];
  } else {
    if (k0 + 4 * p + (t >> 6) < K) v = *reinterpret_cast<const f32x4 *>(src);
  }
  return v;
}

template <int AMODE, int BMODE, bool W, bool L, bool TAIL = false>
__device__ __forceinline__ void slab_1w2(const Bases1w bs, const float *__restrict__ Ao, int64_t lda,
                                         const float *__restrict__ Bo, int64_t ldb, int k2,
                                         unsigned a_goff, unsigned b_goff, Stage<8> &ra, Stage<8> &rb,
                                         f32x16 (&acc)[4][4], int K = 0, int t = 0) {
  f32x4 a[2][4], b[2][4];
#pragma unroll
  for (int x = 0; x < 4; x++) {
    a[0][x] = rd_op<AMODE>(bs.a_rd, x, 0);
    b[0][x] = rd_op<BMODE>(bs.b_rd, x, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mt][c], b[q & 1][nt][c],
                                                             acc[mt][nt], 0, 0, 0);
      if (q < 3 && c < 3) {
#pragma unroll
        for (int x = (c == 0 ? 0 : c + 1); x <= c + 1; x++) {
          a[(q + 1) & 1][x] = rd_op<AMODE>(bs.a_rd, x, q + 1);
          b[(q + 1) & 1][x] = rd_op<BMODE>(bs.b_rd, x, q + 1);
        }
      }
      const int s = 4 * q + c;
      if (s < 8) {
        if (W) wr_stage<AMODE>(bs.a_wr, ra.v[s], s);
        if (L) ra.v[s] = TAIL ? ld_stage_tail<AMODE>(Ao, lda, k2, s, a_goff, K, t)
                              : ld_stage<AMODE>(Ao, lda, k2, s, a_goff);
      } else {
        if (W) wr_stage<BMODE>(bs.b_wr, rb.v[s - 8], s - 8);
        if (L) rb.v[s - 8] = TAIL ? ld_stage_tail<BMODE>(Bo, ldb, k2, s - 8, b_goff, K, t)
                                  : ld_stage<BMODE>(Bo, ldb, k2, s - 8, b_goff);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int AMODE, int BMODE, bool KTAIL = false, class EP = NoEpi>
__global__ void __launch_bounds__(256, 1)
sgemm_tile256_1w2_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                         int64_t ldb, float *__restrict__ C, int64_t ldc, int M, int N, int K,
                         float alpha, float beta, int tiles_m, int tiles_n, EP ep) {
  constexpr int LDS_A = (AMODE == XMAJOR) ? 256 * XLD : BK * 256;
  constexpr int LDS_B = (BMODE == XMAJOR) ? 256 * XLD : BK * 256;
  constexpr int LDS_BUF = LDS_A + LDS_B;
  __shared__ __attribute__((aligned(16))) float lds[2 * LDS_BUF];

  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * 256, n0 = tn * 256;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // per-thread constants: LDS element offsets inside one buffer, global byte offsets
  const int a_rd = (AMODE == XMAJOR) ? (wm * 128 + i) * XLD + 4 * h : h * 256 + wm * 128 + i;
  const int b_rd = LDS_A + ((BMODE == XMAJOR) ? (wn * 128 + i) * XLD + 4 * h : h * 256 + wn * 128 + i);
  const int a_wr = (AMODE == XMAJOR) ? (t >> 3) * XLD + 8 * ((t & 7) >> 1) + 2 * (t & 1)
                                     : (t >> 6) * 256 + 4 * (t & 63);
  const int b_wr = LDS_A + ((BMODE == XMAJOR) ? (t >> 3) * XLD + 8 * ((t & 7) >> 1) + 2 * (t & 1)
                                              : (t >> 6) * 256 + 4 * (t & 63));
  const unsigned a_goff = 4u * (unsigned) ((AMODE == XMAJOR) ? (t >> 3) * (int) lda + 4 * (t & 7)
                                                              : (t >> 6) * (int) lda + 4 * (t & 63));
  const unsigned b_goff = 4u * (unsigned) ((BMODE == XMAJOR) ? (t >> 3) * (int) ldb + 4 * (t & 7)
                                                              : (t >> 6) * (int) ldb + 4 * (t & 63));
  // uniform tile origins
  const float *Ao = (AMODE == XMAJOR) ? A + (int64_t) m0 * lda : A + m0;
  const float *Bo = (BMODE == XMAJOR) ? B + (int64_t) n0 * ldb : B + n0;
  Bases1w b0, b1;  // multiplying out of buffer 0 / buffer 1
  b0.a_rd = lds + a_rd;            b0.b_rd = lds + b_rd;
  b0.a_wr = lds + LDS_BUF + a_wr;  b0.b_wr = lds + LDS_BUF + b_wr;
  b1.a_rd = lds + LDS_BUF + a_rd;  b1.b_rd = lds + LDS_BUF + b_rd;
  b1.a_wr = lds + a_wr;            b1.b_wr = lds + b_wr;

  f32x16 acc[4][4];
  init_wave_tile_128(acc, ep, m0, n0, wm, wn, h, i);

  // caller guarantees at least two full slabs before a partial one (K >= 64; K >= 96 if K % 32);
  // KTAIL instantiation <=> K % 32 != 0: its last slab is fetched with guarded loads
  const int nkt = (K + BK - 1) / BK;
  Stage<8> ra, rb;
#pragma unroll
  for (int p = 0; p < 8; p++) {
    ra.v[p] = ld_stage<AMODE>(Ao, lda, 0, p, a_goff);
    rb.v[p] = ld_stage<BMODE>(Bo, ldb, 0, p, b_goff);
  }
#pragma unroll
  for (int p = 0; p < 8; p++) {
    wr_stage<AMODE>(b1.a_wr, ra.v[p], p);  // b1's write side is buffer 0
    wr_stage<BMODE>(b1.b_wr, rb.v[p], p);
  }
#pragma unroll
  for (int p = 0; p < 8; p++) {
    ra.v[p] = ld_stage<AMODE>(Ao, lda, BK, p, a_goff);
    rb.v[p] = ld_stage<BMODE>(Bo, ldb, BK, p, b_goff);
  }
  __syncthreads();

  int kt = 0;
  for (; kt + (KTAIL ? 3 : 2) < nkt; kt++) {
    const Bases1w bs = (kt & 1) ? b1 : b0;
    slab_1w2<AMODE, BMODE, true, true>(bs, Ao, lda, Bo, ldb, (kt + 2) * BK, a_goff, b_goff, ra, rb, acc);
    __syncthreads();
  }
  if (KTAIL) {  // slab nkt-3: the slab fetched now is the partial one
    const Bases1w bs = (kt & 1) ? b1 : b0;
    slab_1w2<AMODE, BMODE, true, true, true>(bs, Ao, lda, Bo, ldb, (kt + 2) * BK, a_goff, b_goff, ra, rb, acc,
                                             K, t);
    __syncthreads();
    kt++;
  }
  {
    const Bases1w bs = (kt & 1) ? b1 : b0;
    slab_1w2<AMODE, BMODE, true, false>(bs, Ao, lda, Bo, ldb, 0, a_goff, b_goff, ra, rb, acc);
    __syncthreads();
    kt++;
  }
  {
    const Bases1w bs = (kt & 1) ? b1 : b0;
    slab_1w2<AMODE, BMODE, false, false>(bs, Ao, lda, Bo, ldb, 0, a_goff, b_goff, ra, rb, acc);
  }

  store_wave_tile_128(C, ldc, m0, n0, wm, wn, h, i, acc, alpha, beta, ep);
}


// ---------------------------------------------------------------------------------------
// Variant 4: variant 3 with the two remaining sources of VALU work in the slab loop removed.
//  * LDS operand reads are explicit `ds_read2st64_b32` (two k-rows of the same sub-tile per
//    instruction; the 8-bit offsets count 256-byte units, so every (k-group, k) position of a
//    32 KB operand image is an immediate on one of four per-sub-tile base registers).  The
//    compiler's own pairing chose (sub-tile, sub-tile+1) pairs whose bases need a v_add each:
//    32 VALU adds per slab.
//  * LDS-DMA pieces are issued as `global_load_lds_dwordx4 voff, s[base:base+1]`: the piece
//    origin lives in SGPRs and advances with scalar adds; the builtin form computed a 64-bit
//    per-lane address (3 VALU ops per piece, 48 per slab) in the shadow of the same MFMAs.
//  * even/odd slabs are separate code, so the buffer choice is a constant, not a v_cndmask.
// Loads issued from inline asm are invisible to the compiler's waitcnt insertion: every k-group
// starts with an explicit `s_waitcnt lgkmcnt(0)` that also "produces" the fragment registers,
// which pins the MFMAs behind it.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_rd2st64(uint32_t addr) {
  f32x2 r;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(O0), "n"(O1) : "memory");
  return r;
}
// k-group Q (k = 8Q .. 8Q+7) of one sub-tile: rows 8Q + {0,2,4,6} + h of the [k][256] image
template <int Q>
__device__ __forceinline__ f32x4 rd_frag(uint32_t base) {
  const f32x2 lo = lds_rd2st64<Q * 32, Q * 32 + 8>(base);
  const f32x2 hi = lds_rd2st64<Q * 32 + 16, Q * 32 + 24>(base);
  f32x4 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
  return v;
}
__device__ __forceinline__ void lgkm_fence(f32x4 (&a)[4], f32x4 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
               :: "memory");
}
// (M0 carries the LDS destination; it is declared clobbered so the compiler reloads its own uses)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
#pragma clang diagnostic pop

// ---- synchronisation of the four waves of a workgroup: s_barrier, or progress counters in LDS (SYNC = 1) -----------
// With ONE wave per SIMD a wave that waits at s_barrier leaves its matrix pipe idle, and a barrier makes every slab as
// slow as its slowest wave (sum over slabs of the max over waves).  SYNC = 1 replaces the rendezvous by two counters
// in LDS that give every wave one k-group (~4 k cycles) of slack in both directions:
//   landed:    += 1 by every wave once ITS DMA pieces of the next slab are in LDS (s_waitcnt vmcnt(0) at the start of
//              k-group 2 -- the pieces were issued during group 0); a wave reads the next slab's first fragments (under
//              the MFMAs of group 3) only after landed >= 4 * (slabs so far): RAW.
//   read_done: += 1 by every wave once its last fragment reads of the current buffer have returned (start of group 3);
//              a wave issues DMA pieces into the other buffer (group 0 of the next slab) only after
//              read_done >= 4 * (slabs before this one): WAR.
// A counter cannot be satisfied by a fast wave's NEXT increment standing in for a slow wave's missing one: to publish
// `landed` of slab n + 2 a wave has passed group 3 of slab n, which needed `landed` of slab n + 1 from ALL four (and
// likewise for read_done).  The value a check needs is fetched one slot ahead (a ds_read_b32 behind the fragment reads, covered by
// the same s_waitcnt lgkmcnt(0) that opens the k-group); only a wave that is AHEAD of the others re-reads in a loop.
__device__ __forceinline__ void lds_count_up(uint32_t addr, uint32_t one) {      // one lane adds 1
  asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(addr), "v"(one) : "memory");
}
__device__ __forceinline__ uint32_t lds_peek(uint32_t addr) {                  // issue only; the next lgkmcnt(0) covers it
  uint32_t r;
  asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
__device__ __forceinline__ void lds_wait_count(uint32_t addr, uint32_t peeked, uint32_t target) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(peeked) :: "memory");      // (pins the use behind the peek's return; already 0 here)
  uint32_t v = __builtin_amdgcn_readfirstlane(peeked);
  while (v < target) {       // (a wave ahead of the others: spin on the counter)
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    v = __builtin_amdgcn_readfirstlane(r);
  }
}
struct SlabSync {
  uint32_t landed_addr, read_addr;   // LDS byte addresses of the two counters
  uint32_t one;                      // 1 in a VGPR
  uint32_t slab;                     // index of the slab being multiplied (uniform)
  uint32_t peek_land, peek_read;     // counter values fetched one slot ahead
};

// One 32-deep slab out of LDS buffer BUF.  Slots 0-3 issue the 16 DMA pieces of the next slab
// (origins a_next / b_next, 4-k-row stride a_step4 / b_step4 bytes) into the other buffer; 4 per
// slot is the measured optimum (all in slot 0: 141.9, 8 per slot: 144.5, 4: 146.1-146.7, 2: 145.0,
// 1: 144.6 TFLOP/s at 4096^3).  The barrier sits in front of the LAST k-group instead of after
// it: by then every wave has issued and consumed all its reads of this buffer (the fragments of
// group 3 were fetched during group 2) and the DMA pieces have had 8 slots to land, so the
// fragments of the next slab's group 0 are read under the MFMAs of group 3 and a slab starts
// with its operands in registers.  On entry a[0] / b[0] hold group 0 of this slab.
// ABL (tools/exp/dma2_ablate.hip, timing only -- results are wrong by construction): 1 no barrier, 2 no vmcnt wait,
// 4 no DMA pieces, 8 no fragment reads.
// IL = 1: the side work of a slot (fragment reads of the next k-group, DMA pieces of the next slab) is spread over the
// gaps between the slot's 16 MFMAs, at most one piece per gap, instead of standing in ONE gap behind them.  A wave
// issues an instruction every ~4 cycles and an MFMA keeps the pipe busy for 64: the ~35 instructions of slot 0's
// 8 reads + 4 DMA pieces (v_readlane, s_mov m0, s_nop, global_load_lds, two scalar adds each) took longer than the
// gap they stood in -- the ablation (tools/exp/dma2_ablate, profiles/r6/kernel_ablation.jsonl) prices the DMA pieces
// at 1.9 % and the fragment reads at 0.8 % of the whole-K launch, which is what a drained pipe per slot costs.
template <int BUF, int ABL = 0, int SYNC = 0, int IL = 0>
__device__ __forceinline__ void slab_dma2(const uint32_t (&a_base)[2][4], const uint32_t (&b_base)[2][4],
                                          uint64_t a_next, uint64_t b_next, uint64_t a_step4, uint64_t b_step4,
                                          unsigned a_goff, unsigned b_goff, uint32_t a_dst, uint32_t b_dst,
                                          f32x4 (&a)[2][4], f32x4 (&b)[2][4], f32x16 (&acc)[4][4], SlabSync &sy) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    lgkm_fence(a[q & 1], b[q & 1]);
    if (SYNC == 0) {
      if (q == 3) {
        if (!(ABL & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
        if (!(ABL & 1)) __syncthreads();
      }
    } else {
      if (q == 0) lds_wait_count(sy.read_addr, sy.peek_read, 4u * sy.slab);          // WAR: the other buffer is free
      if (q == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              // my pieces of the next slab are in LDS
        lds_count_up(sy.landed_addr, sy.one);
      }
      if (q == 3) {
        lds_count_up(sy.read_addr, sy.one);                                           // my reads of this buffer are done
        lds_wait_count(sy.landed_addr, sy.peek_land, 4u * (sy.slab + 1u));            // RAW: everybody's pieces are in
      }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int s = 4 * q + c;
      // fragment of sub-tile x of the NEXT k-group (A if !second, B if second): the buffer and k-group follow from q
      auto read_piece = [&](int x, bool second) {
        if (ABL & 8) return;
        if (!second) {
          if (q == 0) a[1][x] = rd_frag<1>(a_base[BUF][x]);
          if (q == 1) a[0][x] = rd_frag<2>(a_base[BUF][x]);
          if (q == 2) a[1][x] = rd_frag<3>(a_base[BUF][x]);
          if (q == 3) a[0][x] = rd_frag<0>(a_base[BUF ^ 1][x]);
        } else {
          if (q == 0) b[1][x] = rd_frag<1>(b_base[BUF][x]);
          if (q == 1) b[0][x] = rd_frag<2>(b_base[BUF][x]);
          if (q == 2) b[1][x] = rd_frag<3>(b_base[BUF][x]);
          if (q == 3) b[0][x] = rd_frag<0>(b_base[BUF ^ 1][x]);
        }
      };
      // DMA piece p (0..3) of slot s: pieces 2s, 2s+1 of A and of B; piece = k-rows 4p..4p+3 (one per wave)
      auto dma_piece = [&](int p) {
        if (ABL & 4) return;
        const int pc = 2 * s + (p >> 1);
        if (!(p & 1)) dma16(a_goff, a_next + (uint64_t) pc * a_step4, a_dst + pc * 4096);
        else dma16(b_goff, b_next + (uint64_t) pc * b_step4, b_dst + pc * 4096);
      };
      if (IL == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
          for (int nt = 0; nt < 4; nt++)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mt][c], b[q & 1][nt][c],
                                                               acc[mt][nt], 0, 0, 0);
        if (c < 3) {
#pragma unroll
          for (int x = (c == 0 ? 0 : c + 1); x <= c + 1; x++) { read_piece(x, false); read_piece(x, true); }
        }
        if (SYNC == 1 && c == 2) {        // behind this slot's fragment reads: what the next k-group's check will look at
          if (q == 2) sy.peek_land = lds_peek(sy.landed_addr);
          if (q == 3) sy.peek_read = lds_peek(sy.read_addr);
        }
        if (s < 4) { dma_piece(0); dma_piece(1); dma_piece(2); dma_piece(3); }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        const int x0 = c == 0 ? 0 : c + 1;        // first sub-tile whose next-group fragments this slot fetches (c < 3)
#pragma unroll
        for (int j = 0; j < 16; j++) {
          acc[j >> 2][j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][j >> 2][c], b[q & 1][j & 3][c],
                                                                     acc[j >> 2][j & 3], 0, 0, 0);
          // gap behind MFMA j: reads in the even gaps 0, 2 (, 4, 6 in slot 0 of a group), DMA pieces in gaps 1, 5, 9, 13,
          // the counter peek in gap 8
          if (c < 3 && (j == 0 || j == 2)) read_piece(x0, j == 2);
          if (c == 0 && (j == 4 || j == 6)) read_piece(1, j == 6);
          if (s < 4 && (j & 3) == 1) dma_piece(j >> 2);
          if (SYNC == 1 && c == 2 && j == 8) {
            if (q == 2) sy.peek_land = lds_peek(sy.landed_addr);
            if (q == 3) sy.peek_read = lds_peek(sy.read_addr);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  sy.slab++;
}

template <class EP = NoEpi, int ABL = 0, int SYNC = 0, int IL = 0>
__global__ void __launch_bounds__(256, 1)
sgemm_tile256_dma2_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                          int64_t ldb, float *__restrict__ C, int64_t ldc, int M, int N, int K,
                          float alpha, float beta, int tiles_m, int tiles_n, EP ep) {
  constexpr int LDS_A = BK * 256, LDS_BUF = 2 * BK * 256;   // floats
  __shared__ __attribute__((aligned(1024))) float lds[2 * LDS_BUF + (SYNC ? 4 : 0)];
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * 256, n0 = tn * 256;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float *) lds;
  uint32_t a_base[2][4], b_base[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; bf++)
#pragma unroll
    for (int x = 0; x < 4; x++) {
      a_base[bf][x] = lds0 + 4u * (unsigned) (bf * LDS_BUF + h * 256 + wm * 128 + x * 32 + i);
      b_base[bf][x] = lds0 + 4u * (unsigned) (bf * LDS_BUF + LDS_A + h * 256 + wn * 128 + x * 32 + i);
    }
  const unsigned a_goff = 4u * (unsigned) ((t >> 6) * (int) lda + 4 * (t & 63));
  const unsigned b_goff = 4u * (unsigned) ((t >> 6) * (int) ldb + 4 * (t & 63));
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // wave-uniform DMA destinations (byte addresses): k-row `wave` of piece 0, per buffer
  const uint32_t a_dst0 = lds0 + 4u * (unsigned) (wv * 256), b_dst0 = a_dst0 + 4u * LDS_A;
  const uint32_t a_dst1 = a_dst0 + 4u * LDS_BUF, b_dst1 = b_dst0 + 4u * LDS_BUF;
  const uint64_t a_org = reinterpret_cast<uint64_t>(A + m0), b_org = reinterpret_cast<uint64_t>(B + n0);
  const uint64_t a_step4 = (uint64_t) lda * 16, b_step4 = (uint64_t) ldb * 16;   // 4 k-rows, bytes
  const uint64_t a_slab = a_step4 * 8, b_slab = b_step4 * 8;                     // 32 k-rows

  f32x16 acc[4][4];
  SlabSync sy{};
  if (SYNC) {
    sy.landed_addr = lds0 + 4u * (unsigned) (2 * LDS_BUF);
    sy.read_addr = sy.landed_addr + 4u;
    sy.one = 1u;
    if (t == 0) { lds[2 * LDS_BUF] = 0.f; lds[2 * LDS_BUF + 1] = 0.f; }      // (bit pattern 0; in place before the barrier below)
  }

  const int nkt = K / BK;
#pragma unroll
  for (int p = 0; p < 8; p++) {
    dma16(a_goff, a_org + (uint64_t) p * a_step4, a_dst0 + p * 4096);
    dma16(b_goff, b_org + (uint64_t) p * b_step4, b_dst0 + p * 4096);
  }
  init_wave_tile_128(acc, ep, m0, n0, wm, wn, h, i);   // (a chain's partial sums arrive under the first slab's DMA)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 fa[2][4], fb[2][4];
#pragma unroll
  for (int x = 0; x < 4; x++) {
    fa[0][x] = rd_frag<0>(a_base[0][x]);
    fb[0][x] = rd_frag<0>(b_base[0][x]);
    fa[1][x] = fa[0][x]; fb[1][x] = fb[0][x];
  }
  // K is a multiple of 64 here (launch_modes), so the slabs go in (buffer 0, buffer 1) pairs and
  // the loop body is the same code for every pair: the last pair simply prefetches the last slab
  // again (valid memory, never used), which keeps the accumulators in one register assignment.
  uint64_t a_next = a_org + a_slab, b_next = b_org + b_slab;
  for (int kt = 0; kt < nkt; kt += 2) {
    slab_dma2<0, ABL, SYNC, IL>(a_base, b_base, a_next, b_next, a_step4, b_step4, a_goff, b_goff, a_dst1, b_dst1, fa, fb, acc, sy);
    const bool more = kt + 2 < nkt;
    a_next += more ? a_slab : 0; b_next += more ? b_slab : 0;
    slab_dma2<1, ABL, SYNC, IL>(a_base, b_base, a_next, b_next, a_step4, b_step4, a_goff, b_goff, a_dst0, b_dst0, fa, fb, acc, sy);
    a_next += more ? a_slab : 0; b_next += more ? b_slab : 0;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the look-ahead fragment reads of the last slab

  store_wave_tile_128(C, ldc, m0, n0, wm, wn, h, i, acc, alpha, beta, ep);
}

// ---------------------------------------------------------------------------------------
// Variant 5 (round 6): x-major operands (k contiguous: A 'N' = row-major [m][k], the reference's own layout; B 'T') through
// LDS-DMA as well, with NO k-major copy.  Described for A; B is the same with n for m.  LDS-DMA writes 16 bytes per lane to CONSECUTIVE LDS addresses, so it cannot
// produce the padded / permuted x-major image the register-staged kernels read with ds_read_b128; but every lane may
// FETCH any 16-byte chunk.  A slab of A is 256 rows x 128 bytes (32 k); one DMA piece is 8 rows = 64 lanes x 16 B, and
// lane (r, p) fetches chunk p ^ r of row r: the image is [row][chunk ^ (row & 7)] (an XOR swizzle of the 16-byte chunks
// inside each row).  Lane (i, h) of a wave needs, per group of four k (one chunk), the floats k = 4g + h and 4g + 2 + h of
// its row: two ds_read_b32 at  row * 128 + ((g ^ (i & 7)) << 4) + 4 h  (+ 8).  The XOR with g cannot be an immediate, so
// it is folded ONCE into eight base registers per buffer (sub-tile 0's chunk of group g); the sub-tile (32 rows =
// 4096 bytes) is the instructions' 16-bit immediate.  The slab loop therefore has no VALU instruction, like the
// k-major kernel's -- measured on the way here: the same loop with one v_xor per read (56 per slab) runs at 0.934 of
// peak, with a ds_read_b128 + two v_cndmask per fragment (120 VALU per slab) at 0.877, without VALU at 0.968: every
// VALU instruction between the MFMAs of a one-wave-per-SIMD loop costs ~10 cycles of the matrix pipe
// (profiles/r6/kernel/dmax_variants.txt).  The loop runs in EIGHT groups of four k (fragments double-buffered in half
// the registers of the k-major loop); MFMA step c of group g consumes k = 4g + 2c + h -- the same k-ordered chain as
// every other kernel: bit-identical results.  Synchronisation and side-work placement as in slab_dma2<SYNC = 1, IL = 1>:
// the DMA pieces go out during groups 1-2, `landed` is published at group 5 and checked at group 7, `read_done`
// published at group 7 and checked at group 1.  4096 x 32768 x 32768: 'N','N' 152.6, 'N','T' 152.4, 'T','T' 152.4 TFLOP/s =
// 0.969-0.970 ('T','N' through the k-major kernel on the same box: 0.965; the register-staged kernels: 0.930 / 0.910 /
// 0.934).
template <int O>
__device__ __forceinline__ float lds_rd32(uint32_t addr) {
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(O) : "memory");
  return r;
}
// the lane's two floats of sub-tile X's chunk, off the group's base register
template <int X>
__device__ __forceinline__ f32x2 rd_frag4_x(uint32_t base_g) {
  f32x2 v;
  v[0] = lds_rd32<X * 4096>(base_g);
  v[1] = lds_rd32<X * 4096 + 8>(base_g);
  return v;
}
__device__ __forceinline__ f32x2 rd_frag4_xs(uint32_t base_g, int x) {
  if (x == 0) return rd_frag4_x<0>(base_g);
  if (x == 1) return rd_frag4_x<1>(base_g);
  if (x == 2) return rd_frag4_x<2>(base_g);
  return rd_frag4_x<3>(base_g);
}
__device__ __forceinline__ void lgkm_fence2(f32x2 (&a)[4], f32x2 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
               :: "memory");
}
// B fragment (k-major image [k][256]) of group G: k-rows 4G + h and 4G + 2 + h (the base holds h and the sub-tile)
template <int G>
__device__ __forceinline__ f32x2 rd_frag4_k(uint32_t base) { return lds_rd2st64<G * 16, G * 16 + 8>(base); }

// AX / BX: the operand is x-major (swizzled image, per-group bases *_gb) or k-major ([k][256] image, per-sub-tile bases
// *_kb); *_step: bytes between two DMA pieces of the operand (32 rows x-major, 4 k-rows k-major)
template <int BUF, bool AX, bool BX>
__device__ __forceinline__ void slab_dmax(const uint32_t (&a_gb)[2][8], const uint32_t (&b_gb)[2][8],
                                          const uint32_t (&a_kb)[2][4], const uint32_t (&b_kb)[2][4],
                                          uint64_t a_next, uint64_t b_next, uint64_t a_step, uint64_t b_step,
                                          unsigned a_goff, unsigned b_goff, uint32_t a_dst, uint32_t b_dst,
                                          f32x2 (&a)[2][4], f32x2 (&b)[2][4], f32x16 (&acc)[4][4], SlabSync &sy) {
  // operand fragment of the NEXT group (g + 1; the other buffer's group 0 behind group 7)
  auto next_frag = [&](bool x_major, const uint32_t (&gb)[2][8], const uint32_t (&kb)[2][4], int g, int x) -> f32x2 {
    if (x_major) return g < 7 ? rd_frag4_xs(gb[BUF][g + 1], x) : rd_frag4_xs(gb[BUF ^ 1][0], x);
    if (g == 0) return rd_frag4_k<1>(kb[BUF][x]);
    if (g == 1) return rd_frag4_k<2>(kb[BUF][x]);
    if (g == 2) return rd_frag4_k<3>(kb[BUF][x]);
    if (g == 3) return rd_frag4_k<4>(kb[BUF][x]);
    if (g == 4) return rd_frag4_k<5>(kb[BUF][x]);
    if (g == 5) return rd_frag4_k<6>(kb[BUF][x]);
    if (g == 6) return rd_frag4_k<7>(kb[BUF][x]);
    return rd_frag4_k<0>(kb[BUF ^ 1][x]);
  };
#pragma unroll
  for (int g = 0; g < 8; g++) {
    lgkm_fence2(a[g & 1], b[g & 1]);
    if (g == 1) lds_wait_count(sy.read_addr, sy.peek_read, 4u * sy.slab);            // WAR: the other buffer is free
    if (g == 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // my pieces of the next slab are in LDS
      lds_count_up(sy.landed_addr, sy.one);
    }
    if (g == 7) {
      lds_count_up(sy.read_addr, sy.one);                                             // my reads of this buffer are done
      lds_wait_count(sy.landed_addr, sy.peek_land, 4u * (sy.slab + 1u));              // RAW: everybody's pieces are in
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        acc[j >> 2][j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][j >> 2][c], b[g & 1][j & 3][c],
                                                                   acc[j >> 2][j & 3], 0, 0, 0);
        // gaps behind the MFMAs of step 0: the fragments of the NEXT group (A in gaps 0, 4, 8, 12, B in 2, 6, 10, 14);
        // of step 1: the DMA pieces (groups 1-2) and the counter peeks
        if (c == 0 && (j & 1) == 0) {
          const int x = j >> 2;
          if (!(j & 2)) a[(g + 1) & 1][x] = next_frag(AX, a_gb, a_kb, g, x);
          else b[(g + 1) & 1][x] = next_frag(BX, b_gb, b_kb, g, x);
        }
        if ((g == 1 || g == 2) && (j & 3) == 1) {       // 16 DMA pieces over the four steps of groups 1-2, one per fourth gap
          const int s = 2 * (g - 1) + c, pc = 2 * s + (j >> 3);
          if (!((j >> 2) & 1)) dma16(a_goff, a_next + (uint64_t) pc * a_step, a_dst + pc * 4096);
          else dma16(b_goff, b_next + (uint64_t) pc * b_step, b_dst + pc * 4096);
        }
        if (c == 1 && j == 8) {
          if (g == 6) sy.peek_land = lds_peek(sy.landed_addr);
          if (g == 0) sy.peek_read = lds_peek(sy.read_addr);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  sy.slab++;
}

template <int AMODE, int BMODE, class EP = NoEpi>
__global__ void __launch_bounds__(256, 1)
sgemm_tile256_dmax_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                          int64_t ldb, float *__restrict__ C, int64_t ldc, int M, int N, int K,
                          float alpha, float beta, int tiles_m, int tiles_n, EP ep) {
  constexpr bool AX = AMODE == XMAJOR, BX = BMODE == XMAJOR;      // at least one of them (launch_modes)
  constexpr int LDS_A = BK * 256, LDS_BUF = 2 * BK * 256;   // floats
  __shared__ __attribute__((aligned(1024))) float lds[2 * LDS_BUF + 4];
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * 256, n0 = tn * 256;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float *) lds;
  // x-major image: [row][chunk ^ (row & 7)], 128 bytes a row.  *_gb[buffer][g]: chunk g of this lane's row in sub-tile
  // 0, its float h -- row * 128 + ((g ^ (i & 7)) << 4) + 4 h; sub-tile x is + x * 4096 (an immediate of the reads).
  // k-major image: [k][256]; *_kb[buffer][x]: k-row h, this lane's column of sub-tile x.
  uint32_t a_gb[2][8], b_gb[2][8], a_kb[2][4], b_kb[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; bf++) {
#pragma unroll
    for (int g = 0; g < 8; g++) {
      a_gb[bf][g] = AX ? lds0 + 4u * (unsigned) (bf * LDS_BUF) + (unsigned) (wm * 128 + i) * 128u + (unsigned) ((g ^ (i & 7)) << 4) + 4u * (unsigned) h : 0u;
      b_gb[bf][g] = BX ? lds0 + 4u * (unsigned) (bf * LDS_BUF + LDS_A) + (unsigned) (wn * 128 + i) * 128u + (unsigned) ((g ^ (i & 7)) << 4) + 4u * (unsigned) h : 0u;
    }
#pragma unroll
    for (int x = 0; x < 4; x++) {
      a_kb[bf][x] = AX ? 0u : lds0 + 4u * (unsigned) (bf * LDS_BUF + h * 256 + wm * 128 + x * 32 + i);
      b_kb[bf][x] = BX ? 0u : lds0 + 4u * (unsigned) (bf * LDS_BUF + LDS_A + h * 256 + wn * 128 + x * 32 + i);
    }
  }
  // DMA.  x-major operand: a piece = 8 rows (lane = row r8, position p: fetches chunk p ^ r8), wave w takes rows 8w.. of
  // every 32; k-major operand: a piece = 4 k-rows (one per wave), as in the k-major kernel
  const int r8 = lane >> 3, pp = lane & 7;
  const unsigned a_goff = AX ? 4u * (unsigned) ((8 * wave + r8) * (int) lda + ((pp ^ r8) << 2)) : 4u * (unsigned) ((t >> 6) * (int) lda + 4 * (t & 63));
  const unsigned b_goff = BX ? 4u * (unsigned) ((8 * wave + r8) * (int) ldb + ((pp ^ r8) << 2)) : 4u * (unsigned) ((t >> 6) * (int) ldb + 4 * (t & 63));
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t a_dst0 = lds0 + 4u * (unsigned) (wv * 256), b_dst0 = a_dst0 + 4u * LDS_A;
  const uint32_t a_dst1 = a_dst0 + 4u * LDS_BUF, b_dst1 = b_dst0 + 4u * LDS_BUF;
  const uint64_t a_org = reinterpret_cast<uint64_t>(AX ? A + (int64_t) m0 * lda : A + m0);
  const uint64_t b_org = reinterpret_cast<uint64_t>(BX ? B + (int64_t) n0 * ldb : B + n0);
  // bytes between two pieces (32 rows x-major / 4 k-rows k-major) and between two slabs (32 k)
  const uint64_t a_step = AX ? (uint64_t) lda * 128 : (uint64_t) lda * 16, b_step = BX ? (uint64_t) ldb * 128 : (uint64_t) ldb * 16;
  const uint64_t a_slab = AX ? 128 : (uint64_t) lda * 128, b_slab = BX ? 128 : (uint64_t) ldb * 128;

  f32x16 acc[4][4];
  SlabSync sy{};
  sy.landed_addr = lds0 + 4u * (unsigned) (2 * LDS_BUF);
  sy.read_addr = sy.landed_addr + 4u;
  sy.one = 1u;
  if (t == 0) { lds[2 * LDS_BUF] = 0.f; lds[2 * LDS_BUF + 1] = 0.f; }

  const int nkt = K / BK;
#pragma unroll
  for (int p = 0; p < 8; p++) {
    dma16(a_goff, a_org + (uint64_t) p * a_step, a_dst0 + p * 4096);
    dma16(b_goff, b_org + (uint64_t) p * b_step, b_dst0 + p * 4096);
  }
  init_wave_tile_128(acc, ep, m0, n0, wm, wn, h, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x2 fa[2][4], fb[2][4];
#pragma unroll
  for (int x = 0; x < 4; x++) {
    fa[0][x] = AX ? rd_frag4_xs(a_gb[0][0], x) : rd_frag4_k<0>(a_kb[0][x]);
    fb[0][x] = BX ? rd_frag4_xs(b_gb[0][0], x) : rd_frag4_k<0>(b_kb[0][x]);
    fa[1][x] = fa[0][x]; fb[1][x] = fb[0][x];
  }
  uint64_t a_next = a_org + a_slab, b_next = b_org + b_slab;
  for (int kt = 0; kt < nkt; kt += 2) {
    slab_dmax<0, AX, BX>(a_gb, b_gb, a_kb, b_kb, a_next, b_next, a_step, b_step, a_goff, b_goff, a_dst1, b_dst1, fa, fb, acc, sy);
    const bool more = kt + 2 < nkt;
    a_next += more ? a_slab : 0; b_next += more ? b_slab : 0;
    slab_dmax<1, AX, BX>(a_gb, b_gb, a_kb, b_kb, a_next, b_next, a_step, b_step, a_goff, b_goff, a_dst0, b_dst0, fa, fb, acc, sy);
    a_next += more ? a_slab : 0; b_next += more ? b_slab : 0;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  store_wave_tile_128(C, ldc, m0, n0, wm, wn, h, i, acc, alpha, beta, ep);
}

// ---------------------------------------------------------------------------------------
// Variant 4 for the layouts with an x-major operand (register staging: LDS-DMA cannot produce
// the permuted x-major image): the same hand-scheduled loop as sgemm_tile256_dma2_kernel --
// explicit LDS reads/writes with immediate offsets, global loads in the "SGPR base + VGPR
// offset" form, even/odd slabs as separate code, barrier in front of the last k-group.
// Data flow per slab n (buffer n & 1): slots 0-11 store the staged registers (slab n+1, fetched
// during slab n-1) into the other buffer and refill them with slab n+2; slot s waits with
// vmcnt(15): exactly the 15 younger loads may still be in flight.
template <int O>
__device__ __forceinline__ f32x4 lds_rd128(uint32_t addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(O) : "memory");
  return r;
}
template <int O>
__device__ __forceinline__ void lds_wr128(uint32_t addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(O) : "memory");
}
template <int O>
__device__ __forceinline__ void lds_wr64(uint32_t addr, f32x2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(O) : "memory");
}
__device__ __forceinline__ f32x4 gld128(uint32_t voff, uint64_t sbase) {
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
__device__ __forceinline__ void vm_fence15(f32x4 &v) { asm volatile("s_waitcnt vmcnt(15)" : "+v"(v) :: "memory"); }

// fragment of k-group Q, sub-tile X out of an operand image whose per-thread read base is `base`
template <int MODE, int Q, int X>
__device__ __forceinline__ f32x4 rd_frag_m(uint32_t base) {
  if (MODE == XMAJOR) return lds_rd128<4 * (X * 32 * XLD + 8 * Q)>(base);
  const f32x2 lo = lds_rd2st64<Q * 32, Q * 32 + 8>(base + 128 * X);
  const f32x2 hi = lds_rd2st64<Q * 32 + 16, Q * 32 + 24>(base + 128 * X);
  f32x4 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
  return v;
}
// staged piece P -> LDS image (x-major: permuted 8-groups [k0 k2 | k1 k3] + [k4 k6 | k5 k7] halves)
template <int MODE, int P>
__device__ __forceinline__ void wr_piece(uint32_t base, f32x4 v) {
  if (MODE == XMAJOR) {
    f32x2 lo, hi;
    lo[0] = v[0]; lo[1] = v[2]; hi[0] = v[1]; hi[1] = v[3];
    lds_wr64<4 * (P * 32 * XLD)>(base, lo);
    lds_wr64<4 * (P * 32 * XLD + 4)>(base, hi);
  } else {
    lds_wr128<4 * (P * 4 * 256)>(base, v);
  }
}

template <int Q, int MODE>
__device__ __forceinline__ void rd_group_x(uint32_t base, f32x4 (&dst)[4], int x) {
  if (x == 0) dst[0] = rd_frag_m<MODE, Q, 0>(base);
  if (x == 1) dst[1] = rd_frag_m<MODE, Q, 1>(base);
  if (x == 2) dst[2] = rd_frag_m<MODE, Q, 2>(base);
  if (x == 3) dst[3] = rd_frag_m<MODE, Q, 3>(base);
}

template <int MODE, int S>
__device__ __forceinline__ void stage_slot(uint32_t wr_base, f32x4 &reg, unsigned goff, uint64_t next, uint64_t step) {
  vm_fence15(reg);
  wr_piece<MODE, S>(wr_base, reg);
  reg = gld128(goff, next + (uint64_t) S * step);
}

// slab out of buffer BUF; a_rd/b_rd: read bases per buffer, a_wr/b_wr: write bases per buffer
template <int AMODE, int BMODE, int BUF>
__device__ __forceinline__ void slab_1w3(const uint32_t (&a_rd)[2], const uint32_t (&b_rd)[2],
                                         const uint32_t (&a_wr)[2], const uint32_t (&b_wr)[2],
                                         uint64_t a_next, uint64_t b_next, uint64_t a_step, uint64_t b_step,
                                         unsigned a_goff, unsigned b_goff, f32x4 (&ra)[8], f32x4 (&rb)[8],
                                         f32x4 (&a)[2][4], f32x4 (&b)[2][4], f32x16 (&acc)[4][4]) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    lgkm_fence(a[q & 1], b[q & 1]);   // lgkmcnt(0): fragments of this group AND this wave's LDS stores
    if (q == 3) __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mt][c], b[q & 1][nt][c],
                                                             acc[mt][nt], 0, 0, 0);
      if (c < 3) {
#pragma unroll
        for (int x = (c == 0 ? 0 : c + 1); x <= c + 1; x++) {
          if (q == 0) { rd_group_x<1, AMODE>(a_rd[BUF], a[1], x); rd_group_x<1, BMODE>(b_rd[BUF], b[1], x); }
          if (q == 1) { rd_group_x<2, AMODE>(a_rd[BUF], a[0], x); rd_group_x<2, BMODE>(b_rd[BUF], b[0], x); }
          if (q == 2) { rd_group_x<3, AMODE>(a_rd[BUF], a[1], x); rd_group_x<3, BMODE>(b_rd[BUF], b[1], x); }
          if (q == 3) { rd_group_x<0, AMODE>(a_rd[BUF ^ 1], a[0], x); rd_group_x<0, BMODE>(b_rd[BUF ^ 1], b[0], x); }
        }
      }
      const int s = 4 * q + c;   // staging: 16 pieces over slots 0-11 (A pieces 0-7, then B pieces 0-7)
      if (s == 0)  { stage_slot<AMODE, 0>(a_wr[BUF ^ 1], ra[0], a_goff, a_next, a_step); stage_slot<AMODE, 1>(a_wr[BUF ^ 1], ra[1], a_goff, a_next, a_step); }
      if (s == 1)  { stage_slot<AMODE, 2>(a_wr[BUF ^ 1], ra[2], a_goff, a_next, a_step); }
      if (s == 2)  { stage_slot<AMODE, 3>(a_wr[BUF ^ 1], ra[3], a_goff, a_next, a_step); }
      if (s == 3)  { stage_slot<AMODE, 4>(a_wr[BUF ^ 1], ra[4], a_goff, a_next, a_step); stage_slot<AMODE, 5>(a_wr[BUF ^ 1], ra[5], a_goff, a_next, a_step); }
      if (s == 4)  { stage_slot<AMODE, 6>(a_wr[BUF ^ 1], ra[6], a_goff, a_next, a_step); }
      if (s == 5)  { stage_slot<AMODE, 7>(a_wr[BUF ^ 1], ra[7], a_goff, a_next, a_step); }
      if (s == 6)  { stage_slot<BMODE, 0>(b_wr[BUF ^ 1], rb[0], b_goff, b_next, b_step); stage_slot<BMODE, 1>(b_wr[BUF ^ 1], rb[1], b_goff, b_next, b_step); }
      if (s == 7)  { stage_slot<BMODE, 2>(b_wr[BUF ^ 1], rb[2], b_goff, b_next, b_step); }
      if (s == 8)  { stage_slot<BMODE, 3>(b_wr[BUF ^ 1], rb[3], b_goff, b_next, b_step); }
      if (s == 9)  { stage_slot<BMODE, 4>(b_wr[BUF ^ 1], rb[4], b_goff, b_next, b_step); stage_slot<BMODE, 5>(b_wr[BUF ^ 1], rb[5], b_goff, b_next, b_step); }
      if (s == 10) { stage_slot<BMODE, 6>(b_wr[BUF ^ 1], rb[6], b_goff, b_next, b_step); }
      if (s == 11) { stage_slot<BMODE, 7>(b_wr[BUF ^ 1], rb[7], b_goff, b_next, b_step); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int AMODE, int BMODE, class EP = NoEpi>
__global__ void __launch_bounds__(256, 1)
sgemm_tile256_1w3_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                         int64_t ldb, float *__restrict__ C, int64_t ldc, int M, int N, int K,
                         float alpha, float beta, int tiles_m, int tiles_n, EP ep) {
  constexpr int LDS_A = (AMODE == XMAJOR) ? 256 * XLD : BK * 256;
  constexpr int LDS_B = (BMODE == XMAJOR) ? 256 * XLD : BK * 256;
  constexpr int LDS_BUF = LDS_A + LDS_B;
  __shared__ __attribute__((aligned(1024))) float lds[2 * LDS_BUF];
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * 256, n0 = tn * 256;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float *) lds;
  // per-thread element offsets inside one buffer (same maps as the 1w2 kernel)
  const int a_rd_e = (AMODE == XMAJOR) ? (wm * 128 + i) * XLD + 4 * h : h * 256 + wm * 128 + i;
  const int b_rd_e = LDS_A + ((BMODE == XMAJOR) ? (wn * 128 + i) * XLD + 4 * h : h * 256 + wn * 128 + i);
  const int a_wr_e = (AMODE == XMAJOR) ? (t >> 3) * XLD + 8 * ((t & 7) >> 1) + 2 * (t & 1)
                                       : (t >> 6) * 256 + 4 * (t & 63);
  const int b_wr_e = LDS_A + ((BMODE == XMAJOR) ? (t >> 3) * XLD + 8 * ((t & 7) >> 1) + 2 * (t & 1)
                                                : (t >> 6) * 256 + 4 * (t & 63));
  uint32_t a_rd[2], b_rd[2], a_wr[2], b_wr[2];
#pragma unroll
  for (int bf = 0; bf < 2; bf++) {
    a_rd[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + a_rd_e);
    b_rd[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + b_rd_e);
    a_wr[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + a_wr_e);
    b_wr[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + b_wr_e);
  }
  const unsigned a_goff = 4u * (unsigned) ((AMODE == XMAJOR) ? (t >> 3) * (int) lda + 4 * (t & 7)
                                                              : (t >> 6) * (int) lda + 4 * (t & 63));
  const unsigned b_goff = 4u * (unsigned) ((BMODE == XMAJOR) ? (t >> 3) * (int) ldb + 4 * (t & 7)
                                                              : (t >> 6) * (int) ldb + 4 * (t & 63));
  // piece origins: x-major piece p = rows 32p.. (step 32 rows), slab = +32 floats along the row;
  // k-major piece p = k-rows 4p.. (step 4 rows), slab = +32 rows
  const uint64_t a_org = reinterpret_cast<uint64_t>((AMODE == XMAJOR) ? A + (int64_t) m0 * lda : A + m0);
  const uint64_t b_org = reinterpret_cast<uint64_t>((BMODE == XMAJOR) ? B + (int64_t) n0 * ldb : B + n0);
  const uint64_t a_step = (AMODE == XMAJOR) ? (uint64_t) lda * 128 : (uint64_t) lda * 16;
  const uint64_t b_step = (BMODE == XMAJOR) ? (uint64_t) ldb * 128 : (uint64_t) ldb * 16;
  const uint64_t a_slab = (AMODE == XMAJOR) ? 128 : (uint64_t) lda * 128;
  const uint64_t b_slab = (BMODE == XMAJOR) ? 128 : (uint64_t) ldb * 128;

  f32x16 acc[4][4];
  init_wave_tile_128(acc, ep, m0, n0, wm, wn, h, i);

  const int nkt = K / BK;   // even, >= 2 (launch_modes)
  f32x4 ra[8], rb[8];
  // prologue: slab 0 -> buffer 0 through the registers, slab 1 -> registers
#pragma unroll
  for (int p = 0; p < 8; p++) {
    ra[p] = gld128(a_goff, a_org + (uint64_t) p * a_step);
    rb[p] = gld128(b_goff, b_org + (uint64_t) p * b_step);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]),
               "+v"(ra[6]), "+v"(ra[7]) :: "memory");
  asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]),
               "+v"(rb[7]) :: "memory");
  wr_piece<AMODE, 0>(a_wr[0], ra[0]); wr_piece<AMODE, 1>(a_wr[0], ra[1]); wr_piece<AMODE, 2>(a_wr[0], ra[2]);
  wr_piece<AMODE, 3>(a_wr[0], ra[3]); wr_piece<AMODE, 4>(a_wr[0], ra[4]); wr_piece<AMODE, 5>(a_wr[0], ra[5]);
  wr_piece<AMODE, 6>(a_wr[0], ra[6]); wr_piece<AMODE, 7>(a_wr[0], ra[7]);
  wr_piece<BMODE, 0>(b_wr[0], rb[0]); wr_piece<BMODE, 1>(b_wr[0], rb[1]); wr_piece<BMODE, 2>(b_wr[0], rb[2]);
  wr_piece<BMODE, 3>(b_wr[0], rb[3]); wr_piece<BMODE, 4>(b_wr[0], rb[4]); wr_piece<BMODE, 5>(b_wr[0], rb[5]);
  wr_piece<BMODE, 6>(b_wr[0], rb[6]); wr_piece<BMODE, 7>(b_wr[0], rb[7]);
  // the stores above read ra/rb: make the refills below wait for them (asm order is program order)
#pragma unroll
  for (int p = 0; p < 8; p++) ra[p] = gld128(a_goff, a_org + a_slab + (uint64_t) p * a_step);
#pragma unroll
  for (int p = 0; p < 8; p++) rb[p] = gld128(b_goff, b_org + b_slab + (uint64_t) p * b_step);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 fa[2][4], fb[2][4];
  fa[0][0] = rd_frag_m<AMODE, 0, 0>(a_rd[0]); fa[0][1] = rd_frag_m<AMODE, 0, 1>(a_rd[0]);
  fa[0][2] = rd_frag_m<AMODE, 0, 2>(a_rd[0]); fa[0][3] = rd_frag_m<AMODE, 0, 3>(a_rd[0]);
  fb[0][0] = rd_frag_m<BMODE, 0, 0>(b_rd[0]); fb[0][1] = rd_frag_m<BMODE, 0, 1>(b_rd[0]);
  fb[0][2] = rd_frag_m<BMODE, 0, 2>(b_rd[0]); fb[0][3] = rd_frag_m<BMODE, 0, 3>(b_rd[0]);
#pragma unroll
  for (int x = 0; x < 4; x++) { fa[1][x] = fa[0][x]; fb[1][x] = fb[0][x]; }

  // slab n stores slab n+1 (already in ra/rb) and fetches slab n+2; past the end the fetch
  // address stops advancing (valid memory, data never used)
  uint64_t a_next = a_org + 2 * a_slab, b_next = b_org + 2 * b_slab;
  if (nkt <= 2) { a_next = a_org + a_slab; b_next = b_org + b_slab; }
  for (int kt = 0; kt < nkt; kt += 2) {
    slab_1w3<AMODE, BMODE, 0>(a_rd, b_rd, a_wr, b_wr, a_next, b_next, a_step, b_step, a_goff, b_goff, ra, rb,
                              fa, fb, acc);
    const bool m1 = kt + 3 < nkt;
    a_next += m1 ? a_slab : 0; b_next += m1 ? b_slab : 0;
    slab_1w3<AMODE, BMODE, 1>(a_rd, b_rd, a_wr, b_wr, a_next, b_next, a_step, b_step, a_goff, b_goff, ra, rb,
                              fa, fb, acc);
    const bool m2 = kt + 4 < nkt;
    a_next += m2 ? a_slab : 0; b_next += m2 ? b_slab : 0;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // look-ahead loads/reads of the last slabs

  store_wave_tile_128(C, ldc, m0, n0, wm, wn, h, i, acc, alpha, beta, ep);
}

// ---------------------------------------------------------------------------------------
// Persistent form of sgemm_tile256_1w3_kernel for SHORT K (flash::kmeans: K = the point dimension).
// At K = 256 a 256 x 256 tile is 8 slabs = 131 k cycles of matrix-pipe time per wave; launched one
// workgroup per tile the kernel above spends a comparable time in its prologue (first slabs
// through an empty pipeline) and its 256 KB store with nothing else resident on the CU (one
// workgroup per CU), which is why short K used to go to the 128 x 128 kernel (three workgroups per
// CU, 97-106 TFLOP/s).  Here one workgroup per CU walks a run of tiles and the hand-scheduled slab
// pipeline simply continues across tile boundaries; between two tiles there is only the store
// (256 instructions per wave) and the clearing of the accumulators.  1M x 1024 x 256 'N','T':
// 116 -> 126 TFLOP/s (128 with 'N','N'), K = 128: 107 -> 111; without the store the loop runs at 139.6 / 135.3,
// i.e. what is left is the issue time of the store itself (256 KB per CU and tile through one
// 64 B/clk path, ~16 k cycles of a tile's 147 k), which cannot overlap the next tile's MFMAs
// because they overwrite the accumulators it reads.  Measured and not kept: a 128 x 128 persistent
// double-buffered kernel (2 workgroups per CU: 114, below the single-buffer kernel's 117-121 in
// one big launch), per-lane store offsets instead of scalar row addresses (the compiler hoists 64
// row offsets out of the tile loop into scratch: 94), non-temporal stores (no change), 16-byte stores after
// a 4 x 4 DPP transposition inside each lane quad (64 instead of 256 store instructions per wave: 124, no
// gain), workgroups started in eight phases an eighth of a tile apart so that the CUs do not all store at
// the same moment (no gain either).  What the numbers say: a CU stores one dword per lane at ~10-16 B/clk
// (the guide's 6 TB/s chip-wide for this store shape = 9.9 B/clk/CU), so a tile's 256 KB are 16-26 k cycles
// whatever the instruction width or the neighbours do.
template <int AMODE, int BMODE, class EP = NoEpi>
__global__ void __launch_bounds__(256, 1)
sgemm_tile256_p1w3_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                         int64_t ldb, float *__restrict__ C, int64_t ldc, int M, int N, int K,
                         float alpha, float beta, int tiles_m, int tiles_n, EP ep) {
  constexpr int LDS_A = (AMODE == XMAJOR) ? 256 * XLD : BK * 256;
  constexpr int LDS_B = (BMODE == XMAJOR) ? 256 * XLD : BK * 256;
  constexpr int LDS_BUF = LDS_A + LDS_B;
  __shared__ __attribute__((aligned(1024))) float lds[2 * LDS_BUF];
  // persistent: this workgroup walks the tiles [t0, t1) of the linear order tm * tiles_n + tn (tn fastest:
  // the tiles of a run share their A row block); workgroups b and b + 8 share an XCD and get adjacent runs
  const int G = (int) gridDim.x, w = ((int) blockIdx.x & 7) * (G >> 3) + ((int) blockIdx.x >> 3);
  const int64_t nt_all = (int64_t) tiles_m * tiles_n;
  const int t0 = (int) (nt_all * w / G), t1 = (int) (nt_all * (w + 1) / G);
  if (t0 >= t1) return;
  const int m0 = (t0 / tiles_n) * 256, n0 = (t0 % tiles_n) * 256;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float *) lds;
  // per-thread element offsets inside one buffer (same maps as the 1w2 kernel)
  const int a_rd_e = (AMODE == XMAJOR) ? (wm * 128 + i) * XLD + 4 * h : h * 256 + wm * 128 + i;
  const int b_rd_e = LDS_A + ((BMODE == XMAJOR) ? (wn * 128 + i) * XLD + 4 * h : h * 256 + wn * 128 + i);
  const int a_wr_e = (AMODE == XMAJOR) ? (t >> 3) * XLD + 8 * ((t & 7) >> 1) + 2 * (t & 1)
                                       : (t >> 6) * 256 + 4 * (t & 63);
  const int b_wr_e = LDS_A + ((BMODE == XMAJOR) ? (t >> 3) * XLD + 8 * ((t & 7) >> 1) + 2 * (t & 1)
                                                : (t >> 6) * 256 + 4 * (t & 63));
  uint32_t a_rd[2], b_rd[2], a_wr[2], b_wr[2];
#pragma unroll
  for (int bf = 0; bf < 2; bf++) {
    a_rd[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + a_rd_e);
    b_rd[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + b_rd_e);
    a_wr[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + a_wr_e);
    b_wr[bf] = lds0 + 4u * (unsigned) (bf * LDS_BUF + b_wr_e);
  }
  const unsigned a_goff = 4u * (unsigned) ((AMODE == XMAJOR) ? (t >> 3) * (int) lda + 4 * (t & 7)
                                                              : (t >> 6) * (int) lda + 4 * (t & 63));
  const unsigned b_goff = 4u * (unsigned) ((BMODE == XMAJOR) ? (t >> 3) * (int) ldb + 4 * (t & 7)
                                                              : (t >> 6) * (int) ldb + 4 * (t & 63));
  // piece origins: x-major piece p = rows 32p.. (step 32 rows), slab = +32 floats along the row;
  // k-major piece p = k-rows 4p.. (step 4 rows), slab = +32 rows
  auto a_origin = [&](int mm) { return reinterpret_cast<uint64_t>((AMODE == XMAJOR) ? A + (int64_t) mm * lda : A + mm); };
  auto b_origin = [&](int nn) { return reinterpret_cast<uint64_t>((BMODE == XMAJOR) ? B + (int64_t) nn * ldb : B + nn); };
  const uint64_t a_org = a_origin(m0), b_org = b_origin(n0);
  const uint64_t a_step = (AMODE == XMAJOR) ? (uint64_t) lda * 128 : (uint64_t) lda * 16;
  const uint64_t b_step = (BMODE == XMAJOR) ? (uint64_t) ldb * 128 : (uint64_t) ldb * 16;
  const uint64_t a_slab = (AMODE == XMAJOR) ? 128 : (uint64_t) lda * 128;
  const uint64_t b_slab = (BMODE == XMAJOR) ? 128 : (uint64_t) ldb * 128;

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  const int nkt = K / BK;   // even, >= 2 (launch_modes)
  f32x4 ra[8], rb[8];
  // prologue: slab 0 -> buffer 0 through the registers, slab 1 -> registers
#pragma unroll
  for (int p = 0; p < 8; p++) {
    ra[p] = gld128(a_goff, a_org + (uint64_t) p * a_step);
    rb[p] = gld128(b_goff, b_org + (uint64_t) p * b_step);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]),
               "+v"(ra[6]), "+v"(ra[7]) :: "memory");
  asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]),
               "+v"(rb[7]) :: "memory");
  wr_piece<AMODE, 0>(a_wr[0], ra[0]); wr_piece<AMODE, 1>(a_wr[0], ra[1]); wr_piece<AMODE, 2>(a_wr[0], ra[2]);
  wr_piece<AMODE, 3>(a_wr[0], ra[3]); wr_piece<AMODE, 4>(a_wr[0], ra[4]); wr_piece<AMODE, 5>(a_wr[0], ra[5]);
  wr_piece<AMODE, 6>(a_wr[0], ra[6]); wr_piece<AMODE, 7>(a_wr[0], ra[7]);
  wr_piece<BMODE, 0>(b_wr[0], rb[0]); wr_piece<BMODE, 1>(b_wr[0], rb[1]); wr_piece<BMODE, 2>(b_wr[0], rb[2]);
  wr_piece<BMODE, 3>(b_wr[0], rb[3]); wr_piece<BMODE, 4>(b_wr[0], rb[4]); wr_piece<BMODE, 5>(b_wr[0], rb[5]);
  wr_piece<BMODE, 6>(b_wr[0], rb[6]); wr_piece<BMODE, 7>(b_wr[0], rb[7]);
  // the stores above read ra/rb: make the refills below wait for them (asm order is program order)
#pragma unroll
  for (int p = 0; p < 8; p++) ra[p] = gld128(a_goff, a_org + a_slab + (uint64_t) p * a_step);
#pragma unroll
  for (int p = 0; p < 8; p++) rb[p] = gld128(b_goff, b_org + b_slab + (uint64_t) p * b_step);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 fa[2][4], fb[2][4];
  fa[0][0] = rd_frag_m<AMODE, 0, 0>(a_rd[0]); fa[0][1] = rd_frag_m<AMODE, 0, 1>(a_rd[0]);
  fa[0][2] = rd_frag_m<AMODE, 0, 2>(a_rd[0]); fa[0][3] = rd_frag_m<AMODE, 0, 3>(a_rd[0]);
  fb[0][0] = rd_frag_m<BMODE, 0, 0>(b_rd[0]); fb[0][1] = rd_frag_m<BMODE, 0, 1>(b_rd[0]);
  fb[0][2] = rd_frag_m<BMODE, 0, 2>(b_rd[0]); fb[0][3] = rd_frag_m<BMODE, 0, 3>(b_rd[0]);
#pragma unroll
  for (int x = 0; x < 4; x++) { fa[1][x] = fa[0][x]; fb[1][x] = fb[0][x]; }

  // Slab n stores slab n+1 (already in ra/rb) and fetches slab n+2 -- of THIS tile or, at a tile's end, the
  // first slabs of the NEXT one: when the last slab of a tile has been multiplied, buffer 0 holds the next
  // tile's slab 0, the staging registers its slab 1 and the fragment registers its first k-group, i.e. the
  // state the prologue above creates, so the next tile starts without one.  Past the run's last tile the
  // fetch address stops advancing (valid memory, data never used).
  // running prefetch cursor: the slab two ahead of the one being multiplied
  int pf_tile = t0, pf_kt = 2;
  uint64_t pf_a = a_org + 2 * a_slab, pf_b = b_org + 2 * b_slab;
  if (nkt <= 2) {                       // the tile has only slabs 0 and 1: two ahead is the next tile's slab 0
    pf_kt = 0;
    pf_tile = t0 + 1 < t1 ? t0 + 1 : t0;
    pf_a = a_origin((pf_tile / tiles_n) * 256);
    pf_b = b_origin((pf_tile % tiles_n) * 256);
  }
  auto pf_advance = [&]() {
    if (++pf_kt < nkt) { pf_a += a_slab; pf_b += b_slab; return; }
    if (pf_tile + 1 < t1) {
      pf_tile++;
      pf_kt = 0;
      pf_a = a_origin((pf_tile / tiles_n) * 256);
      pf_b = b_origin((pf_tile % tiles_n) * 256);
    } else {
      pf_kt = nkt - 1;                  // past the run: stay on valid memory
    }
  };
  for (int tile = t0; tile < t1; tile++) {
    for (int kt = 0; kt < nkt; kt += 2) {
      slab_1w3<AMODE, BMODE, 0>(a_rd, b_rd, a_wr, b_wr, pf_a, pf_b, a_step, b_step, a_goff, b_goff, ra, rb, fa, fb, acc);
      pf_advance();
      slab_1w3<AMODE, BMODE, 1>(a_rd, b_rd, a_wr, b_wr, pf_a, pf_b, a_step, b_step, a_goff, b_goff, ra, rb, fa, fb, acc);
      pf_advance();
    }
    // the look-ahead loads were issued during the last slab and have had most of it to land
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    store_wave_tile_128(C, ldc, (tile / tiles_n) * 256, (tile % tiles_n) * 256, wm, wn, h, i, acc, alpha, beta, ep);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  }
}

// per-device pools of call-lifetime HIP events (flash_support.cpp)
hipError_t pooled_event(hipEvent_t *e, bool timing);
void pooled_event_return(hipEvent_t e);
hipError_t pooled_stream(hipStream_t *s, bool copy_priority);
void pooled_stream_return(hipStream_t s);
// the stream the ragged strips of a big launch run on beside its interior kernel: one per host thread and device, taken
// from the process's stream pool at the thread's first such launch and handed back when the thread ends
static hipStream_t strip_stream() {
  struct Held {
    hipStream_t s[64] = {};
    ~Held() { for (hipStream_t x : s) pooled_stream_return(x); }
  };
  static thread_local Held held;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void) hipGetLastError(); return nullptr; }
  if (!held.s[dev] && pooled_stream(&held.s[dev], false) != hipSuccess) { (void) hipGetLastError(); held.s[dev] = nullptr; }
  return held.s[dev];
}

// debug / test knobs of the kernel choice, read at every launch
static int knob(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

template <int AMODE, int BMODE, class EP>
static hipError_t launch_guarded(const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                                 int64_t ldc, int M, int N, int K, float alpha, float beta,
                                 hipStream_t st, EP ep) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
  hipLaunchKernelGGL((sgemm_tile_kernel<128, 128, 2, 2, false, AMODE, BMODE, true, EP>),
                     dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, alpha,
                     beta, tiles_m, tiles_n, ep);
  return hipGetLastError();
}

template <int AMODE, int BMODE, class EP>
static hipError_t launch_modes(const float *A, int64_t lda, const float *B, int64_t ldb,
                               float *C, int64_t ldc, int M, int N, int K, float alpha,
                               float beta, hipStream_t st, EP ep) {
  const bool vec_ld = (lda % 4 == 0) && (ldb % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  const bool vec_ok = vec_ld && (K % BK == 0) && (K > 0);
  // 256x256 kernels (one wave per SIMD) for the tile-aligned part of the problem:
  //   K % 64 == 0: hand-scheduled slab loops (explicit LDS reads/writes with immediate offsets,
  //      scalar-addressed global loads / DMA, barrier in front of the last k-group): 'T','N'
  //      through LDS-DMA (sgemm_tile256_dma2_kernel: 147.6-149.5 TFLOP/s at 4096^3, 149.7 at
  //      K = 16384), the layouts with an x-major operand through registers
  //      (sgemm_tile256_1w3_kernel: NN 145.0, TT 145.4, NT 141.8);
  //   any other K: the compiler-scheduled register-staging loop (sgemm_tile256_1w2_kernel,
  //      140.6-141.3), its last partial slab fetched with guarded loads when K % 32 != 0.
  // Ragged sizes (tail-merged tiles of 4096+r, unaligned problem edges): the largest
  // 256-aligned interior runs on the big-tile kernel (its last K slab guarded when K % 32 != 0),
  // the right and bottom strips on the guarded 128x128 kernel.  Every output element is still
  // produced by one kernel as one k-ordered chain, so the split does not change a single bit.
  const int Mi = M - M % 256, Ni = N - N % 256;
  const bool k_ok = (K % BK == 0) ? (K >= 2 * BK) : (K >= 3 * BK);
  // K < 512 (flash::kmeans: K = the point dimension): a 256 x 256 tile then has less MFMA work than
  // its prologue + 256 KB store cost with one workgroup per CU; the 128 x 128 kernel keeps three
  // workgroups per CU in flight (1M x 1024 x 256: 5.7-6.0 ms against 6.5-8.4).  BOF_GEMM_SHORT_K overrides.
  const int short_k = knob("BOF_GEMM_SHORT_K", 512);
  if (vec_ld && k_ok && K >= short_k && (int64_t) (Mi / 256) * (Ni / 256) >= 128 && lda < (1 << 22) && ldb < (1 << 22)) {
    const int tiles_m = Mi / 256, tiles_n = Ni / 256;
    hipEvent_t fork_ev = nullptr;
    if ((N > Ni || M > Mi) && (int64_t) tiles_m * tiles_n >= 1024 && knob("BOF_GEMM_STRIP_STREAM", 1) != 0) {
      if (pooled_event(&fork_ev, false) != hipSuccess || hipEventRecord(fork_ev, st) != hipSuccess) {
        (void) hipGetLastError();
        if (fork_ev) pooled_event_return(fork_ev);
        fork_ev = nullptr;
      }
    }
    if (AMODE == KMAJOR && BMODE == KMAJOR && K % (2 * BK) == 0) {
      // round 6 default: the four waves synchronise through progress counters in LDS and the slot's side work is spread
      // over the MFMA gaps (SYNC = 1, IL = 1: 0.975 of peak on a whole-K panel launch against 0.956 with s_barrier, bit
      // for bit the same C -- profiles/r6/kernel/); BOF_GEMM_DMA2_SYNC=0 launches the s_barrier kernel (A/B)
      if (knob("BOF_GEMM_DMA2_SYNC", 1) != 0)
        hipLaunchKernelGGL((sgemm_tile256_dma2_kernel<EP, 0, 1, 1>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, B,
                           ldb, C, ldc, Mi, Ni, K, alpha, beta, tiles_m, tiles_n, ep);
      else
        hipLaunchKernelGGL((sgemm_tile256_dma2_kernel<EP, 0, 0, 0>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, B,
                           ldb, C, ldc, Mi, Ni, K, alpha, beta, tiles_m, tiles_n, ep);
    }
    else if (K % (2 * BK) == 0 && knob("BOF_GEMM_DMAX", 1) != 0)
      // every layout with an x-major operand ('N','N', 'N','T', 'T','T'): that operand straight from its rows through
      // XOR-swizzled LDS-DMA (variant 5, round 6: 150-152 TFLOP/s in all four layouts; the register-staged kernels ran
      // 'N','N' at 145.0, 'T','T' at 145.4, 'N','T' at 141.8).  $BOF_GEMM_DMAX=0 restores the register-staged kernel.
      hipLaunchKernelGGL((sgemm_tile256_dmax_kernel<AMODE, BMODE, EP>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, B,
                         ldb, C, ldc, Mi, Ni, K, alpha, beta, tiles_m, tiles_n, ep);
    else if (K % (2 * BK) == 0)
      hipLaunchKernelGGL((sgemm_tile256_1w3_kernel<AMODE, BMODE, EP>), dim3(tiles_m * tiles_n), dim3(256), 0, st,
                         A, lda, B, ldb, C, ldc, Mi, Ni, K, alpha, beta, tiles_m, tiles_n, ep);
    else if (K % BK == 0)
      hipLaunchKernelGGL((sgemm_tile256_1w2_kernel<AMODE, BMODE, false, EP>), dim3(tiles_m * tiles_n), dim3(256),
                         0, st, A, lda, B, ldb, C, ldc, Mi, Ni, K, alpha, beta, tiles_m, tiles_n, ep);
    else
      hipLaunchKernelGGL((sgemm_tile256_1w2_kernel<AMODE, BMODE, true, EP>), dim3(tiles_m * tiles_n), dim3(256),
                         0, st, A, lda, B, ldb, C, ldc, Mi, Ni, K, alpha, beta, tiles_m, tiles_n, ep);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || (N == Ni && M == Mi)) {
      if (fork_ev) pooled_event_return(fork_ev);     // (a failed interior launch must not keep the pool's event: ADVICE r5)
      return e;
    }
    // The strips touch other elements of C than the interior and only read A and B: on a big launch they run on a
    // stream of their own BESIDE the interior kernel instead of behind it (forked from `st` in front of the interior,
    // joined behind it) -- two launches of a 128-deep slab loop each, 0.9 ms of a 54 ms product at the paper's
    // 31000-edge shape (profiles/r5/ragged_probe.txt: 146.7 TFLOP/s interior alone, 144.4 with the strips behind it).
    // One such stream per host thread and device, kept for the life of the thread (BOF_GEMM_STRIP_STREAM=0: off).
    hipStream_t ss = st;
    hipEvent_t ev_join = nullptr;
    if (fork_ev) {
      ss = strip_stream();
      if (!ss || hipStreamWaitEvent(ss, fork_ev, 0) != hipSuccess) { (void) hipGetLastError(); ss = st; }
    }
    if (N > Ni) {  // right strip: rows [0, Mi), columns [Ni, N)
      const float *Bs = (BMODE == XMAJOR) ? B + (int64_t) Ni * ldb : B + Ni;
      e = launch_guarded<AMODE, BMODE>(A, lda, Bs, ldb, C + Ni, ldc, Mi, N - Ni, K, alpha, beta, ss,
                                       epi_shift(ep, 0, Ni));
    }
    if (e == hipSuccess && M > Mi) {  // bottom strip: rows [Mi, M), all columns
      const float *As = (AMODE == XMAJOR) ? A + (int64_t) Mi * lda : A + Mi;
      e = launch_guarded<AMODE, BMODE>(As, lda, B, ldb, C + (int64_t) Mi * ldc, ldc, M - Mi, N, K, alpha,
                                       beta, ss, epi_shift(ep, Mi, 0));
    }
    if (ss != st) {       // join: whatever comes behind this sgemm on `st` also waits for the strips
      hipError_t j = pooled_event(&ev_join, false);
      if (j == hipSuccess) j = hipEventRecord(ev_join, ss);
      if (j == hipSuccess) j = hipStreamWaitEvent(st, ev_join, 0);
      if (ev_join) pooled_event_return(ev_join);
      if (e == hipSuccess) e = j;
    }
    if (fork_ev) pooled_event_return(fork_ev);
    return e;
  }
  const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
  dim3 grid(tiles_m * tiles_n), block(256);
  // short K, 256-aligned, at least four tiles for each of the 256 persistent workgroups: the persistent
  // one-wave-per-SIMD kernel (BOF_GEMM_PERSIST=0: off)
  // (BOF_GEMM_PERSIST_MIN_TILES lowers the tile count for tests)
  const bool persist_on = knob("BOF_GEMM_PERSIST", 1) != 0;
  if (!EP::chain && persist_on && vec_ld && K % (2 * BK) == 0 && K >= 4 * BK && K < short_k && M % 256 == 0 && N % 256 == 0 &&
      (int64_t) (M / 256) * (N / 256) >= knob("BOF_GEMM_PERSIST_MIN_TILES", 1024) && lda < (1 << 22) && ldb < (1 << 22)) {
    // one workgroup per CU; BOF_GEMM_PERSIST_WGS (a multiple of 8) makes the runs longer on small test problems
    const int wgs = std::max(8, knob("BOF_GEMM_PERSIST_WGS", 256) / 8 * 8);
    hipLaunchKernelGGL((sgemm_tile256_p1w3_kernel<AMODE, BMODE, EP>), dim3(wgs), dim3(256), 0, st, A, lda, B, ldb, C,
                       ldc, M, N, K, alpha, beta, M / 256, N / 256, ep);
    return hipGetLastError();
  }
  if (vec_ok && M % 128 == 0 && N % 128 == 0)
    hipLaunchKernelGGL((sgemm_tile_kernel<128, 128, 2, 2, false, AMODE, BMODE, false, EP>), grid, block,
                       0, st, A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, tiles_m, tiles_n, ep);
  else
    hipLaunchKernelGGL((sgemm_tile_kernel<128, 128, 2, 2, false, AMODE, BMODE, true, EP>), grid, block,
                       0, st, A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, tiles_m, tiles_n, ep);
  return hipGetLastError();
}

// Row-major core: C[M x N] = alpha*op(A)*op(B) + beta*C.
template <class EP>
static hipError_t sgemm_rm(bool ta, bool tb, int M, int N, int K, float alpha, const float *A,
                           int64_t lda, const float *B, int64_t ldb, float beta, float *C,
                           int64_t ldc, hipStream_t st, EP ep) {
  // cblas_sgemm's quick return (the reference's call site: include/tasks/gemm_task.h:87-90; BLAS: "when
  // alpha is zero or k is zero, A and B are not referenced"): C = beta*C (+ the epilogue's updates), so
  // NaN / Inf in A or B must not reach C -- 0 * NaN would.  The guarded kernel over zero k-slabs with
  // alpha = 0 computes exactly that (acc = 0; c = beta == 0 ? 0 : beta*c) without touching A or B.
  // MKL's behaviour is pinned by tests/golden/mkl_golden_special.npz.
  // An accumulate chain (ChainEpi): the quick return belongs to the FINAL launch, which then also ignores the partial
  // sums (they may hold the 0 * NaN the rule exists to keep out); a raw launch just accumulates.
  // The FINAL launch of a chain over zero k-slabs (K == 0, alpha != 0) still owes the chain's partial sums their
  // alpha: only alpha == 0 is a quick return there (what spot_check_kernel recomputes) -- the kernels run
  // zero slabs from acc_in and apply the store rule with the caller's alpha.
  bool quick = alpha == 0.f || K == 0;
  if constexpr (EP::chain) {
    if (ep.raw_out) quick = false;
    else if (alpha == 0.f) ep.acc_in = nullptr;
    else if (ep.acc_in) quick = false;
  }
  if (quick) {
    K = 0;
    alpha = 0.f;
  }
  // A: 'N' stored [M][K] -> XMAJOR; 'T' stored [K][M] -> KMAJOR
  // B: 'N' stored [K][N] -> KMAJOR; 'T' stored [N][K] -> XMAJOR
  if (!ta && !tb) return launch_modes<XMAJOR, KMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st, ep);
  if (!ta && tb)  return launch_modes<XMAJOR, XMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st, ep);
  if (ta && !tb)  return launch_modes<KMAJOR, KMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st, ep);
  return launch_modes<KMAJOR, XMAJOR>(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st, ep);
}

__global__ void __launch_bounds__(256)
expand_tile_local_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t len, int64_t blk, int64_t nblk) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  const int64_t tile = i / blk < nblk - 1 ? i / blk : nblk - 1;
  dst[i] = src[i - tile * blk];
}
hipError_t expand_tile_local(const float *src, float *dst, int64_t len, int64_t blk, int64_t nblk, hipStream_t st) {
  drop_stale_error();
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(expand_tile_local_kernel, dim3((unsigned) ((len + 255) / 256)), dim3(256), 0, st, src, dst, len,
                     blk, nblk);
  return hipGetLastError();
}

// K % 64 != 0 on a product that is big enough for the 256 x 256 LDS-DMA kernels (round 6): those kernels take K in
// pairs of 32-deep slabs, and the whole launch used to fall to the register-staged kernel with a guarded last slab
// (0.89-0.90 of peak).  A chain cut at ANY k equals one launch bit for bit (ChainEpi), so the product runs as TWO launches
// of one chain: K - K % 64 through the DMA kernel with its raw sums stored, the last < 64 k through the guarded
// 128 x 128 kernel, which starts from those sums and applies the caller's alpha / beta.  `raw`: where the first launch
// may put its raw sums (laid out like C, leading dimension ld_raw) -- C itself when nothing of C is needed any more
// (beta == 0), the chain's own accumulator otherwise; nullptr: no such place, one launch.  $BOF_GEMM_KSPLIT=0: off.
static hipError_t sgemm_rm_ksplit(bool ta, bool tb, int M, int N, int K, float alpha, const float *A, int64_t lda, const float *B,
                                  int64_t ldb, float beta, float *C, int64_t ldc, hipStream_t st, ChainEpi ep, float *raw,
                                  int64_t ld_raw) {
  const int r = K % (2 * BK), K1 = K - r;
  const bool vec_ld = (lda % 4 == 0) && (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  const bool big = vec_ld && (int64_t) (M / 256) * (N / 256) >= 128 && K1 >= std::max(knob("BOF_GEMM_SHORT_K", 512), 16 * 2 * BK) &&
                   lda < (1 << 22) && ldb < (1 << 22);
  if (r == 0 || !big || !raw || alpha == 0.f || knob("BOF_GEMM_KSPLIT", 1) == 0)
    return sgemm_rm(ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, st, ep);
  const ChainEpi head{ep.acc_in, ep.ld_acc, nullptr, 0, 1};
  hipError_t e = sgemm_rm(ta, tb, M, N, K1, alpha, A, lda, B, ldb, beta, raw, ld_raw, st, head);
  if (e != hipSuccess) return e;
  // the operands' k origin moves on by K1: along the row of an x-major image ('N' A, 'T' B), down the rows of a k-major one
  const float *A2 = ta ? A + (int64_t) K1 * lda : A + K1;
  const float *B2 = tb ? B + K1 : B + (int64_t) K1 * ldb;
  const ChainEpi tail{raw, ld_raw, ep.c_in, ep.ld_cin, ep.raw_out};
  return sgemm_rm(ta, tb, M, N, r, alpha, A2, lda, B2, ldb, beta, C, ldc, st, tail);
}

// cblas_sgemm argument meaning.  Column-major: C^T = op(B)^T * op(A)^T.
hipError_t sgemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha,
                 const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                 int64_t ldc, hipStream_t st) {
  drop_stale_error();
  if (m == 0 || n == 0) return hipSuccess;
  if (k % (2 * BK) != 0 && beta == 0.f && k > 0) {       // (C is free to carry the chain's raw sums between the two launches)
    const ChainEpi whole{nullptr, 0, nullptr, 0, 0};
    if (ord == 'C')
      return sgemm_rm_ksplit(tb == 'T', ta == 'T', (int) n, (int) m, (int) k, alpha, b, ldb, a, lda, beta, c, ldc, st, whole, c, ldc);
    return sgemm_rm_ksplit(ta == 'T', tb == 'T', (int) m, (int) n, (int) k, alpha, a, lda, b, ldb, beta, c, ldc, st, whole, c, ldc);
  }
  if (ord == 'C')
    return sgemm_rm(tb == 'T', ta == 'T', (int) n, (int) m, (int) k, alpha, b, ldb, a, lda, beta,
                    c, ldc, st, NoEpi{});
  return sgemm_rm(ta == 'T', tb == 'T', (int) m, (int) n, (int) k, alpha, a, lda, b, ldb, beta, c,
                  ldc, st, NoEpi{});
}

// One k-range of an accumulate chain (ChainEpi above; GemmChain in bof_internal.h).  acc_in / c_in are stored like C
// (same order, their own leading dimensions), so the column-major swap applies to them unchanged.
hipError_t sgemm_chain(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, const float *a,
                       int64_t lda, const float *b, int64_t ldb, float beta, float *c, int64_t ldc, const GemmChain &ch,
                       hipStream_t st) {
  drop_stale_error();
  if (m == 0 || n == 0) return hipSuccess;
  const ChainEpi ep{ch.acc_in, ch.ld_acc, ch.c_in, ch.ld_cin, ch.raw_out ? 1 : 0};
  // where a ragged K's first launch may leave its raw sums (sgemm_rm_ksplit): the launch's own output when that is raw
  // anyway; the chain's accumulator (rewritten in place: its old contents are consumed by this very launch); C when
  // nothing of it is read (beta == 0 and no separate c_in)
  float *raw = ch.raw_out ? c : (ch.acc_in ? const_cast<float *>(ch.acc_in) : ((beta == 0.f && !ch.c_in) ? c : nullptr));
  const int64_t ld_raw = ch.raw_out ? ldc : (ch.acc_in ? ch.ld_acc : ldc);
  if (ord == 'C')
    return sgemm_rm_ksplit(tb == 'T', ta == 'T', (int) n, (int) m, (int) k, alpha, b, ldb, a, lda, beta, c, ldc, st, ep, raw, ld_raw);
  return sgemm_rm_ksplit(ta == 'T', tb == 'T', (int) m, (int) n, (int) k, alpha, a, lda, b, ldb, beta, c, ldc, st, ep, raw, ld_raw);
}

// KMeansTask::execute on one tile (reference include/tasks/kmeans_task.h:53-82):
//   C = alpha*op(A)*op(B) + beta*C;  C[r][c] += u1[r]*v1[c];  C[r][c] += u2[r]*v2[c]
// with r along m and c along n whatever the storage order (the reference passes
// u1 = c_l2sq, v1 = ones, u2 = ones, v2 = p_l2sq).  Column-major runs as the row-major product
// of the swapped operands, where the roles of the row and column vectors swap with them.
hipError_t sgemm_rank1x2(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha,
                         const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                         int64_t ldc, const float *u1, const float *v1, const float *u2, const float *v2,
                         hipStream_t st) {
  drop_stale_error();
  if (m == 0 || n == 0) return hipSuccess;
  if (ord == 'C')
    return sgemm_rm(tb == 'T', ta == 'T', (int) n, (int) m, (int) k, alpha, b, ldb, a, lda, beta,
                    c, ldc, st, Rank1x2{v1, u1, v2, u2});
  return sgemm_rm(ta == 'T', tb == 'T', (int) m, (int) n, (int) k, alpha, a, lda, b, ldb, beta, c,
                  ldc, st, Rank1x2{u1, v1, u2, v2});
}

// ---- BOF_VERIFY: spot check of one launch ---------------------------------------------------------------------------
// 64 output elements of a launch (the corners + pseudo-random positions drawn from `seed`) are recomputed by one wave,
// element by element, in the kernels' own arithmetic -- the k-ordered fmaf chain from 0 or from the chain's raw sums,
// then the launch's store rule -- from the operands as they stand behind the launch, and compared bit for bit with what
// the launch stored.  What it sees and the hand-over sums cannot: a launch that ran with another launch's arguments,
// ran too early (operands that landed later no longer reproduce what it stored), ran twice, or not at all.  The values
// the launch overwrites (raw sums it started from, the C its beta applies to) are captured in front of it.
// Both results go to the verify table as word sums (plain, and weighted by the sample index).
struct SpotView {
  const float *A, *B;
  int64_t lda, ldb;
  int ta, tb;             // row-major core terms: a(r,k) = ta ? A[k*lda + r] : A[r*lda + k]; b(k,c) = tb ? B[c*ldb + k] : B[k*ldb + c]
  int M, N, K;
  float alpha, beta;
  float *C;
  int64_t ldc;
  const float *acc_in, *c_in;
  int64_t ld_acc, ld_cin;
  int raw_out;
  const float *u1, *v1, *u2, *v2;   // Rank1x2 (nullptr: none), indexed by the row-major core's row / column
  uint64_t seed;
};
__device__ __forceinline__ void spot_position(const SpotView &v, int t, int *r, int *c) {
  uint64_t h = v.seed * 0x9E3779B97F4A7C15ull + (uint64_t) (t + 1) * 0xD1B54A32D192ED03ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  *r = (int) ((h & 0xFFFFFFFFu) % (uint32_t) v.M);
  *c = (int) ((h >> 32) % (uint32_t) v.N);
  if (t == 0) { *r = 0; *c = 0; }
  if (t == 1) { *r = v.M - 1; *c = v.N - 1; }
  if (t == 2) { *r = 0; *c = v.N - 1; }
  if (t == 3) { *r = v.M - 1; *c = 0; }
}
__global__ void __launch_bounds__(64) spot_capture_kernel(SpotView v, float *save) {
  int r, c;
  spot_position(v, (int) threadIdx.x, &r, &c);
  save[threadIdx.x] = v.acc_in ? v.acc_in[(int64_t) r * v.ld_acc + c] : 0.f;
  const float *ci = v.c_in ? v.c_in + (int64_t) r * v.ld_cin + c : v.C + (int64_t) r * v.ldc + c;
  save[64 + threadIdx.x] = (v.beta != 0.f && !v.raw_out) ? *ci : 0.f;
}
__global__ void __launch_bounds__(64) spot_check_kernel(SpotView v, const float *save, unsigned long long *exp2,
                                                        unsigned long long *got2) {
  const int t = (int) threadIdx.x;
  int r, c;
  spot_position(v, t, &r, &c);
  float acc = v.acc_in ? save[t] : 0.f;
  const bool quick = !v.raw_out && v.alpha == 0.f;      // cblas_sgemm's quick return: A, B and the sums are not looked at
  if (quick) acc = 0.f;
  else
    for (int k = 0; k < v.K; k++) {
      const float a = v.ta ? v.A[(int64_t) k * v.lda + r] : v.A[(int64_t) r * v.lda + k];
      const float b = v.tb ? v.B[(int64_t) c * v.ldb + k] : v.B[(int64_t) k * v.ldb + c];
      acc = __builtin_fmaf(a, b, acc);
    }
  float want = v.raw_out ? acc : (v.beta == 0.f ? v.alpha * acc : __builtin_fmaf(v.alpha, acc, v.beta * save[64 + t]));
  if (v.u1) {
    want = __fadd_rn(want, __fmul_rn(v.u1[r], v.v1[c]));
    want = __fadd_rn(want, __fmul_rn(v.u2[r], v.v2[c]));
  }
  const float got = v.C[(int64_t) r * v.ldc + c];
  unsigned long long we = __float_as_uint(want), wg = __float_as_uint(got);
  unsigned long long e1 = we, e2 = we * (unsigned long long) (t + 1), g1 = wg, g2 = wg * (unsigned long long) (t + 1);
  for (int off = 32; off > 0; off >>= 1) {
    e1 += __shfl_down(e1, off, 64); e2 += __shfl_down(e2, off, 64);
    g1 += __shfl_down(g1, off, 64); g2 += __shfl_down(g2, off, 64);
  }
  if (t == 0) {
    atomicAdd(&exp2[0], e1); atomicAdd(&exp2[1], e2);
    atomicAdd(&got2[0], g1); atomicAdd(&got2[1], g2);
  }
}
static SpotView spot_view(const SpotArgs &s) {
  SpotView v{};
  const bool cm = s.ord == 'C';
  // column-major: C^T = op(B)^T * op(A)^T, the row-major core sees the operands swapped (as sgemm does)
  v.A = cm ? s.b : s.a; v.lda = cm ? s.ldb : s.lda; v.ta = cm ? s.tb == 'T' : s.ta == 'T';
  v.B = cm ? s.a : s.b; v.ldb = cm ? s.lda : s.ldb; v.tb = cm ? s.ta == 'T' : s.tb == 'T';
  v.M = (int) (cm ? s.n : s.m); v.N = (int) (cm ? s.m : s.n); v.K = (int) s.k;
  v.alpha = s.alpha; v.beta = s.beta; v.C = s.c; v.ldc = s.ldc;
  v.acc_in = s.ch.acc_in; v.ld_acc = s.ch.ld_acc; v.c_in = s.ch.c_in; v.ld_cin = s.ch.ld_cin; v.raw_out = s.ch.raw_out ? 1 : 0;
  if (s.u1) {   // (u by C row, v by C column in the caller's terms; swapped with the operands)
    v.u1 = cm ? s.v1 : s.u1; v.v1 = cm ? s.u1 : s.v1; v.u2 = cm ? s.v2 : s.u2; v.v2 = cm ? s.u2 : s.v2;
  }
  v.seed = s.seed;
  return v;
}
hipError_t sgemm_spot_capture(const SpotArgs &s, float *save128, hipStream_t st) {
  drop_stale_error();
  if (s.m <= 0 || s.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(spot_capture_kernel, dim3(1), dim3(64), 0, st, spot_view(s), save128);
  return hipGetLastError();
}
hipError_t sgemm_spot_check(const SpotArgs &s, const float *save128, unsigned long long *exp2, unsigned long long *got2,
                            hipStream_t st) {
  drop_stale_error();
  if (s.m <= 0 || s.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(spot_check_kernel, dim3(1), dim3(64), 0, st, spot_view(s), save128, exp2, got2);
  return hipGetLastError();
}

}  // namespace bof
