// csr_kernels.hip -- CSR x dense (SpMM) and CSR x vector (SpMV) tiles for gfx950.
//
// Replaces mkl_scsrmm in SimpleCsrmmRmTask/SimpleCsrmmCmTask::execute
// (reference include/tasks/csrmm_task.h:201-229, 278-313) and
// mkl_cspblas_scsrgemv in CsrGemv{NoTrans,Trans}InMem::execute
// (include/tasks/csrgemv_task.h:60-83, 152-179).
//
// These are HBM/gather-bound byte-moving kernels: no MFMA.  What matters is
// coalescing (one wave-instruction fetches one whole B row segment), many
// independent row gathers in flight per wave, and streaming the CSR arrays
// exactly once.  Column indices stay int64 as stored on disk; they are only
// read, never rewritten.
//
// Numerics: each output element is the fmaf chain over the row's non-zeros in
// storage order, then c = beta==0 ? alpha*acc : fmaf(alpha, acc, beta*c):
// identical to oracle/bof_oracle.c::orc_scsrmm / orc_scsrgemv ('N').
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "bof_hip.h"
#include "bof_internal.h"

namespace bof {

constexpr int CSR_WAVES = 4;  // waves per block, one CSR row per wave at a time

template <int VEC> struct vecf;
template <> struct vecf<1> { typedef float type; };
template <> struct vecf<2> { typedef float2 type; };
template <> struct vecf<4> { typedef float4 type; };

template <int VEC>
__device__ __forceinline__ void vfma(float v, const typename vecf<VEC>::type &b, float (&acc)[VEC]);
template <> __device__ __forceinline__ void vfma<1>(float v, const float &b, float (&acc)[1]) {
  acc[0] = __builtin_fmaf(v, b, acc[0]);
}
template <> __device__ __forceinline__ void vfma<2>(float v, const float2 &b, float (&acc)[2]) {
  acc[0] = __builtin_fmaf(v, b.x, acc[0]);
  acc[1] = __builtin_fmaf(v, b.y, acc[1]);
}
template <> __device__ __forceinline__ void vfma<4>(float v, const float4 &b, float (&acc)[4]) {
  acc[0] = __builtin_fmaf(v, b.x, acc[0]);
  acc[1] = __builtin_fmaf(v, b.y, acc[1]);
  acc[2] = __builtin_fmaf(v, b.z, acc[2]);
  acc[3] = __builtin_fmaf(v, b.w, acc[3]);
}

// Row-major B/C, contiguous columns.  One wave per CSR row; lane owns VEC
// consecutive output columns of a 64*VEC-wide column pass.  The row's (col,val)
// pairs are loaded 64 at a time with coalesced loads and broadcast by
// v_readlane; the B-row gathers are issued UNROLL at a time before the fmas.
template <int VEC, int UNROLL>
__global__ void __launch_bounds__(64 * CSR_WAVES)
csrmm_rowmajor_kernel(int64_t m, int n, float alpha, const float *__restrict__ val,
                      const int64_t *__restrict__ col, const int64_t *__restrict__ ptr,
                      const float *__restrict__ B, int64_t ldb, float beta,
                      float *__restrict__ C, int64_t ldc, unsigned *__restrict__ seen) {
  typedef typename vecf<VEC>::type V;
  const int lane = threadIdx.x & 63;
  if (seen && threadIdx.x == 0) atomicAdd(&seen[blockIdx.x], 1u);      // launch receipt (BOF_VERIFY)
  const int64_t row = (int64_t) blockIdx.x * CSR_WAVES + (threadIdx.x >> 6);
  if (row >= m) return;
  const int64_t base = ptr[0];
  const int64_t p0 = ptr[row] - base, p1 = ptr[row + 1] - base;

  for (int j0 = 0; j0 < n; j0 += 64 * VEC) {
    const int j = j0 + lane * VEC;
    const bool active = j < n;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = 0.f;

    for (int64_t p = p0; p < p1; p += 64) {
      const int cnt = (int) min((int64_t) 64, p1 - p);
      int mycol = 0;
      float myval = 0.f;
      if (lane < cnt) {
        mycol = (int) col[p + lane];
        myval = val[p + lane];
      }
      int tI = 0;
      for (; tI + UNROLL <= cnt; tI += UNROLL) {
        V bv[UNROLL];
        float vv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
          const int c = __builtin_amdgcn_readlane(mycol, tI + u);
          vv[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), tI + u));
          if (active) bv[u] = *reinterpret_cast<const V *>(B + (int64_t) c * ldb + j);
        }
        if (active) {
#pragma unroll
          for (int u = 0; u < UNROLL; u++) vfma<VEC>(vv[u], bv[u], acc);
        }
      }
      // the last cnt % UNROLL entries of the chunk: ONE more batch with the absent slots switched off (wave-uniform
      // tests), not one gather at a time -- at 100 non-zeros per row (cfg3: 64 + 36) the four single gathers of the
      // second chunk were four exposed latencies out of sixteen per row
      if (tI < cnt) {
        const int rem = cnt - tI;
        V bv[UNROLL];
        float vv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL - 1; u++) {
          if (u < rem) {
            const int c = __builtin_amdgcn_readlane(mycol, tI + u);
            vv[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), tI + u));
            if (active) bv[u] = *reinterpret_cast<const V *>(B + (int64_t) c * ldb + j);
          }
        }
        if (active) {
#pragma unroll
          for (int u = 0; u < UNROLL - 1; u++)
            if (u < rem) vfma<VEC>(vv[u], bv[u], acc);
        }
      }
    }
    if (active) {
      float *cp = C + row * ldc + j;
      if (beta == 0.f) {
#pragma unroll
        for (int e = 0; e < VEC; e++) acc[e] = alpha * acc[e];
      } else {
        float old[VEC];
        *reinterpret_cast<V *>(old) = *reinterpret_cast<const V *>(cp);
#pragma unroll
        for (int e = 0; e < VEC; e++) acc[e] = __builtin_fmaf(alpha, acc[e], beta * old[e]);
      }
      *reinterpret_cast<V *>(cp) = *reinterpret_cast<V *>(acc);
    }
  }
}

// Generic element-strided B/C (column-major 'C' layout, or unaligned row-major):
// B(c, j) = B[c*rsb + j*csb], C(i, j) = C[i*rsc + j*csc].  Lanes over rows for
// column-major so C stores coalesce; one thread owns one (row, column) pair.
__global__ void __launch_bounds__(256)
csrmm_strided_kernel(int64_t m, int n, float alpha, const float *__restrict__ val,
                     const int64_t *__restrict__ col, const int64_t *__restrict__ ptr,
                     const float *__restrict__ B, int64_t rsb, int64_t csb, float beta,
                     float *__restrict__ C, int64_t rsc, int64_t csc, int lanes_over_rows) {
  int64_t row;
  int j;
  if (lanes_over_rows) {
    row = (int64_t) blockIdx.x * 256 + threadIdx.x;
    j = blockIdx.y;
  } else {
    row = blockIdx.x;
    j = blockIdx.y * 256 + threadIdx.x;
  }
  if (row >= m || j >= n) return;
  const int64_t base = ptr[0];
  float acc = 0.f;
  for (int64_t p = ptr[row] - base; p < ptr[row + 1] - base; p++)
    acc = __builtin_fmaf(val[p], B[col[p] * rsb + (int64_t) j * csb], acc);
  float *cp = C + row * rsc + (int64_t) j * csc;
  *cp = (beta == 0.f) ? alpha * acc : __builtin_fmaf(alpha, acc, beta * (*cp));
}

// out[c * ld_out + r] = in[r * ld_in + c] for r < rows, c < cols: 64 x 64 tiles through a padded
// LDS image, 256-byte coalesced segments on both sides.  Used to run column-major ('C') CSRMM
// on the row-major kernel: B is transposed once per call, every C block on its way in/out.
__global__ void __launch_bounds__(256)
transpose_kernel(const float *__restrict__ in, int64_t ld_in, int64_t rows, int64_t cols,
                 float *__restrict__ out, int64_t ld_out) {
  __shared__ float tile[64][65];
  const int64_t r0 = (int64_t) blockIdx.y * 64, c0 = (int64_t) blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(r0 + i) * ld_in + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < cols && r0 + tx < rows) out[(c0 + i) * ld_out + r0 + tx] = tile[tx][i];
}

hipError_t transpose_f32(const float *in, int64_t ld_in, int64_t rows, int64_t cols, float *out,
                         int64_t ld_out, hipStream_t st) {
  drop_stale_error();
  if (rows == 0 || cols == 0) return hipSuccess;
  dim3 grid((unsigned) ((cols + 63) / 64), (unsigned) ((rows + 63) / 64)), block(256);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, in, ld_in, rows, cols, out, ld_out);
  return hipGetLastError();
}

// launch receipts (bof_internal.h): entries a launch marks, and the checker
int64_t scsrmm_receipt_entries(char ord_b, int64_t m) { return ord_b == 'R' ? (m + CSR_WAVES - 1) / CSR_WAVES : 0; }
__global__ void __launch_bounds__(256) csr_receipt_check_kernel(unsigned *seen, int64_t n, unsigned *flag) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (seen[i] != 1u) atomicAdd(flag, 1u);
  seen[i] = 0u;
}
hipError_t csr_receipt_check(unsigned *seen, int64_t n, unsigned *flag, hipStream_t st) {
  drop_stale_error();
  if (!seen || n <= 0) return hipSuccess;
  hipLaunchKernelGGL(csr_receipt_check_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, seen, n, flag);
  return hipGetLastError();
}

hipError_t scsrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha, const float *val,
                  const int64_t *col, const int64_t *ptr, const float *b, int64_t ldb, float beta,
                  float *c, int64_t ldc, hipStream_t st, unsigned *seen) {
  drop_stale_error();
  (void) k;
  if (m == 0 || n == 0) return hipSuccess;
  if (ord_b == 'R') {
    const bool al16 = (n % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(b) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(c) & 15) == 0);
    const bool al8 = (n % 2 == 0) && (ldb % 2 == 0) && (ldc % 2 == 0) &&
                     ((reinterpret_cast<uintptr_t>(b) & 7) == 0) &&
                     ((reinterpret_cast<uintptr_t>(c) & 7) == 0);
    dim3 grid((unsigned) ((m + CSR_WAVES - 1) / CSR_WAVES)), block(64 * CSR_WAVES);
    if (al16 && n > 128)
      hipLaunchKernelGGL((csrmm_rowmajor_kernel<4, 8>), grid, block, 0, st, m, (int) n, alpha, val,
                         col, ptr, b, ldb, beta, c, ldc, seen);
    else if (al8 && n > 64)
      hipLaunchKernelGGL((csrmm_rowmajor_kernel<2, 8>), grid, block, 0, st, m, (int) n, alpha, val,
                         col, ptr, b, ldb, beta, c, ldc, seen);
    else
      hipLaunchKernelGGL((csrmm_rowmajor_kernel<1, 8>), grid, block, 0, st, m, (int) n, alpha, val,
                         col, ptr, b, ldb, beta, c, ldc, seen);
  } else {
    dim3 grid((unsigned) ((m + 255) / 256), (unsigned) n), block(256);
    hipLaunchKernelGGL(csrmm_strided_kernel, grid, block, 0, st, m, (int) n, alpha, val, col, ptr,
                       b, (int64_t) 1, ldb, beta, c, (int64_t) 1, ldc, 1);
  }
  return hipGetLastError();
}

// ---- SpMV --------------------------------------------------------------------
// 'N': y[row] = fmaf chain over the row's entries in storage order (bit-exact vs oracle).
// A wave owns 64 consecutive rows.  Their entries are one contiguous range of val/col, so
// the wave streams it with fully coalesced 64-entry loads, gathers x[col] (several gathers
// in flight per lane) and parks the (val, x) pairs in LDS; then lane i walks row i's
// segment of the LDS image sequentially.  A thread-per-row walk of global memory instead
// touches 64 lines per load instruction and thrashes the 32 KB L1 (11.7 ms -> see DESIGN).
constexpr int GEMV_CAP = 1024;  // LDS entries per wave (8 KB): 64 rows x 16 nnz on average
__global__ void __launch_bounds__(256)
csrgemv_n_kernel(int64_t m, const float *__restrict__ val, const int64_t *__restrict__ ptr,
                 const int64_t *__restrict__ col, const float *__restrict__ x,
                 float *__restrict__ y, unsigned *__restrict__ seen) {
  __shared__ float2 sh[4][GEMV_CAP];
  if (seen && threadIdx.x == 0) atomicAdd(&seen[blockIdx.x], 1u);      // launch receipt (BOF_VERIFY)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r0 = ((int64_t) blockIdx.x * 4 + w) * 64;
  const int nrows = (int) max((int64_t) 0, min((int64_t) 64, m - r0));
  const int64_t base = ptr[0];
  int64_t p0 = 0, cnt = 0;
  if (nrows > 0) {
    p0 = ptr[r0] - base;
    cnt = ptr[r0 + nrows] - base - p0;
  }
  const bool staged = cnt <= GEMV_CAP;
  if (staged) {
    int64_t o = lane;
    for (; o + 192 < cnt; o += 256) {  // four coalesced chunks, four gathers in flight
      const int64_t c0 = col[p0 + o], c1 = col[p0 + o + 64], c2 = col[p0 + o + 128], c3 = col[p0 + o + 192];
      const float v0 = val[p0 + o], v1 = val[p0 + o + 64], v2 = val[p0 + o + 128], v3 = val[p0 + o + 192];
      const float x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
      sh[w][o] = make_float2(v0, x0);
      sh[w][o + 64] = make_float2(v1, x1);
      sh[w][o + 128] = make_float2(v2, x2);
      sh[w][o + 192] = make_float2(v3, x3);
    }
    // the last (up to three) chunks: their gathers in flight together too, absent ones switched off per lane -- at
    // 10 non-zeros per row (cfg5: 640 entries per wave = two batches of four + two chunks) the two single chunks were
    // two exposed gather latencies out of four
    if (o < cnt) {
      const bool h1 = o + 64 < cnt, h2 = o + 128 < cnt;
      const int64_t c0 = col[p0 + o], c1 = h1 ? col[p0 + o + 64] : 0, c2 = h2 ? col[p0 + o + 128] : 0;
      const float v0 = val[p0 + o], v1 = h1 ? val[p0 + o + 64] : 0.f, v2 = h2 ? val[p0 + o + 128] : 0.f;
      const float x0 = x[c0], x1 = h1 ? x[c1] : 0.f, x2 = h2 ? x[c2] : 0.f;
      sh[w][o] = make_float2(v0, x0);
      if (h1) sh[w][o + 64] = make_float2(v1, x1);
      if (h2) sh[w][o + 128] = make_float2(v2, x2);
    }
  }
  __syncthreads();
  if (lane >= nrows) return;
  const int64_t row = r0 + lane;
  const int64_t q0 = ptr[row] - base - p0, q1 = ptr[row + 1] - base - p0;
  float acc = 0.f;
  if (staged) {
    for (int64_t q = q0; q < q1; q++) {
      const float2 e = sh[w][q];
      acc = __builtin_fmaf(e.x, e.y, acc);
    }
  } else {  // a 64-row group heavier than the LDS image: walk global memory directly
    for (int64_t q = q0; q < q1; q++) acc = __builtin_fmaf(val[p0 + q], x[col[p0 + q]], acc);
  }
  y[row] = acc;
}

// 'T': y[col[p]] += val[p] * x[row]; one thread per row, fp32 atomics (the
// reference's mutex-guarded vector add, csrgemv_task.h:169-176, becomes
// memory-side atomic adds; result is order-independent for integer data).
__global__ void __launch_bounds__(256)
csrgemv_t_kernel(int64_t m, const float *__restrict__ val, const int64_t *__restrict__ ptr,
                 const int64_t *__restrict__ col, const float *__restrict__ x,
                 float *__restrict__ y, unsigned *__restrict__ seen) {
  if (seen && threadIdx.x == 0) atomicAdd(&seen[blockIdx.x], 1u);      // launch receipt (BOF_VERIFY)
  const int64_t row = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (row >= m) return;
  const int64_t base = ptr[0];
  const float xv = x[row];
  for (int64_t p = ptr[row] - base; p < ptr[row + 1] - base; p++)
    atomicAdd(y + col[p], val[p] * xv);
}

int64_t scsrgemv_receipt_entries(int64_t m) { return (m + 255) / 256; }       // (both kernels: a workgroup = 256 rows)
hipError_t scsrgemv(char trans, int64_t m, int64_t n, const float *val, const int64_t *ptr,
                    const int64_t *col, const float *x, float *y, hipStream_t st, unsigned *seen) {
  drop_stale_error();
  (void) n;
  if (m == 0) return hipSuccess;
  dim3 grid((unsigned) ((m + 255) / 256)), block(256);
  if (trans == 'N')  // same grid: a block = 4 waves x 64 rows
    hipLaunchKernelGGL(csrgemv_n_kernel, grid, block, 0, st, m, val, ptr, col, x, y, seen);
  else
    hipLaunchKernelGGL(csrgemv_t_kernel, grid, block, 0, st, m, val, ptr, col, x, y, seen);
  return hipGetLastError();
}

// dst[i] = srcs[0][i] + srcs[1][i] + ... (fixed order), srcs may live in the HBM of peer devices
// (peer access: the loads cross xGMI) and dst may be one of them.  The multi-device form of the
// reference's mutex-guarded vector add (include/tasks/csrgemv_task.h:169-176): device d owns
// segment d of y and sums that segment of every device's partial.
struct PartialPtrs { const float *p[BOF_MAX_DEVICES]; };
__global__ void __launch_bounds__(256)
sum_partials_kernel(float *dst, PartialPtrs src, int n_src, int64_t len) {
  const int64_t stride = (int64_t) gridDim.x * 256;
  for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < len; i += stride) {
    float acc = src.p[0][i];
    for (int s = 1; s < n_src; s++) acc += src.p[s][i];
    dst[i] = acc;
  }
}
hipError_t sum_partials(float *dst, const float *const *srcs, int n_src, int64_t len, hipStream_t st) {
  drop_stale_error();
  if (len <= 0 || n_src <= 0) return hipSuccess;
  if (n_src > BOF_MAX_DEVICES) return hipErrorInvalidValue;
  PartialPtrs pp;
  for (int s = 0; s < BOF_MAX_DEVICES; s++) pp.p[s] = s < n_src ? srcs[s] : nullptr;
  const unsigned blocks = (unsigned) std::min<int64_t>((len + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(blocks), dim3(256), 0, st, dst, pp, n_src, len);
  return hipGetLastError();
}

}  // namespace bof
