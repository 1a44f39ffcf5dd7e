// gen_kernels.hip -- the reference's synthetic-input generators, run in HBM so
// BASELINE-size inputs (4-16 GiB dense, 12 GB CSR) never cross PCIe.
//   dense:  misc/dense_create.cpp:28-37   (mode 's': x[i] = i % 10, 'z': 0)
//   sparse: misc/sparse_create.cpp:50-81  (per-row glibc rand_r stream, sort,
//           unique, keep the smallest nnz_per_row columns; val = (i % 9) + 1)
// Checked against oracle/bof_oracle.c (itself pinned to the reference tools'
// known-answer hashes) in tests/test_gpu_generators.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "bof_internal.h"

namespace bof {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256)
gen_dense_kernel(float *__restrict__ d, int64_t first, int64_t count, int mode, uint64_t seed) {
  const int64_t stride = (int64_t) gridDim.x * 256;
  for (int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x; t < count; t += stride) {
    const int64_t g = first + t;
    float v;
    if (mode == 's') v = (float) (g % 10);
    else if (mode == 'u') {  // uniform [-1,1): 24 random mantissa bits
      const uint32_t r = (uint32_t) (splitmix64(seed ^ (uint64_t) g * 0xD1342543DE82EF95ull) >> 40);
      v = (float) r * (1.0f / 8388608.0f) - 1.0f;
    } else v = 0.f;
    d[t] = v;
  }
}

hipError_t gen_dense(float *d, int64_t first, int64_t count, char mode, uint64_t seed,
                     hipStream_t st) {
  drop_stale_error();
  if (count == 0) return hipSuccess;
  int64_t blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gen_dense_kernel, dim3((unsigned) blocks), dim3(256), 0, st, d, first, count,
                     (int) mode, seed);
  return hipGetLastError();
}

// glibc rand_r is three steps of s <- s*1103515245 + 12345 (mod 2^32) whose
// outputs contribute 11, 10 and 10 bits.  Draw t of a row consumes LCG steps
// 6t+1 .. 6t+6, so lane t jumps straight to its state with the affine power
// (A^(6t), C_(6t)) computed by repeated squaring.
__device__ __forceinline__ void lcg_jump(uint32_t steps, uint32_t &mul, uint32_t &add) {
  uint32_t a = 1103515245u, c = 12345u;  // one step: s -> a*s + c
  mul = 1u; add = 0u;
  while (steps) {
    if (steps & 1u) { add = add * a + c; mul = mul * a; }
    c = c * a + c;  // (a,c) o (a,c): s -> a*(a*s+c)+c
    a = a * a;
    steps >>= 1;
  }
}
__device__ __forceinline__ int rand_r_dev(uint32_t &s) {
  s = s * 1103515245u + 12345u;
  int r = (int) ((s >> 16) & 2047u);
  s = s * 1103515245u + 12345u;
  r = (r << 10) ^ (int) ((s >> 16) & 1023u);
  s = s * 1103515245u + 12345u;
  r = (r << 10) ^ (int) ((s >> 16) & 1023u);
  return r;
}

// One block per row; bitonic sort of the ndraw candidates in LDS.
template <int NP2>
__global__ void __launch_bounds__(256)
gen_sparse_kernel(int64_t row0, int64_t nrows, int64_t ncols, int nnz_per_row,
                  float *__restrict__ csr, int64_t *__restrict__ col, int64_t *__restrict__ off) {
  __shared__ int64_t keys[NP2];
  __shared__ int flag[NP2];
  const int ndraw = nnz_per_row + 40;
  for (int64_t rr = blockIdx.x; rr < nrows; rr += gridDim.x) {
    const int64_t r = row0 + rr;
    for (int t = threadIdx.x; t < NP2; t += 256) {
      int64_t key = INT64_MAX;
      if (t < ndraw) {
        uint32_t mul, add;
        lcg_jump(6u * (uint32_t) t, mul, add);
        uint32_t s = mul * (uint32_t) r + add;
        const int64_t lo = rand_r_dev(s);
        const int64_t hi = rand_r_dev(s);
        key = (lo + hi * (int64_t) 2147483647) % ncols;
      }
      keys[t] = key;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= NP2; k2 <<= 1)
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < NP2; t += 256) {
          const int ixj = t ^ j;
          if (ixj > t) {
            const int64_t a = keys[t], b = keys[ixj];
            const bool up = (t & k2) == 0;
            if ((a > b) == up) { keys[t] = b; keys[ixj] = a; }
          }
        }
        __syncthreads();
      }
    // unique: position of each first occurrence = exclusive count of earlier firsts
    for (int t = threadIdx.x; t < NP2; t += 256)
      flag[t] = (t < ndraw && (t == 0 || keys[t] != keys[t - 1])) ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {  // ndraw <= 2048: a serial scan is negligible next to the sort
      int run = 0;
      for (int t = 0; t < ndraw; t++) { const int f = flag[t]; flag[t] = f ? run : -1; run += f; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ndraw; t += 256) {
      const int pos = flag[t];
      if (pos >= 0 && pos < nnz_per_row) {
        const int64_t g = r * nnz_per_row + pos;
        col[rr * nnz_per_row + pos] = keys[t];
        csr[rr * nnz_per_row + pos] = (float) ((g % 9) + 1);
      }
    }
    if (threadIdx.x == 0) {
      off[rr] = r * nnz_per_row;
      if (rr == nrows - 1) off[nrows] = (row0 + nrows) * nnz_per_row;
    }
    __syncthreads();
  }
}

hipError_t gen_sparse_rows(int64_t row0, int64_t nrows, int64_t ncols, int64_t nnz_per_row,
                           float *csr, int64_t *col, int64_t *off, hipStream_t st) {
  drop_stale_error();
  if (nrows == 0) return hipSuccess;
  const int64_t ndraw = nnz_per_row + 40;
  if (ndraw > 2048) return hipErrorInvalidValue;
  int64_t blocks = nrows < 65536 ? nrows : 65536;
  dim3 grid((unsigned) blocks), block(256);
#define BOF_GEN(NP2)                                                                          \
  hipLaunchKernelGGL((gen_sparse_kernel<NP2>), grid, block, 0, st, row0, nrows, ncols,        \
                     (int) nnz_per_row, csr, col, off)
  if (ndraw <= 64) BOF_GEN(64);
  else if (ndraw <= 128) BOF_GEN(128);
  else if (ndraw <= 256) BOF_GEN(256);
  else if (ndraw <= 512) BOF_GEN(512);
  else if (ndraw <= 1024) BOF_GEN(1024);
  else BOF_GEN(2048);
#undef BOF_GEN
  return hipGetLastError();
}

// ---- BOF_VERIFY: word sums of a 2-D region (instrumentation, bof_options.verify) -----------------------------
// out[0] += sum of the region's 32-bit words; out[1] += sum of word * (logical index + 1), both modulo 2^64.
// The region is `rows` rows of `row_words` words, `pitch_words` apart.  Logical index of word (r, c):
// index_base + r * row_words + c, or -- t_pitch > 0: the region is a TRANSPOSED image of the logical object
// (the k-major copy of a panel) -- index_base + c * t_pitch + r.  The host computes the same two sums over the
// same object wherever it sits in a pinned slot or a file (flash_common.h: host_word_sums), so equal sums at two
// hand-over points mean the words arrived, all of them, each in its place.
__global__ void __launch_bounds__(256)
verify_sum_kernel(const uint32_t *__restrict__ p, int64_t rows, int64_t row_words, int64_t pitch_words,
                  uint64_t index_base, int64_t t_pitch, unsigned long long *__restrict__ out) {
  const int64_t total = rows * row_words;
  unsigned long long s1 = 0, s2 = 0;
  for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t) gridDim.x * 256) {
    const int64_t r = i / row_words, c = i - r * row_words;
    const unsigned long long w = p[r * pitch_words + c];
    const unsigned long long li = index_base + (t_pitch > 0 ? (unsigned long long) (c * t_pitch + r) : (unsigned long long) i);
    s1 += w;
    s2 += w * (li + 1ull);
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  __shared__ unsigned long long part[2][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { part[0][wave] = s1; part[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], part[0][0] + part[0][1] + part[0][2] + part[0][3]);
    atomicAdd(&out[1], part[1][0] + part[1][1] + part[1][2] + part[1][3]);
  }
}

hipError_t verify_sum(const void *p, int64_t rows, int64_t row_words, int64_t pitch_words, uint64_t index_base,
                      int64_t t_pitch, unsigned long long *out2, hipStream_t st) {
  drop_stale_error();
  if (rows <= 0 || row_words <= 0) return hipSuccess;
  const int64_t total = rows * row_words;
  const unsigned blocks = (unsigned) std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(verify_sum_kernel, dim3(blocks), dim3(256), 0, st, (const uint32_t *) p, rows, row_words, pitch_words,
                     index_base, t_pitch, out2);
  return hipGetLastError();
}

}  // namespace bof
