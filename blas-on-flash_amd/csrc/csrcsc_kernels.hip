// csrcsc_kernels.hip -- CSR -> CSR-of-the-transpose on the GPU (gfx950).
//
// What it replaces: mkl_csrcsc(job = {0,0,0,-1,-1,1}) as called per row block by the
// reference (include/tasks/csrcsc_task.h:66-75) plus the column-block merge of
// src/blas/csrcsc.cpp:100-146.  Both together are a STABLE transposition: inside every
// output row the source rows appear in ascending order.  Here the whole matrix sits in HBM,
// so there are no row blocks and no merge: the non-zeros are sorted by column with a stable
// least-significant-digit radix sort (8-bit digits, ceil(log2(n)/8) passes) whose payload is
// (source row, value).  Pure index / byte moving work, bound by HBM; bit-exact by construction.
//
//   pass structure (per digit):  radix_hist_kernel  -> per-tile digit histograms
//                                exclusive_scan     -> global base of every (digit, tile)
//                                radix_scatter_kernel -> stable placement
//   the first pass reads the CSR arrays directly (row ids recovered from the offsets by a
//   binary search bounded to the tile's row range) and also counts the columns (-> ia_tr);
//   the last pass writes ja_tr (int64 row ids) and val_tr directly.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bof_internal.h"

namespace bof {
namespace {

// ---------------------------------------------------------------------------------------
// exclusive scan (uint32 or int64 input, int64 output), chunks of 4096 per workgroup
// ---------------------------------------------------------------------------------------
constexpr int SCAN_T = 256, SCAN_IPT = 16, SCAN_CH = SCAN_T * SCAN_IPT;

__device__ inline int64_t wave_sum(int64_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;  // valid in lane 0
}

template <typename InT>
__global__ __launch_bounds__(SCAN_T) void scan_reduce_kernel(const InT *__restrict__ in, int64_t n,
                                                             int64_t *__restrict__ totals) {
  __shared__ int64_t part[SCAN_T / 64];
  const int64_t base = (int64_t) blockIdx.x * SCAN_CH;
  int64_t s = 0;
  for (int i = threadIdx.x; i < SCAN_CH; i += SCAN_T) {
    const int64_t p = base + i;
    if (p < n) s += (int64_t) in[p];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t t = 0;
    for (int w = 0; w < SCAN_T / 64; w++) t += part[w];
    totals[blockIdx.x] = t;
  }
}

// out[i] = offsets[block] + sum(in[chunk start .. i)); safe in place (in == out for int64)
template <typename InT>
__global__ __launch_bounds__(SCAN_T) void scan_apply_kernel(const InT *in, int64_t n,
                                                            const int64_t *__restrict__ offsets,
                                                            int64_t *out) {
  __shared__ int64_t wtot[SCAN_T / 64];
  const int64_t first = (int64_t) blockIdx.x * SCAN_CH + (int64_t) threadIdx.x * SCAN_IPT;
  int64_t v[SCAN_IPT];
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_IPT; i++) {
    v[i] = (first + i < n) ? (int64_t) in[first + i] : 0;
    s += v[i];
  }
  // inclusive scan of the thread sums inside the wave, then across the 4 waves
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t inc = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  int64_t run = offsets ? offsets[blockIdx.x] : 0;
  for (int i = 0; i < w; i++) run += wtot[i];
  run += inc - s;
#pragma unroll
  for (int i = 0; i < SCAN_IPT; i++) {
    if (first + i < n) out[first + i] = run;
    run += v[i];
  }
}

template <typename InT>
hipError_t exclusive_scan(const InT *in, int64_t *out, int64_t n, int64_t *tmp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const int64_t nch = (n + SCAN_CH - 1) / SCAN_CH;
  if (nch == 1) {
    scan_apply_kernel<InT><<<1, SCAN_T, 0, st>>>(in, n, nullptr, out);
    return hipGetLastError();
  }
  scan_reduce_kernel<InT><<<(unsigned) nch, SCAN_T, 0, st>>>(in, n, tmp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = exclusive_scan<int64_t>(tmp, tmp, nch, tmp + nch, st);
  if (e != hipSuccess) return e;
  scan_apply_kernel<InT><<<(unsigned) nch, SCAN_T, 0, st>>>(in, n, tmp, out);
  return hipGetLastError();
}

inline size_t scan_tmp_elems(int64_t n) {  // int64 elements needed by exclusive_scan(n)
  size_t tot = 0;
  while (n > SCAN_CH) {
    n = (n + SCAN_CH - 1) / SCAN_CH;
    tot += (size_t) n;
  }
  return tot + 8;
}

// ---------------------------------------------------------------------------------------
// stable LSD radix sort of the non-zeros by column
// ---------------------------------------------------------------------------------------
constexpr int RS_T = 256, RS_WAVES = RS_T / 64, RS_SUB = 4096, RS_TILE = RS_WAVES * RS_SUB;

struct SortArgs {
  // source: pass 0 reads the CSR arrays, later passes the (key,row,val) records
  const int64_t *col;   // pass 0
  const int64_t *ptr;   // pass 0 (m + 1 offsets, any base)
  const float *val_in;  // pass 0: CSR values; later: record values
  const uint32_t *key_in, *row_in;
  // destination: records, or (last pass) the transposed CSR arrays
  uint32_t *key_out, *row_out;
  float *val_out;
  int64_t *col_tr;
  int64_t m, nnz;
  int shift, wbits;
  int64_t nblocks;
};

// row of position p (0-based among the non-zeros): largest r in [lo, hi] with ptr[r] - ptr[0] <= p
__device__ inline int64_t row_of(const int64_t *__restrict__ ptr, int64_t z, int64_t p, int64_t lo,
                                 int64_t hi) {
  const int64_t want = z + p;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (ptr[mid] <= want) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

template <bool FIRST>
__global__ __launch_bounds__(RS_T) void radix_hist_kernel(SortArgs a, uint32_t *__restrict__ hist,
                                                          uint32_t *__restrict__ colcnt) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t t0 = (int64_t) blockIdx.x * RS_TILE;
  const int64_t t1 = t0 + RS_TILE < a.nnz ? t0 + RS_TILE : a.nnz;
  const uint32_t mask = (1u << a.wbits) - 1u;
  for (int64_t p = t0 + threadIdx.x; p < t1; p += RS_T) {
    uint32_t key;
    if (FIRST) {
      key = (uint32_t) a.col[p];
      atomicAdd(&colcnt[key], 1u);  // column populations -> ia_tr
    } else {
      key = a.key_in[p];
    }
    atomicAdd(&h[(key >> a.shift) & mask], 1u);
  }
  __syncthreads();
  hist[(int64_t) threadIdx.x * a.nblocks + blockIdx.x] = h[threadIdx.x];  // digit-major
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(RS_T) void radix_scatter_kernel(SortArgs a,
                                                             const int64_t *__restrict__ bases) {
  __shared__ uint32_t cnt[RS_WAVES][256];
  __shared__ int64_t run[RS_WAVES][256];
  __shared__ int64_t rng[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t mask = (1u << a.wbits) - 1u;
  const int64_t t0 = (int64_t) blockIdx.x * RS_TILE;
  const int64_t t1 = t0 + RS_TILE < a.nnz ? t0 + RS_TILE : a.nnz;
  const int64_t s0 = t0 + (int64_t) w * RS_SUB;                 // this wave's sub-tile
  const int64_t s1 = s0 + RS_SUB < t1 ? s0 + RS_SUB : t1;
  for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_T) (&cnt[0][0])[i] = 0;
  int64_t z = 0;
  if (FIRST) {
    z = a.ptr[0];
    if (threadIdx.x < 2) {
      const int64_t p = threadIdx.x == 0 ? t0 : t1 - 1;
      rng[threadIdx.x] = row_of(a.ptr, z, p, 0, a.m - 1);
    }
  }
  __syncthreads();
  // phase 1: digit counts of every wave's sub-tile
  for (int64_t p = s0 + lane; p < s1; p += 64) {
    const uint32_t key = FIRST ? (uint32_t) a.col[p] : a.key_in[p];
    atomicAdd(&cnt[w][(key >> a.shift) & mask], 1u);
  }
  __syncthreads();
  {
    const int d = threadIdx.x;  // one digit per thread: running base of each wave
    int64_t base = bases[(int64_t) d * a.nblocks + blockIdx.x];
    for (int i = 0; i < RS_WAVES; i++) {
      run[i][d] = base;
      base += cnt[i][d];
    }
  }
  __syncthreads();
  const int64_t r_lo = FIRST ? rng[0] : 0, r_hi = FIRST ? rng[1] : 0;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // phase 2: chunks of 64 in order; inside a chunk the lane order is the element order
  for (int64_t c0 = s0; c0 < s1; c0 += 64) {
    const int64_t p = c0 + lane;
    const bool ok = p < s1;
    uint32_t key = 0, row = 0;
    float v = 0.f;
    if (ok) {
      if (FIRST) {
        key = (uint32_t) a.col[p];
        row = (uint32_t) row_of(a.ptr, z, p, r_lo, r_hi);
      } else {
        key = a.key_in[p];
        row = a.row_in[p];
      }
      v = a.val_in[p];
    }
    const uint32_t d = (key >> a.shift) & mask;
    uint64_t peers = __ballot(ok);
    for (int b = 0; b < a.wbits; b++) {
      const uint64_t vote = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? vote : ~vote;
    }
    const int rank = __popcll(peers & lt);
    int64_t pos = 0;
    if (ok) pos = run[w][d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (ok && rank == 0) run[w][d] += __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      if (LAST) {
        a.col_tr[pos] = (int64_t) row;
        a.val_out[pos] = v;
      } else {
        a.key_out[pos] = key;
        a.row_out[pos] = row;
        a.val_out[pos] = v;
      }
    }
  }
}

template <bool FIRST, bool LAST>
hipError_t scatter_launch(const SortArgs &a, const int64_t *bases, hipStream_t st) {
  radix_scatter_kernel<FIRST, LAST><<<(unsigned) a.nblocks, RS_T, 0, st>>>(a, bases);
  return hipGetLastError();
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct Layout {
  int passes, wbits;
  int64_t nblocks;
  size_t off_hist, off_bases, off_colcnt, off_scan, off_rec[2], total;
};

Layout make_layout(int64_t n, int64_t nnz) {
  Layout L{};
  int bits = 1;
  while (bits < 32 && ((int64_t) 1 << bits) < n) bits++;
  L.passes = (bits + 7) / 8;
  L.wbits = (bits + L.passes - 1) / L.passes;
  L.nblocks = (nnz + RS_TILE - 1) / RS_TILE;
  const size_t nh = (size_t) 256 * (size_t) (L.nblocks > 0 ? L.nblocks : 1);
  size_t o = 0;
  L.off_hist = o;   o += align256(nh * 4);
  L.off_bases = o;  o += align256(nh * 8);
  L.off_colcnt = o; o += align256((size_t) (n + 1) * 4);
  const size_t scan_n = nh > (size_t) (n + 1) ? nh : (size_t) (n + 1);
  L.off_scan = o;   o += align256(scan_tmp_elems((int64_t) scan_n) * 8);
  const int n_rec = L.passes == 1 ? 0 : (L.passes == 2 ? 1 : 2);
  for (int i = 0; i < 2; i++) {
    L.off_rec[i] = o;
    if (i < n_rec) o += 3 * align256((size_t) nnz * 4);
  }
  L.total = o;
  return L;
}

}  // namespace

size_t csrcsc_workspace_bytes(int64_t n, int64_t nnz) { return make_layout(n, nnz).total; }

hipError_t scsrcsc(int64_t m, int64_t n, int64_t nnz, const float *val, const int64_t *ptr,
                   const int64_t *col, float *val_tr, int64_t *ptr_tr, int64_t *col_tr,
                   void *workspace, hipStream_t st) {
  hipError_t e;
  if (m <= 0 || nnz <= 0) return hipMemsetAsync(ptr_tr, 0, (size_t) (n + 1) * 8, st);
  const Layout L = make_layout(n, nnz);
  char *ws = (char *) workspace;
  uint32_t *hist = (uint32_t *) (ws + L.off_hist);
  int64_t *bases = (int64_t *) (ws + L.off_bases);
  uint32_t *colcnt = (uint32_t *) (ws + L.off_colcnt);
  int64_t *scan_tmp = (int64_t *) (ws + L.off_scan);
  const size_t rec = align256((size_t) nnz * 4);
  e = hipMemsetAsync(colcnt, 0, (size_t) (n + 1) * 4, st);
  if (e != hipSuccess) return e;

  SortArgs a{};
  a.col = col; a.ptr = ptr; a.m = m; a.nnz = nnz; a.nblocks = L.nblocks; a.wbits = L.wbits;
  a.col_tr = col_tr;
  for (int pass = 0; pass < L.passes; pass++) {
    const bool first = pass == 0, last = pass == L.passes - 1;
    a.shift = pass * L.wbits;
    if (first) {
      a.val_in = val; a.key_in = nullptr; a.row_in = nullptr;
    } else {
      char *src = ws + L.off_rec[(pass - 1) & 1];
      a.key_in = (const uint32_t *) src;
      a.row_in = (const uint32_t *) (src + rec);
      a.val_in = (const float *) (src + 2 * rec);
    }
    if (last) {
      a.key_out = nullptr; a.row_out = nullptr; a.val_out = val_tr;
    } else {
      char *dst = ws + L.off_rec[pass & 1];
      a.key_out = (uint32_t *) dst;
      a.row_out = (uint32_t *) (dst + rec);
      a.val_out = (float *) (dst + 2 * rec);
    }
    if (first) radix_hist_kernel<true><<<(unsigned) L.nblocks, RS_T, 0, st>>>(a, hist, colcnt);
    else radix_hist_kernel<false><<<(unsigned) L.nblocks, RS_T, 0, st>>>(a, hist, colcnt);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = exclusive_scan<uint32_t>(hist, bases, 256 * L.nblocks, scan_tmp, st);
    if (e != hipSuccess) return e;
    if (first && last) e = scatter_launch<true, true>(a, bases, st);
    else if (first) e = scatter_launch<true, false>(a, bases, st);
    else if (last) e = scatter_launch<false, true>(a, bases, st);
    else e = scatter_launch<false, false>(a, bases, st);
    if (e != hipSuccess) return e;
  }
  // ia_tr = exclusive scan of the column populations (colcnt[n] = 0 -> ia_tr[n] = nnz)
  return exclusive_scan<uint32_t>(colcnt, ptr_tr, n + 1, scan_tmp, st);
}

}  // namespace bof
