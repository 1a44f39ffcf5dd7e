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
//   binary search bounded to the tile's row range); the last pass writes ja_tr (int64 row
//   ids), val_tr and the sorted keys, from which ia_tr follows (no global atomics anywhere:
//   device-scope atomics execute at the memory side on MI355X and cost 37 ms per 1e9).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
// A tile is 4096 consecutive records handled by one 256-thread workgroup: wave w owns the
// w-th quarter, lane l of its c-th chunk owns record 64*c + l, so "wave, chunk, lane" order is
// the input order.  The tile is first sorted by digit inside LDS (stable), then written out
// run by run, so every global store instruction covers consecutive addresses.
constexpr int RS_T = 256, RS_WAVES = RS_T / 64, RS_CHUNKS = 16, RS_SUB = 64 * RS_CHUNKS,
              RS_TILE = RS_WAVES * RS_SUB;
// Digits of at most 8 bits.  10-bit digits (two passes instead of three at 1M columns) were built
// and measured: 41.3 ms against 28.5 ms for the cfg3 matrix -- a 4096-record tile then holds 4
// records per digit, the scatter's runs shrink to 16 bytes and both passes take 14-17 ms instead of
// 8-9, and the 1024-entry LDS tables halve the occupancy of the record-only (GEMV) passes.
constexpr int RS_MAXBITS = 8, RS_MAXD = 1 << RS_MAXBITS;
#ifndef RS_OCC
#define RS_OCC 4
#endif

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8), each with its own
// L2.  Tile t appends its run of every digit right behind tile t-1's, so when consecutive TILES run
// on the same XCD at about the same time their short runs (32 records per digit and tile on
// average at 7 bits) meet in that L2 and leave it as full lines.  tile_of() gives XCD x the
// contiguous tile range [x*per, (x+1)*per).
__device__ inline int64_t tile_of(int64_t b, int64_t nblocks, int xcd_order) {
  if (!xcd_order) return b;
  const int64_t per = (nblocks + 7) / 8;
  return (b % 8) * per + b / 8;   // may be >= nblocks for the last XCD: caller skips
}

struct SortArgs {
  // source: pass 0 reads the CSR arrays, later passes the (key,row,val) records
  const int64_t *col;   // pass 0
  const int64_t *ptr;   // pass 0 (m + 1 offsets, any base)
  const float *val_in;  // pass 0: CSR values; later: record values
  const uint32_t *key_in, *row_in;
  // destination: records; the last pass writes the transposed CSR arrays + the sorted keys
  uint32_t *key_out, *row_out;
  float *val_out;
  int64_t *col_tr;
  int64_t m, nnz;
  int shift, wbits;
  int64_t nblocks;
  const int64_t *tile_row;  // pass 0: row holding the first record of every tile (nblocks + 1)
  const float *x;           // GEMV mode, pass 0: the record value becomes val * x[row]
  int xcd_order;            // 1: tiles are dealt to XCDs in contiguous ranges (tile_of)
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

// tile_row[t] = row of record min(t * RS_TILE, nnz - 1): one full-depth search per tile here
// instead of a serial one at the head of every scatter workgroup
__global__ __launch_bounds__(256) void tile_rows_kernel(const int64_t *__restrict__ ptr, int64_t m,
                                                        int64_t nnz, int64_t nblocks,
                                                        int64_t *__restrict__ tile_row) {
  const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (t > nblocks) return;
  const int64_t p = t * RS_TILE < nnz - 1 ? t * RS_TILE : nnz - 1;
  tile_row[t] = row_of(ptr, ptr[0], p, 0, m - 1);
}

template <bool FIRST>
__global__ __launch_bounds__(RS_T) void radix_hist_kernel(SortArgs a, uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[RS_MAXD];
  const uint32_t mask = (1u << a.wbits) - 1u;
  for (uint32_t d = threadIdx.x; d <= mask; d += RS_T) h[d] = 0;
  __syncthreads();
  // same dealing of tiles to XCDs as the scatter: the counters of neighbouring tiles are
  // neighbouring dwords of the digit-major table, and only meet as full lines inside one L2
  // (written from all 8 XCDs the 125 MB table cost 1.3 GB of partial-line writes per pass)
  const int64_t tile = tile_of(blockIdx.x, a.nblocks, a.xcd_order);
  if (tile >= a.nblocks) return;
  const int64_t t0 = tile * RS_TILE;
  const int64_t t1 = t0 + RS_TILE < a.nnz ? t0 + RS_TILE : a.nnz;
  // all 16 loads of a lane in flight before the first LDS add (a load-add-load-add loop was
  // latency-bound: 1.28 ms per 1e9 keys, 3.1 TB/s)
  uint32_t key[RS_TILE / RS_T];
#pragma unroll
  for (int c = 0; c < RS_TILE / RS_T; c++) {
    const int64_t p = t0 + c * RS_T + threadIdx.x;
    key[c] = 0;
    if (p < t1) key[c] = FIRST ? (uint32_t) a.col[p] : a.key_in[p];
  }
#pragma unroll
  for (int c = 0; c < RS_TILE / RS_T; c++)
    if (t0 + c * RS_T + threadIdx.x < t1) atomicAdd(&h[(key[c] >> a.shift) & mask], 1u);
  __syncthreads();
  for (uint32_t d = threadIdx.x; d <= mask; d += RS_T) hist[(int64_t) d * a.nblocks + tile] = h[d];  // digit-major
}

// GEMV mode (the partition passes of A^T x): records are (column, product) only -- no row
// payload is carried or staged.
template <bool FIRST, bool LAST, bool GEMV>
__global__ __launch_bounds__(RS_T, RS_OCC) void radix_scatter_kernel(SortArgs a,
                                                             const int64_t *__restrict__ bases) {
  // 40 KB of LDS per workgroup = four workgroups per CU (a separate 16 KB image per record
  // field allowed two).  skey: the row offsets while the row ids are painted (first pass), then
  // the key image.  spay: the painted row ids (first pass), then the payload image -- the row ids
  // and, once those have been written out, the values (GEMV records carry no row: values only).
  __shared__ uint32_t skey[RS_TILE], spay[RS_TILE];
  __shared__ uint32_t run[RS_WAVES][RS_MAXD];  // counts, then running local positions
  __shared__ int64_t gbase[RS_MAXD];           // global start of this tile's run of a digit MINUS its local start
  __shared__ uint32_t wsum[RS_WAVES];
  uint32_t *const srow = spay;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t mask = (1u << a.wbits) - 1u;
  const int64_t tile = tile_of(blockIdx.x, a.nblocks, a.xcd_order);
  if (tile >= a.nblocks) return;
  const int64_t t0 = tile * RS_TILE;
  const int64_t t1 = t0 + RS_TILE < a.nnz ? t0 + RS_TILE : a.nnz;
  const int64_t s0 = t0 + (int64_t) w * RS_SUB;  // this wave's quarter
  for (int i = threadIdx.x; i < (RS_WAVES << a.wbits); i += RS_T) run[i >> a.wbits][i & mask] = 0;
  for (uint32_t d = threadIdx.x; d <= mask; d += RS_T) gbase[d] = bases[(int64_t) d * a.nblocks + tile];
  const int64_t z = FIRST ? a.ptr[0] : 0;
  __syncthreads();
  // phase 0/1: records into registers, digit counts per wave
  uint32_t key[RS_CHUNKS], row[RS_CHUNKS];
  float v[RS_CHUNKS];
  // rows that can own a record of this tile (the upper one may start exactly at the tile end)
  const int64_t r_lo = FIRST ? a.tile_row[tile] : 0, r_hi = FIRST ? a.tile_row[tile + 1] : 0;
#pragma unroll
  for (int c = 0; c < RS_CHUNKS; c++) {
    const int64_t p = s0 + c * 64 + lane;
    key[c] = 0; row[c] = 0; v[c] = 0.f;
    if (p < t1) {
      key[c] = FIRST ? (uint32_t) a.col[p] : a.key_in[p];
      v[c] = a.val_in[p];
      if (!FIRST && !GEMV) row[c] = a.row_in[p];
    }
  }
  // first pass: row ids.  The offsets of the tile's row range [r_lo, r_hi] are copied into LDS
  // (tile-relative, in the still unused skey area); every wave then paints the row id over the
  // positions of "its" rows in the srow area (64 positions per LDS store) and each record picks
  // its id up from there.  A range that does not fit (thousands of empty rows inside one tile)
  // is searched in global memory instead.
  const int64_t nr = r_hi - r_lo + 1;
  const bool paint = FIRST && nr < RS_TILE;
  if (paint) {
    const int cnt = (int) (t1 - t0);
    for (int64_t i = threadIdx.x; i <= nr; i += RS_T) {
      int64_t rel = a.ptr[r_lo + i] - z - t0;  // first row may start before, last end after the tile
      rel = rel < 0 ? 0 : (rel > cnt ? cnt : rel);
      skey[i] = (uint32_t) rel;
    }
    __syncthreads();
    if (nr * 16 > cnt) {
      // short rows (fewer than 16 records on average: the 10-per-row matrices of csrgemv): one
      // row per LANE -- a wave per row would walk ~100 rows one after the other, each for a
      // handful of positions (that serial walk was 40 % of the first pass at 10 records per row)
      for (int ri = threadIdx.x; ri < (int) nr; ri += RS_T) {
        const int b = (int) skey[ri], e = (int) skey[ri + 1];
        for (int q = b; q < e; q++) srow[q] = (uint32_t) (r_lo + ri);
      }
    } else {
      for (int ri = w; ri < (int) nr; ri += RS_WAVES) {
        const int b = (int) skey[ri], e = (int) skey[ri + 1];
        for (int q = b + lane; q < e; q += 64) srow[q] = (uint32_t) (r_lo + ri);
      }
    }
    __syncthreads();
  }
  if (FIRST) {
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; c++) {
      const int64_t p = s0 + c * 64 + lane;
      if (p < t1) row[c] = paint ? srow[w * RS_SUB + c * 64 + lane] : (uint32_t) row_of(a.ptr, z, p, r_lo, r_hi);
    }
  }
  if (FIRST && GEMV) {  // all 16 x loads in flight before the first product
    float xv[RS_CHUNKS];
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; c++) xv[c] = s0 + c * 64 + lane < t1 ? a.x[row[c]] : 0.f;
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; c++) v[c] *= xv[c];
  }
#pragma unroll
  for (int c = 0; c < RS_CHUNKS; c++)
    if (s0 + c * 64 + lane < t1) atomicAdd(&run[w][(key[c] >> a.shift) & mask], 1u);
  __syncthreads();
  // local layout: digit-major, wave-minor (= input order inside a digit).  Thread t owns the
  // dpt consecutive digits [t*dpt, (t+1)*dpt) (dpt = 1, 2 or 4), so the scan over threads is the
  // scan over digits.
  {
    const int dpt = (int) ((mask + 1 + RS_T - 1) / RS_T);
    uint32_t tot = 0;
    for (int j = 0; j < dpt; j++) {
      const uint32_t d = threadIdx.x * dpt + j;
      if (d <= mask)
#pragma unroll
        for (int i = 0; i < RS_WAVES; i++) tot += run[i][d];
    }
    uint32_t inc = tot;  // inclusive scan of tot over the 256 threads
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t start = inc - tot;
    for (int i = 0; i < w; i++) start += wsum[i];
    for (int j = 0; j < dpt; j++) {
      const uint32_t d = threadIdx.x * dpt + j;
      if (d <= mask) {
        gbase[d] -= (int64_t) start;  // position i of the image goes to gbase[d] + i
#pragma unroll
        for (int i = 0; i < RS_WAVES; i++) { const uint32_t cw = run[i][d]; run[i][d] = start; start += cw; }
      }
    }
  }
  __syncthreads();
  // phase 2: stable placement into the LDS image
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint32_t pos[RS_CHUNKS];
#pragma unroll
  for (int c = 0; c < RS_CHUNKS; c++) {
    const bool ok = s0 + c * 64 + lane < t1;
    const uint32_t d = (key[c] >> a.shift) & mask;
    uint64_t peers = __ballot(ok);
    for (int b = 0; b < a.wbits; b++) {
      const uint64_t vote = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? vote : ~vote;
    }
    const int rank = __popcll(peers & lt);
    pos[c] = 0;
    if (ok) pos[c] = run[w][d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (ok && rank == 0) run[w][d] += __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      skey[pos[c]] = key[c];
      spay[pos[c]] = GEMV ? __float_as_uint(v[c]) : row[c];
    }
  }
  __syncthreads();
  // phase 3: run-by-run write-out, consecutive threads -> consecutive addresses
  const int cnt = (int) (t1 - t0);
  int64_t g[RS_CHUNKS];
#pragma unroll
  for (int j = 0; j < RS_CHUNKS; j++) {
    const int i = threadIdx.x + j * RS_T;
    g[j] = 0;
    if (i < cnt) {
      const uint32_t k = skey[i];
      g[j] = gbase[(k >> a.shift) & mask] + i;
      a.key_out[g[j]] = k;  // last pass: the sorted keys, from which the offsets are derived
      if (GEMV) a.val_out[g[j]] = __uint_as_float(spay[i]);
      else if (LAST) a.col_tr[g[j]] = (int64_t) spay[i];
      else a.row_out[g[j]] = spay[i];
    }
  }
  if (!GEMV) {  // the values take the place of the row ids
    __syncthreads();
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; c++)
      if (s0 + c * 64 + lane < t1) spay[pos[c]] = __float_as_uint(v[c]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_CHUNKS; j++) {
      const int i = threadIdx.x + j * RS_T;
      if (i < cnt) a.val_out[g[j]] = __uint_as_float(spay[i]);
    }
  }
}

template <bool FIRST, bool LAST, bool GEMV = false>
hipError_t scatter_launch(const SortArgs &a, const int64_t *bases, hipStream_t st) {
  const int64_t grid = a.xcd_order ? (a.nblocks + 7) / 8 * 8 : a.nblocks;
  radix_scatter_kernel<FIRST, LAST, GEMV><<<(unsigned) grid, RS_T, 0, st>>>(a, bases);
  return hipGetLastError();
}

// offsets of the transpose from the sorted keys: one lower_bound per output row.  (A linear
// boundary scan over the keys was tried first: a run of empty rows -- the reference generator
// leaves the top 29 % of the columns empty -- serialises in one thread there: 586 ms.)
__global__ __launch_bounds__(256) void offsets_by_search_kernel(const uint32_t *__restrict__ key,
                                                                 int64_t nnz, int64_t n,
                                                                 int64_t *__restrict__ ptr_tr,
                                                                 int kshift = 0) {
  const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (c > n) return;
  int64_t lo = 0, hi = nnz;  // first i with (key[i] >> kshift) >= c
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t) (key[mid] >> kshift) < c) lo = mid + 1;
    else hi = mid;
  }
  ptr_tr[c] = lo;
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// digit width of the transposition sort: at most RS_MAXBITS (BOF_SORT_BITS narrows it for
// experiments)
inline int sort_digit_bits() {
  const int v = getenv("BOF_SORT_BITS") ? atoi(getenv("BOF_SORT_BITS")) : 8;
  return v < 4 ? 4 : (v > RS_MAXBITS ? RS_MAXBITS : v);
}

inline int sort_xcd_order() {
  const int v = getenv("BOF_SORT_XCD") ? atoi(getenv("BOF_SORT_XCD")) : 1;
  return v;
}

struct Layout {
  int passes, wbits;
  int64_t nblocks;
  size_t off_hist, off_bases, off_scan, off_keys, off_tilerow, off_rec[2], total;
};

Layout make_layout(int64_t n, int64_t nnz) {
  Layout L{};
  int bits = 1;
  while (bits < 32 && ((int64_t) 1 << bits) < n) bits++;
  const int per = sort_digit_bits();
  L.passes = (bits + per - 1) / per;
  L.wbits = (bits + L.passes - 1) / L.passes;
  L.nblocks = (nnz + RS_TILE - 1) / RS_TILE;
  const size_t nh = ((size_t) 1 << L.wbits) * (size_t) (L.nblocks > 0 ? L.nblocks : 1);
  size_t o = 0;
  L.off_hist = o;   o += align256(nh * 4);
  L.off_bases = o;  o += align256(nh * 8);
  L.off_scan = o;   o += align256(scan_tmp_elems((int64_t) nh) * 8);
  L.off_keys = o;   o += align256((size_t) nnz * 4);
  L.off_tilerow = o; o += align256((size_t) (L.nblocks + 1) * 8);
  const int n_rec = L.passes == 1 ? 0 : (L.passes == 2 ? 1 : 2);
  for (int i = 0; i < 2; i++) {
    L.off_rec[i] = o;
    if (i < n_rec) o += 3 * align256((size_t) nnz * 4);
  }
  L.total = o;
  return L;
}

// ---------------------------------------------------------------------------------------
// y = A^T x without global atomics: partition the products by column bin, then sum per bin
// ---------------------------------------------------------------------------------------
// Bins of GT_W consecutive columns.  The radix passes above sort the (column, val * x[row])
// records by BIN only (the bits above log2(GT_W)); one workgroup per bin then adds its records
// into an LDS image of its slice of y and stores the slice.  Replaces one device-scope
// `global_atomic_add_f32` per non-zero (memory-side, ~20 G/s on scattered columns).
constexpr int GT_LOGW = 13, GT_W = 1 << GT_LOGW;

// LDS float add as a compare-and-swap loop.  `ds_add_f32` runs at 0.8 lane-operations per ns and
// CU on gfx950 whatever the addresses (203 G/s chip-wide; `ds_add_u32` 10.6, this loop 5.5 on
// random addresses: tools/exp/lds_atomic_bench.hip) and was the whole cost of the per-bin sums.
__device__ inline void lds_add_f32(float *p, float v) {
  uint32_t *q = reinterpret_cast<uint32_t *>(p);
  uint32_t old = *q, assumed;
  do {
    assumed = old;
    old = atomicCAS(q, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (old != assumed);
}

__global__ __launch_bounds__(256) void gemv_t_accumulate_kernel(const uint32_t *__restrict__ key,
                                                                const float *__restrict__ prod,
                                                                const int64_t *__restrict__ bin_off,
                                                                int64_t n, float *__restrict__ y) {
  __shared__ float ys[GT_W];
  for (int i = threadIdx.x; i < GT_W; i += 256) ys[i] = 0.f;
  __syncthreads();
  const int64_t b = blockIdx.x, s = bin_off[b], e = bin_off[b + 1];
  // head up to the first 16-byte boundary, then four consecutive records per lane and load
  // (dwordx4, two of each array in flight), then the tail
  const int64_t a0 = (s + 3) & ~(int64_t) 3, a1 = e & ~(int64_t) 3;
  if (a0 < a1) {
    for (int64_t i = s + threadIdx.x; i < a0; i += 256) lds_add_f32(&ys[key[i] & (GT_W - 1)], prod[i]);
    const uint4 *k4 = reinterpret_cast<const uint4 *>(key);
    const float4 *p4 = reinterpret_cast<const float4 *>(prod);
    int64_t q = a0 / 4 + threadIdx.x;
    const int64_t q1 = a1 / 4;
    for (; q + 256 < q1; q += 512) {
      const uint4 ka = k4[q], kb = k4[q + 256];
      const float4 pa = p4[q], pb = p4[q + 256];
      lds_add_f32(&ys[ka.x & (GT_W - 1)], pa.x); lds_add_f32(&ys[ka.y & (GT_W - 1)], pa.y);
      lds_add_f32(&ys[ka.z & (GT_W - 1)], pa.z); lds_add_f32(&ys[ka.w & (GT_W - 1)], pa.w);
      lds_add_f32(&ys[kb.x & (GT_W - 1)], pb.x); lds_add_f32(&ys[kb.y & (GT_W - 1)], pb.y);
      lds_add_f32(&ys[kb.z & (GT_W - 1)], pb.z); lds_add_f32(&ys[kb.w & (GT_W - 1)], pb.w);
    }
    for (; q < q1; q += 256) {
      const uint4 ka = k4[q];
      const float4 pa = p4[q];
      lds_add_f32(&ys[ka.x & (GT_W - 1)], pa.x); lds_add_f32(&ys[ka.y & (GT_W - 1)], pa.y);
      lds_add_f32(&ys[ka.z & (GT_W - 1)], pa.z); lds_add_f32(&ys[ka.w & (GT_W - 1)], pa.w);
    }
    for (int64_t i = a1 + threadIdx.x; i < e; i += 256) lds_add_f32(&ys[key[i] & (GT_W - 1)], prod[i]);
  } else {
    for (int64_t i = s + threadIdx.x; i < e; i += 256) lds_add_f32(&ys[key[i] & (GT_W - 1)], prod[i]);
  }
  __syncthreads();
  const int64_t c0 = b * GT_W;
  for (int i = threadIdx.x; i < GT_W && c0 + i < n; i += 256) y[c0 + i] = ys[i];
}

struct GemvLayout {
  int passes, wbits;
  int64_t nblocks, nbins;
  size_t off_hist, off_bases, off_scan, off_tilerow, off_binoff, off_rec[2], total;
};

GemvLayout make_gemv_layout(int64_t n, int64_t nnz) {
  GemvLayout L{};
  L.nbins = (n + GT_W - 1) / GT_W;
  int bits = 1;
  while (bits < 32 && ((int64_t) 1 << bits) < L.nbins) bits++;
  L.passes = (bits + 7) / 8;
  L.wbits = (bits + L.passes - 1) / L.passes;
  L.nblocks = (nnz + RS_TILE - 1) / RS_TILE;
  const size_t nh = ((size_t) 1 << L.wbits) * (size_t) (L.nblocks > 0 ? L.nblocks : 1);
  size_t o = 0;
  L.off_hist = o;    o += align256(nh * 4);
  L.off_bases = o;   o += align256(nh * 8);
  L.off_scan = o;    o += align256(scan_tmp_elems((int64_t) nh) * 8);
  L.off_tilerow = o; o += align256((size_t) (L.nblocks + 1) * 8);
  L.off_binoff = o;  o += align256((size_t) (L.nbins + 1) * 8);
  for (int i = 0; i < 2; i++) {
    L.off_rec[i] = o;
    if (i == 0 || L.passes > 1) o += 2 * align256((size_t) nnz * 4);
  }
  L.total = o;
  return L;
}

}  // namespace

// Column-block merge of the out-of-core transposition (reference BlockMergeTask,
// include/tasks/csrcsc_task.h:93-163): for every output row c of the block, the runs that the
// row blocks b = 0..nb-1 hold for it are concatenated in block order (= ascending source rows);
// block-local row ids get the block's first row added.  One wave per output row.
__global__ __launch_bounds__(256) void csc_merge_kernel(int nb, int64_t cw, const int64_t *__restrict__ blk_ptr,
                                                        const int64_t *__restrict__ seg_base,
                                                        const int64_t *__restrict__ row0,
                                                        const int64_t *__restrict__ out_ptr,
                                                        const float *__restrict__ val_in,
                                                        const int64_t *__restrict__ col_in,
                                                        float *__restrict__ val_out, int64_t *__restrict__ col_out) {
  const int64_t c = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= cw) return;
  const int lane = threadIdx.x & 63;
  int64_t cursor = out_ptr[c];
  for (int b = 0; b < nb; b++) {
    const int64_t s = blk_ptr[(int64_t) b * (cw + 1) + c], e = blk_ptr[(int64_t) b * (cw + 1) + c + 1];
    const int64_t src = seg_base[b], r0 = row0[b];
    for (int64_t i = s + lane; i < e; i += 64) {
      val_out[cursor + (i - s)] = val_in[src + i];
      col_out[cursor + (i - s)] = col_in[src + i] + r0;
    }
    cursor += e - s;
  }
}

hipError_t csc_merge(int nb, int64_t cw, const int64_t *blk_ptr, const int64_t *seg_base, const int64_t *row0,
                     const int64_t *out_ptr, const float *val_in, const int64_t *col_in, float *val_out,
                     int64_t *col_out, hipStream_t st) {
  if (cw <= 0 || nb <= 0) return hipSuccess;
  csc_merge_kernel<<<(unsigned) ((cw + 3) / 4), 256, 0, st>>>(nb, cw, blk_ptr, seg_base, row0, out_ptr, val_in, col_in,
                                                              val_out, col_out);
  return hipGetLastError();
}

size_t csrgemv_t_workspace_bytes(int64_t n, int64_t nnz) { return make_gemv_layout(n, nnz).total; }

// y[0..n) = A^T x for A = CSR(val, ptr[m+1] (any base), col), m x n; y is overwritten.
hipError_t scsrgemv_t_partitioned(int64_t m, int64_t n, int64_t nnz, const float *val, const int64_t *ptr,
                                  const int64_t *col, const float *x, float *y, void *workspace,
                                  hipStream_t st) {
  drop_stale_error();
  hipError_t e;
  if (n <= 0) return hipSuccess;
  if (m <= 0 || nnz <= 0) return hipMemsetAsync(y, 0, (size_t) n * 4, st);
  const GemvLayout L = make_gemv_layout(n, nnz);
  char *ws = (char *) workspace;
  uint32_t *hist = (uint32_t *) (ws + L.off_hist);
  int64_t *bases = (int64_t *) (ws + L.off_bases);
  int64_t *scan_tmp = (int64_t *) (ws + L.off_scan);
  int64_t *tile_row = (int64_t *) (ws + L.off_tilerow);
  int64_t *bin_off = (int64_t *) (ws + L.off_binoff);
  const size_t rec = align256((size_t) nnz * 4);
  tile_rows_kernel<<<(unsigned) ((L.nblocks + 1 + 255) / 256), 256, 0, st>>>(ptr, m, nnz, L.nblocks, tile_row);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  SortArgs a{};
  a.xcd_order = sort_xcd_order();
  a.tile_row = tile_row; a.x = x;
  a.col = col; a.ptr = ptr; a.m = m; a.nnz = nnz; a.nblocks = L.nblocks; a.wbits = L.wbits;
  const char *sorted = nullptr;
  for (int pass = 0; pass < L.passes; pass++) {
    const bool first = pass == 0;
    a.shift = GT_LOGW + pass * L.wbits;
    if (first) {
      a.val_in = val; a.key_in = nullptr;
    } else {
      const char *src = ws + L.off_rec[(pass - 1) & 1];
      a.key_in = (const uint32_t *) src;
      a.val_in = (const float *) (src + rec);
    }
    char *dst = ws + L.off_rec[pass & 1];
    a.key_out = (uint32_t *) dst;
    a.val_out = (float *) (dst + rec);
    sorted = dst;
    const unsigned hgrid = (unsigned) (a.xcd_order ? (L.nblocks + 7) / 8 * 8 : L.nblocks);
    if (first) radix_hist_kernel<true><<<hgrid, RS_T, 0, st>>>(a, hist);
    else radix_hist_kernel<false><<<hgrid, RS_T, 0, st>>>(a, hist);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = exclusive_scan<uint32_t>(hist, bases, ((int64_t) 1 << L.wbits) * L.nblocks, scan_tmp, st);
    if (e != hipSuccess) return e;
    e = first ? scatter_launch<true, false, true>(a, bases, st) : scatter_launch<false, false, true>(a, bases, st);
    if (e != hipSuccess) return e;
  }
  offsets_by_search_kernel<<<(unsigned) ((L.nbins + 1 + 255) / 256), 256, 0, st>>>(
      (const uint32_t *) sorted, nnz, L.nbins, bin_off, GT_LOGW);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  gemv_t_accumulate_kernel<<<(unsigned) L.nbins, 256, 0, st>>>((const uint32_t *) sorted,
                                                               (const float *) (sorted + rec), bin_off, n, y);
  return hipGetLastError();
}

size_t csrcsc_workspace_bytes(int64_t n, int64_t nnz) { return make_layout(n, nnz).total; }

hipError_t scsrcsc(int64_t m, int64_t n, int64_t nnz, const float *val, const int64_t *ptr,
                   const int64_t *col, float *val_tr, int64_t *ptr_tr, int64_t *col_tr,
                   void *workspace, hipStream_t st) {
  drop_stale_error();
  hipError_t e;
  if (m <= 0 || nnz <= 0) return hipMemsetAsync(ptr_tr, 0, (size_t) (n + 1) * 8, st);
  const Layout L = make_layout(n, nnz);
  char *ws = (char *) workspace;
  uint32_t *hist = (uint32_t *) (ws + L.off_hist);
  int64_t *bases = (int64_t *) (ws + L.off_bases);
  int64_t *scan_tmp = (int64_t *) (ws + L.off_scan);
  uint32_t *keys_sorted = (uint32_t *) (ws + L.off_keys);
  const size_t rec = align256((size_t) nnz * 4);

  int64_t *tile_row = (int64_t *) (ws + L.off_tilerow);
  tile_rows_kernel<<<(unsigned) ((L.nblocks + 1 + 255) / 256), 256, 0, st>>>(ptr, m, nnz, L.nblocks, tile_row);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  SortArgs a{};
  a.xcd_order = sort_xcd_order();
  a.tile_row = tile_row;
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
      a.key_out = keys_sorted; a.row_out = nullptr; a.val_out = val_tr;
    } else {
      char *dst = ws + L.off_rec[pass & 1];
      a.key_out = (uint32_t *) dst;
      a.row_out = (uint32_t *) (dst + rec);
      a.val_out = (float *) (dst + 2 * rec);
    }
    const unsigned hgrid = (unsigned) (a.xcd_order ? (L.nblocks + 7) / 8 * 8 : L.nblocks);
    if (first) radix_hist_kernel<true><<<hgrid, RS_T, 0, st>>>(a, hist);
    else radix_hist_kernel<false><<<hgrid, RS_T, 0, st>>>(a, hist);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = exclusive_scan<uint32_t>(hist, bases, ((int64_t) 1 << L.wbits) * L.nblocks, scan_tmp, st);
    if (e != hipSuccess) return e;
    if (first && last) e = scatter_launch<true, true>(a, bases, st);
    else if (first) e = scatter_launch<true, false>(a, bases, st);
    else if (last) e = scatter_launch<false, true>(a, bases, st);
    else e = scatter_launch<false, false>(a, bases, st);
    if (e != hipSuccess) return e;
  }
  offsets_by_search_kernel<<<(unsigned) ((n + 1 + 255) / 256), 256, 0, st>>>(keys_sorted, nnz, n, ptr_tr);
  return hipGetLastError();
}

}  // namespace bof
