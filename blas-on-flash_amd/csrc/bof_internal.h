// bof_internal.h -- declarations shared by the translation units of libbof_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>

#include "bof_hip.h"

namespace bof {

// ---- error plumbing -----------------------------------------------------------
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);  // records message, returns BOF_EHIP/BOF_ENODEV
#define BOF_HIP_TRY(expr)                                             \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) return ::bof::hip_fail(_e, #expr);          \
  } while (0)
// hipGetLastError() after a launch returns the thread's last unconsumed error, whichever call left it
// there -- a cleanup call whose status was dropped, perhaps in an earlier library call on this thread.
// The launch wrappers drop such leftovers first, so the status they return is the launch's own.
inline void drop_stale_error() { (void) hipGetLastError(); }

// ---- kernels (gemm_f32_mfma.hip, csr_kernels.hip, gen_kernels.hip) -------------
hipError_t sgemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha,
                 const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                 int64_t ldc, hipStream_t st);
// sgemm + the two rank-1 updates of KMeansTask::execute fused into the store (u by C row, v by C column)
hipError_t sgemm_rank1x2(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha,
                         const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                         int64_t ldc, const float *u1, const float *v1, const float *u2, const float *v2,
                         hipStream_t st);
// One k-range of an ACCUMULATE CHAIN (gemm_f32_mfma.hip, ChainEpi): the chain carries its raw fp32 accumulators from
// launch to launch -- raw_out stores them unscaled into c, acc_in starts the next range from them -- and only the final
// launch (raw_out = false) applies  c = beta == 0 ? alpha*acc : fmaf(alpha, acc, beta*c_in)  (c_in == nullptr: c itself).
// A chain cut at any k positions therefore equals ONE sgemm over the whole K bit for bit.
struct GemmChain {
  const float *acc_in = nullptr;   // laid out like C, leading dimension ld_acc
  int64_t ld_acc = 0;
  const float *c_in = nullptr;     // laid out like C, leading dimension ld_cin
  int64_t ld_cin = 0;
  bool raw_out = false;
};
hipError_t sgemm_chain(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k, float alpha, const float *a,
                       int64_t lda, const float *b, int64_t ldb, float beta, float *c, int64_t ldc, const GemmChain &ch,
                       hipStream_t st);
// BOF_VERIFY: spot check of one launch (gemm_f32_mfma.hip): 64 sampled outputs recomputed in the kernels' arithmetic.
// The arguments are the launch's own (sgemm / sgemm_chain / sgemm_rank1x2 terms); capture runs in FRONT of the launch
// (it saves the values the launch overwrites), check behind it; exp2 / got2: two 64-bit sums each (Verify entries).
struct SpotArgs {
  char ord, ta, tb;
  int64_t m, n, k;
  float alpha;
  const float *a;
  int64_t lda;
  const float *b;
  int64_t ldb;
  float beta;
  float *c;
  int64_t ldc;
  GemmChain ch;
  const float *u1 = nullptr, *v1 = nullptr, *u2 = nullptr, *v2 = nullptr;
  uint64_t seed = 0;
};
hipError_t sgemm_spot_capture(const SpotArgs &s, float *save128, hipStream_t st);
hipError_t sgemm_spot_check(const SpotArgs &s, const float *save128, unsigned long long *exp2, unsigned long long *got2,
                            hipStream_t st);
// dst[i] = src[i - min(i / blk, nblk - 1) * blk]: a vector indexed inside a tile (the tiler's blocks, last one
// tail-merged) unrolled over the whole dimension
hipError_t expand_tile_local(const float *src, float *dst, int64_t len, int64_t blk, int64_t nblk, hipStream_t st);
// the norm vectors of flash::kmeans in HBM (null pointer to the struct: plain gemm)
struct KmeansVecs {
  const float *c_l2sq, *p_l2sq, *ones;
};
// One tile task of the gemm / kmeans tilers: row0 / col0 = the tile's first row / column of C
// (kmeans.cpp:115-118 offsets c_l2sq and p_l2sq by them; `ones` is passed un-offset).
inline hipError_t tile_sgemm(char ord, char ta, char tb, int64_t M, int64_t N, int64_t K, float alpha,
                             const float *a, int64_t lda, const float *b, int64_t ldb, float beta, float *c,
                             int64_t ldc, const KmeansVecs *kv, int64_t row0, int64_t col0, hipStream_t st) {
  if (!kv) return sgemm(ord, ta, tb, M, N, K, alpha, a, lda, b, ldb, beta, c, ldc, st);
  return sgemm_rank1x2(ord, ta, tb, M, N, K, alpha, a, lda, b, ldb, beta, c, ldc, kv->c_l2sq + row0, kv->ones,
                       kv->ones, kv->p_l2sq + col0, st);
}
// `seen` (optional, BOF_VERIFY): a launch receipt -- every workgroup adds 1 to seen[blockIdx.x] (row-major kernels: one
// entry per 4 rows); csr_receipt_check compares the first n entries with 1, adds the number of misses to *flag and
// zeroes them again.  What it catches: workgroups that ran under a neighbour's ID (profiles/r6/incident_csrmm).
hipError_t scsrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha, const float *val,
                  const int64_t *col, const int64_t *ptr, const float *b, int64_t ldb, float beta,
                  float *c, int64_t ldc, hipStream_t st, unsigned *seen = nullptr);
int64_t scsrmm_receipt_entries(char ord_b, int64_t m);      // how many entries of `seen` a launch over m rows marks
hipError_t csr_receipt_check(unsigned *seen, int64_t n, unsigned *flag, hipStream_t st);
hipError_t transpose_f32(const float *in, int64_t ld_in, int64_t rows, int64_t cols, float *out,
                         int64_t ld_out, hipStream_t st);
size_t csrcsc_workspace_bytes(int64_t n, int64_t nnz);
hipError_t scsrcsc(int64_t m, int64_t n, int64_t nnz, const float *val, const int64_t *ptr,
                   const int64_t *col, float *val_tr, int64_t *ptr_tr, int64_t *col_tr,
                   void *workspace, hipStream_t st);
hipError_t csc_merge(int nb, int64_t cw, const int64_t *blk_ptr, const int64_t *seg_base, const int64_t *row0,
                     const int64_t *out_ptr, const float *val_in, const int64_t *col_in, float *val_out,
                     int64_t *col_out, hipStream_t st);
// y = A^T x by partitioning the products by column bin (no global atomics); y overwritten
size_t csrgemv_t_workspace_bytes(int64_t n, int64_t nnz);
hipError_t scsrgemv_t_partitioned(int64_t m, int64_t n, int64_t nnz, const float *val, const int64_t *ptr,
                                  const int64_t *col, const float *x, float *y, void *workspace,
                                  hipStream_t st);
// grow-only per-device scratch, freed by bof_flash_release.  Slots: 0 row-major copy of a
// column-major B; 1..16 per-stream row-major C blocks; 17 csrcsc workspace; 18..20 transposed
// CSR (values, indices, offsets) of csrmm 'T'
// 21, 22 k-major copies of GEMM operands (bof_gemm_resident); 23 `ones` of flash::kmeans unrolled over rows / columns
enum { SCR_B_RM = 0, SCR_C_RM0 = 1, SCR_CSRCSC = 17, SCR_TR_VAL = 18, SCR_TR_COL = 19, SCR_TR_PTR = 20,
       SCR_GEMM_A = 21, SCR_GEMM_B = 22, SCR_KM_ONES = 23, SCR_COUNT = 24 };
int scratch_get(int which, size_t bytes, void **ptr);
void scratch_release_all();
// (`seen`: a launch receipt as for scsrmm -- one entry per workgroup of 256 rows, both kernels)
hipError_t scsrgemv(char trans, int64_t m, int64_t n, const float *val, const int64_t *ptr,
                    const int64_t *col, const float *x, float *y, hipStream_t st, unsigned *seen = nullptr);
int64_t scsrgemv_receipt_entries(int64_t m);
// dst = sum of n_src vectors (fixed order; sources may be peer-device memory, dst may alias one of them)
hipError_t sum_partials(float *dst, const float *const *srcs, int n_src, int64_t len, hipStream_t st);
hipError_t gen_dense(float *d, int64_t first, int64_t count, char mode, uint64_t seed,
                     hipStream_t st);
hipError_t gen_sparse_rows(int64_t row0, int64_t nrows, int64_t ncols, int64_t nnz_per_row,
                           float *csr, int64_t *col, int64_t *off, hipStream_t st);

// BOF_VERIFY (gen_kernels.hip): out2[0] += sum of the 32-bit words of a 2-D region, out2[1] += sum of
// word * (logical index + 1); t_pitch > 0: the region is the transposed image of the logical object
hipError_t verify_sum(const void *p, int64_t rows, int64_t row_words, int64_t pitch_words, uint64_t index_base,
                      int64_t t_pitch, unsigned long long *out2, hipStream_t st);

// ---- tilers (plan.cpp) ----------------------------------------------------------
struct GemmGeometry {
  int64_t size[3];  // m, k, n
  int64_t blk[3];   // block edge per dim
  int64_t nblk[3];  // block count per dim (tail-merge rule applied)
  int rdim[3], cdim[3];  // stored (row, col) dim of A, B, C
  int64_t ld[3];    // leading dims in elements
};
GemmGeometry gemm_geometry(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                           int64_t lda, int64_t ldb, int64_t ldc, int64_t blk);
void gemm_task_at(const GemmGeometry &g, int64_t l, int64_t i, int64_t j, float beta,
                  bof_gemm_task *t);

// row-panel layout of flash::gemm for a budget (plan.cpp; see bof_panel_plan in bof_hip.h)
// full_dC: when g describes ONE DEVICE'S SLAB of a larger problem (size along the C panel dimension D
// cut down), the stored width of an operand whose columns run along D is still the whole problem's
// (its panels are shared by all devices); 0 = g is the whole problem.
bof_panel_plan plan_panels(const GemmGeometry &g, uint64_t budget, int64_t group, int64_t full_dC = 0, bool with_acc = false);

// ---- fork/join of compute streams (c_api.hip) -----------------------------------
struct StreamSet {
  int n = 0;
  hipStream_t s[16];
  hipEvent_t fork_ev = nullptr, join_ev[16];
  int init(int n_streams);
  int fork(hipStream_t parent);  // every stream waits for work already queued on parent
  int join(hipStream_t parent);  // parent waits for every stream
};
StreamSet *stream_set(int n_streams);  // per-device singleton
// The pipelines of a REPEATED ordinal of the device list ([0,0,0]: the one-GPU stand-in for three devices) get compute
// streams of their own instead of feeding the ordinal's shared set from several dispatcher threads (default since
// round 5; $BOF_STREAMS_PER_REP=0 shares them again: profiles/r4/fuzz_thread_bisect.md section 6); the repetition is
// announced per thread
extern thread_local int t_ordinal_rep;

bof_options resolved(const bof_options *o);

// Level-2 and level-3 entry points share per-device state (stream set, scratch slots, pinned
// rings, the HBM tile slab): host threads entering them on one device are serialised for the
// whole call.  Recursive: csrmm 'T' re-enters the 'N' path.
std::recursive_mutex &device_call_mutex();

}  // namespace bof
