/*
 * bof_oracle.h -- CPU restatement of the BLAS-on-flash _gemm / _csrmm / _csrgemv
 * hot path.  TEST INFRASTRUCTURE ONLY: nothing under blas-on-flash_amd/ (the
 * product) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - generators: pinned to the known-answer hashes SURVEY.md App. A-3 records
 *     from the compiled reference tools (tests/golden/kat.json), and
 *     dense_create additionally to oracle/_ref/dense_create (the reference's own
 *     source file compiled unmodified).
 *   - arithmetic: the reference's arithmetic lives in Intel MKL (closed source,
 *     not vendored; reference pins "MKL 2017+", README.md:19).  Pinned to golden
 *     vectors produced in the build container by calling MKL 2021.4's
 *     cblas_sgemm / mkl_scsrmm / mkl_cspblas_scsrgemv with the reference's
 *     call-site arguments (tests/golden/make_golden_mkl.py) and to the full-size
 *     known answers of SURVEY.md App. A-3.
 *   - the reference library itself (scheduler + libaio) is NOT buildable here
 *     without writing stand-in mkl.h / libaio.h headers, so there is no
 *     oracle/_ref build of flash::gemm; its tiling rules are restated from
 *     source and pinned by hand-derived KATs (SURVEY.md App. D examples).
 *
 * All integer types are 64-bit as in the reference (-DMKL_ILP64,
 * include/bof_types.h:11-28).
 */
#ifndef BOF_ORACLE_H
#define BOF_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- generators (misc/dense_create.cpp, misc/sparse_create.cpp) ---------- */
/* glibc rand_r restated so the generator is libc-independent
 * (misc/sparse_create.cpp:66-70 calls rand_r(&seed)). */
int orc_rand_r(unsigned int *seed);
/* misc/dense_create.cpp:21-38: mode 's' -> x[i] = i % 10, 'z' -> 0.
 * `first` is the global element index of out[0] (for chunked generation). */
void orc_dense_fill(float *out, int64_t first, int64_t count, char mode);
/* misc/sparse_create.cpp:23 */
int64_t orc_sparse_nnz_per_row(int64_t ncols, double sparsity);
/* misc/sparse_create.cpp:50-81 for rows [row0,row0+nrows_chunk): writes
 * csr/col for those rows (nnz_per_row each) and off[0..nrows_chunk] where
 * off[i] = (row0+i)*nnz_per_row.  Returns 0, or -1 if a row had fewer than
 * nnz_per_row distinct columns (the reference asserts). */
int orc_sparse_create_rows(int64_t row0, int64_t nrows_chunk, int64_t ncols,
                           int64_t nnz_per_row, float *csr, int64_t *col,
                           int64_t *off);

/* ---- tiling rules ---------------------------------------------------------- */
/* One GemmTask as src/blas/gemm.cpp:83-129 builds it. Offsets/LDs in elements. */
typedef struct {
  int64_t l, i, j;            /* k-block, m-block, n-block index               */
  int64_t M, K, N;            /* tile extents handed to sgemm                  */
  int64_t off[3];             /* element offset of tile in A, B, C files       */
  int64_t nrows[3];           /* StrideInfo.n_strides  (stored rows)           */
  int64_t ncols[3];           /* StrideInfo.len_per_stride / 4 (stored cols)   */
  int64_t ld_file[3];         /* StrideInfo.stride / 4                          */
  float   beta;               /* beta for l==0, 1 for l>0                      */
  int64_t parent;             /* index of task (l-1,i,j) in the list, or -1    */
} orc_gemm_task;
/* Returns number of tasks (N_k*N_m*N_n) in l-major injection order; fills
 * `out` if non-NULL (capacity `cap`).  nblk[3] receives {N_m, N_k, N_n}. */
int64_t orc_gemm_plan(char ord, char ta, char tb, int64_t m, int64_t n,
                      int64_t k, float beta, int64_t lda, int64_t ldb,
                      int64_t ldc, int64_t blk, orc_gemm_task *out, int64_t cap,
                      int64_t nblk[3]);
/* include/blas_utils.h:72-97 (get_next_blk_size + fill_blocks) with the
 * rows-remaining clamp.  Returns number of blocks. */
int64_t orc_csr_blocks(const int64_t *ia, int64_t m, int64_t min_rows,
                       int64_t max_rows, int64_t max_nnz, int64_t *starts,
                       int64_t *sizes, int64_t cap);

/* ---- arithmetic (the MKL call sites) ------------------------------------- */
/* cblas_sgemm semantics (include/tasks/gemm_task.h:87-90,
 * drivers/in_mem_gemm.cpp:64-67).  fp32, per output element a k-ordered fmaf
 * chain starting from 0, then c = beta==0 ? alpha*acc : fmaf(alpha,acc,beta*c). */
void orc_sgemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
               float alpha, const float *a, int64_t lda, const float *b,
               int64_t ldb, float beta, float *c, int64_t ldc);
/* flash::gemm restated: tile plan (blk) + per-tile orc_sgemm on packed tiles +
 * accumulate chains, operating on in-memory images of the three files. */
void orc_flash_gemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                    float alpha, float beta, const float *a, const float *b,
                    float *c, int64_t lda, int64_t ldb, int64_t ldc,
                    int64_t blk);
/* drivers/in_mem_gemm.cpp:63-70 restated: ONE sgemm over the whole matrices (ld 0 = stored width). */
void orc_in_mem_gemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                     float alpha, float beta, const float *a, const float *b,
                     float *c, int64_t lda, int64_t ldb, int64_t ldc);
/* KMeansTask::execute (include/tasks/kmeans_task.h:53-82) and flash::kmeans
 * (src/blas/kmeans.cpp:27-198); see the notes in bof_oracle.c about row-major. */
void orc_skmeans_task(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                      float alpha, const float *a, int64_t lda, const float *b,
                      int64_t ldb, float beta, float *c, int64_t ldc,
                      const float *c_l2sq, const float *p_l2sq, const float *ones);
void orc_flash_kmeans(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                      float alpha, float beta, const float *a, const float *b,
                      float *c, int64_t lda, int64_t ldb, int64_t ldc, int64_t blk,
                      const float *c_l2sq, const float *p_l2sq, const float *ones);

/* mkl_scsrmm('N', ..., "GXXC"/"GXXF") semantics (include/tasks/csrmm_task.h:
 * 226-228, 310-312; drivers/in_mem_csrmm.cpp:100-120) with 0-based indices in
 * both layouts: C[m x n] = alpha * A[m x k](CSR) * B[k x n] + beta * C.
 * ord_b 'R': B,C row-major with ldb, ldc; 'C': column-major. */
void orc_scsrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha,
                const float *val, const int64_t *col, const int64_t *ptrb,
                const int64_t *ptre, const float *b, int64_t ldb, float beta,
                float *c, int64_t ldc);
/* flash::csrmm 'N' restated (src/blas/csrmm.cpp:64-126, 203-266): row blocks by
 * orc_csr_blocks, column panels of cblk, per-task orc_scsrmm. */
void orc_flash_csrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha,
                     float beta, const float *val, const int64_t *ia,
                     const int64_t *ja, const float *b, float *c,
                     int64_t max_rows, int64_t max_nnz, int64_t cblk);
/* mkl_cspblas_scsrgemv semantics as used by src/blas/csrgemv.cpp:82-97:
 * 'N': y[0..m) = A x ; 'T': y[0..n) = A^T x  (y overwritten). */
void orc_scsrgemv(char trans, int64_t m, int64_t n, const float *val,
                  const int64_t *ia, const int64_t *ja, const float *x,
                  float *y);
/* flash::csrgemv restated with row blocking ('T' accumulates per block). */
void orc_flash_csrgemv(char trans, int64_t m, int64_t n, const float *val,
                       const int64_t *ia, const int64_t *ja, const float *x,
                       float *y, int64_t max_rows, int64_t max_nnz);

/* mkl_csrcsc(job = {0,0,0,-1,-1,1}) as called by include/tasks/csrcsc_task.h:
 * 66-75: A (m x n CSR, 0-based, ia may carry a base) -> A^T as CSR (n x m):
 * ia_tr[n+1] 0-based, ja_tr = original row ids ascending inside each output row
 * (the sequential count / scan / place algorithm, i.e. a stable transposition),
 * val_tr moved with them.  src/blas/csrcsc.cpp:32-159 produces the same arrays
 * through row-block transposes + a column-block merge. */
void orc_csrcsc(int64_t m, int64_t n, const float *val, const int64_t *ia,
                const int64_t *ja, float *val_tr, int64_t *ia_tr, int64_t *ja_tr);
/* mkl_scsrmm('T', "GXXC") semantics (src/blas/csrmm.cpp:355-422 is the broken
 * caller): C[n x k] = alpha * A^T * B[m x k] + beta * C with A an
 * m x n CSR, B/C row-major.  Every output element is an fmaf chain over the
 * source rows in ascending order. */
void orc_scsrmm_t(int64_t m, int64_t n, int64_t k, float alpha, const float *val,
                  const int64_t *ia, const int64_t *ja, const float *b, int64_t ldb,
                  float beta, float *c, int64_t ldc);

/* ---- helpers --------------------------------------------------------------- */
/* src/utils.cpp:31-43 */
uint64_t orc_fnv64a(const char *s, uint64_t n);
/* src/utils.cpp:48-53 */
uint64_t orc_buf_size(uint64_t n_strides, uint64_t len_per_stride);
/* OpenMP-parallel variants used only by bench.py's cpu_baseline leg. */
void orc_sgemm_mt(int64_t m, int64_t n, int64_t k, const float *a, const float *b,
                  float *c, int nthreads);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
