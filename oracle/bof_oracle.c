/*
 * bof_oracle.c -- CPU restatement of the BLAS-on-flash hot path.
 * TEST INFRASTRUCTURE ONLY (see bof_oracle.h header for pinning status).
 * Plain C, no dependencies.  Every function cites the reference lines it
 * restates (paths relative to the upstream tree).
 */
#include "bof_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_SECTOR 512 /* CMakeLists.txt:65 SECTOR_LEN */

/* ------------------------------------------------------------------------- */
/* generators                                                                */
/* ------------------------------------------------------------------------- */

/* glibc stdlib/rand_r.c: three rounds of the 1103515245/12345 LCG producing
 * 11 + 10 + 10 bits.  misc/sparse_create.cpp:69 depends on exactly this. */
int orc_rand_r(unsigned int *seed) {
  unsigned int next = *seed;
  int result;
  next = next * 1103515245u + 12345u;
  result = (int) ((next / 65536u) % 2048u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int) ((next / 65536u) % 1024u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int) ((next / 65536u) % 1024u);
  *seed = next;
  return result;
}

/* misc/dense_create.cpp:28-37 */
void orc_dense_fill(float *out, int64_t first, int64_t count, char mode) {
  if (mode == 's') {
    for (int64_t t = 0; t < count; t++) out[t] = (float) ((first + t) % 10);
  } else {
    for (int64_t t = 0; t < count; t++) out[t] = 0.0f;
  }
}

/* misc/sparse_create.cpp:23 : nnz_per_row = ceil(ncols * sparsity) */
int64_t orc_sparse_nnz_per_row(int64_t ncols, double sparsity) {
  return (int64_t) ceil((double) ncols * sparsity);
}

static int cmp_i64(const void *x, const void *y) {
  int64_t a = *(const int64_t *) x, b = *(const int64_t *) y;
  return (a > b) - (a < b);
}

/* misc/sparse_create.cpp:50-81.  Per row r: seed = r; draw nnz_per_row + 40
 * columns (lo + hi * RAND_MAX) % ncols with lo drawn FIRST (verified
 * against the compiled tool); sort; unique; keep the smallest nnz_per_row.
 * values: csr[i] = (i % 9) + 1 with i the global nnz index (:50-53). */
int orc_sparse_create_rows(int64_t row0, int64_t nrows_chunk, int64_t ncols,
                           int64_t nnz_per_row, float *csr, int64_t *col,
                           int64_t *off) {
  const int64_t ndraw = nnz_per_row + 40;
  int64_t *tmp = (int64_t *) malloc(sizeof(int64_t) * (size_t) ndraw);
  int rc = 0;
  for (int64_t rr = 0; rr < nrows_chunk; rr++) {
    const int64_t r = row0 + rr;
    off[rr] = r * nnz_per_row;
    unsigned int seed = (unsigned int) r;
    for (int64_t t = 0; t < ndraw; t++) {
      int64_t lo = orc_rand_r(&seed);
      int64_t hi = orc_rand_r(&seed);
      tmp[t] = (lo + hi * (int64_t) 2147483647) % ncols;
    }
    qsort(tmp, (size_t) ndraw, sizeof(int64_t), cmp_i64);
    int64_t u = 0;
    for (int64_t t = 0; t < ndraw; t++)
      if (t == 0 || tmp[t] != tmp[t - 1]) tmp[u++] = tmp[t];
    if (u < nnz_per_row) { rc = -1; u = nnz_per_row; }
    for (int64_t t = 0; t < nnz_per_row; t++) {
      const int64_t g = r * nnz_per_row + t;
      col[rr * nnz_per_row + t] = tmp[t];
      csr[rr * nnz_per_row + t] = (float) ((g % 9) + 1);
    }
  }
  off[nrows_chunk] = (row0 + nrows_chunk) * nnz_per_row;
  free(tmp);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* tiling                                                                    */
/* ------------------------------------------------------------------------- */

/* src/blas/gemm.cpp:39-129.  Dim order d: 0=m, 1=k, 2=n;
 * matrices 0=A (m,k), 1=B (k,n), 2=C (m,n). */
int64_t orc_gemm_plan(char ord, char ta, char tb, int64_t m, int64_t n,
                      int64_t k, float beta, int64_t lda, int64_t ldb,
                      int64_t ldc, int64_t blk, orc_gemm_task *out, int64_t cap,
                      int64_t nblk[3]) {
  const int transA = (ta == 'T'), transB = (tb == 'T'), colMajor = (ord == 'C');
  int64_t S[3] = {m, k, n}, LD[3] = {lda, ldb, ldc};
  int64_t row[3], colv[3], bsz[3], nb[3];
  for (int d = 0; d < 3; d++) {
    /* gemm.cpp:46-49: A -> (0,1), B -> (1,2), C -> (0,2) */
    int x = d, y = (d + 1) % 3;
    row[d] = x < y ? x : y;
    colv[d] = x < y ? y : x;
    bsz[d] = S[d] < blk ? S[d] : blk;
  }
  /* gemm.cpp:52-61 */
  int swap[3] = {transA ^ colMajor, transB ^ colMajor, colMajor};
  for (int d = 0; d < 3; d++)
    if (swap[d]) { int64_t t = row[d]; row[d] = colv[d]; colv[d] = t; }
  /* gemm.cpp:63-67 */
  for (int d = 0; d < 3; d++)
    if (LD[d] == 0) LD[d] = S[colv[d]];
  /* gemm.cpp:69-75: remainder < 128 elements merges into the last block */
  for (int d = 0; d < 3; d++) {
    int64_t q = S[d] / bsz[d];
    nb[d] = (S[d] - q * bsz[d] < (int64_t) (ORC_SECTOR / sizeof(float))) ? q : q + 1;
  }
  if (nblk) { nblk[0] = nb[0]; nblk[1] = nb[1]; nblk[2] = nb[2]; }
  const int64_t total = nb[0] * nb[1] * nb[2];
  if (!out) return total;
  int64_t t = 0;
  for (int64_t l = 0; l < nb[1]; l++)
    for (int64_t i = 0; i < nb[0]; i++)
      for (int64_t j = 0; j < nb[2]; j++, t++) {
        if (t >= cap) return total;
        orc_gemm_task *T = &out[t];
        int64_t idx[3] = {i, l, j}, ext[3];
        for (int d = 0; d < 3; d++) /* gemm.cpp:90-95 */
          ext[d] = (idx[d] == nb[d] - 1) ? S[d] - idx[d] * bsz[d] : bsz[d];
        for (int mat = 0; mat < 3; mat++) { /* gemm.cpp:97-112 */
          int64_t r0 = idx[row[mat]] * bsz[row[mat]];
          int64_t c0 = idx[colv[mat]] * bsz[colv[mat]];
          T->nrows[mat] = ext[row[mat]];
          T->ncols[mat] = ext[colv[mat]];
          T->ld_file[mat] = LD[mat];
          T->off[mat] = r0 * LD[mat] + c0;
        }
        T->l = l; T->i = i; T->j = j;
        T->M = ext[0]; T->K = ext[1]; T->N = ext[2];
        T->beta = (l > 0) ? 1.0f : beta; /* gemm.cpp:114-115 */
        T->parent = (l > 0) ? t - nb[0] * nb[2] : -1; /* gemm.cpp:122-126 */
      }
  return total;
}

/* include/blas_utils.h:72-97.  Reference starts at min_rows and grows while the
 * block's nnz <= max_nnz (so it ends one row past the budget), caps at
 * max_rows.  We clamp to the rows remaining (the reference over-runs `ia` when
 * fewer than min_rows rows remain). */
int64_t orc_csr_blocks(const int64_t *ia, int64_t m, int64_t min_rows,
                       int64_t max_rows, int64_t max_nnz, int64_t *starts,
                       int64_t *sizes, int64_t cap) {
  int64_t cur = 0, nb = 0;
  while (cur < m) {
    const int64_t left = m - cur;
    int64_t b = min_rows;
    while (b < left && (ia[cur + b] - ia[cur]) <= max_nnz) b++;
    if (b > max_rows) b = max_rows;
    if (b > left) b = left;
    if (nb < cap) {
      if (starts) starts[nb] = cur;
      if (sizes) sizes[nb] = b;
    }
    nb++;
    cur += b;
  }
  return nb;
}

/* ------------------------------------------------------------------------- */
/* arithmetic                                                                */
/* ------------------------------------------------------------------------- */

static inline float epilogue(float alpha, float acc, float beta, float cold) {
  return (beta == 0.0f) ? alpha * acc : fmaf(alpha, acc, beta * cold);
}

/* row-major core: C[m x n] = alpha*op(A)*op(B) + beta*C, i-k-j order so each
 * output element is a k-ordered fmaf chain (what v_mfma_f32_32x32x2_f32
 * produces) while the j loop vectorises. */
static void sgemm_rm(int ta, int tb, int64_t m, int64_t n, int64_t k, float alpha,
                     const float *a, int64_t lda, const float *b, int64_t ldb,
                     float beta, float *c, int64_t ldc) {
  /* cblas_sgemm's quick return (BLAS: "when alpha is zero or k is zero, A and B are not
   * referenced"): C = beta*C, so NaN / Inf in A or B do not reach C.  Pinned by
   * tests/golden/mkl_golden_special.npz (MKL 2021.4 at the reference's call site,
   * include/tasks/gemm_task.h:87-90).  mkl_scsrmm has NO such path (same fixture): there
   * alpha = 0 still multiplies the row sums. */
  if (alpha == 0.0f || k == 0) {
    k = 0;
    alpha = 0.0f;
  }
  float *acc = (float *) malloc(sizeof(float) * (size_t) (n > 0 ? n : 1));
  /* B stored [n][k] ('T'): one [k][n] copy, so that the j loop below runs over contiguous memory and vectorises in
   * both layouts (the fuzz tests spend most of their time here); the arithmetic per element is untouched */
  float *bt = NULL;
  if (tb && k > 0 && n > 0) {
    bt = (float *) malloc(sizeof(float) * (size_t) k * (size_t) n);
    for (int64_t j = 0; j < n; j++)
      for (int64_t kk = 0; kk < k; kk++) bt[kk * n + j] = b[j * ldb + kk];
  }
  for (int64_t i = 0; i < m; i++) {
    for (int64_t j = 0; j < n; j++) acc[j] = 0.0f;
    for (int64_t kk = 0; kk < k; kk++) {
      const float av = ta ? a[kk * lda + i] : a[i * lda + kk];
      const float *restrict brow = bt ? bt + kk * n : b + kk * ldb;
      float *restrict ac = acc;
      for (int64_t j = 0; j < n; j++) ac[j] = fmaf(av, brow[j], ac[j]);
    }
    float *crow = c + i * ldc;
    for (int64_t j = 0; j < n; j++) crow[j] = epilogue(alpha, acc[j], beta, crow[j]);
  }
  free(bt);
  free(acc);
}

/* cblas_sgemm as called at include/tasks/gemm_task.h:87-90 and
 * drivers/in_mem_gemm.cpp:64-67.  Column-major is the row-major product of the
 * swapped operands: C^T = op(B)^T op(A)^T. */
void orc_sgemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
               float alpha, const float *a, int64_t lda, const float *b,
               int64_t ldb, float beta, float *c, int64_t ldc) {
  if (ord == 'C')
    sgemm_rm(tb == 'T', ta == 'T', n, m, k, alpha, b, ldb, a, lda, beta, c, ldc);
  else
    sgemm_rm(ta == 'T', tb == 'T', m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}

static void gather_tile(float *dst, const float *src, int64_t off, int64_t nrows,
                        int64_t ncols, int64_t ld) {
  for (int64_t r = 0; r < nrows; r++)
    memcpy(dst + r * ncols, src + off + r * ld, sizeof(float) * (size_t) ncols);
}

/* src/blas/gemm.cpp:27-202: every task reads packed tiles (ld = stored column
 * count, :117-120), runs sgemm with the caller's flags, C tile chained over l
 * with beta=1, written back strided. */
void orc_flash_gemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                    float alpha, float beta, const float *a, const float *b,
                    float *c, int64_t lda, int64_t ldb, int64_t ldc,
                    int64_t blk) {
  int64_t nblk[3];
  int64_t nt = orc_gemm_plan(ord, ta, tb, m, n, k, beta, lda, ldb, ldc, blk, NULL, 0, nblk);
  orc_gemm_task *tasks = (orc_gemm_task *) malloc(sizeof(orc_gemm_task) * (size_t) nt);
  orc_gemm_plan(ord, ta, tb, m, n, k, beta, lda, ldb, ldc, blk, tasks, nt, nblk);
  for (int64_t t = 0; t < nt; t++) {
    orc_gemm_task *T = &tasks[t];
    float *ta_ = (float *) malloc(sizeof(float) * (size_t) (T->nrows[0] * T->ncols[0]));
    float *tb_ = (float *) malloc(sizeof(float) * (size_t) (T->nrows[1] * T->ncols[1]));
    float *tc_ = (float *) malloc(sizeof(float) * (size_t) (T->nrows[2] * T->ncols[2]));
    gather_tile(ta_, a, T->off[0], T->nrows[0], T->ncols[0], T->ld_file[0]);
    gather_tile(tb_, b, T->off[1], T->nrows[1], T->ncols[1], T->ld_file[1]);
    if (T->beta != 0.0f) /* gemm_task.h:49-53: C read iff beta != 0 */
      gather_tile(tc_, c, T->off[2], T->nrows[2], T->ncols[2], T->ld_file[2]);
    orc_sgemm(ord, ta, tb, T->M, T->N, T->K, alpha, ta_, T->ncols[0], tb_,
              T->ncols[1], T->beta, tc_, T->ncols[2]);
    for (int64_t r = 0; r < T->nrows[2]; r++)
      memcpy(c + T->off[2] + r * T->ld_file[2], tc_ + r * T->ncols[2],
             sizeof(float) * (size_t) T->ncols[2]);
    free(ta_); free(tb_); free(tc_);
  }
  free(tasks);
}

/* drivers/in_mem_gemm.cpp:63-70: the whole product as ONE cblas_sgemm call on the in-memory images of the three
 * files (leading dimension 0 = the stored width, the flash API's default: include/flash_blas.h:14-18).  Per output
 * element one k-ordered fmaf chain over the WHOLE K -- no rounding at the tiler's k-block boundaries, which is where
 * it differs (by ~1e-7 relative) from orc_flash_gemm's chain of per-tile calls.  The north-star bar is stated
 * against this driver ("outputs match drivers/in_mem_gemm"), and the product's default arithmetic
 * (bof_options.gemm_chain = 0) reproduces it bit for bit whatever the tile size. */
void orc_in_mem_gemm(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                     float alpha, float beta, const float *a, const float *b,
                     float *c, int64_t lda, int64_t ldb, int64_t ldc) {
  const int colmajor = ord == 'C';
  /* stored width of A, B, C (src/blas/gemm.cpp:39-60: a matrix stored as the transpose of its logical shape
   * swaps its dimensions) */
  if (!lda) lda = ((ta == 'T') != colmajor) ? m : k;
  if (!ldb) ldb = ((tb == 'T') != colmajor) ? k : n;
  if (!ldc) ldc = colmajor ? m : n;
  orc_sgemm(ord, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}

/* KMeansTask::execute (include/tasks/kmeans_task.h:53-82): the tile product, then two K = 1
 * products with alpha = beta = 1:
 *   mkl_gemm(ord, NoTrans, Trans, a_nrows, b_ncols, 1, 1.0, c_l2sq, a_nrows, ones, b_ncols, 1.0, C, ldc)
 *   mkl_gemm(ord, NoTrans, Trans, a_nrows, b_ncols, 1, 1.0, ones, a_nrows, p_l2sq, b_ncols, 1.0, C, ldc)
 * i.e. C[r][c] += c_l2sq[r]*ones[c], then C[r][c] += ones[r]*p_l2sq[c].
 * Column-major ('C', the only order the reference's driver uses, drivers/kmeans.cpp:37-39) is
 * restated call for call.  Row-major is NOT: there the reference's leading dimensions
 * (a_nrows for an a_nrows x 1 row-major operand) make cblas read c_l2sq[r * a_nrows], out of
 * bounds for every r > 0, and its tiler offsets c_l2sq by the COLUMN block (kmeans.cpp:115-118);
 * for 'R' this oracle states the evident intent -- the same two updates, r along m, c along n
 * (leading dimensions 1) -- and the product documents the deviation (include/bof_hip.h). */
void orc_skmeans_task(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                      float alpha, const float *a, int64_t lda, const float *b,
                      int64_t ldb, float beta, float *c, int64_t ldc,
                      const float *c_l2sq, const float *p_l2sq, const float *ones) {
  orc_sgemm(ord, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
  if (ord == 'C') {
    orc_sgemm('C', 'N', 'T', m, n, 1, 1.0f, c_l2sq, m, ones, n, 1.0f, c, ldc);
    orc_sgemm('C', 'N', 'T', m, n, 1, 1.0f, ones, m, p_l2sq, n, 1.0f, c, ldc);
  } else {
    orc_sgemm('R', 'N', 'T', m, n, 1, 1.0f, c_l2sq, 1, ones, 1, 1.0f, c, ldc);
    orc_sgemm('R', 'N', 'T', m, n, 1, 1.0f, ones, 1, p_l2sq, 1, 1.0f, c, ldc);
  }
}

/* src/blas/kmeans.cpp:27-198: flash::gemm's tiler (same blocks, same tail-merge rule :76-81,
 * same packed tiles :128-131, beta = 1 for l > 0 :121-122) with KMeansTask tasks.  Tile
 * (l, i, j) gets c_l2sq + i*blk_m and p_l2sq + j*blk_n (:115-118 for 'C'; see above for 'R')
 * and the un-offset `ones`; every l runs the whole task, so the two updates are added once
 * per k-block. */
void orc_flash_kmeans(char ord, char ta, char tb, int64_t m, int64_t n, int64_t k,
                      float alpha, float beta, const float *a, const float *b,
                      float *c, int64_t lda, int64_t ldb, int64_t ldc, int64_t blk,
                      const float *c_l2sq, const float *p_l2sq, const float *ones) {
  int64_t nblk[3];
  int64_t nt = orc_gemm_plan(ord, ta, tb, m, n, k, beta, lda, ldb, ldc, blk, NULL, 0, nblk);
  if (nt <= 0) return;
  orc_gemm_task *tasks = (orc_gemm_task *) malloc(sizeof(orc_gemm_task) * (size_t) nt);
  orc_gemm_plan(ord, ta, tb, m, n, k, beta, lda, ldb, ldc, blk, tasks, nt, nblk);
  const int64_t blk_m = blk < m ? blk : m, blk_n = blk < n ? blk : n;
  for (int64_t t = 0; t < nt; t++) {
    orc_gemm_task *T = &tasks[t];
    float *ta_ = (float *) malloc(sizeof(float) * (size_t) (T->nrows[0] * T->ncols[0]));
    float *tb_ = (float *) malloc(sizeof(float) * (size_t) (T->nrows[1] * T->ncols[1]));
    float *tc_ = (float *) malloc(sizeof(float) * (size_t) (T->nrows[2] * T->ncols[2]));
    gather_tile(ta_, a, T->off[0], T->nrows[0], T->ncols[0], T->ld_file[0]);
    gather_tile(tb_, b, T->off[1], T->nrows[1], T->ncols[1], T->ld_file[1]);
    if (T->beta != 0.0f)
      gather_tile(tc_, c, T->off[2], T->nrows[2], T->ncols[2], T->ld_file[2]);
    orc_skmeans_task(ord, ta, tb, T->M, T->N, T->K, alpha, ta_, T->ncols[0], tb_, T->ncols[1],
                     T->beta, tc_, T->ncols[2], c_l2sq + T->i * blk_m, p_l2sq + T->j * blk_n, ones);
    for (int64_t r = 0; r < T->nrows[2]; r++)
      memcpy(c + T->off[2] + r * T->ld_file[2], tc_ + r * T->ncols[2],
             sizeof(float) * (size_t) T->ncols[2]);
    free(ta_); free(tb_); free(tc_);
  }
  free(tasks);
}

/* mkl_scsrmm('N', m, n, k, alpha, "GXXC"|"GXXF", val, col, pntrb, pntre, B, ldb,
 * beta, C, ldc): include/tasks/csrmm_task.h:226-228 (row-major, 0-based) and
 * :310-312 (column-major; the reference converts to 1-based first, we keep
 * 0-based indices and only change the dense layout).  Per output element a
 * fmaf chain in nnz order. */
void orc_scsrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha,
                const float *val, const int64_t *col, const int64_t *ptrb,
                const int64_t *ptre, const float *b, int64_t ldb, float beta,
                float *c, int64_t ldc) {
  (void) k;
  float *acc = (float *) malloc(sizeof(float) * (size_t) (n > 0 ? n : 1));
  const int64_t base = ptrb[0];
  for (int64_t i = 0; i < m; i++) {
    for (int64_t j = 0; j < n; j++) acc[j] = 0.0f;
    for (int64_t p = ptrb[i] - base; p < ptre[i] - base; p++) {
      const float v = val[p];
      const int64_t cc = col[p];
      if (ord_b == 'R') {
        const float *brow = b + cc * ldb;
        for (int64_t j = 0; j < n; j++) acc[j] = fmaf(v, brow[j], acc[j]);
      } else {
        for (int64_t j = 0; j < n; j++) acc[j] = fmaf(v, b[j * ldb + cc], acc[j]);
      }
    }
    if (ord_b == 'R') {
      float *crow = c + i * ldc;
      for (int64_t j = 0; j < n; j++) crow[j] = epilogue(alpha, acc[j], beta, crow[j]);
    } else {
      for (int64_t j = 0; j < n; j++)
        c[j * ldc + i] = epilogue(alpha, acc[j], beta, c[j * ldc + i]);
    }
  }
  free(acc);
}

/* src/blas/csrmm.cpp:64-126 ('R') and :203-266 ('C'): row blocks from
 * fill_blocks(min = 128 rows), column panels of width cblk; every task is one
 * scsrmm on a packed B panel / C block.  Tasks of one row block write disjoint
 * C columns, so running them in place on the full arrays (ld = full k) gives
 * the same result as the reference's packed copies. */
void orc_flash_csrmm(char ord_b, int64_t m, int64_t n, int64_t k, float alpha,
                     float beta, const float *val, const int64_t *ia,
                     const int64_t *ja, const float *b, float *c,
                     int64_t max_rows, int64_t max_nnz, int64_t cblk) {
  int64_t nb = orc_csr_blocks(ia, m, ORC_SECTOR / sizeof(float), max_rows, max_nnz, NULL, NULL, 0);
  int64_t *st = (int64_t *) malloc(sizeof(int64_t) * (size_t) nb);
  int64_t *sz = (int64_t *) malloc(sizeof(int64_t) * (size_t) nb);
  orc_csr_blocks(ia, m, ORC_SECTOR / sizeof(float), max_rows, max_nnz, st, sz, nb);
  for (int64_t bi = 0; bi < nb; bi++) {
    const int64_t s = st[bi], r = sz[bi], z = ia[s]; /* absolute: ja + ia[start], src/blas/csrmm.cpp:97-98 */
    for (int64_t j0 = 0; j0 < k; j0 += cblk) {
      const int64_t w = (k - j0 < cblk) ? k - j0 : cblk;
      if (ord_b == 'R')
        orc_scsrmm('R', r, w, n, alpha, val + z, ja + z, ia + s, ia + s + 1,
                   b + j0, k, beta, c + s * k + j0, k);
      else
        orc_scsrmm('C', r, w, n, alpha, val + z, ja + z, ia + s, ia + s + 1,
                   b + j0 * n, n, beta, c + j0 * m + s, m);
    }
  }
  free(st); free(sz);
}

/* include/tasks/csrcsc_task.h:66-75 (mkl_csrcsc on one row block) and
 * src/blas/csrcsc.cpp:32-159 (blocks + merge): both amount to the stable
 * counting-sort transposition below. */
void orc_csrcsc(int64_t m, int64_t n, const float *val, const int64_t *ia,
                const int64_t *ja, float *val_tr, int64_t *ia_tr, int64_t *ja_tr) {
  const int64_t base = m > 0 ? ia[0] : 0;
  for (int64_t j = 0; j <= n; j++) ia_tr[j] = 0;
  for (int64_t i = 0; i < m; i++)
    for (int64_t p = ia[i] - base; p < ia[i + 1] - base; p++) ia_tr[ja[p] + 1]++;
  for (int64_t j = 0; j < n; j++) ia_tr[j + 1] += ia_tr[j];
  int64_t *cur = (int64_t *) malloc(sizeof(int64_t) * (size_t) (n > 0 ? n : 1));
  for (int64_t j = 0; j < n; j++) cur[j] = ia_tr[j];
  for (int64_t i = 0; i < m; i++)
    for (int64_t p = ia[i] - base; p < ia[i + 1] - base; p++) {
      const int64_t q = cur[ja[p]]++;
      ja_tr[q] = i;
      val_tr[q] = val[p];
    }
  free(cur);
}

void orc_scsrmm_t(int64_t m, int64_t n, int64_t k, float alpha, const float *val,
                  const int64_t *ia, const int64_t *ja, const float *b, int64_t ldb,
                  float beta, float *c, int64_t ldc) {
  const int64_t base = m > 0 ? ia[0] : 0;
  float *acc = (float *) calloc((size_t) (n > 0 ? n : 1) * (size_t) (k > 0 ? k : 1), sizeof(float));
  for (int64_t i = 0; i < m; i++) {
    const float *brow = b + i * ldb;
    for (int64_t p = ia[i] - base; p < ia[i + 1] - base; p++) {
      const float v = val[p];
      float *arow = acc + ja[p] * k;
      for (int64_t j = 0; j < k; j++) arow[j] = fmaf(v, brow[j], arow[j]);
    }
  }
  for (int64_t r = 0; r < n; r++)
    for (int64_t j = 0; j < k; j++)
      c[r * ldc + j] = epilogue(alpha, acc[r * k + j], beta, c[r * ldc + j]);
  free(acc);
}

/* mkl_cspblas_scsrgemv (0-based CSR) as used through
 * include/tasks/csrgemv_task.h:74 ('N') and :165 ('T'); the reference pads the
 * block to a square `dim` (:36-44) which contributes only zeros. */
void orc_scsrgemv(char trans, int64_t m, int64_t n, const float *val,
                  const int64_t *ia, const int64_t *ja, const float *x,
                  float *y) {
  const int64_t base = ia[0];
  if (trans == 'N') {
    for (int64_t i = 0; i < m; i++) {
      float acc = 0.0f;
      for (int64_t p = ia[i] - base; p < ia[i + 1] - base; p++)
        acc = fmaf(val[p], x[ja[p]], acc);
      y[i] = acc;
    }
  } else {
    for (int64_t j = 0; j < n; j++) y[j] = 0.0f;
    for (int64_t i = 0; i < m; i++)
      for (int64_t p = ia[i] - base; p < ia[i + 1] - base; p++)
        y[ja[p]] = fmaf(val[p], x[i], y[ja[p]]);
  }
}

/* src/blas/csrgemv.cpp:14-97: blocks by get_next_blk_size(min 128,
 * CSRMM_RM_RBLK_SIZE, MAX_NNZS); 'N' writes disjoint slices; 'T' zeroes y
 * (:64) then adds each block's partial vector (csrgemv_task.h:169-176). */
void orc_flash_csrgemv(char trans, int64_t m, int64_t n, const float *val,
                       const int64_t *ia, const int64_t *ja, const float *x,
                       float *y, int64_t max_rows, int64_t max_nnz) {
  int64_t nb = orc_csr_blocks(ia, m, ORC_SECTOR / sizeof(float), max_rows, max_nnz, NULL, NULL, 0);
  int64_t *st = (int64_t *) malloc(sizeof(int64_t) * (size_t) nb);
  int64_t *sz = (int64_t *) malloc(sizeof(int64_t) * (size_t) nb);
  orc_csr_blocks(ia, m, ORC_SECTOR / sizeof(float), max_rows, max_nnz, st, sz, nb);
  float *part = NULL;
  if (trans == 'T') {
    part = (float *) malloc(sizeof(float) * (size_t) n);
    for (int64_t j = 0; j < n; j++) y[j] = 0.0f;
  }
  for (int64_t bi = 0; bi < nb; bi++) {
    const int64_t s = st[bi], r = sz[bi], z = ia[s]; /* absolute: ja + ia[start], src/blas/csrmm.cpp:97-98 */
    if (trans == 'N') {
      orc_scsrgemv('N', r, n, val + z, ia + s, ja + z, x, y + s);
    } else {
      orc_scsrgemv('T', r, n, val + z, ia + s, ja + z, x + s, part);
      for (int64_t j = 0; j < n; j++) y[j] += part[j];
    }
  }
  free(part); free(st); free(sz);
}

/* ------------------------------------------------------------------------- */
/* helpers                                                                   */
/* ------------------------------------------------------------------------- */

/* src/utils.cpp:31-43 */
uint64_t orc_fnv64a(const char *s, uint64_t n) {
  uint64_t h = 14695981039346656037ull;
  for (uint64_t i = 0; i < n; i++) {
    h ^= (uint64_t) (int64_t) s[i]; /* reference xors the (signed) char */
    h *= 0x100000001b3ull;
  }
  return h;
}

/* src/utils.cpp:48-53 */
uint64_t orc_buf_size(uint64_t n_strides, uint64_t len_per_stride) {
  if (n_strides == 1)
    return ((len_per_stride + ORC_SECTOR - 1) / ORC_SECTOR) * ORC_SECTOR + ORC_SECTOR;
  return n_strides * len_per_stride;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Row-major NN sgemm, alpha=1 beta=0, OpenMP over row panels, cache-blocked
 * i-k-j.  Only bench.py's cpu_baseline ("port") times this. */
void orc_sgemm_mt(int64_t m, int64_t n, int64_t k, const float *a, const float *b,
                  float *c, int nthreads) {
  const int64_t BI = 64, BK = 256, BJ = 1024;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) collapse(2)
#endif
  for (int64_t i0 = 0; i0 < m; i0 += BI)
    for (int64_t j0 = 0; j0 < n; j0 += BJ) {
      const int64_t i1 = i0 + BI < m ? i0 + BI : m;
      const int64_t j1 = j0 + BJ < n ? j0 + BJ : n;
      for (int64_t i = i0; i < i1; i++)
        for (int64_t j = j0; j < j1; j++) c[i * n + j] = 0.0f;
      for (int64_t k0 = 0; k0 < k; k0 += BK) {
        const int64_t k1 = k0 + BK < k ? k0 + BK : k;
        for (int64_t i = i0; i < i1; i++) {
          float *crow = c + i * n;
          for (int64_t kk = k0; kk < k1; kk++) {
            const float av = a[i * k + kk];
            const float *brow = b + kk * n;
            for (int64_t j = j0; j < j1; j++) crow[j] = fmaf(av, brow[j], crow[j]);
          }
        }
      }
    }
  (void) nthreads;
}
