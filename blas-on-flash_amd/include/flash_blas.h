// flash_blas.h -- the flash BLAS kernel API, hot-path subset, with the reference's
// names, argument order and defaults (include/flash_blas.h:14-18, 37-46, 55-57).
// Implemented in src/flash_api.cpp on top of the C ABI of libbof_hip.so
// (include/bof_hip.h): tiles stream file -> pinned ring -> HBM and are computed
// by hand-written gfx950 kernels.  kmeans (include/flash_blas.h:20-25) is the gemm pipeline
// with KMeansTask's two rank-1 updates fused into the tile store.  Out-of-scope kernels of the
// reference (sort, map, reduce, the never-defined gemv) are not declared.
#pragma once
#include <functional>

#include "bof_logger.h"
#include "bof_types.h"
#include "pointers/allocator.h"
#include "pointers/pointer.h"

namespace flash {
  // C = alpha*op(A)*op(B) + beta*C ; mat_ord 'R'|'C', trans_* 'N'|'T'; leading dims in
  // elements (0 = tight).  Returns 0.
  FBLAS_INT gemm(CHAR mat_ord, CHAR trans_a, CHAR trans_b, FBLAS_UINT m, FBLAS_UINT n,
                 FBLAS_UINT k, FPTYPE alpha, FPTYPE beta, flash_ptr<FPTYPE> a,
                 flash_ptr<FPTYPE> b, flash_ptr<FPTYPE> c, FBLAS_UINT lda_a = 0,
                 FBLAS_UINT lda_b = 0, FBLAS_UINT lda_c = 0);

  // gemm's tiler with KMeansTask tasks (src/blas/kmeans.cpp, include/tasks/kmeans_task.h:53-82):
  // every tile task computes C = alpha*op(A)*op(B) + beta*C (beta = 1 after the first k-block)
  // and adds c_l2sq[r]*ones[c] + ones[r]*p_l2sq[c] (r along m, c along n; tile-local indices
  // into `ones`).  c_l2sq (m), p_l2sq (n), ones (largest tile edge) are host arrays.  With
  // ('C','T','N', ncenters, npoints, dim, -2, 0, centers, points, dist, dim, dim, ncenters, ...)
  // C is the matrix of squared distances (drivers/kmeans.cpp:37-39).  Returns 0.
  FBLAS_INT kmeans(CHAR mat_ord, CHAR trans_a, CHAR trans_b, FBLAS_UINT m, FBLAS_UINT n,
                   FBLAS_UINT k, FPTYPE alpha, FPTYPE beta, flash_ptr<FPTYPE> a,
                   flash_ptr<FPTYPE> b, flash_ptr<FPTYPE> c, FBLAS_UINT lda_a, FBLAS_UINT lda_b,
                   FBLAS_UINT lda_c, FPTYPE* c_l2sq, FPTYPE* p_l2sq, FPTYPE* ones);

  // trans_a 'N': C = alpha*A*B + beta*C with A (m x n) in CSR {a, ia, ja}, B (n x k) and
  // C (m x k) dense row- ('R') or column-major ('C').  trans_a 'T': C (n x k) =
  // alpha*A^T*B + beta*C with B (m x k).  Returns 0, or -1 for unrecognised flags.
  FBLAS_INT csrmm(CHAR trans_a, FBLAS_UINT m, FBLAS_UINT n, FBLAS_UINT k, FPTYPE alpha,
                  FPTYPE beta, flash_ptr<FPTYPE> a, flash_ptr<MKL_INT> ia, flash_ptr<MKL_INT> ja,
                  CHAR ord_b, flash_ptr<FPTYPE> b, flash_ptr<FPTYPE> c);

  // variant with B and C in host memory (row- or column-major; returns 0 on success)
  FBLAS_INT csrmm(CHAR trans_a, FBLAS_UINT m, FBLAS_UINT n, FBLAS_UINT k, FPTYPE alpha,
                  FPTYPE beta, flash_ptr<FPTYPE> a, flash_ptr<MKL_INT> ia, flash_ptr<MKL_INT> ja,
                  CHAR ord_b, FPTYPE* b, FPTYPE* c);

  // A : CSR(ia, ja, a, m, n) -> A^T : CSR(ia_tr, ja_tr, a_tr, n, m); the three output files
  // must exist (ia_tr: n+1 offsets, ja_tr / a_tr: nnz entries).  Returns 0.
  FBLAS_INT csrcsc(FBLAS_UINT m, FBLAS_UINT n, flash_ptr<MKL_INT> ia, flash_ptr<MKL_INT> ja,
                   flash_ptr<FPTYPE> a, flash_ptr<MKL_INT> ia_tr, flash_ptr<MKL_INT> ja_tr,
                   flash_ptr<FPTYPE> a_tr);

  // c = A*b ('N') or A^T*b ('T'); b and c are host vectors
  FBLAS_INT csrgemv(CHAR trans_a, FBLAS_UINT m, FBLAS_UINT n, flash_ptr<FPTYPE> a,
                    flash_ptr<MKL_INT> ia, flash_ptr<MKL_INT> ja, FPTYPE* b, FPTYPE* c);
}  // namespace flash
