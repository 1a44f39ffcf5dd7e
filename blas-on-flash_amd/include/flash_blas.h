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
  // shorthands for this header only (the signatures are the reference's)
  using DenseFile = flash_ptr<FPTYPE>;    // fp32 values in a file
  using IndexFile = flash_ptr<MKL_INT>;   // int64 CSR offsets / column indices in a file

  // ---- sparse: A is rows x cols in CSR {values, offsets, columns} --------------------------------
  // op 'N': C (rows x width) = alpha*A*B + beta*C, B is cols x width.  op 'T': C (cols x width) =
  // alpha*A^T*B + beta*C, B is rows x width.  dense_order 'R' | 'C' is the layout of B and C.
  // Returns 0, or -1 for unrecognised flags.
  FBLAS_INT csrmm(CHAR op, FBLAS_UINT rows, FBLAS_UINT cols, FBLAS_UINT width, FPTYPE alpha, FPTYPE beta,
                  DenseFile values, IndexFile offsets, IndexFile columns, CHAR dense_order, DenseFile B,
                  DenseFile C);
  // the same with B and C in host memory
  FBLAS_INT csrmm(CHAR op, FBLAS_UINT rows, FBLAS_UINT cols, FBLAS_UINT width, FPTYPE alpha, FPTYPE beta,
                  DenseFile values, IndexFile offsets, IndexFile columns, CHAR dense_order, FPTYPE* B_host,
                  FPTYPE* C_host);
  // y = A*x ('N') or A^T*x ('T'); x and y are host vectors
  FBLAS_INT csrgemv(CHAR op, FBLAS_UINT rows, FBLAS_UINT cols, DenseFile values, IndexFile offsets,
                    IndexFile columns, FPTYPE* x_host, FPTYPE* y_host);
  // CSR(offsets, columns, values; rows x cols) -> its transpose (cols x rows); the three output
  // files must exist (t_offsets: cols + 1 entries, t_columns / t_values: nnz entries).  Returns 0.
  FBLAS_INT csrcsc(FBLAS_UINT rows, FBLAS_UINT cols, IndexFile offsets, IndexFile columns, DenseFile values,
                   IndexFile t_offsets, IndexFile t_columns, DenseFile t_values);

  // ---- dense -----------------------------------------------------------------------------------
  // C (height x width) = alpha*op(A)*op(B) + beta*C, inner dimension `depth`; order 'R' | 'C',
  // op_* 'N' | 'T'; leading dimensions in elements (0 = tight).  Returns 0.
  FBLAS_INT gemm(CHAR order, CHAR op_a, CHAR op_b, FBLAS_UINT height, FBLAS_UINT width, FBLAS_UINT depth,
                 FPTYPE alpha, FPTYPE beta, DenseFile A, DenseFile B, DenseFile C, FBLAS_UINT ld_a = 0,
                 FBLAS_UINT ld_b = 0, FBLAS_UINT ld_c = 0);
  // gemm's tiler with KMeansTask tasks (src/blas/kmeans.cpp, include/tasks/kmeans_task.h:53-82):
  // every tile task computes C = alpha*op(A)*op(B) + beta*C (beta = 1 after the first depth block)
  // and adds row_norms[r]*ones[c] + ones[r]*col_norms[c] (r along height, c along width;
  // tile-local indices into `ones`).  row_norms (height), col_norms (width) and ones (largest
  // tile edge) are host arrays -- the reference calls them c_l2sq, p_l2sq, ones.  With
  // ('C','T','N', ncenters, npoints, dim, -2, 0, centers, points, dist, dim, dim, ncenters, ...)
  // C is the matrix of squared distances (drivers/kmeans.cpp:37-39).  Returns 0.
  FBLAS_INT kmeans(CHAR order, CHAR op_a, CHAR op_b, FBLAS_UINT height, FBLAS_UINT width, FBLAS_UINT depth,
                   FPTYPE alpha, FPTYPE beta, DenseFile A, DenseFile B, DenseFile C, FBLAS_UINT ld_a,
                   FBLAS_UINT ld_b, FBLAS_UINT ld_c, FPTYPE* row_norms, FPTYPE* col_norms, FPTYPE* ones);
}  // namespace flash
