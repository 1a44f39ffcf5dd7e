// in_mem_gemm driver -- command line and timing line of the reference's drivers/in_mem_gemm.cpp:17-83:
//   in_mem_gemm_driver <A> <B> <C> <A_nrows> <A_ncols> <B_ncols> <alpha> <beta> <ta> <tb> <ord>
//                      <lda_a> <lda_b> <lda_c>
// The reference reads the three matrices into host arrays and makes ONE cblas_sgemm call (:63-67); here the
// matrices go into HBM and ONE bof_sgemm call (the same kernel every tile task of flash::gemm runs) covers
// the whole problem.  misc/gemm_run.sh:21-23 compares this binary's C with gemm_driver's.
#include "in_mem_util.h"

int main(int argc, char** argv) {
  if (argc != 15)
    GLOG_FATAL("Usage Mode : <exec> <mat_A_file> <mat_B_file> <mat_C_file> <A_nrows> <A_ncols> <B_ncols> <alpha> <beta> "
               "<a transpose?> <b transpose?> <matr order> <lda_a> <lda_b> <lda_c>");
  const FBLAS_UINT m = std::stoull(argv[4]), k = std::stoull(argv[5]), n = std::stoull(argv[6]);
  const FPTYPE alpha = std::stof(argv[7]), beta = std::stof(argv[8]);
  const CHAR ta = argv[9][0], tb = argv[10][0], ord = argv[11][0];
  const FBLAS_UINT lda = std::stoull(argv[12]), ldb = std::stoull(argv[13]), ldc = std::stoull(argv[14]);
  inmem::need_gpu();
  // stored shapes as cblas interprets them: rows x leading dimension
  const bool a_rows_m = (ta == 'T') == (ord == 'C'), b_rows_k = (tb == 'T') == (ord == 'C');
  const FBLAS_UINT a_rows = a_rows_m ? m : k, b_rows = b_rows_k ? k : n, c_rows = ord == 'R' ? m : n;
  inmem::DeviceArray A, B, C;
  GLOG_INFO("Reading matrix A into memory");
  A.load(argv[1], a_rows * lda * sizeof(FPTYPE));
  GLOG_INFO("Reading matrix B into memory");
  B.load(argv[2], b_rows * ldb * sizeof(FPTYPE));
  GLOG_INFO("Reading matrix C into memory");
  C.load(argv[3], c_rows * ldc * sizeof(FPTYPE));
  GLOG_INFO("dimensions : A = ", m, "x", k, ", B = ", k, "x", n);
  GLOG_INFO("Starting sgemm call");
  inmem::must(bof_stream_sync(nullptr), "sync");
  flash::Timer timer;
  inmem::must(bof_sgemm(ord, ta, tb, (int64_t) m, (int64_t) n, (int64_t) k, alpha, A.as<float>(), (int64_t) lda, B.as<float>(),
                        (int64_t) ldb, beta, C.as<float>(), (int64_t) ldc, nullptr), "bof_sgemm");
  inmem::must(bof_stream_sync(nullptr), "sync");
  GLOG_INFO("gemm() took ", timer.elapsed() / 1000);
  GLOG_INFO("Writing C to file");
  C.store(argv[3], c_rows * ldc * sizeof(FPTYPE));
  return 0;
}
