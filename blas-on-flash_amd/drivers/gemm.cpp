// gemm driver -- same command line and timing line as the reference's
// drivers/gemm.cpp:17-23,57-62:
//   gemm_driver <A> <B> <C> <A_nrows> <A_ncols> <B_ncols> <alpha> <beta> <ta> <tb> <ord>
//               <lda_a> <lda_b> <lda_c>
// C is updated in place in its file.
#include <chrono>
#include <string>

#include "bof_utils.h"
#include "flash_blas.h"
#include "lib_funcs.h"

static flash::Logger logger("gemm_driver");

int main(int argc, char** argv) {
  if (argc != 15) {
    LOG_INFO(logger, "Usage Mode : <exec> <mat_A_file> <mat_B_file> <mat_C_file> <A_nrows> "
                     "<A_ncols> <B_ncols> <alpha> <beta> <a transpose?> <b transpose?> "
                     "<matr order> <lda_a> <lda_b> <lda_c>");
    LOG_FATAL(logger, "expected 14 args, got ", argc - 1);
  }
  flash::flash_setup("/tmp/gemm_driver_temps");

  const FBLAS_UINT m = std::stoull(argv[4]), k = std::stoull(argv[5]), n = std::stoull(argv[6]);
  const FPTYPE alpha = std::stof(argv[7]), beta = std::stof(argv[8]);
  const CHAR ta = argv[9][0], tb = argv[10][0], ord = argv[11][0];
  const FBLAS_UINT lda = std::stoull(argv[12]), ldb = std::stoull(argv[13]), ldc = std::stoull(argv[14]);

  auto A = flash::map_file<FPTYPE>(argv[1], flash::Mode::READWRITE);
  auto B = flash::map_file<FPTYPE>(argv[2], flash::Mode::READWRITE);
  auto C = flash::map_file<FPTYPE>(argv[3], flash::Mode::READWRITE);
  LOG_INFO(logger, "dimensions : A = ", m, "x", k, ", B = ", k, "x", n);

  const auto t0 = std::chrono::steady_clock::now();
  const FBLAS_INT res = flash::gemm(ord, ta, tb, m, n, k, alpha, beta, A, B, C, lda, ldb, ldc);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  LOG_INFO(logger, "gemm() took ", secs);
  LOG_INFO(logger, "flash::gemm() returned with ", res);

  flash::unmap_file(A);
  flash::unmap_file(B);
  flash::unmap_file(C);
  flash::flash_destroy();
  return res == 0 ? 0 : 1;
}
