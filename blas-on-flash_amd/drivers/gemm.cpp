// gemm driver -- command line and timing line of the reference's drivers/gemm.cpp:17-23,57-62:
//   gemm_driver <A> <B> <C> <A_nrows> <A_ncols> <B_ncols> <alpha> <beta> <ta> <tb> <ord>
//               <lda_a> <lda_b> <lda_c>
// (note the m, k, n order of the sizes).  C is updated in place in its file.
#include "driver_util.h"

int main(int argc, char** argv) {
  const drv::Args arg(argc, argv, 14,
                      "<mat_A_file> <mat_B_file> <mat_C_file> <A_nrows> <A_ncols> <B_ncols> <alpha> <beta> "
                      "<a transpose?> <b transpose?> <matr order> <lda_a> <lda_b> <lda_c>");
  FBLAS_INT res;
  {
    drv::Session lib("/tmp/gemm_driver_temps");
    auto A = lib.map<FPTYPE>(arg.str(1)), B = lib.map<FPTYPE>(arg.str(2)), C = lib.map<FPTYPE>(arg.str(3));
    const FBLAS_UINT m = arg.u(4), k = arg.u(5), n = arg.u(6);
    GLOG_INFO("dimensions : A = ", m, "x", k, ", B = ", k, "x", n);
    flash::Timer timer;
    res = flash::gemm(arg.c(11), arg.c(9), arg.c(10), m, n, k, arg.f(7), arg.f(8), A, B, C, arg.u(12), arg.u(13),
                      arg.u(14));
    GLOG_INFO("gemm() took ", timer.elapsed() / 1000);
    GLOG_INFO("flash::gemm() returned with ", res);
  }
  return res == 0 ? 0 : 1;
}
