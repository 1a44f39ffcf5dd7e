// csrmm driver -- command line and timing line of the reference's drivers/csrmm.cpp:12-15,62-65:
//   csrmm_driver <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <B_ncols>
//                <alpha> <beta> <trans_a> <ord_b>
#include "driver_util.h"

int main(int argc, char** argv) {
  const drv::Args arg(argc, argv, 12,
                      "<vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <B_ncols> <alpha> <beta> "
                      "<trans_a> <ord_b>");
  FBLAS_INT res;
  {
    drv::Session lib;
    auto vals = lib.map<FPTYPE>(arg.str(1));
    auto cols = lib.map<MKL_INT>(arg.str(2));
    auto offs = lib.map<MKL_INT>(arg.str(3));
    auto B = lib.map<FPTYPE>(arg.str(4)), C = lib.map<FPTYPE>(arg.str(5));
    GLOG_INFO("Starting csrmm call");
    flash::Timer timer;
    res = flash::csrmm(arg.c(11), arg.u(6), arg.u(7), arg.u(8), arg.f(9), arg.f(10), vals, offs, cols, arg.c(12), B, C);
    GLOG_INFO("csrmm() took ", timer.elapsed() / 1000);
    GLOG_INFO("Finished csrmm");
  }
  return res == 0 ? 0 : 1;
}
