// csrcsc driver -- command line of the reference's drivers/csrcsc.cpp:17-28:
//   csrcsc_driver <vals_a> <indices_a> <offsets_a> <vals_a_tr> <indices_a_tr> <offsets_a_tr>
//                 <n_rows> <n_cols>
// The three output files must exist (nnz fp32 / nnz int64 / n_cols+1 int64).
#include "driver_util.h"

int main(int argc, char** argv) {
  const drv::Args arg(argc, argv, 8,
                      "<vals_a> <indices_a> <offsets_a> <vals_a_tr> <indices_a_tr> <offsets_a_tr> <n_rows> <n_cols>");
  FBLAS_INT res;
  {
    drv::Session lib;
    auto vals = lib.map<FPTYPE>(arg.str(1)), vals_tr = lib.map<FPTYPE>(arg.str(4));
    auto cols = lib.map<MKL_INT>(arg.str(2)), cols_tr = lib.map<MKL_INT>(arg.str(5));
    auto offs = lib.map<MKL_INT>(arg.str(3)), offs_tr = lib.map<MKL_INT>(arg.str(6));
    GLOG_INFO("Starting csrcsc call");
    flash::Timer timer;
    res = flash::csrcsc(arg.u(7), arg.u(8), offs, cols, vals, offs_tr, cols_tr, vals_tr);
    GLOG_INFO("csrcsc() took ", timer.elapsed() / 1000);
    GLOG_INFO("Finished csrcsc");
  }
  return res == 0 ? 0 : 1;
}
