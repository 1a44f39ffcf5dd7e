// csrmm driver -- same command line and timing line as the reference's
// drivers/csrmm.cpp:12-15,62-65:
//   csrmm_driver <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols>
//                <B_ncols> <alpha> <beta> <trans_a> <ord_b>
#include <string>

#include "bof_timer.h"
#include "bof_utils.h"
#include "flash_blas.h"
#include "lib_funcs.h"

int main(int argc, char** argv) {
  if (argc != 13)
    GLOG_FATAL("usage : <exec> <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> "
               "<A_ncols> <B_ncols> <alpha> <beta> <trans_a> <ord_b>");
  flash::flash_setup("");
  const FBLAS_UINT a_nrows = std::stoull(argv[6]), a_ncols = std::stoull(argv[7]),
                   b_ncols = std::stoull(argv[8]);
  const FPTYPE alpha = std::stof(argv[9]), beta = std::stof(argv[10]);
  const CHAR trans_a = argv[11][0], ord_b = argv[12][0];

  auto vals = flash::map_file<FPTYPE>(argv[1], flash::Mode::READWRITE);
  auto idxs = flash::map_file<MKL_INT>(argv[2], flash::Mode::READWRITE);
  auto offs = flash::map_file<MKL_INT>(argv[3], flash::Mode::READWRITE);
  auto B = flash::map_file<FPTYPE>(argv[4], flash::Mode::READWRITE);
  auto C = flash::map_file<FPTYPE>(argv[5], flash::Mode::READWRITE);

  GLOG_INFO("Starting csrmm call");
  flash::Timer timer;
  const FBLAS_INT res = flash::csrmm(trans_a, a_nrows, a_ncols, b_ncols, alpha, beta, vals, offs,
                                     idxs, ord_b, B, C);
  GLOG_INFO("csrmm() took ", timer.elapsed() / 1000);
  GLOG_INFO("Finished csrmm");

  flash::unmap_file(vals);
  flash::unmap_file(idxs);
  flash::unmap_file(offs);
  flash::unmap_file(B);
  flash::unmap_file(C);
  flash::flash_destroy();
  return res == 0 ? 0 : 1;
}
