// csrcsc driver -- same command line as the reference's drivers/csrcsc.cpp:17-28:
//   csrcsc_driver <vals_a> <indices_a> <offsets_a> <vals_a_tr> <indices_a_tr> <offsets_a_tr>
//                 <n_rows> <n_cols>
// The three output files must exist (nnz fp32 / nnz int64 / n_cols+1 int64).
#include <string>

#include "bof_timer.h"
#include "bof_utils.h"
#include "flash_blas.h"
#include "lib_funcs.h"

int main(int argc, char** argv) {
  if (argc != 9)
    GLOG_FATAL("usage : <exec> <vals_a> <indices_a> <offsets_a> <vals_a_tr> <indices_a_tr> "
               "<offsets_a_tr> <n_rows> <n_cols>");
  flash::flash_setup("");
  const FBLAS_UINT n_rows = std::stoull(argv[7]), n_cols = std::stoull(argv[8]);

  auto vals = flash::map_file<FPTYPE>(argv[1], flash::Mode::READWRITE);
  auto idxs = flash::map_file<MKL_INT>(argv[2], flash::Mode::READWRITE);
  auto offs = flash::map_file<MKL_INT>(argv[3], flash::Mode::READWRITE);
  auto vals_tr = flash::map_file<FPTYPE>(argv[4], flash::Mode::READWRITE);
  auto idxs_tr = flash::map_file<MKL_INT>(argv[5], flash::Mode::READWRITE);
  auto offs_tr = flash::map_file<MKL_INT>(argv[6], flash::Mode::READWRITE);

  GLOG_INFO("Starting csrcsc call");
  flash::Timer timer;
  const FBLAS_INT res = flash::csrcsc(n_rows, n_cols, offs, idxs, vals, offs_tr, idxs_tr, vals_tr);
  GLOG_INFO("csrcsc() took ", timer.elapsed() / 1000);
  GLOG_INFO("Finished csrcsc");

  flash::unmap_file(vals);
  flash::unmap_file(idxs);
  flash::unmap_file(offs);
  flash::unmap_file(vals_tr);
  flash::unmap_file(idxs_tr);
  flash::unmap_file(offs_tr);
  flash::flash_destroy();
  return res == 0 ? 0 : 1;
}
