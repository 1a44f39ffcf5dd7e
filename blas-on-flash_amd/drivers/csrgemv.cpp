// csrgemv driver -- same command line as the reference's drivers/csrgemv.cpp:13-15:
//   csrgemv_driver <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <trans_a>
// b and c are host vectors read from / written to plain files (:44-74).  Unlike the
// shipped reference driver this one calls flash_setup() and prints
// a timing line.
#include <fstream>
#include <string>
#include <vector>

#include "bof_timer.h"
#include "bof_utils.h"
#include "flash_blas.h"
#include "lib_funcs.h"

static flash::Logger logger("csrgemv");

int main(int argc, char** argv) {
  if (argc != 9)
    LOG_FATAL(logger, "usage : <exec> <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> "
                      "<A_nrows> <A_ncols> <trans_a>");
  flash::flash_setup("./");
  const FBLAS_UINT a_nrows = std::stoull(argv[6]), a_ncols = std::stoull(argv[7]);
  const CHAR trans_a = argv[8][0];
  const FBLAS_UINT b_len = trans_a == 'N' ? a_ncols : a_nrows;
  const FBLAS_UINT c_len = trans_a == 'N' ? a_nrows : a_ncols;

  auto vals = flash::map_file<FPTYPE>(argv[1], flash::Mode::READWRITE);
  auto idxs = flash::map_file<MKL_INT>(argv[2], flash::Mode::READWRITE);
  auto offs = flash::map_file<MKL_INT>(argv[3], flash::Mode::READWRITE);

  std::vector<FPTYPE> b(b_len), c(c_len);
  std::ifstream(argv[4], std::ios::binary).read((char*) b.data(), b_len * sizeof(FPTYPE));
  std::ifstream(argv[5], std::ios::binary).read((char*) c.data(), c_len * sizeof(FPTYPE));

  LOG_INFO(logger, "Starting csrgemv call");
  flash::Timer timer;
  const FBLAS_INT res = flash::csrgemv(trans_a, a_nrows, a_ncols, vals, offs, idxs, b.data(), c.data());
  LOG_INFO(logger, "csrgemv() took ", timer.elapsed() / 1000);

  flash::unmap_file(vals);
  flash::unmap_file(idxs);
  flash::unmap_file(offs);
  std::ofstream(argv[5], std::ios::binary).write((char*) c.data(), c_len * sizeof(FPTYPE));
  flash::flash_destroy();
  return res == 0 ? 0 : 1;
}
