// csrgemv driver -- command line of the reference's drivers/csrgemv.cpp:13-15:
//   csrgemv_driver <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <trans_a>
// b (input) and c (output) are host vectors kept in plain files (:44-74).  Unlike the shipped
// reference driver this one binds the library first and prints a timing line.
#include <fstream>

#include "driver_util.h"

static std::vector<FPTYPE> load_vector(const std::string& path, FBLAS_UINT len) {
  std::vector<FPTYPE> v(len);
  std::ifstream(path, std::ios::binary).read(reinterpret_cast<char*>(v.data()), (std::streamsize) (len * sizeof(FPTYPE)));
  return v;
}

int main(int argc, char** argv) {
  const drv::Args arg(argc, argv, 8, "<vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <trans_a>");
  const FBLAS_UINT rows = arg.u(6), cols = arg.u(7);
  const CHAR trans = arg.c(8);
  const bool plain = trans == 'N';
  std::vector<FPTYPE> b = load_vector(arg.str(4), plain ? cols : rows);
  std::vector<FPTYPE> c = load_vector(arg.str(5), plain ? rows : cols);
  FBLAS_INT res;
  {
    drv::Session lib("./");
    auto vals = lib.map<FPTYPE>(arg.str(1));
    auto idxs = lib.map<MKL_INT>(arg.str(2));
    auto offs = lib.map<MKL_INT>(arg.str(3));
    GLOG_INFO("Starting csrgemv call");
    flash::Timer timer;
    res = flash::csrgemv(trans, rows, cols, vals, offs, idxs, b.data(), c.data());
    GLOG_INFO("csrgemv() took ", timer.elapsed() / 1000);
  }
  std::ofstream(arg.str(5), std::ios::binary).write(reinterpret_cast<const char*>(c.data()),
                                                    (std::streamsize) (c.size() * sizeof(FPTYPE)));
  return res == 0 ? 0 : 1;
}
