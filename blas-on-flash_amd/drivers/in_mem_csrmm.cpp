// in_mem_csrmm driver -- command line and timing line of the reference's drivers/in_mem_csrmm.cpp:1-141:
//   in_mem_csrmm_driver <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <B_ncols>
//                       <alpha> <beta> <trans_a> <ord_b>
// The reference reads A (CSR), B and C into host arrays and makes ONE mkl_csrmm call (:116-121; for 'C' it
// first converts the index arrays to 1-based in place, :100-114 -- not needed here); here everything goes
// into HBM and ONE bof_csrmm_resident call computes C.  B is a_ncols x b_ncols ('N') or a_nrows x b_ncols
// ('T'), C the other one (:60-84).
#include "in_mem_util.h"

int main(int argc, char** argv) {
  if (argc != 13)
    GLOG_FATAL("usage : <exec> <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <B_ncols> <alpha> <beta> "
               "<trans_a> <ord_b>");
  const FBLAS_UINT a_nrows = std::stoull(argv[6]), a_ncols = std::stoull(argv[7]), b_ncols = std::stoull(argv[8]);
  const FPTYPE alpha = std::stof(argv[9]), beta = std::stof(argv[10]);
  const CHAR trans_a = argv[11][0], ord_b = argv[12][0];
  inmem::need_gpu();
  GLOG_INFO("Reading offs_a from file");
  std::vector<MKL_INT> offs = inmem::load_offsets(argv[3], a_nrows);
  const FBLAS_UINT nnzs = (FBLAS_UINT) (offs[a_nrows] - offs[0]);
  GLOG_INFO("Using nnzs=", nnzs);
  inmem::DeviceArray d_offs, idxs, vals, B, C;
  d_offs.alloc((a_nrows + 1) * sizeof(MKL_INT));
  inmem::must(bof_memcpy_h2d(d_offs.d, offs.data(), (a_nrows + 1) * sizeof(MKL_INT), nullptr), "offsets to HBM");
  GLOG_INFO("Reading idxs_a from file");
  idxs.load(argv[2], nnzs * sizeof(MKL_INT));
  GLOG_INFO("Reading vals_a from file");
  vals.load(argv[1], nnzs * sizeof(FPTYPE));
  const FBLAS_UINT b_rows = trans_a == 'N' ? a_ncols : a_nrows, c_rows = trans_a == 'N' ? a_nrows : a_ncols;
  GLOG_INFO("Reading vals_b from file");
  B.load(argv[4], b_rows * b_ncols * sizeof(FPTYPE));
  GLOG_INFO("Reading vals_c from file");
  C.load(argv[5], c_rows * b_ncols * sizeof(FPTYPE));
  bof_options o;
  bof_default_options(&o);
  GLOG_INFO("Starting mkl_csrmm call");
  inmem::must(bof_stream_sync(nullptr), "sync");
  flash::Timer timer;
  // element 0 of the value / index arrays is the first non-zero of row 0 whatever base the offsets carry
  inmem::must(bof_csrmm_resident(trans_a, (int64_t) a_nrows, (int64_t) a_ncols, (int64_t) b_ncols, alpha, beta,
                                 vals.as<float>() - offs[0], reinterpret_cast<const int64_t*>(offs.data()), d_offs.as<int64_t>(), idxs.as<int64_t>() - offs[0],
                                 ord_b, B.as<float>(), C.as<float>(), &o, nullptr), "bof_csrmm_resident");
  inmem::must(bof_stream_sync(nullptr), "sync");
  GLOG_INFO("mkl_csrmm() took ", timer.elapsed() / 1000);
  GLOG_INFO("Write vals_c to file");
  C.store(argv[5], c_rows * b_ncols * sizeof(FPTYPE));
  bof_flash_release();
  return 0;
}
