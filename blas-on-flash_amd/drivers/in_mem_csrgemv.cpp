// in_mem_csrgemv driver -- command line of the reference's drivers/in_mem_csrgemv.cpp:1-100:
//   in_mem_csrgemv_driver <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <trans_a>
// The reference pads the matrix to a square of dim = max(rows, cols) (offsets repeated, vectors zero-filled,
// :31-60) because mkl_cspblas_scsrgemv only takes square matrices, and makes ONE call (:63-64); the padding
// rows are empty and the padding entries of the vectors are never written out, so the rectangular product
// below is the same function of the files.  'N': c = A b; 'T': c = A^T b (c is overwritten: MKL's csrgemv
// has no beta).
#include "in_mem_util.h"

int main(int argc, char** argv) {
  if (argc != 9) GLOG_FATAL("usage : <exec> <vals_A> <indices_A> <offsets_A> <vals_B> <vals_C> <A_nrows> <A_ncols> <trans_a>");
  const FBLAS_UINT a_nrows = std::stoull(argv[6]), a_ncols = std::stoull(argv[7]);
  const CHAR trans_a = argv[8][0];
  inmem::need_gpu();
  GLOG_INFO("Reading a_offs from file");
  std::vector<MKL_INT> offs = inmem::load_offsets(argv[3], a_nrows);
  const FBLAS_UINT nnzs = (FBLAS_UINT) (offs[a_nrows] - offs[0]);
  GLOG_INFO("Using nnzs=", nnzs);
  inmem::DeviceArray d_offs, idxs, vals, b, c;
  d_offs.alloc((a_nrows + 1) * sizeof(MKL_INT));
  inmem::must(bof_memcpy_h2d(d_offs.d, offs.data(), (a_nrows + 1) * sizeof(MKL_INT), nullptr), "offsets to HBM");
  GLOG_INFO("Reading a_vals from file");
  vals.load(argv[1], nnzs * sizeof(FPTYPE));
  GLOG_INFO("Reading a_idxs from file");
  idxs.load(argv[2], nnzs * sizeof(MKL_INT));
  const FBLAS_UINT b_len = trans_a == 'N' ? a_ncols : a_nrows, c_len = trans_a == 'N' ? a_nrows : a_ncols;
  GLOG_INFO("Reading vector b from file");
  b.load(argv[4], b_len * sizeof(FPTYPE));
  c.alloc(c_len * sizeof(FPTYPE));
  bof_options o;
  bof_default_options(&o);
  GLOG_INFO("Starting mkl_csrgemv call");
  inmem::must(bof_stream_sync(nullptr), "sync");
  flash::Timer timer;
  inmem::must(bof_csrgemv_resident(trans_a, (int64_t) a_nrows, (int64_t) a_ncols, vals.as<float>() - offs[0], reinterpret_cast<const int64_t*>(offs.data()),
                                   d_offs.as<int64_t>(), idxs.as<int64_t>() - offs[0], b.as<float>(), c.as<float>(), &o, nullptr),
              "bof_csrgemv_resident");
  inmem::must(bof_stream_sync(nullptr), "sync");
  GLOG_INFO("csrgemv() took ", timer.elapsed() / 1000);
  GLOG_INFO("Writing vector c to file");
  c.store(argv[5], c_len * sizeof(FPTYPE));
  bof_flash_release();
  return 0;
}
