"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference's
hot path).  TEST INFRASTRUCTURE: importable only from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke(); never from the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle.so")

i64, f32, chr_ = C.c_int64, C.c_float, C.c_char
P = C.c_void_p


class GemmTask(C.Structure):
    _fields_ = [("l", i64), ("i", i64), ("j", i64), ("M", i64), ("K", i64), ("N", i64),
                ("off", i64 * 3), ("nrows", i64 * 3), ("ncols", i64 * 3),
                ("ld_file", i64 * 3), ("beta", f32), ("parent", i64)]


def build():
    """Compile the oracle (and oracle/_ref when the reference tree is present)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_rand_r.argtypes = [C.POINTER(C.c_uint)]
        L.orc_rand_r.restype = C.c_int
        L.orc_dense_fill.argtypes = [P, i64, i64, chr_]
        L.orc_sparse_nnz_per_row.argtypes = [i64, C.c_double]
        L.orc_sparse_nnz_per_row.restype = i64
        L.orc_sparse_create_rows.argtypes = [i64, i64, i64, i64, P, P, P]
        L.orc_gemm_plan.argtypes = [chr_, chr_, chr_, i64, i64, i64, f32, i64, i64, i64, i64,
                                    P, i64, P]
        L.orc_gemm_plan.restype = i64
        L.orc_csr_blocks.argtypes = [P, i64, i64, i64, i64, P, P, i64]
        L.orc_csr_blocks.restype = i64
        L.orc_sgemm.argtypes = [chr_, chr_, chr_, i64, i64, i64, f32, P, i64, P, i64, f32, P, i64]
        L.orc_flash_gemm.argtypes = [chr_, chr_, chr_, i64, i64, i64, f32, f32, P, P, P,
                                     i64, i64, i64, i64]
        L.orc_in_mem_gemm.argtypes = [chr_, chr_, chr_, i64, i64, i64, f32, f32, P, P, P, i64, i64, i64]
        L.orc_skmeans_task.argtypes = [chr_, chr_, chr_, i64, i64, i64, f32, P, i64, P, i64, f32, P, i64,
                                       P, P, P]
        L.orc_flash_kmeans.argtypes = [chr_, chr_, chr_, i64, i64, i64, f32, f32, P, P, P,
                                       i64, i64, i64, i64, P, P, P]
        L.orc_scsrmm.argtypes = [chr_, i64, i64, i64, f32, P, P, P, P, P, i64, f32, P, i64]
        L.orc_flash_csrmm.argtypes = [chr_, i64, i64, i64, f32, f32, P, P, P, P, P, i64, i64, i64]
        L.orc_scsrgemv.argtypes = [chr_, i64, i64, P, P, P, P, P]
        L.orc_flash_csrgemv.argtypes = [chr_, i64, i64, P, P, P, P, P, i64, i64]
        L.orc_csrcsc.argtypes = [i64, i64, P, P, P, P, P, P]
        L.orc_scsrmm_t.argtypes = [i64, i64, i64, f32, P, P, P, P, i64, f32, P, i64]
        L.orc_fnv64a.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_fnv64a.restype = C.c_uint64
        L.orc_buf_size.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_buf_size.restype = C.c_uint64
        L.orc_sgemm_mt.argtypes = [i64, i64, i64, P, P, P, C.c_int]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(P)


def _c(ch):
    return ch.encode() if isinstance(ch, str) else ch


# reference defaults (CMakeLists.txt:38-63; BASELINE cfg2 uses GEMM_BLK_SIZE=4096)
MAX_NNZS = 10_000_000
CSRMM_RBLK = 131072
CSRMM_CBLK = 1024


def dense_fill(nrows, ncols, mode="s"):
    out = np.empty(nrows * ncols, np.float32)
    lib().orc_dense_fill(_p(out), 0, out.size, _c(mode))
    return out.reshape(nrows, ncols)


def sparse_create(nrows, ncols, sparsity):
    """-> (vals f32[nnz], cols i64[nnz], offs i64[nrows+1]) as misc/sparse_create.cpp."""
    npr = lib().orc_sparse_nnz_per_row(ncols, sparsity)
    vals = np.empty(nrows * npr, np.float32)
    cols = np.empty(nrows * npr, np.int64)
    offs = np.empty(nrows + 1, np.int64)
    rc = lib().orc_sparse_create_rows(0, nrows, ncols, npr, _p(vals), _p(cols), _p(offs))
    assert rc == 0
    return vals, cols, offs


def gemm_plan(ord_, ta, tb, m, n, k, beta, lda, ldb, ldc, blk):
    nblk = (i64 * 3)()
    nt = lib().orc_gemm_plan(_c(ord_), _c(ta), _c(tb), m, n, k, beta, lda, ldb, ldc, blk,
                             None, 0, nblk)
    arr = (GemmTask * nt)()
    lib().orc_gemm_plan(_c(ord_), _c(ta), _c(tb), m, n, k, beta, lda, ldb, ldc, blk,
                        arr, nt, nblk)
    return list(arr), list(nblk)


def csr_blocks(ia, m, min_rows=128, max_rows=CSRMM_RBLK, max_nnz=MAX_NNZS):
    ia = np.ascontiguousarray(ia, np.int64)
    nb = lib().orc_csr_blocks(_p(ia), m, min_rows, max_rows, max_nnz, None, None, 0)
    st = np.empty(nb, np.int64)
    sz = np.empty(nb, np.int64)
    lib().orc_csr_blocks(_p(ia), m, min_rows, max_rows, max_nnz, _p(st), _p(sz), nb)
    return st, sz


def sgemm(ord_, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """In place on c (flat float32 array)."""
    lib().orc_sgemm(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, _p(a), lda, _p(b), ldb, beta,
                    _p(c), ldc)
    return c


def flash_gemm_tiles(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda, ldb, ldc, blk):
    """The reference's flash::gemm restated task by task (src/blas/gemm.cpp:83-129): per-tile sgemm on packed
    tiles, accumulate chains with one rounding per k-block.  What the product computes with gemm_chain=1."""
    lib().orc_flash_gemm(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, beta, _p(a), _p(b), _p(c),
                         lda, ldb, ldc, blk)
    return c


def in_mem_gemm(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda=0, ldb=0, ldc=0):
    """drivers/in_mem_gemm.cpp:63-70 restated: ONE sgemm over the whole matrices, in place on c."""
    lib().orc_in_mem_gemm(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, beta, _p(a), _p(b), _p(c), lda, ldb, ldc)
    return c


def flash_gemm(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda, ldb, ldc, blk=0, chain=0):
    """What the product's flash::gemm must produce bit for bit.  chain = 0 (bof_options.gemm_chain default): one
    k-ordered chain per element over the whole K, independent of the tile size `blk` -- the in-memory driver's
    result; chain = 1: the reference's task-by-task arithmetic (needs blk)."""
    if chain == 1:
        return flash_gemm_tiles(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda, ldb, ldc, blk)
    return in_mem_gemm(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda, ldb, ldc)


def skmeans_task(ord_, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, c_l2sq, p_l2sq, ones):
    """KMeansTask::execute restated; in place on c."""
    lib().orc_skmeans_task(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, _p(a), lda, _p(b), ldb, beta,
                           _p(c), ldc, _p(c_l2sq), _p(p_l2sq), _p(ones))
    return c


def flash_kmeans(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda, ldb, ldc, blk, c_l2sq, p_l2sq, ones):
    lib().orc_flash_kmeans(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, beta, _p(a), _p(b), _p(c),
                           lda, ldb, ldc, blk, _p(c_l2sq), _p(p_l2sq), _p(ones))
    return c


def scsrmm(ord_b, m, n, k, alpha, val, col, ptr, b, ldb, beta, c, ldc):
    ptr = np.ascontiguousarray(ptr, np.int64)
    lib().orc_scsrmm(_c(ord_b), m, n, k, alpha, _p(val), _p(col), _p(ptr), _p(ptr[1:]),
                     _p(b), ldb, beta, _p(c), ldc)
    return c


def flash_csrmm(ord_b, m, n, k, alpha, beta, val, ia, ja, b, c, max_rows=CSRMM_RBLK,
                max_nnz=MAX_NNZS, cblk=CSRMM_CBLK):
    lib().orc_flash_csrmm(_c(ord_b), m, n, k, alpha, beta, _p(val), _p(ia), _p(ja), _p(b),
                          _p(c), max_rows, max_nnz, cblk)
    return c


def scsrgemv(trans, m, n, val, ia, ja, x, y):
    lib().orc_scsrgemv(_c(trans), m, n, _p(val), _p(ia), _p(ja), _p(x), _p(y))
    return y


def flash_csrgemv(trans, m, n, val, ia, ja, x, y, max_rows=CSRMM_RBLK, max_nnz=MAX_NNZS):
    lib().orc_flash_csrgemv(_c(trans), m, n, _p(val), _p(ia), _p(ja), _p(x), _p(y),
                            max_rows, max_nnz)
    return y


def csrcsc(m, n, val, ia, ja):
    """-> (val_tr, ia_tr, ja_tr) of the n x m transpose."""
    ia = np.ascontiguousarray(ia, np.int64)
    nnz = int(ia[m] - ia[0]) if m > 0 else 0
    val_tr = np.zeros(max(nnz, 1), np.float32)
    ja_tr = np.zeros(max(nnz, 1), np.int64)
    ia_tr = np.zeros(n + 1, np.int64)
    lib().orc_csrcsc(m, n, _p(val), _p(ia), _p(ja), _p(val_tr), _p(ia_tr), _p(ja_tr))
    return val_tr[:nnz], ia_tr, ja_tr[:nnz]


def scsrmm_t(m, n, k, alpha, val, ia, ja, b, ldb, beta, c, ldc):
    lib().orc_scsrmm_t(m, n, k, alpha, _p(val), _p(ia), _p(ja), _p(b), ldb, beta, _p(c), ldc)
    return c
