"""ctypes binding of libbof_hip.so (the C ABI declared in include/bof_hip.h).

Used by tests/, bench.py and __graft_entry__.py.  Device memory is passed as raw
pointers (e.g. torch.Tensor.data_ptr()); streams as hipStream_t handles
(torch.cuda.current_stream().cuda_stream).  There is no CPU fallback: every
compute entry point raises BofError when the library or a HIP device is missing.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libbof_hip.so")

i64, u64, f32, chr_, P = C.c_int64, C.c_uint64, C.c_float, C.c_char, C.c_void_p


class BofError(RuntimeError):
    pass


class Options(C.Structure):
    """bof_options (reference compile-time tunables, CMakeLists.txt:38-63)."""
    _fields_ = [("gemm_blk", i64), ("max_nnzs", i64), ("csrmm_rblk", i64), ("csrmm_cblk", i64),
                ("hbm_budget", i64), ("n_io_threads", C.c_int32), ("n_streams", C.c_int32),
                ("use_odirect", C.c_int32), ("pinned_slots", C.c_int32), ("gemm_path", C.c_int32),
                ("io_chunk_mib", C.c_int32),
                # ABI v3: in-process device list + per-call forms of the environment knobs
                ("n_devices", C.c_int32), ("devices", C.c_int32 * 16), ("io_engine", C.c_int32),
                ("io_request_kib", C.c_int32), ("panel_group", C.c_int32), ("panel_streams", C.c_int32),
                ("panel_writers", C.c_int32), ("panel_kmajor", C.c_int32),
                # one process per GPU: the shared operand read once per node (see include/bof_hip.h)
                ("share_world", C.c_int32), ("share_rank", C.c_int32), ("share_name", C.c_char * 48),
                # ABI v4: instrumentation + device-to-device broadcast of a shared operand
                ("kernel_timing", C.c_int32), ("verify", C.c_int32), ("peer_bcast", C.c_int32),
                # ABI v5: how the k-blocks of a C tile are combined (0 / 2: one chain over the whole K; 1: the reference's)
                ("gemm_chain", C.c_int32), ("reserved_", C.c_int32 * 4)]


class GemmTask(C.Structure):
    _fields_ = [("l", i64), ("i", i64), ("j", i64), ("M", i64), ("K", i64), ("N", i64),
                ("off", i64 * 3), ("nrows", i64 * 3), ("ncols", i64 * 3),
                ("ld_file", i64 * 3), ("beta", f32), ("parent", i64)]


class FPtr(C.Structure):
    """bof_fptr == flash_ptr<T> {file, byte offset} (include/pointers/pointer.h:15-18)."""
    _fields_ = [("fd", C.c_int), ("foffset", u64)]


class PanelPlan(C.Structure):
    """bof_panel_plan: which path bof_flash_gemm takes for a budget, and the row-panel layout."""
    _fields_ = [("eligible", C.c_int32), ("why", C.c_int32), ("streamed", C.c_int32), ("resident", C.c_int32 * 3),
                ("n_panels", i64 * 3), ("n_slots", i64 * 3), ("slot_bytes", u64 * 3), ("need_bytes", u64),
                ("groups", i64), ("first_group", i64), ("acc_bytes", u64)]


class FlashStats(C.Structure):
    _fields_ = [("bytes_read", u64), ("bytes_written", u64), ("bytes_h2d", u64),
                ("bytes_d2h", u64), ("tasks", u64), ("tile_hits", u64), ("tile_misses", u64),
                ("seconds", C.c_double), ("read_ops", u64), ("write_ops", u64), ("bytes_peer", u64),
                ("kernel_launches", u64), ("kernel_seconds", C.c_double), ("bytes_p2p", u64),
                ("verify_checks", u64)]


# every symbol include/bof_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bof_abi_version", C.c_int, []),
    ("bof_last_error", C.c_char_p, []),
    ("bof_device_count", C.c_int, []),
    ("bof_set_device", C.c_int, [C.c_int]),
    ("bof_default_options", None, [C.POINTER(Options)]),
    ("bof_malloc", C.c_int, [C.POINTER(P), C.c_size_t]),
    ("bof_free", C.c_int, [P]),
    ("bof_host_alloc", C.c_int, [C.POINTER(P), C.c_size_t]),
    ("bof_host_free", C.c_int, [P]),
    ("bof_memcpy_h2d", C.c_int, [P, P, C.c_size_t, P]),
    ("bof_memcpy_d2h", C.c_int, [P, P, C.c_size_t, P]),
    ("bof_memset", C.c_int, [P, C.c_int, C.c_size_t, P]),
    ("bof_stream_create", C.c_int, [C.POINTER(P)]),
    ("bof_stream_destroy", C.c_int, [P]),
    ("bof_stream_sync", C.c_int, [P]),
    ("bof_mem_info", C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("bof_sgemm", C.c_int, [chr_, chr_, chr_, i64, i64, i64, f32, P, i64, P, i64, f32, P, i64, P]),
    ("bof_skmeans_task", C.c_int, [chr_, chr_, chr_, i64, i64, i64, f32, P, i64, P, i64, f32, P, i64, P, P, P, P]),
    ("bof_scsrmm", C.c_int, [chr_, i64, i64, i64, f32, P, P, P, P, i64, f32, P, i64, P]),
    ("bof_scsrgemv", C.c_int, [chr_, i64, i64, P, P, P, P, P, P]),
    ("bof_scsrcsc", C.c_int, [i64, i64, i64, P, P, P, P, P, P, P]),
    ("bof_csrcsc_workspace_bytes", u64, [i64, i64]),
    ("bof_gemm_plan", i64, [chr_, chr_, chr_, i64, i64, i64, f32, i64, i64, i64, i64,
                            C.POINTER(GemmTask), i64, C.POINTER(i64)]),
    ("bof_csr_blocks", i64, [P, i64, i64, i64, i64, P, P, i64]),
    ("bof_gemm_resident", C.c_int, [chr_, chr_, chr_, i64, i64, i64, f32, f32, P, P, P, i64, i64,
                                    i64, C.POINTER(Options), P]),
    ("bof_kmeans_resident", C.c_int, [chr_, chr_, chr_, i64, i64, i64, f32, f32, P, P, P, i64, i64,
                                      i64, P, P, P, C.POINTER(Options), P]),
    ("bof_csrmm_resident", C.c_int, [chr_, i64, i64, i64, f32, f32, P, P, P, P, chr_, P, P,
                                     C.POINTER(Options), P]),
    ("bof_csrgemv_resident", C.c_int, [chr_, i64, i64, P, P, P, P, P, P, C.POINTER(Options), P]),
    ("bof_flash_gemm", C.c_int, [chr_, chr_, chr_, u64, u64, u64, f32, f32, FPtr, FPtr, FPtr,
                                 u64, u64, u64, C.POINTER(Options)]),
    ("bof_flash_kmeans", C.c_int, [chr_, chr_, chr_, u64, u64, u64, f32, f32, FPtr, FPtr, FPtr,
                                   u64, u64, u64, P, P, P, C.POINTER(Options)]),
    ("bof_flash_csrmm", C.c_int, [chr_, u64, u64, u64, f32, f32, FPtr, FPtr, FPtr, chr_, FPtr,
                                  FPtr, C.POINTER(Options)]),
    ("bof_flash_csrcsc", C.c_int, [u64, u64, FPtr, FPtr, FPtr, FPtr, FPtr, FPtr, C.POINTER(Options)]),
    ("bof_flash_csrmm_inmem", C.c_int, [chr_, u64, u64, u64, f32, f32, FPtr, FPtr, FPtr, chr_, P, P,
                                        C.POINTER(Options)]),
    ("bof_flash_csrgemv", C.c_int, [chr_, u64, u64, FPtr, FPtr, FPtr, P, P, C.POINTER(Options)]),
    ("bof_file_to_device", C.c_int, [FPtr, u64, P, C.POINTER(Options), P]),
    ("bof_device_to_file", C.c_int, [FPtr, u64, P, C.POINTER(Options), P]),
    ("bof_flash_gemm_panel_plan", C.c_int, [chr_, chr_, chr_, u64, u64, u64, u64, u64, u64, i64, u64, i64,
                                            C.POINTER(PanelPlan)]),
    ("bof_flash_last_stats", C.c_int, [C.POINTER(FlashStats)]),
    ("bof_flash_last_device_stats", C.c_int, [C.POINTER(FlashStats), C.c_int]),
    ("bof_flash_release", C.c_int, []),
    ("bof_share_cleanup", C.c_int, [C.c_char_p]),
    ("bof_share_selftest", i64, [C.c_char_p, C.c_int, C.c_int, i64, i64, C.c_int, C.c_double]),
    ("bof_flash_gemm_simulate", C.c_int, [chr_, chr_, chr_, u64, u64, u64, f32, u64, u64, u64, i64, i64,
                                          C.c_int32, C.POINTER(FlashStats)]),
    ("bof_event_dump", u64, [C.c_char_p]),
    ("bof_flash_last_c_file", C.c_int, [C.POINTER(C.c_uint64)]),
    ("bof_flash_last_launch_mix", C.c_int, [C.POINTER(C.c_uint64)]),
    ("bof_file_sread", C.c_int, [C.c_int, u64, u64, u64, u64, P, C.c_int]),
    ("bof_file_swrite", C.c_int, [C.c_int, u64, u64, u64, u64, P, C.c_int]),
    ("bof_file_forget", C.c_int, [C.c_int]),
    ("bof_file_set_request_bytes", C.c_int, [u64]),
    ("bof_gen_dense", C.c_int, [P, i64, i64, chr_, u64, P]),
    ("bof_gen_sparse_rows", C.c_int, [i64, i64, i64, i64, P, P, P, P]),
]


def build(verbose=False):
    """Compile libbof_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", HERE, "-j8"] + ([] if verbose else ["-s"]), check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        # One HIP runtime per process: torch wheels bundle their own libamdhip64.so.7.
        # Loading torch FIRST makes the dynamic linker satisfy our DT_NEEDED
        # libamdhip64.so.7 with that already-loaded copy (same SONAME), so torch
        # tensors/streams and our kernels share one runtime.  Loaded the other way
        # round the process ends up with two runtimes and torch sees no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise BofError(f"{LIB_PATH} is missing: run `make -C {HERE}` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise BofError(f"{what} failed rc={rc}: {lib().bof_last_error().decode(errors='replace')}")


def _c(ch):
    return ch.encode() if isinstance(ch, str) else ch


def default_options(**kw):
    o = Options()
    lib().bof_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "share_name":
            o.share_name = v.encode() if isinstance(v, str) else v
        elif k == "devices":        # devices=[0, 1, ...]: the in-process device list (repeats allowed)
            v = list(v)
            o.n_devices = len(v)
            for i, d in enumerate(v):
                o.devices[i] = d
        else:
            setattr(o, k, v)
    return o


def require_device():
    n = lib().bof_device_count()
    if n <= 0:
        raise BofError("no HIP device visible: the bof_hip compute path needs an MI355X "
                       "(there is no CPU fallback)")
    return n


# ---- level 1 ---------------------------------------------------------------------
def sgemm(ord_, ta, tb, m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc, stream=0):
    check(lib().bof_sgemm(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta,
                          c_ptr, ldc, stream), "bof_sgemm")


def skmeans_task(ord_, ta, tb, m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc, c_l2sq, p_l2sq, ones,
                 stream=0):
    check(lib().bof_skmeans_task(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta,
                                 c_ptr, ldc, c_l2sq, p_l2sq, ones, stream), "bof_skmeans_task")


def scsrmm(ord_b, m, n, k, alpha, val, col, ptr, b, ldb, beta, c, ldc, stream=0):
    check(lib().bof_scsrmm(_c(ord_b), m, n, k, alpha, val, col, ptr, b, ldb, beta, c, ldc, stream),
          "bof_scsrmm")


def scsrgemv(trans, m, n, val, ptr, col, x, y, stream=0):
    check(lib().bof_scsrgemv(_c(trans), m, n, val, ptr, col, x, y, stream), "bof_scsrgemv")


def scsrcsc(m, n, nnz, val, ptr, col, val_tr, ptr_tr, col_tr, stream=0):
    check(lib().bof_scsrcsc(m, n, nnz, val, ptr, col, val_tr, ptr_tr, col_tr, stream), "bof_scsrcsc")


# ---- planning ----------------------------------------------------------------------
def gemm_plan(ord_, ta, tb, m, n, k, beta, lda, ldb, ldc, blk):
    nblk = (i64 * 3)()
    nt = lib().bof_gemm_plan(_c(ord_), _c(ta), _c(tb), m, n, k, beta, lda, ldb, ldc, blk, None, 0,
                             nblk)
    if nt < 0:
        raise BofError("bof_gemm_plan: bad argument")
    arr = (GemmTask * max(nt, 1))()
    lib().bof_gemm_plan(_c(ord_), _c(ta), _c(tb), m, n, k, beta, lda, ldb, ldc, blk, arr, nt, nblk)
    return list(arr)[:nt], list(nblk)


def csr_blocks(ia, m, min_rows=128, max_rows=131072, max_nnz=10_000_000):
    import numpy as np
    ia = np.ascontiguousarray(ia, np.int64)
    nb = lib().bof_csr_blocks(ia.ctypes.data, m, min_rows, max_rows, max_nnz, None, None, 0)
    if nb < 0:
        raise BofError("bof_csr_blocks: bad argument")
    st = np.empty(nb, np.int64)
    sz = np.empty(nb, np.int64)
    lib().bof_csr_blocks(ia.ctypes.data, m, min_rows, max_rows, max_nnz, st.ctypes.data,
                         sz.ctypes.data, nb)
    return st, sz


# ---- level 2 -----------------------------------------------------------------------
def gemm_resident(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda=0, ldb=0, ldc=0, opts=None,
                  stream=0):
    check(lib().bof_gemm_resident(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, beta, a, b, c, lda, ldb,
                                  ldc, C.byref(opts) if opts is not None else None, stream),
          "bof_gemm_resident")


def kmeans_resident(ord_, ta, tb, m, n, k, alpha, beta, a, b, c, lda, ldb, ldc, c_l2sq, p_l2sq, ones,
                    opts=None, stream=0):
    check(lib().bof_kmeans_resident(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, beta, a, b, c, lda, ldb, ldc,
                                    c_l2sq, p_l2sq, ones, C.byref(opts) if opts is not None else None, stream),
          "bof_kmeans_resident")


def csrmm_resident(trans_a, m, n, k, alpha, beta, val, ia_host, ia_dev, ja, ord_b, b, c, opts=None,
                   stream=0):
    check(lib().bof_csrmm_resident(_c(trans_a), m, n, k, alpha, beta, val, ia_host, ia_dev, ja,
                                   _c(ord_b), b, c, C.byref(opts) if opts is not None else None,
                                   stream), "bof_csrmm_resident")


def csrgemv_resident(trans_a, m, n, val, ia_host, ia_dev, ja, x, y, opts=None, stream=0):
    check(lib().bof_csrgemv_resident(_c(trans_a), m, n, val, ia_host, ia_dev, ja, x, y,
                                     C.byref(opts) if opts is not None else None, stream),
          "bof_csrgemv_resident")


# ---- level 3 -----------------------------------------------------------------------
def flash_gemm(ord_, ta, tb, m, n, k, alpha, beta, fa, fb, fc, lda=0, ldb=0, ldc=0, opts=None):
    check(lib().bof_flash_gemm(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, beta, fa, fb, fc, lda, ldb,
                               ldc, C.byref(opts) if opts is not None else None), "bof_flash_gemm")


def flash_kmeans(ord_, ta, tb, m, n, k, alpha, beta, fa, fb, fc, lda, ldb, ldc, c_l2sq, p_l2sq, ones, opts=None):
    """c_l2sq / p_l2sq / ones: host addresses (numpy .ctypes.data)."""
    check(lib().bof_flash_kmeans(_c(ord_), _c(ta), _c(tb), m, n, k, alpha, beta, fa, fb, fc, lda, ldb, ldc,
                                 c_l2sq, p_l2sq, ones, C.byref(opts) if opts is not None else None),
          "bof_flash_kmeans")


def flash_csrmm(trans_a, m, n, k, alpha, beta, fa, fia, fja, ord_b, fb, fc, opts=None):
    check(lib().bof_flash_csrmm(_c(trans_a), m, n, k, alpha, beta, fa, fia, fja, _c(ord_b), fb, fc,
                                C.byref(opts) if opts is not None else None), "bof_flash_csrmm")


def flash_csrcsc(m, n, fia, fja, fa, fia_tr, fja_tr, fa_tr, opts=None):
    check(lib().bof_flash_csrcsc(m, n, fia, fja, fa, fia_tr, fja_tr, fa_tr,
                                 C.byref(opts) if opts is not None else None), "bof_flash_csrcsc")


def flash_csrmm_inmem(trans_a, m, n, k, alpha, beta, fa, fia, fja, ord_b, b_host, c_host, opts=None):
    check(lib().bof_flash_csrmm_inmem(_c(trans_a), m, n, k, alpha, beta, fa, fia, fja, _c(ord_b), b_host,
                                      c_host, C.byref(opts) if opts is not None else None),
          "bof_flash_csrmm_inmem")


def flash_csrgemv(trans_a, m, n, fa, fia, fja, b_host, c_host, opts=None):
    check(lib().bof_flash_csrgemv(_c(trans_a), m, n, fa, fia, fja, b_host, c_host,
                                  C.byref(opts) if opts is not None else None),
          "bof_flash_csrgemv")


def file_to_device(f, nbytes, dptr, opts=None, stream=0):
    """Blocking; ordered behind the work already queued on `stream` (raw hipStream_t)."""
    check(lib().bof_file_to_device(f, nbytes, dptr, C.byref(opts) if opts is not None else None, stream),
          "bof_file_to_device")


def device_to_file(f, nbytes, dptr, opts=None, stream=0):
    check(lib().bof_device_to_file(f, nbytes, dptr, C.byref(opts) if opts is not None else None, stream),
          "bof_device_to_file")


def flash_gemm_simulate(ord_, ta, tb, m, n, k, beta, blk, n_slots, lookahead=16, lda=0, ldb=0, ldc=0):
    s = FlashStats()
    check(lib().bof_flash_gemm_simulate(_c(ord_), _c(ta), _c(tb), m, n, k, beta, lda, ldb, ldc, blk,
                                        n_slots, lookahead, C.byref(s)), "bof_flash_gemm_simulate")
    return {f: getattr(s, f) for f, _ in s._fields_}


def flash_gemm_panel_plan(ord_, ta, tb, m, n, k, blk, hbm_budget, lda=0, ldb=0, ldc=0, group=1):
    p = PanelPlan()
    check(lib().bof_flash_gemm_panel_plan(_c(ord_), _c(ta), _c(tb), m, n, k, lda, ldb, ldc, blk, hbm_budget, group,
                                          C.byref(p)), "bof_flash_gemm_panel_plan")
    return {"eligible": bool(p.eligible), "why": p.why, "streamed": p.streamed, "resident": list(p.resident),
            "n_panels": list(p.n_panels), "n_slots": list(p.n_slots), "slot_bytes": list(p.slot_bytes),
            "need_bytes": p.need_bytes, "groups": p.groups, "first_group": p.first_group, "acc_bytes": p.acc_bytes}


def flash_last_stats():
    s = FlashStats()
    check(lib().bof_flash_last_stats(C.byref(s)), "bof_flash_last_stats")
    return {f: getattr(s, f) for f, _ in s._fields_}


def flash_last_launch_mix():
    """Compute launches of the last row-panel flash_gemm by kind: chain k-ranges / whole-K panels / whole-K row slices."""
    out = (C.c_uint64 * 3)()
    check(lib().bof_flash_last_launch_mix(out), "bof_flash_last_launch_mix")
    return {"chain_k_ranges": out[0], "whole_k_panels": out[1], "whole_k_row_slices": out[2]}


def flash_last_c_file():
    """How the last flash_csrmm call treated its C file: (mode, bytes of whole row blocks through the buffered twin);
    mode 1 O_DIRECT, 2 O_DIRECT with widened reads / page-split writes, 0 buffered twin, -1 not an O_DIRECT file."""
    tw = C.c_uint64(0)
    mode = lib().bof_flash_last_c_file(C.byref(tw))
    return mode, tw.value


def flash_last_device_stats():
    """Per-device counters of the last level-3 call, in device-list order."""
    arr = (FlashStats * 16)()
    n = lib().bof_flash_last_device_stats(arr, 16)
    if n < 0:
        raise BofError("bof_flash_last_device_stats failed")
    return [{f: getattr(arr[i], f) for f, _ in FlashStats._fields_} for i in range(min(n, 16))]


# ---- generators --------------------------------------------------------------------
def gen_dense(ptr, first, count, mode="s", seed=0, stream=0):
    check(lib().bof_gen_dense(ptr, first, count, _c(mode), seed, stream), "bof_gen_dense")


def gen_sparse_rows(row0, nrows, ncols, nnz_per_row, csr, col, off, stream=0):
    check(lib().bof_gen_sparse_rows(row0, nrows, ncols, nnz_per_row, csr, col, off, stream),
          "bof_gen_sparse_rows")
