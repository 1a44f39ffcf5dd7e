#!/usr/bin/env python3
"""MKL goldens for the kernels that carry the benchmark (tests/golden/mkl_golden_big.npz).

The small fixtures of make_golden_mkl.py (<= 150 in every dimension) only reach the guarded
128x128 kernel.  Here cblas_sgemm (MKL 2021.4, ILP64 -- the routine the reference calls at
include/tasks/gemm_task.h:87-90 and drivers/in_mem_gemm.cpp:64-67) multiplies 4096 x 2048 x 1024
problems for all 8 (order, transA, transB) combinations, alpha = 0.5, beta = 2, on uniform
[-1,1) inputs -- large and aligned enough for the 256x256 MFMA kernels (LDS-DMA and register
staging) and, through bof_gemm_resident with a 1024 tile, for 8-task accumulate chains.

The INPUTS are not stored: they come from the library's counter-based generator
(bof_gen_dense mode 'u', blas-on-flash_amd/csrc/gen_kernels.hip), restated below in numpy
(`dense_u`; tests check that the two agree bit for bit).  Stored: six 64 x 64 sub-blocks of
MKL's C per case (corners, tile boundaries, interior), ~100 KB per case.
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("MKL_INTERFACE_LAYER", "ILP64")
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from gen_u import dense_u  # noqa: E402

M, N, K = 4096, 2048, 1024
ALPHA, BETA = 0.5, 2.0
SEEDS = (11, 12, 13)                     # a, b, c0
# (row, col) origins of the 64 x 64 sub-blocks of the LOGICAL m x n result
BLOCKS = [(0, 0), (M - 64, N - 64), (224, 992), (1024 - 32, 1024 - 32), (2048, 256), (3333, 1777)]


def stored_shapes(ord_, ta, tb):
    a = (M, K) if (ta == "T") == (ord_ == "C") else (K, M)
    b = (K, N) if (tb == "T") == (ord_ == "C") else (N, K)
    c = (M, N) if ord_ == "R" else (N, M)
    return a, b, c


def main():
    mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so", mode=C.RTLD_GLOBAL)
    i64, f32, P = C.c_int64, C.c_float, C.c_void_p
    mkl.cblas_sgemm.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, f32, P, i64, P, i64, f32, P, i64]
    buf = C.create_string_buffer(256)
    mkl.MKL_Get_Version_String(buf, 256)
    out, meta = {}, ["mkl=" + buf.value.decode(), f"shape {M} {N} {K} alpha {ALPHA} beta {BETA} seeds {SEEDS}"]
    for ord_ in "RC":
        for ta in "NT":
            for tb in "NT":
                sa, sb, sc = stored_shapes(ord_, ta, tb)
                a = dense_u(0, sa[0] * sa[1], SEEDS[0]).reshape(sa)
                b = dense_u(0, sb[0] * sb[1], SEEDS[1]).reshape(sb)
                c = dense_u(0, sc[0] * sc[1], SEEDS[2]).reshape(sc).copy()
                mkl.cblas_sgemm(101 if ord_ == "R" else 102, 112 if ta == "T" else 111, 112 if tb == "T" else 111,
                                M, N, K, ALPHA, a.ctypes.data_as(P), sa[1], b.ctypes.data_as(P), sb[1], BETA,
                                c.ctypes.data_as(P), sc[1])
                logical = c if ord_ == "R" else c.T       # m x n view
                key = f"{ord_}{ta}{tb}"
                out[key] = np.stack([np.ascontiguousarray(logical[r:r + 64, q:q + 64]) for r, q in BLOCKS])
                # float64 check of the fixture itself
                a64 = (a if sa == (M, K) else a.T).astype(np.float64)
                b64 = (b if sb == (K, N) else b.T).astype(np.float64)
                c0 = dense_u(0, sc[0] * sc[1], SEEDS[2]).reshape(sc)
                c0l = (c0 if ord_ == "R" else c0.T).astype(np.float64)
                worst = 0.0
                for (r, q), blk in zip(BLOCKS, out[key]):
                    ref = ALPHA * a64[r:r + 64] @ b64[:, q:q + 64] + BETA * c0l[r:r + 64, q:q + 64]
                    worst = max(worst, float(np.abs(blk - ref).max() / np.abs(ref).max()))
                meta.append(f"{key} rel_err_vs_float64 {worst:.3e}")
                assert worst < 1e-5
    out["blocks"] = np.array(BLOCKS, np.int64)
    out["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "mkl_golden_big.npz"), **out)
    print("\n".join(meta))


if __name__ == "__main__":
    main()
