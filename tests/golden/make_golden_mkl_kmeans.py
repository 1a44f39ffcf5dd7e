#!/usr/bin/env python3
"""Golden vectors for flash::kmeans' tile task (run ONCE in the build container; the .npz is
committed with this script).

KMeansTask::execute (reference include/tasks/kmeans_task.h:53-82) is three cblas_sgemm calls:
the tile product, then two K = 1 products with alpha = beta = 1 that add c_l2sq[r]*ones[c] and
ones[r]*p_l2sq[c].  The fixtures are produced by making exactly those three calls into the
MKL 2021.4 runtime of this container (ILP64, as the reference links it), in column-major order
-- the only order the reference's driver uses (drivers/kmeans.cpp:37-39; for row-major the
task's leading dimensions read out of bounds, see oracle/bof_oracle.c) -- for the four
transposition pairs, the driver's own call shape ('C','T','N', alpha = -2, beta = 0: squared
distances between centers and points) among them.  Inputs are stored too.
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("MKL_INTERFACE_LAYER", "ILP64")
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
HERE = os.path.dirname(os.path.abspath(__file__))

mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so", mode=C.RTLD_GLOBAL)
i64, f32, P = C.c_int64, C.c_float, C.c_void_p
mkl.cblas_sgemm.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, f32, P, i64, P, i64, f32, P, i64]


def p(a):
    return a.ctypes.data_as(P)


def sgemm_cm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    mkl.cblas_sgemm(102, 112 if ta == "T" else 111, 112 if tb == "T" else 111, m, n, k, alpha, p(a), lda,
                    p(b), ldb, beta, p(c), ldc)


def kmeans_task_cm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, c_l2sq, p_l2sq, ones):
    """kmeans_task.h:68-81, argument for argument."""
    sgemm_cm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc)
    sgemm_cm("N", "T", m, n, 1, 1.0, c_l2sq, m, ones, n, 1.0, c, ldc)
    sgemm_cm("N", "T", m, n, 1, 1.0, ones, m, p_l2sq, n, 1.0, c, ldc)


def main():
    rng = np.random.default_rng(20260)
    out = {}
    cases = []
    # (ta, tb, m, n, k, alpha, beta, pad): column-major; pad = extra leading-dimension elements
    for ta, tb in (("T", "N"), ("N", "N"), ("N", "T"), ("T", "T")):
        cases.append((ta, tb, 37, 53, 29, -2.0, 0.0, 0))
        cases.append((ta, tb, 130, 150, 64, 0.5, 2.0, 3))
    cases.append(("T", "N", 64, 300, 48, -2.0, 0.0, 0))     # drivers/kmeans.cpp:37-39: centers x points
    cases.append(("T", "N", 1, 17, 5, -2.0, 0.0, 0))
    cases.append(("T", "N", 19, 1, 7, -2.0, 1.0, 0))
    for idx, (ta, tb, m, n, k, alpha, beta, pad) in enumerate(cases):
        # column-major: A is (ta == 'N' ? m x k : k x m) with lda >= rows
        ra, ca = (m, k) if ta == "N" else (k, m)
        rb, cb = (k, n) if tb == "N" else (n, k)
        lda, ldb, ldc = ra + pad, rb + pad, m + pad
        a = rng.uniform(-1, 1, lda * ca).astype(np.float32)
        b = rng.uniform(-1, 1, ldb * cb).astype(np.float32)
        c0 = rng.uniform(-1, 1, ldc * n).astype(np.float32)
        cl = rng.uniform(0, 4, m).astype(np.float32)
        pl = rng.uniform(0, 4, n).astype(np.float32)
        for variant in ("ones", "weights"):
            # "weights": a caller-supplied vector that is not all ones pins which factor goes where
            ones = np.ones(max(m, n), np.float32) if variant == "ones" else rng.uniform(0.5, 1.5, max(m, n)).astype(np.float32)
            c = c0.copy()
            kmeans_task_cm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, cl, pl, ones)
            key = f"k{idx}_{variant}"
            out[key + "_meta"] = np.array([ord(ta), ord(tb), m, n, k, lda, ldb, ldc], np.int64)
            out[key + "_ab"] = np.array([alpha, beta], np.float32)
            for name, arr in (("a", a), ("b", b), ("c0", c0), ("cl", cl), ("pl", pl), ("ones", ones), ("c", c)):
                out[f"{key}_{name}"] = arr
    buf = C.create_string_buffer(256)
    mkl.MKL_Get_Version_String(buf, 256)
    out["mkl_version"] = np.frombuffer(buf.value, np.uint8)
    np.savez_compressed(os.path.join(HERE, "mkl_golden_kmeans.npz"), **out)
    print(len(cases) * 2, "cases;", buf.value.decode())


if __name__ == "__main__":
    main()
