#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run ONCE in the build
container; the outputs are committed, this script is committed with them).

The reference's arithmetic for the hot path is three Intel MKL routines
(SURVEY.md 2.2): cblas_sgemm (include/tasks/gemm_task.h:87-90,
drivers/in_mem_gemm.cpp:64-67), mkl_scsrmm (include/tasks/csrmm_task.h:226-228,
drivers/in_mem_csrmm.cpp:116-121) and mkl_cspblas_scsrgemv
(include/tasks/csrgemv_task.h:74,165; drivers/in_mem_csrgemv.cpp).  MKL is a
closed-source third-party dependency that is not vendored in the reference tree;
the build container carries MKL 2021.4 runtime libraries (/opt/conda/lib,
ILP64), so the goldens are produced by calling those very routines with the
argument conventions of the reference's call sites (ILP64 integers, "GXXC" /
"GXXF" descriptors, 1-based conversion for column-major exactly as
drivers/in_mem_csrmm.cpp:100-114, zero-padding to a square matrix exactly as
drivers/in_mem_csrgemv.cpp / csrgemv_task.h:36-44).

Inputs are stored in the fixtures too, so nothing depends on RNG stability.
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

os.environ.setdefault("MKL_INTERFACE_LAYER", "ILP64")
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402  (generator restatement, KAT-pinned)

mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so", mode=C.RTLD_GLOBAL)
i64, f32, P = C.c_int64, C.c_float, C.c_void_p
mkl.cblas_sgemm.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, f32, P, i64, P, i64,
                            f32, P, i64]


def ver():
    buf = C.create_string_buffer(256)
    mkl.MKL_Get_Version_String(buf, 256)
    return buf.value.decode()


def p(a):
    return a.ctypes.data_as(P)


def mkl_sgemm(ord_, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    mkl.cblas_sgemm(101 if ord_ == "R" else 102, 112 if ta == "T" else 111,
                    112 if tb == "T" else 111, m, n, k, alpha, p(a), lda, p(b), ldb, beta,
                    p(c), ldc)


def mkl_scsrmm(ord_b, m, n, k, alpha, val, col, ptr, b, ldb, beta, c, ldc):
    """drivers/in_mem_csrmm.cpp:96-121: 'R' -> "GXXC" 0-based; 'C' -> "GXXF" with
    offsets and indices converted to 1-based."""
    col = col.copy()
    ptr = ptr.copy()
    if ord_b == "C":
        ptr = ptr - ptr[0] + 1
        col = col + 1
    desc = C.create_string_buffer(b"GXXF" if ord_b == "C" else b"GXXC", 6)
    tr = C.c_char(b"N")
    M, N, K, LDB, LDC = i64(m), i64(n), i64(k), i64(ldb), i64(ldc)
    al, be = f32(alpha), f32(beta)
    ptre = ptr[1:]
    mkl.mkl_scsrmm(C.byref(tr), C.byref(M), C.byref(N), C.byref(K), C.byref(al), desc,
                   p(val), p(col), p(ptr), p(ptre), p(b), C.byref(LDB), C.byref(be), p(c),
                   C.byref(LDC))


def mkl_scsrgemv(trans, m, n, val, ia, ja, x):
    """csrgemv_task.h:36-44,126-134: pad offsets to dim=max(m,n)+1 entries, pad
    x to dim; result y has dim entries (caller slices)."""
    dim = max(m, n)
    iap = np.empty(dim + 1, np.int64)
    iap[: m + 1] = ia - ia[0]
    iap[m + 1:] = iap[m]
    xin = np.zeros(dim, np.float32)
    xin[: x.size] = x
    y = np.zeros(dim, np.float32)
    tr = C.c_char(trans.encode())
    D = i64(dim)
    mkl.mkl_cspblas_scsrgemv(C.byref(tr), C.byref(D), p(val), p(iap), p(ja), p(xin), p(y))
    return y


def rand_csr(rng, m, n, density):
    """Random CSR with sorted unique columns per row and random fp32 values."""
    ia = [0]
    ja = []
    for _ in range(m):
        cnt = rng.binomial(n, density)
        cols = np.sort(rng.choice(n, size=cnt, replace=False))
        ja.append(cols)
        ia.append(ia[-1] + cnt)
    ja = np.concatenate(ja).astype(np.int64) if ja else np.zeros(0, np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    return val, np.array(ia, np.int64), ja


def main():
    rng = np.random.default_rng(20260210)
    out = {}
    meta = ["mkl=" + ver()]

    # ---- gemm: 8 layouts x 2 (alpha,beta) x 2 shapes, unaligned leading dims ----
    cases = []
    for ord_ in "RC":
        for ta in "NT":
            for tb in "NT":
                for (alpha, beta) in [(1.0, 0.0), (0.5, 2.0)]:
                    cases.append((64, 80, 48, ord_, ta, tb, alpha, beta))
    for (ord_, ta, tb) in [("R", "N", "N"), ("R", "T", "N"), ("C", "N", "T"), ("C", "T", "T")]:
        cases.append((130, 150, 140, ord_, ta, tb, 0.5, 2.0))
    for idx, (m, n, k, ord_, ta, tb, alpha, beta) in enumerate(cases):
        # stored shapes exactly as cblas interprets them
        ar, ac = (m, k) if (ta == "T") == (ord_ == "C") else (k, m)
        br, bc = (k, n) if (tb == "T") == (ord_ == "C") else (n, k)
        cr, cc = (m, n) if ord_ == "R" else (n, m)
        lda, ldb, ldc = ac + 3, bc + 5, cc + 7   # padded, non-multiple-of-128 LDs
        a = rng.uniform(-1, 1, (ar, lda)).astype(np.float32)
        b = rng.uniform(-1, 1, (br, ldb)).astype(np.float32)
        c0 = rng.uniform(-1, 1, (cr, ldc)).astype(np.float32)
        c1 = c0.copy()
        mkl_sgemm(ord_, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c1, ldc)
        key = f"gemm{idx:02d}"
        out[key + "_a"], out[key + "_b"], out[key + "_c0"], out[key + "_c1"] = a, b, c0, c1
        meta.append(f"{key} {m} {n} {k} {ord_} {ta} {tb} {alpha} {beta} {lda} {ldb} {ldc}")

    # ---- csrmm: random CSR x dense, R and C layouts ---------------------------
    idx = 0
    for mi, (m, n, k, dens) in enumerate([(300, 700, 40, 0.02), (257, 600, 128, 0.01)]):
        val, ia, ja = rand_csr(rng, m, n, dens)
        out[f"csrmat{mi}_val"], out[f"csrmat{mi}_ia"], out[f"csrmat{mi}_ja"] = val, ia, ja
        for ord_b in "RC":
            for (alpha, beta) in [(1.0, 0.0), (0.5, 2.0)]:
                if ord_b == "R":
                    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)
                    c0 = rng.uniform(-1, 1, (m, k)).astype(np.float32)
                    ldb, ldc = k, k
                else:
                    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)   # col-major n x k
                    c0 = rng.uniform(-1, 1, (k, m)).astype(np.float32)  # col-major m x k
                    ldb, ldc = n, m
                c1 = c0.copy()
                # MKL naming: m rows, n = dense cols (k here), k = A cols (n here)
                mkl_scsrmm(ord_b, m, k, n, alpha, val, ja, ia, b, ldb, beta, c1, ldc)
                key = f"csrmm{idx:02d}"
                out[key + "_b"], out[key + "_c0"], out[key + "_c1"] = b, c0, c1
                meta.append(f"{key} {m} {n} {k} {ord_b} {alpha} {beta} csrmat{mi}")
                idx += 1

    # ---- csrgemv: N and T, rectangular both ways -------------------------------
    idx = 0
    for mi, (m, n, dens) in enumerate([(400, 300, 0.03), (250, 900, 0.02)]):
        val, ia, ja = rand_csr(rng, m, n, dens)
        out[f"gemvmat{mi}_val"], out[f"gemvmat{mi}_ia"], out[f"gemvmat{mi}_ja"] = val, ia, ja
        for trans in "NT":
            x = rng.uniform(-1, 1, n if trans == "N" else m).astype(np.float32)
            y = mkl_scsrgemv(trans, m, n, val, ia, ja, x)[: (m if trans == "N" else n)]
            key = f"csrgemv{idx:02d}"
            out[key + "_x"], out[key + "_y"] = x, y
            meta.append(f"{key} {m} {n} {trans} gemvmat{mi}")
            idx += 1

    # ---- integer-data goldens on the reference generators (exact in fp32) -----
    # sparse_create(4096,2048,0.01) x dense_create(2048,128,'s') through mkl_scsrmm
    val, ja, ia = orc.sparse_create(4096, 2048, 0.01)
    b = orc.dense_fill(2048, 128, "s")
    c = np.zeros((4096, 128), np.float32)
    mkl_scsrmm("R", 4096, 128, 2048, 1.0, val, ja, ia, b, 128, 0.0, c, 128)
    exact = {"gen_csrmm_c": c}
    x = (np.arange(2048) % 10).astype(np.float32)
    exact["gen_csrgemv_N"] = mkl_scsrgemv("N", 4096, 2048, val, ia, ja, x)[:4096]
    x = (np.arange(4096) % 10).astype(np.float32)
    exact["gen_csrgemv_T"] = mkl_scsrgemv("T", 4096, 2048, val, ia, ja, x)[:2048]
    a = orc.dense_fill(512, 512, "s")
    c = np.zeros((512, 512), np.float32)
    mkl_sgemm("R", "N", "N", 512, 512, 512, 1.0, a, 512, a, 512, 0.0, c, 512)
    exact["gen_gemm512_c"] = c
    # integer-valued results are exact in fp32 under any summation order, so a
    # hash pins them (plus a few leading values for debuggability)
    for name, arr in exact.items():
        meta.append(f"exact {name} {hashlib.sha256(arr.tobytes()).hexdigest()} "
                    + " ".join(str(float(v)) for v in arr.ravel()[:4]))

    out["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "mkl_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "mkl_golden.npz"), len(out), "arrays;", ver())


if __name__ == "__main__":
    main()
