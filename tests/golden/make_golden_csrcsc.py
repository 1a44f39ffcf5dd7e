#!/usr/bin/env python3
"""Golden vectors for the transposition row of the scope table:
flash::csrcsc and csrmm with trans_a='T'.  Run ONCE in the build container; the
output (mkl_golden_csrcsc.npz) is committed together with this script.

The reference transposes a row block with mkl_csrcsc(job={0,0,0,-1,-1,1}) on the
block padded to a square of edge max(nrows, ncols) (include/tasks/csrcsc_task.h:
45-78) and multiplies with mkl_scsrmm('T', "GXXC") (include/tasks/csrmm_task.h).
Both are called here through the MKL 2021.4 ILP64 runtime in /opt/conda/lib with
exactly those argument conventions.  Inputs are stored beside the outputs.
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
import orc  # noqa: E402

mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so", mode=C.RTLD_GLOBAL)
i64, f32, P = C.c_int64, C.c_float, C.c_void_p


def p(a):
    return a.ctypes.data_as(P)


def ver():
    buf = C.create_string_buffer(256)
    mkl.MKL_Get_Version_String(buf, 256)
    return buf.value.decode()


def mkl_csrcsc(m, n, val, ia, ja):
    """csrcsc_task.h:45-78: offsets padded with empty rows up to pdim = max(m, n)."""
    pdim = max(m, n)
    iap = np.empty(pdim + 1, np.int64)
    iap[: m + 1] = ia - ia[0]
    iap[m + 1:] = iap[m]
    nnz = int(iap[m])
    job = np.array([0, 0, 0, -1, -1, 1], np.int64)
    val_tr = np.zeros(max(nnz, 1), np.float32)
    ja_tr = np.zeros(max(nnz, 1), np.int64)
    ia_tr = np.zeros(pdim + 1, np.int64)
    dim, info = i64(pdim), i64(-1)
    v = val.copy() if nnz else np.zeros(1, np.float32)
    j = ja.copy() if nnz else np.zeros(1, np.int64)
    mkl.mkl_scsrcsc(p(job), C.byref(dim), p(v), p(j), p(iap), p(val_tr), p(ja_tr), p(ia_tr),
                    C.byref(info))
    return val_tr[:nnz], ia_tr[: n + 1].copy(), ja_tr[:nnz]


def mkl_scsrmm_t(m, n, k, alpha, val, col, ptr, b, beta, c):
    """C (n x k) = alpha * A^T (A is m x n CSR) * B (m x k) + beta * C, row-major "GXXC"."""
    desc = C.create_string_buffer(b"GXXC", 6)
    tr = C.c_char(b"T")
    M, N, K, LDB, LDC = i64(m), i64(k), i64(n), i64(k), i64(k)
    al, be = f32(alpha), f32(beta)
    ptr = ptr - ptr[0]
    ptre = ptr[1:]
    mkl.mkl_scsrmm(C.byref(tr), C.byref(M), C.byref(N), C.byref(K), C.byref(al), desc,
                   p(val), p(col), p(ptr), p(ptre), p(b), C.byref(LDB), C.byref(be), p(c),
                   C.byref(LDC))


def rand_csr(rng, m, n, density, empty_rows=()):
    ia = [0]
    ja = []
    for r in range(m):
        cnt = 0 if r in empty_rows else rng.binomial(n, density)
        cols = np.sort(rng.choice(n, size=cnt, replace=False))
        ja.append(cols)
        ia.append(ia[-1] + cnt)
    ja = np.concatenate(ja).astype(np.int64) if ja else np.zeros(0, np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    return val, np.array(ia, np.int64), ja


def main():
    rng = np.random.default_rng(20260211)
    out = {}
    meta = ["mkl=" + ver()]
    shapes = [(300, 700, 0.02, (0, 17, 299)), (640, 200, 0.03, ()), (257, 70000, 0.0005, (5,)),
              (1, 50, 0.3, ()), (40, 1, 0.5, ())]
    for mi, (m, n, dens, empty) in enumerate(shapes):
        val, ia, ja = rand_csr(rng, m, n, dens, empty)
        vt, iat, jat = mkl_csrcsc(m, n, val, ia, ja)
        key = f"tr{mi}"
        out[key + "_val"], out[key + "_ia"], out[key + "_ja"] = val, ia, ja
        out[key + "_val_tr"], out[key + "_ia_tr"], out[key + "_ja_tr"] = vt, iat, jat
        meta.append(f"{key} {m} {n}")
        if mi < 3:
            for ci, (k, alpha, beta) in enumerate([(40, 1.0, 0.0), (128, 0.5, 2.0)]):
                b = rng.uniform(-1, 1, (m, k)).astype(np.float32)
                c0 = rng.uniform(-1, 1, (n, k)).astype(np.float32)
                c1 = c0.copy()
                mkl_scsrmm_t(m, n, k, alpha, val, ja, ia, b, beta, c1)
                ck = f"csrmmT{mi}{ci}"
                if n > 10000:   # keep the fixture small: store rows 0..255 of C only
                    c0s, c1s = c0[:256].copy(), c1[:256].copy()
                else:
                    c0s, c1s = c0, c1
                out[ck + "_b"], out[ck + "_c0"], out[ck + "_c1"] = b, c0s, c1s
                meta.append(f"{ck} {m} {n} {k} {alpha} {beta} {key} {c0s.shape[0]}")

    # integer data on the reference generator: exact, pinned by hash
    val, ja, ia = orc.sparse_create(4096, 2048, 0.01)
    vt, iat, jat = mkl_csrcsc(4096, 2048, val, ia, ja)
    for name, arr in (("gen_tr_val", vt), ("gen_tr_ia", iat), ("gen_tr_ja", jat)):
        meta.append(f"exact {name} {hashlib.sha256(arr.tobytes()).hexdigest()} "
                    + " ".join(str(float(v)) for v in arr.ravel()[:4]))
    b = orc.dense_fill(4096, 128, "s")
    c = np.zeros((2048, 128), np.float32)
    mkl_scsrmm_t(4096, 2048, 128, 1.0, val, ja, ia, b, 0.0, c)
    meta.append(f"exact gen_csrmmT_c {hashlib.sha256(c.tobytes()).hexdigest()} "
                + " ".join(str(float(v)) for v in c.ravel()[:4]))

    out["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "mkl_golden_csrcsc.npz"), **out)
    print("wrote mkl_golden_csrcsc.npz", len(out), "arrays;", ver())


if __name__ == "__main__":
    main()
