#!/usr/bin/env python3
"""MKL goldens for (a) the reference's own accuracy check and (b) special values / quick returns.

Run ONCE in the build container (MKL 2021.4 runtime under /opt/conda/lib, ILP64); the outputs are
committed with this script: tests/golden/mkl_golden_special.npz, tests/golden/mkl_golden_gemm_run.npz.

(a) misc/gemm_run.sh:21-23 -- the only place the reference pins gemm results: uniform [0,1) 3072 x 3072
    inputs (np.random.rand), alpha = 1, beta = 0, all eight (transA, transB, order) combinations, flash
    driver against in-memory driver, `max(|a - b| / b)` ELEMENT-WISE.  Here: cblas_sgemm on the same
    configuration.  Inputs are not stored: A = |u(seed 21)|, B = |u(seed 22)| with u the library's
    counter-based generator (tests/gen_u.py = bof_gen_dense mode 'u'; |.| of a uniform [-1,1) value is
    uniform [0,1] and exact).  Stored per layout: eight 64 x 64 sub-blocks of MKL's C, and the float64
    row sums and column sums of the whole C (every element is covered by one of each).
(b) the BLAS edge semantics at the reference's three MKL call sites (include/tasks/gemm_task.h:87-90,
    csrmm_task.h:226-228, csrgemv_task.h:74,165): alpha == 0 with NaN in A / B (cblas_sgemm does not
    reference A and B: C = beta*C; mkl_scsrmm has no such path: 0 * NaN reaches C), beta == 0 with NaN in C
    (C is overwritten), k == 0, Inf * 0, denormal operands and results (not flushed), negative zeros,
    explicit zero CSR values (multiplied like any other), empty rows.  Inputs and outputs stored.
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("MKL_INTERFACE_LAYER", "ILP64")
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from gen_u import dense_u  # noqa: E402
import make_golden_mkl as mg  # noqa: E402  (ctypes wrappers with the reference's call-site conventions)

NAN, INF = np.float32("nan"), np.float32("inf")
DIM = 3072
RUN_BLOCKS = [(0, 0), (DIM - 64, DIM - 64), (0, DIM - 64), (DIM - 64, 0), (992, 224), (1024 - 32, 2048 - 32),
              (1500, 1500), (2999, 77)]


def gemm_run():
    out, meta = {}, ["mkl=" + mg.ver(), f"gemm_run.sh configuration: {DIM}^3, alpha 1, beta 0, seeds 21 22"]
    a = np.abs(dense_u(0, DIM * DIM, 21)).reshape(DIM, DIM)
    b = np.abs(dense_u(0, DIM * DIM, 22)).reshape(DIM, DIM)
    for ord_ in "RC":
        for ta in "NT":
            for tb in "NT":
                c = np.full((DIM, DIM), NAN, np.float32)       # beta = 0: never read
                mg.mkl_sgemm(ord_, ta, tb, DIM, DIM, DIM, 1.0, a, DIM, b, DIM, 0.0, c, DIM)
                logical = c if ord_ == "R" else c.T
                key = f"{ord_}{ta}{tb}"
                out[key + "_blocks"] = np.stack([np.ascontiguousarray(logical[r:r + 64, q:q + 64]) for r, q in RUN_BLOCKS])
                out[key + "_rowsum"] = logical.astype(np.float64).sum(axis=1)
                out[key + "_colsum"] = logical.astype(np.float64).sum(axis=0)
                # the stored matrices as cblas interprets them -> logical operands, float64 check of the fixture
                al = (a if (ta == "T") == (ord_ == "C") else a.T).astype(np.float64)
                bl = (b if (tb == "T") == (ord_ == "C") else b.T).astype(np.float64)
                worst = 0.0
                for (r, q), blk in zip(RUN_BLOCKS, out[key + "_blocks"]):
                    ref = al[r:r + 64] @ bl[:, q:q + 64]
                    worst = max(worst, float((np.abs(blk - ref) / np.abs(ref)).max()))
                meta.append(f"{key} elementwise_rel_err_vs_float64 {worst:.3e}")
                assert worst < 1e-5
    out["blocks"] = np.array(RUN_BLOCKS, np.int64)
    out["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "mkl_golden_gemm_run.npz"), **out)
    print("\n".join(meta))


def special():
    rng = np.random.default_rng(20261003)
    out, names = {}, []

    # ---- cblas_sgemm ------------------------------------------------------------------------------
    def gemm_case(name, ord_, ta, tb, m, n, k, alpha, beta, a, b, c):
        lda, ldb, ldc = a.shape[1], b.shape[1], c.shape[1]
        got = c.copy()
        mg.mkl_sgemm(ord_, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, got, ldc)
        key = "gemm_" + name
        names.append(key)
        out[key + "_args"] = np.array([ord(ord_), ord(ta), ord(tb), m, n, k, lda, ldb, ldc], np.int64)
        out[key + "_ab"] = np.array([alpha, beta], np.float32)
        out[key + "_a"], out[key + "_b"], out[key + "_c0"], out[key + "_c"] = a, b, c, got

    def shapes(ord_, ta, tb, m, n, k):
        ar = (m, max(k, 1)) if (ta == "T") == (ord_ == "C") else (max(k, 1), m)
        br = (max(k, 1), n) if (tb == "T") == (ord_ == "C") else (n, max(k, 1))
        cr = (m, n) if ord_ == "R" else (n, m)
        return ar, br, cr

    for (ord_, ta, tb) in (("R", "N", "N"), ("C", "T", "N"), ("R", "T", "T")):
        for (m, n, k) in ((40, 24, 32), (160, 136, 72)):
            tag = f"{ord_}{ta}{tb}_{m}"
            ar, br, cr = shapes(ord_, ta, tb, m, n, k)
            a = rng.uniform(-1, 1, ar).astype(np.float32)
            b = rng.uniform(-1, 1, br).astype(np.float32)
            c = rng.uniform(-1, 1, cr).astype(np.float32)
            an, bn, cn = a.copy(), b.copy(), c.copy()
            an[1, 2] = NAN
            an[3, 0] = INF
            bn[2, 1] = NAN
            bn[0, 3] = -INF
            cn[1, 1] = NAN
            cn[2, 3] = INF
            gemm_case("alpha0_nanAB_beta2_" + tag, ord_, ta, tb, m, n, k, 0.0, 2.0, an, bn, c)
            if m > 100:      # the larger shape (several 128 x 128 workgroups) only for the cases above / below
                gemm_case("nan_inf_propagate_" + tag, ord_, ta, tb, m, n, k, 1.0, 0.5, an, bn, c)
                gemm_case("beta0_nanC_" + tag, ord_, ta, tb, m, n, k, 0.75, 0.0, a, b, cn)
                continue
            gemm_case("alpha0_nanAB_beta0_" + tag, ord_, ta, tb, m, n, k, 0.0, 0.0, an, bn, cn)
            gemm_case("alpha0_beta1_nanC_" + tag, ord_, ta, tb, m, n, k, 0.0, 1.0, an, bn, cn)
            gemm_case("beta0_nanC_" + tag, ord_, ta, tb, m, n, k, 0.75, 0.0, a, b, cn)
            gemm_case("nan_inf_propagate_" + tag, ord_, ta, tb, m, n, k, 1.0, 0.5, an, bn, c)
            gemm_case("k0_beta2_" + tag, ord_, ta, tb, m, n, 0, 1.0, 2.0, an[:, :1].copy() if ar[1] == 1 else an, bn, c)
            gemm_case("k0_beta0_nanC_" + tag, ord_, ta, tb, m, n, 0, 1.0, 0.0, an, bn, cn)
            # Inf * 0 -> NaN
            ai, bz = a.copy(), b.copy()
            ai[0, :] = INF
            bz[:, :] = 0
            gemm_case("inf_times_zero_" + tag, ord_, ta, tb, m, n, k, 1.0, 0.0, ai, bz, c)
            # denormal operands (sums stay denormal) and denormal results of normal operands
            ad = (a * np.float32(1e-39)).astype(np.float32)
            gemm_case("denormal_operand_" + tag, ord_, ta, tb, m, n, k, 1.0, 0.0, ad, b, c)
            a20 = (a * np.float32(1e-20)).astype(np.float32)
            b20 = (b * np.float32(1e-20)).astype(np.float32)
            gemm_case("denormal_result_" + tag, ord_, ta, tb, m, n, k, 1.0, 0.0, a20, b20, c)
            # negative zeros: -0 operands, alpha = -1 on a zero product
            az = np.full(ar, -0.0, np.float32)
            gemm_case("negzero_operand_" + tag, ord_, ta, tb, m, n, k, 1.0, 0.0, az, np.abs(b), c)
            gemm_case("negalpha_zero_product_" + tag, ord_, ta, tb, m, n, k, -1.0, 0.0, np.zeros(ar, np.float32), np.abs(b), c)

    # ---- mkl_scsrmm / mkl_cspblas_scsrgemv --------------------------------------------------------
    def csr(m, n, density):
        val, ia, ja = mg.rand_csr(rng, m, n, density)
        return val, ia, ja

    def mm_case(name, ord_b, m, n, k, alpha, beta, val, ia, ja, b, c):
        got = c.copy()
        ldb = k if ord_b == "R" else n
        ldc = k if ord_b == "R" else m
        mg.mkl_scsrmm(ord_b, m, k, n, alpha, val, ja, ia, b, ldb, beta, got, ldc)
        key = "csrmm_" + name
        names.append(key)
        out[key + "_args"] = np.array([ord(ord_b), m, n, k], np.int64)
        out[key + "_ab"] = np.array([alpha, beta], np.float32)
        out[key + "_val"], out[key + "_ia"], out[key + "_ja"] = val, ia, ja
        out[key + "_b"], out[key + "_c0"], out[key + "_c"] = b, c, got

    def mv_case(name, trans, m, n, val, ia, ja, x):
        y = mg.mkl_scsrgemv(trans, m, n, val, ia, ja, x)[: (m if trans == "N" else n)]
        key = "csrgemv_" + name
        names.append(key)
        out[key + "_args"] = np.array([ord(trans), m, n], np.int64)
        out[key + "_val"], out[key + "_ia"], out[key + "_ja"], out[key + "_x"], out[key + "_y"] = val, ia, ja, x, y

    m, n = 96, 160
    val, ia, ja = csr(m, n, 0.08)
    # empty rows: cut rows 5 and 6 out; explicit zeros: every 7th stored value
    keep = np.ones(ja.size, bool)
    keep[ia[5]:ia[7]] = False
    cnt = np.diff(ia)
    cnt[5:7] = 0
    val, ja = val[keep], ja[keep]
    ia = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    val[::7] = 0.0
    hit = np.unique(ja[::7])            # B rows / x entries met by an explicit zero
    for ord_b in "RC":
        for k in (8, 130):
            tag = f"{ord_b}_{k}"
            bs = (n, k) if ord_b == "R" else (k, n)
            cs = (m, k) if ord_b == "R" else (k, m)
            b = rng.uniform(-1, 1, bs).astype(np.float32)
            c = rng.uniform(-1, 1, cs).astype(np.float32)
            bn = b.copy()
            if ord_b == "R":
                bn[hit[0], :] = NAN
                bn[hit[1], 0] = INF
            else:
                bn[:, hit[0]] = NAN
                bn[0, hit[1]] = INF
            cn = c.copy()
            cn[0, 0] = NAN
            cn[-1, -1] = INF
            mm_case("plain_" + tag, ord_b, m, n, k, 0.5, 2.0, val, ia, ja, b, c)
            mm_case("explicit_zero_times_nan_" + tag, ord_b, m, n, k, 1.0, 0.0, val, ia, ja, bn, c)
            mm_case("alpha0_nanB_beta2_" + tag, ord_b, m, n, k, 0.0, 2.0, val, ia, ja, bn, c)
            if k > 128:      # the wide-lane kernel (4 columns per lane): the three cases above only
                continue
            mm_case("alpha0_nanB_beta0_" + tag, ord_b, m, n, k, 0.0, 0.0, val, ia, ja, bn, cn)
            mm_case("beta0_nanC_" + tag, ord_b, m, n, k, 1.0, 0.0, val, ia, ja, b, cn)
            mm_case("beta1_nanC_" + tag, ord_b, m, n, k, 1.0, 1.0, val, ia, ja, b, cn)
            vd = (val * np.float32(1e-39)).astype(np.float32)
            mm_case("denormal_values_" + tag, ord_b, m, n, k, 1.0, 0.0, vd, ia, ja, b, c)
            vz = val.copy()
            vz[1::7] = -0.0
            mm_case("negzero_values_" + tag, ord_b, m, n, k, 1.0, 0.0, vz, ia, ja, np.abs(b), c)
    # csrgemv needs a square pad (csrgemv_task.h:36-44): use m x n with m <= n for 'N', both for 'T'
    x = rng.uniform(-1, 1, n).astype(np.float32)
    xn = x.copy()
    xn[hit[0]] = NAN
    xn[hit[1]] = INF
    mv_case("plain_N", "N", m, n, val, ia, ja, x)
    mv_case("explicit_zero_times_nan_N", "N", m, n, val, ia, ja, xn)
    mv_case("denormal_values_N", "N", m, n, (val * np.float32(1e-39)).astype(np.float32), ia, ja, x)
    xt = rng.uniform(-1, 1, m).astype(np.float32)
    xtn = xt.copy()
    xtn[0] = NAN
    xtn[5] = INF        # an empty row: must not reach y
    mv_case("plain_T", "T", m, n, val, ia, ja, xt)
    mv_case("nan_x_T", "T", m, n, val, ia, ja, xtn)
    out["names"] = np.array(names)
    out["meta"] = np.array(["mkl=" + mg.ver()])
    np.savez_compressed(os.path.join(HERE, "mkl_golden_special.npz"), **out)
    print(f"{len(names)} special-value cases")


if __name__ == "__main__":
    special()
    gemm_run()
