"""-m gpu parity of the transposition row: bof_scsrcsc (stable CSR
transposition by radix sort) and csrmm with trans_a='T', against the MKL golden
vectors (mkl_csrcsc / mkl_scsrmm 'T'), the oracle and size-independent properties."""
import hashlib

import numpy as np
import pytest
import torch

import bofhip
import orc
from gpu_util import ptr, rel_err, stream, to_dev

pytestmark = pytest.mark.gpu
TOL = 1e-4


def gpu_csrcsc(m, n, val, ia, ja):
    nnz = int(ia[m] - ia[0]) if m > 0 else 0
    dv, di, dj = to_dev(val if nnz else np.zeros(1, np.float32)), to_dev(ia), \
        to_dev(ja if nnz else np.zeros(1, np.int64))
    vt = torch.full((max(nnz, 1),), -7.0, dtype=torch.float32, device="cuda")
    jt = torch.full((max(nnz, 1),), -7, dtype=torch.int64, device="cuda")
    it = torch.full((n + 1,), -7, dtype=torch.int64, device="cuda")
    bofhip.scsrcsc(m, n, nnz, ptr(dv), ptr(di), ptr(dj), ptr(vt), ptr(it), ptr(jt), stream())
    torch.cuda.synchronize()
    return vt.cpu().numpy()[:nnz], it.cpu().numpy(), jt.cpu().numpy()[:nnz]


def rand_csr(rng, m, n, max_per_row, heavy=()):
    counts = rng.integers(0, max_per_row + 1, m)
    for r, c in heavy:
        counts[r] = c
    counts = np.minimum(counts, n)
    ia = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts] +
                        [np.zeros(0, np.int64)]).astype(np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    return val, ia, ja


def test_csrcsc_vs_mkl_golden(dev, golden_tr):
    seen = 0
    for t in golden_tr["meta"]:
        f = t.split()
        if not (f[0].startswith("tr") and len(f) == 3):
            continue
        key, m, n = f[0], int(f[1]), int(f[2])
        vt, it, jt = gpu_csrcsc(m, n, golden_tr[key + "_val"], golden_tr[key + "_ia"],
                                golden_tr[key + "_ja"])
        assert np.array_equal(it, golden_tr[key + "_ia_tr"]), key
        assert np.array_equal(jt, golden_tr[key + "_ja_tr"]), key
        assert np.array_equal(vt, golden_tr[key + "_val_tr"]), key
        seen += 1
    assert seen == 5


# digit passes: n=200 -> 1, 40000 -> 2, 300000 -> 3, 2^24+5 -> 4; several 16384-entry tiles
@pytest.mark.parametrize("m,n,per_row", [(20000, 200, 12), (3000, 40000, 40), (5000, 300000, 30),
                                         (900, (1 << 24) + 5, 60), (70000, 257, 3)])
def test_csrcsc_vs_oracle_all_pass_counts(dev, m, n, per_row):
    rng = np.random.default_rng(m + n)
    val, ia, ja = rand_csr(rng, m, n, per_row, heavy=[(1, 0), (m // 2, min(n, 150))])
    assert ia[m] > 16384
    want = orc.csrcsc(m, n, val, ia, ja)
    got = gpu_csrcsc(m, n, val, ia, ja)
    assert np.array_equal(got[1], want[1])
    assert np.array_equal(got[2], want[2])
    assert np.array_equal(got[0], want[0])


def test_csrcsc_offsets_with_base_and_empty(dev):
    """A row range of a bigger matrix (absolute offsets, as the flash level passes them), an
    all-empty matrix and m = 0."""
    rng = np.random.default_rng(5)
    val, ia, ja = rand_csr(rng, 4000, 1000, 20)
    s, e = 1000, 3500
    z = int(ia[s])
    want = orc.csrcsc(e - s, 1000, val[z:], ia[s:e + 1], ja[z:])
    got = gpu_csrcsc(e - s, 1000, val[z:int(ia[e])], ia[s:e + 1], ja[z:int(ia[e])])
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    _, it, _ = gpu_csrcsc(50, 70, np.zeros(0, np.float32), np.zeros(51, np.int64), np.zeros(0, np.int64))
    assert np.array_equal(it, np.zeros(71, np.int64))
    _, it, _ = gpu_csrcsc(0, 9, np.zeros(0, np.float32), np.zeros(1, np.int64), np.zeros(0, np.int64))
    assert np.array_equal(it, np.zeros(10, np.int64))


def test_csrcsc_generator_matrix_hashes(dev, golden_tr):
    want = {t.split()[1]: t.split()[2] for t in golden_tr["meta"] if t.startswith("exact")}
    val, ja, ia = orc.sparse_create(4096, 2048, 0.01)
    vt, it, jt = gpu_csrcsc(4096, 2048, val, ia, ja)
    assert hashlib.sha256(vt.tobytes()).hexdigest() == want["gen_tr_val"]
    assert hashlib.sha256(it.tobytes()).hexdigest() == want["gen_tr_ia"]
    assert hashlib.sha256(jt.tobytes()).hexdigest() == want["gen_tr_ja"]


def run_csrmm_t(m, n, k, alpha, beta, val, ia, ja, b, c0, ord_b="R", opts=None):
    """C[n x k] = alpha A^T B[m x k] + beta C through bof_csrmm_resident('T')."""
    ia = np.ascontiguousarray(ia, np.int64)
    dv, di, dj = to_dev(val), to_dev(ia), to_dev(ja)
    if ord_b == "C":
        db, dc = to_dev(np.ascontiguousarray(b.T)), to_dev(np.ascontiguousarray(c0.T))
    else:
        db, dc = to_dev(b), to_dev(c0)
    bofhip.csrmm_resident("T", m, n, k, alpha, beta, ptr(dv), ia.ctypes.data, ptr(di), ptr(dj), ord_b,
                          ptr(db), ptr(dc), opts, stream())
    torch.cuda.synchronize()
    out = dc.cpu().numpy()
    return out.T.copy() if ord_b == "C" else out


def test_csrmm_t_vs_mkl_golden(dev, golden_tr):
    seen = 0
    for t in golden_tr["meta"]:
        f = t.split()
        if not f[0].startswith("csrmmT"):
            continue
        ck, m, n, k, alpha, beta, key, rows = f[0], int(f[1]), int(f[2]), int(f[3]), float(f[4]), \
            float(f[5]), f[6], int(f[7])
        val, ia, ja = golden_tr[key + "_val"], golden_tr[key + "_ia"], golden_tr[key + "_ja"]
        b = golden_tr[ck + "_b"]
        c0 = np.zeros((n, k), np.float32)
        c0[:rows] = golden_tr[ck + "_c0"]
        ref = orc.scsrmm_t(m, n, k, alpha, val, ia, ja, b, k, beta, c0.copy(), k)
        for ord_b in "RC":
            got = run_csrmm_t(m, n, k, alpha, beta, val, ia, ja, b, c0, ord_b)
            assert rel_err(got[:rows], golden_tr[ck + "_c1"]) < TOL, (ck, ord_b)
            assert np.array_equal(got, ref), (ck, ord_b)   # the oracle's source-row-ordered chain
        seen += 1
    assert seen == 6


def test_csrmm_t_blocks_panels_and_generator_hash(dev, golden_tr):
    """Several row blocks of A^T and two column panels (k = 1030 > CBLK); the k = 128 product on
    the reference generator's matrix is pinned by the MKL hash."""
    want = {t.split()[1]: t.split()[2] for t in golden_tr["meta"] if t.startswith("exact")}
    val, ja, ia = orc.sparse_create(4096, 2048, 0.01)
    b = orc.dense_fill(4096, 128, "s")
    opts = bofhip.default_options(max_nnzs=9000, csrmm_rblk=300)
    got = run_csrmm_t(4096, 2048, 128, 1.0, 0.0, val, ia, ja, b, np.zeros((2048, 128), np.float32),
                      "R", opts)
    assert hashlib.sha256(got.tobytes()).hexdigest() == want["gen_csrmmT_c"]
    rng = np.random.default_rng(11)
    k = 1030
    b = rng.uniform(-1, 1, (4096, k)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (2048, k)).astype(np.float32)
    ref = orc.scsrmm_t(4096, 2048, k, 0.5, val, ia, ja, b, k, 2.0, c0.copy(), k)
    for ord_b in "RC":
        got = run_csrmm_t(4096, 2048, k, 0.5, 2.0, val, ia, ja, b, c0, ord_b, opts)
        assert np.array_equal(got, ref), ord_b


def test_csrcsc_involution_at_scale(dev):
    """1M x 1M, 100 nnz/row (1e8 non-zeros, the BASELINE matrix family): transposing twice
    returns the input bit for bit (its columns are sorted), and the transpose keeps the
    multiset of values per column (checked through column sums of values and row ids)."""
    m = n = 1_000_000
    npr = 100
    nnz = m * npr
    val = torch.empty(nnz, dtype=torch.float32, device="cuda")
    col = torch.empty(nnz, dtype=torch.int64, device="cuda")
    off = torch.empty(m + 1, dtype=torch.int64, device="cuda")
    bofhip.gen_sparse_rows(0, m, n, npr, ptr(val), ptr(col), ptr(off), stream())
    vt = torch.empty_like(val); ct = torch.empty_like(col)
    pt = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    bofhip.scsrcsc(m, n, nnz, ptr(val), ptr(off), ptr(col), ptr(vt), ptr(pt), ptr(ct), stream())
    torch.cuda.synchronize()
    assert int(pt[0]) == 0 and int(pt[n]) == nnz
    counts = torch.bincount(col, minlength=n)
    assert torch.equal(pt[1:] - pt[:-1], counts)
    # inside every output row the source rows ascend strictly
    d = ct[1:] - ct[:-1]
    is_start = torch.zeros(nnz, dtype=torch.bool, device="cuda")
    starts = pt[:-1][counts > 0]
    is_start[starts] = True
    assert bool(((d > 0) | is_start[1:]).all())
    # per-column sums (integers < 2^24 would not hold here, so use float64)
    want = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, col, val.double())
    rows_of_out = torch.repeat_interleave(torch.arange(n, device="cuda"), counts)
    got = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, rows_of_out, vt.double())
    assert torch.equal(want, got)
    del want, got, rows_of_out, is_start, d
    v2 = torch.empty_like(val); c2 = torch.empty_like(col)
    p2 = torch.empty(m + 1, dtype=torch.int64, device="cuda")
    bofhip.scsrcsc(n, m, nnz, ptr(vt), ptr(pt), ptr(ct), ptr(v2), ptr(p2), ptr(c2), stream())
    torch.cuda.synchronize()
    assert torch.equal(p2, off) and torch.equal(c2, col) and torch.equal(v2, val)


@pytest.mark.parametrize("nnz_target", [1, 63, 64, 65, 4095, 4096, 4097, 8192])
def test_csrcsc_tile_boundaries(dev, nnz_target):
    """Non-zero counts around the 64-record chunk and the 4096-record tile of the sort (one entry,
    one short of / exactly / one past a boundary), including a dense single row and single column."""
    rng = np.random.default_rng(nnz_target)
    for (m, n) in [(max(1, nnz_target // 7 + 1), 50), (1, nnz_target), (nnz_target, 1)]:
        cap = m * n
        want_nnz = min(nnz_target, cap)
        flat = np.sort(rng.choice(cap, want_nnz, replace=False))
        rows, cols = flat // n, flat % n
        ia = np.zeros(m + 1, np.int64)
        np.add.at(ia, rows + 1, 1)
        ia = np.cumsum(ia)
        ja = cols.astype(np.int64)
        val = rng.uniform(-1, 1, want_nnz).astype(np.float32)
        want = orc.csrcsc(m, n, val, ia, ja)
        got = gpu_csrcsc(m, n, val, ia, ja)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), (m, n, want_nnz)
