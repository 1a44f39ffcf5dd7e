"""-m gpu parity of the CSR kernels (SpMM / SpMV) against the oracle, the MKL
golden vectors and the generator known answers."""
import hashlib

import numpy as np
import pytest
import torch

import bofhip
import orc
from gpu_util import ptr, rel_err, stream, to_dev

pytestmark = pytest.mark.gpu
TOL = 1e-4


def meta_rows(golden, prefix):
    for line in golden["meta"]:
        t = line.split()
        if t[0].startswith(prefix):
            yield t


def exact_hash(golden, name):
    for t in meta_rows(golden, "exact"):
        if t[1] == name:
            return t[2]
    raise KeyError(name)


def run_scsrmm(ord_b, m, n, k, alpha, beta, val, ia, ja, b, ldb, c0, ldc):
    dv, di, dj, db, dc = to_dev(val), to_dev(ia), to_dev(ja), to_dev(b), to_dev(c0)
    bofhip.scsrmm(ord_b, m, n, k, alpha, ptr(dv), ptr(dj), ptr(di), ptr(db), ldb, beta, ptr(dc),
                  ldc, stream())
    torch.cuda.synchronize()
    assert np.array_equal(dj.cpu().numpy(), ja)  # indices never modified (App. B-15)
    return dc.cpu().numpy()


def test_golden_mkl_csrmm(dev, golden):
    for t in meta_rows(golden, "csrmm"):
        key, mat = t[0], t[7]
        m, n, k = map(int, t[1:4])      # A m x n, B n x k
        ord_b, alpha, beta = t[4], float(t[5]), float(t[6])
        val, ia, ja = golden[mat + "_val"], golden[mat + "_ia"], golden[mat + "_ja"]
        ldb, ldc = (k, k) if ord_b == "R" else (n, m)
        got = run_scsrmm(ord_b, m, k, n, alpha, beta, val, ia, ja, golden[key + "_b"], ldb,
                         golden[key + "_c0"], ldc)
        assert rel_err(got, golden[key + "_c1"]) < TOL, key
        ref = orc.scsrmm(ord_b, m, k, n, alpha, val, ja, ia, golden[key + "_b"], ldb, beta,
                         golden[key + "_c0"].copy(), ldc)
        assert np.array_equal(got, ref), key   # same fmaf chain as the oracle


@pytest.mark.parametrize("ncol", [128, 40, 1030, 7, 256, 64])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (0.5, 2.0)])
def test_scsrmm_rowmajor_widths(dev, ncol, alpha, beta):
    """Every vector width / column-pass path, ragged rows incl. empty rows and a
    row longer than one 64-entry segment."""
    rng = np.random.default_rng(ncol)
    m, n = 333, 900
    counts = rng.integers(0, 30, m)
    counts[5] = 0
    counts[17] = 200
    counts[m - 1] = 65
    ia = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts]).astype(np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    b = rng.uniform(-1, 1, (n, ncol)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, ncol)).astype(np.float32)
    ref = orc.scsrmm("R", m, ncol, n, alpha, val, ja, ia, b, ncol, beta, c0.copy(), ncol)
    got = run_scsrmm("R", m, ncol, n, alpha, beta, val, ia, ja, b, ncol, c0, ncol)
    assert np.array_equal(got, ref)


def test_csrmm_resident_blocks_and_panels(dev, golden):
    """flash::csrmm structure: nnz-budget row blocks x column panels (k=1030 > CBLK
    forces two panels), R and C layouts, on the reference generator's matrix; the
    k=128 'R' result is pinned by the MKL golden hash."""
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    dv, dj, di = to_dev(val), to_dev(ja), to_dev(ia)
    opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000, csrmm_cblk=1024)
    for k, ord_b, alpha, beta in [(128, "R", 1.0, 0.0), (1030, "R", 0.5, 2.0), (1030, "C", 0.5, 2.0),
                                  (128, "C", 1.0, 0.0)]:
        b = orc.dense_fill(n, k, "s")
        rng = np.random.default_rng(k)
        c0 = rng.integers(0, 5, (m, k)).astype(np.float32) if beta else np.zeros((m, k), np.float32)
        if ord_b == "C":
            b = np.ascontiguousarray(b.T)      # col-major n x k
            c0 = np.ascontiguousarray(c0.T)
        ref = orc.flash_csrmm(ord_b, m, n, k, alpha, beta, val, ia, ja, b, c0.copy(),
                              max_rows=1000, max_nnz=5000, cblk=1024)
        db, dc = to_dev(b), to_dev(c0)
        bofhip.csrmm_resident("N", m, n, k, alpha, beta, ptr(dv), ia.ctypes.data, ptr(di), ptr(dj),
                              ord_b, ptr(db), ptr(dc), opts, stream())
        torch.cuda.synchronize()
        got = dc.cpu().numpy()
        assert np.array_equal(got, ref), (k, ord_b)
        if (k, ord_b, beta) == (128, "R", 0.0):
            assert hashlib.sha256(got.tobytes()).hexdigest() == exact_hash(golden, "gen_csrmm_c")
    with pytest.raises(bofhip.BofError):   # reference returns -1 for bad flags too
        bofhip.csrmm_resident("N", m, n, 128, 1.0, 0.0, ptr(dv), ia.ctypes.data, ptr(di), ptr(dj),
                              "X", ptr(db), ptr(dc), opts, stream())


@pytest.mark.parametrize("k", [64, 128, 200])
def test_scsrmm_every_row_length(dev, k):
    """Rows of 0 .. 139 non-zeros, one of each: every remainder of the kernel's batches of eight gathers (the last
    1 .. 7 entries of a 64-entry chunk go out as ONE predicated batch), one and two chunks, and the chunk boundary
    itself (64, 65, 128, 129); random values -- the fmaf chain per output element in storage order, bit for bit."""
    rng = np.random.default_rng(100 + k)
    m, n = 140, 3000
    counts = np.arange(m)
    ia = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts]).astype(np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    ref = orc.scsrmm("R", m, k, n, 0.5, val, ja, ia, b, k, 2.0, c0.copy(), k)
    dv, di, dj, db, dc = to_dev(val), to_dev(ia), to_dev(ja), to_dev(b), to_dev(c0)
    bofhip.scsrmm("R", m, k, n, 0.5, ptr(dv), ptr(dj), ptr(di), ptr(db), k, 2.0, ptr(dc), k, stream())
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)


def test_golden_mkl_csrgemv(dev, golden):
    for t in meta_rows(golden, "csrgemv"):
        key, mat = t[0], t[4]
        m, n, trans = int(t[1]), int(t[2]), t[3]
        val, ia, ja = golden[mat + "_val"], golden[mat + "_ia"], golden[mat + "_ja"]
        x = golden[key + "_x"]
        dv, di, dj, dx = to_dev(val), to_dev(ia), to_dev(ja), to_dev(x)
        dy = torch.zeros(m if trans == "N" else n, dtype=torch.float32, device=dev)
        bofhip.scsrgemv(trans, m, n, ptr(dv), ptr(di), ptr(dj), ptr(dx), ptr(dy), stream())
        torch.cuda.synchronize()
        got = dy.cpu().numpy()
        assert rel_err(got, golden[key + "_y"]) < TOL, key
        if trans == "N":
            ref = orc.scsrgemv("N", m, n, val, ia, ja, x, np.zeros(m, np.float32))
            assert np.array_equal(got, ref)


@pytest.mark.parametrize("trans", ["N", "T"])
def test_csrgemv_resident_generator_known_answer(dev, golden, trans):
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    x = (np.arange(n if trans == "N" else m) % 10).astype(np.float32)
    dv, dj, di, dx = to_dev(val), to_dev(ja), to_dev(ia), to_dev(x)
    dy = torch.full((m if trans == "N" else n,), 7.0, dtype=torch.float32, device=dev)
    opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000)
    bofhip.csrgemv_resident(trans, m, n, ptr(dv), ia.ctypes.data, ptr(di), ptr(dj), ptr(dx),
                            ptr(dy), opts, stream())
    torch.cuda.synchronize()
    got = dy.cpu().numpy()
    ref = orc.flash_csrgemv(trans, m, n, val, ia, ja, x, np.zeros_like(got), 1000, 5000)
    assert np.array_equal(got, ref)
    assert hashlib.sha256(got.tobytes()).hexdigest() == exact_hash(golden, "gen_csrgemv_" + trans)


def test_scsrgemv_n_ragged_heavy_rows(dev):
    """Groups of 64 rows above and below the 1024-entry LDS image, empty rows, m not a
    multiple of 64 or 256."""
    rng = np.random.default_rng(5)
    m, n = 1000, 5000
    counts = rng.integers(0, 12, m)
    counts[64:128] = 40            # 2560 entries in one 64-row group -> direct path
    counts[300] = 1500             # one very heavy row
    counts[500:564] = 0            # an empty group
    ia = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts]).astype(np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    ref = orc.scsrgemv("N", m, n, val, ia, ja, x, np.zeros(m, np.float32))
    dv, di, dj, dx = to_dev(val), to_dev(ia), to_dev(ja), to_dev(x)
    dy = torch.full((m,), 9.0, dtype=torch.float32, device=dev)
    bofhip.scsrgemv("N", m, n, ptr(dv), ptr(di), ptr(dj), ptr(dx), ptr(dy), stream())
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), ref)
    # a sub-block with a base offset in ptr (row block of a larger matrix)
    s0, r = 200, 333
    z = ia[s0]
    dy2 = torch.zeros(r, dtype=torch.float32, device=dev)
    bofhip.scsrgemv("N", r, n, ptr(dv) + 4 * int(z), ptr(di) + 8 * s0, ptr(dj) + 8 * int(z), ptr(dx), ptr(dy2), stream())
    torch.cuda.synchronize()
    assert np.array_equal(dy2.cpu().numpy(), ref[s0:s0 + r])


@pytest.mark.parametrize("m,n,per_row", [(4096, 2048, 21), (3000, 70000, 30), (2000, 5_000_000, 40),
                                         (60000, 300, 5), (500, 3_000_000_0 // 10 + 7, 50)])
def test_csrgemv_t_partitioned_path(dev, golden, m, n, per_row, monkeypatch):
    """y = A^T x through the bin-partition path of bof_csrgemv_resident (forced for small inputs):
    one bin (n <= 8192), one digit pass, two digit passes (n > 2M columns); random data within
    the BASELINE tolerance of the oracle's chain, integer data exact, untouched columns zero."""
    monkeypatch.setenv("BOF_GEMV_T_PARTITION_MIN_NNZ", "1")
    rng = np.random.default_rng(m + n)
    counts = rng.integers(0, per_row + 1, m)
    counts[0] = 0
    counts[m // 2] = min(n, 700)
    ia = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts]).astype(np.int64)
    for data in ("random", "integer"):
        if data == "random":
            val = rng.uniform(-1, 1, ja.size).astype(np.float32)
            x = rng.uniform(-1, 1, m).astype(np.float32)
        else:
            val = rng.integers(1, 10, ja.size).astype(np.float32)
            x = (np.arange(m) % 10).astype(np.float32)
        ref = orc.scsrgemv("T", m, n, val, ia, ja, x, np.zeros(n, np.float32))
        dv, dj, di, dx = to_dev(val), to_dev(ja), to_dev(ia), to_dev(x)
        dy = torch.full((n,), 7.0, dtype=torch.float32, device=dev)
        bofhip.csrgemv_resident("T", m, n, ptr(dv), ia.ctypes.data, ptr(di), ptr(dj), ptr(dx), ptr(dy),
                                None, stream())
        torch.cuda.synchronize()
        got = dy.cpu().numpy()
        if data == "integer":
            assert np.array_equal(got, ref)
        else:
            assert rel_err(got, ref) < TOL
            assert np.array_equal(got == 0, ref == 0)   # columns without entries are exactly zero


def test_csrgemv_t_partitioned_generator_hash_and_row_range(dev, golden, monkeypatch):
    monkeypatch.setenv("BOF_GEMV_T_PARTITION_MIN_NNZ", "1")
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    x = (np.arange(m) % 10).astype(np.float32)
    dv, dj, di, dx = to_dev(val), to_dev(ja), to_dev(ia), to_dev(x)
    dy = torch.full((n,), 7.0, dtype=torch.float32, device=dev)
    bofhip.csrgemv_resident("T", m, n, ptr(dv), ia.ctypes.data, ptr(di), ptr(dj), ptr(dx), ptr(dy), None, stream())
    torch.cuda.synchronize()
    assert hashlib.sha256(dy.cpu().numpy().tobytes()).hexdigest() == exact_hash(golden, "gen_csrgemv_T")
    # a row range with absolute offsets (how a row shard of a bigger matrix is passed)
    s0, r = 1000, 2500
    ia_s = np.ascontiguousarray(ia[s0:s0 + r + 1])
    ref = orc.scsrgemv("T", r, n, val[ia[s0]:], ia_s, ja[ia[s0]:], x[s0:s0 + r], np.zeros(n, np.float32))
    bofhip.csrgemv_resident("T", r, n, ptr(dv), ia_s.ctypes.data, ptr(di) + 8 * s0, ptr(dj), ptr(dx) + 4 * s0,
                            ptr(dy), None, stream())
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy(), ref)


def test_csr_maximum_column_index(dev, monkeypatch):
    """Maximum sizes: A with n = 2^31 - 1 columns (the largest the 31-bit column arithmetic of the
    kernels admits; the entry points reject more) and non-zeros in column 0, 2^30 and n - 1.
    csrgemv 'N' / 'T' (x resp. y of 8.6 GB) and csrmm with k = 2 (B of 17 GB) on an MI355X;
    values chosen so that every result is a small integer."""
    n = 2 ** 31 - 1
    m = 1000
    cols = np.array([0, 2 ** 30, n - 1], np.int64)
    ia = (np.arange(m + 1, dtype=np.int64) * 3)
    ja = np.tile(cols, m)
    val = np.tile(np.array([1.0, 2.0, 3.0], np.float32), m) + (np.arange(3 * m) // 3 % 5).astype(np.float32)
    d_val, d_ja, d_ia = to_dev(val), to_dev(ja), to_dev(ia)
    # 'N': y[i] = sum_j val[i,j] * x[col j]
    x = torch.zeros(n, dtype=torch.float32, device=dev)
    x[0], x[2 ** 30], x[n - 1] = 1.0, 10.0, 100.0
    y = torch.full((m,), -1.0, dtype=torch.float32, device=dev)
    bofhip.scsrgemv("N", m, n, ptr(d_val), ptr(d_ia), ptr(d_ja), ptr(x), ptr(y), stream())
    torch.cuda.synchronize()
    v = val.reshape(m, 3)
    assert np.array_equal(y.cpu().numpy(), v[:, 0] * 1 + v[:, 1] * 10 + v[:, 2] * 100)
    # the whole-matrix level-2 call ('N' row blocks; 'T' through the column-bin partition)
    bofhip.csrgemv_resident("N", m, n, ptr(d_val), ia.ctypes.data, ptr(d_ia), ptr(d_ja), ptr(x), ptr(y),
                            bofhip.default_options(), stream())
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), v[:, 0] * 1 + v[:, 1] * 10 + v[:, 2] * 100)
    del x
    xt = torch.ones(m, dtype=torch.float32, device=dev)
    yt = torch.full((n,), 7.0, dtype=torch.float32, device=dev)
    def check_t():
        torch.cuda.synchronize()
        for c, j in ((0, 0), (2 ** 30, 1), (n - 1, 2)):
            assert float(yt[c]) == float(v[:, j].sum())
        assert float(yt.double().sum()) == float(v.astype(np.float64).sum())     # nothing landed anywhere else
    bofhip.scsrgemv("T", m, n, ptr(d_val), ptr(d_ia), ptr(d_ja), ptr(xt), ptr(yt.zero_()), stream())   # atomic form
    check_t()
    monkeypatch.setenv("BOF_GEMV_T_PARTITION_MIN_NNZ", "1")     # the column-bin partition: 262144 bins, 3 digit passes
    yt.fill_(7.0)
    bofhip.csrgemv_resident("T", m, n, ptr(d_val), ia.ctypes.data, ptr(d_ia), ptr(d_ja), ptr(xt), ptr(yt),
                            bofhip.default_options(), stream())
    check_t()
    del yt, xt
    torch.cuda.empty_cache()
    # csrmm k = 2: B is 2^31 - 1 rows of two floats
    b = torch.zeros(n * 2, dtype=torch.float32, device=dev)
    B = b.view(n, 2)
    B[0] = torch.tensor([1.0, 2.0], device=dev)
    B[2 ** 30] = torch.tensor([10.0, 20.0], device=dev)
    B[n - 1] = torch.tensor([100.0, 200.0], device=dev)
    c = torch.full((m, 2), -1.0, dtype=torch.float32, device=dev)
    bofhip.scsrmm("R", m, 2, n, 1.0, ptr(d_val), ptr(d_ja), ptr(d_ia), ptr(b), 2, 0.0, ptr(c), 2, stream())
    torch.cuda.synchronize()
    want = np.stack([v[:, 0] * 1 + v[:, 1] * 10 + v[:, 2] * 100, v[:, 0] * 2 + v[:, 1] * 20 + v[:, 2] * 200], 1)
    assert np.array_equal(c.cpu().numpy(), want)
    # one column more is refused, not wrapped
    with pytest.raises(bofhip.BofError):
        bofhip.scsrmm("R", m, 2, n + 1, 1.0, ptr(d_val), ptr(d_ja), ptr(d_ia), ptr(b), 2, 0.0, ptr(c), 2, stream())
    del b, c
    torch.cuda.empty_cache()
