"""-m gpu parity of the fp32 MFMA tile GEMM and the resident tile DAG against the
oracle (bit-exact: both are k-ordered fmaf chains) and the MKL golden vectors
(1e-4 relative, BASELINE.json north_star)."""
import itertools
import os

import numpy as np
import pytest
import torch

import bofhip
import orc
from gpu_util import ptr, rel_err, stream, to_dev

pytestmark = pytest.mark.gpu
TOL = 1e-4  # north_star: fp32 within 1e-4 relative


def stored_shapes(ord_, ta, tb, m, n, k):
    a = (m, k) if (ta == "T") == (ord_ == "C") else (k, m)
    b = (k, n) if (tb == "T") == (ord_ == "C") else (n, k)
    c = (m, n) if ord_ == "R" else (n, m)
    return a, b, c


def run_sgemm(ord_, ta, tb, m, n, k, alpha, beta, a, lda, b, ldb, c, ldc):
    da, db, dc = to_dev(a), to_dev(b), to_dev(c)
    bofhip.sgemm(ord_, ta, tb, m, n, k, alpha, ptr(da), lda, ptr(db), ldb, beta, ptr(dc), ldc,
                 stream())
    torch.cuda.synchronize()
    return dc.cpu().numpy()


def test_golden_mkl_gemm(dev, golden):
    """The reference's cblas_sgemm outputs (tests/golden) for all 8 layouts."""
    for line in golden["meta"]:
        t = line.split()
        if not t[0].startswith("gemm"):
            continue
        key = t[0]
        m, n, k = map(int, t[1:4])
        ord_, ta, tb = t[4:7]
        alpha, beta = float(t[7]), float(t[8])
        lda, ldb, ldc = map(int, t[9:12])
        got = run_sgemm(ord_, ta, tb, m, n, k, alpha, beta, golden[key + "_a"], lda,
                        golden[key + "_b"], ldb, golden[key + "_c0"], ldc)
        ref = golden[key + "_c1"]
        assert rel_err(got, ref) < TOL, (key, ord_, ta, tb)
        # padding columns of C (beyond the logical width) must be untouched
        cc = n if ord_ == "R" else m
        assert np.array_equal(got[:, cc:], golden[key + "_c0"][:, cc:]), key


@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
@pytest.mark.parametrize("m,n,k,alpha,beta", [
    (256, 384, 320, 1.0, 0.0),      # aligned fast path (multiples of the 128x128x32 tile)
    (300, 200, 500, 0.5, 2.0),      # ragged: guarded path, m/n/k tails
    (128, 128, 32, 1.0, 1.0),       # single block, single K slab
    (1, 1, 1, 2.0, 0.0),            # degenerate
    (129, 127, 33, -1.5, 0.25),     # one past / one short of the tile edges
])
def test_sgemm_bit_exact_vs_oracle(dev, ord_, ta, tb, m, n, k, alpha, beta):
    rng = np.random.default_rng(hash((ord_, ta, tb, m, n, k)) & 0xFFFF)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    pad = 0 if (m % 128 == 0) else 3
    lda, ldb, ldc = sa[1] + pad, sb[1] + pad, sc[1] + pad
    a = rng.uniform(-1, 1, (sa[0], lda)).astype(np.float32)
    b = rng.uniform(-1, 1, (sb[0], ldb)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (sc[0], ldc)).astype(np.float32)
    ref = orc.sgemm(ord_, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c0.copy(), ldc)
    got = run_sgemm(ord_, ta, tb, m, n, k, alpha, beta, a, lda, b, ldb, c0, ldc)
    assert np.array_equal(got, ref), (rel_err(got, ref))
    # and against float64 within the north-star tolerance
    A = a[:, :sa[1]].astype(np.float64)
    B = b[:, :sb[1]].astype(np.float64)
    if ord_ == "C":
        opA = A.T if ta == "N" else A
        opB = B.T if tb == "N" else B
    else:
        opA = A if ta == "N" else A.T
        opB = B if tb == "N" else B.T
    full = alpha * (opA @ opB)
    c64 = c0[:, :sc[1]].astype(np.float64)
    full = full + beta * (c64 if ord_ == "R" else c64.T)
    gotl = got[:, :sc[1]] if ord_ == "R" else got[:, :sc[1]].T
    assert rel_err(gotl, full) < TOL


@pytest.mark.parametrize("ta,tb", list(itertools.product("NT", "NT")))
def test_sgemm_big_tile_kernel_bit_exact(dev, ta, tb):
    """4096 x 2048 x 96 is 16 x 8 = 128 blocks of 256 x 256: the double-buffered 8-wave
    kernel the BASELINE tile size runs on (3 K-slabs: prologue, steady state, drain)."""
    m, n, k = 4096, 2048, 96
    rng = np.random.default_rng(42)
    sa, sb, sc = stored_shapes("R", ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    for alpha, beta in [(1.0, 0.0), (0.5, 2.0)]:
        ref = orc.sgemm("R", ta, tb, m, n, k, alpha, a, sa[1], b, sb[1], beta, c0.copy(), sc[1])
        got = run_sgemm("R", ta, tb, m, n, k, alpha, beta, a, sa[1], b, sb[1], c0, sc[1])
        assert np.array_equal(got, ref), rel_err(got, ref)


@pytest.mark.parametrize("ta,tb", list(itertools.product("NT", "NT")))
@pytest.mark.parametrize("m,n,k", [(4096 + 100, 2048 + 37, 116), (4096, 2048 + 255, 96 + 31), (4096 + 1, 2048, 128)])
def test_sgemm_ragged_interior_plus_strips_bit_exact(dev, ta, tb, m, n, k):
    """Tail-merged tile shapes: 256-aligned interior on the big-tile kernel (guarded last K slab),
    right/bottom strips on the guarded kernel -- still one k-ordered chain per element."""
    rng = np.random.default_rng(m * 7 + n)
    sa, sb, sc = stored_shapes("R", ta, tb, m, n, k)
    pad = lambda c: (c + 3) // 4 * 4 + 4          # leading dims: multiples of 4, not tight
    lda, ldb, ldc = pad(sa[1]), pad(sb[1]), sc[1] + 5
    a = rng.uniform(-1, 1, (sa[0], lda)).astype(np.float32)
    b = rng.uniform(-1, 1, (sb[0], ldb)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (sc[0], ldc)).astype(np.float32)
    ref = orc.sgemm("R", ta, tb, m, n, k, 0.5, a, lda, b, ldb, 2.0, c0.copy(), ldc)
    got = run_sgemm("R", ta, tb, m, n, k, 0.5, 2.0, a, lda, b, ldb, c0, ldc)
    assert np.array_equal(got, ref), rel_err(got, ref)


@pytest.mark.parametrize("ta,tb", list(itertools.product("NT", "NT")))
@pytest.mark.parametrize("k", [64, 128, 192, 1024])
def test_sgemm_big_tile_slab_pairs(dev, ta, tb, k):
    """K % 64 == 0 runs the hand-scheduled kernels whose slabs go in (buffer 0, buffer 1) pairs
    ('T','N': LDS-DMA staging; the other layouts: register staging): one pair, two, three, many;
    padded leading dimensions; bit-exact against the oracle's k-ordered chain."""
    m, n = 4096, 2048
    rng = np.random.default_rng(k)
    sa, sb, sc = stored_shapes("R", ta, tb, m, n, k)
    lda, ldb, ldc = sa[1] + 8, sb[1] + 4, n + 12
    a = rng.uniform(-1, 1, (sa[0], lda)).astype(np.float32)
    b = rng.uniform(-1, 1, (sb[0], ldb)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, ldc)).astype(np.float32)
    for alpha, beta in [(1.0, 0.0), (0.5, 2.0)]:
        ref = orc.sgemm("R", ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c0.copy(), ldc)
        got = run_sgemm("R", ta, tb, m, n, k, alpha, beta, a, lda, b, ldb, c0, ldc)
        assert np.array_equal(got, ref), rel_err(got, ref)


def test_sgemm_k_zero_and_empty(dev):
    c0 = np.arange(12, dtype=np.float32).reshape(3, 4)
    a = np.zeros((3, 1), np.float32)
    b = np.zeros((1, 4), np.float32)
    got = run_sgemm("R", "N", "N", 3, 4, 0, 1.0, 2.0, a, 1, b, 4, c0, 4)
    assert np.array_equal(got, 2.0 * c0)
    got = run_sgemm("R", "N", "N", 0, 4, 5, 1.0, 2.0, a, 5, b, 4, c0, 4)
    assert np.array_equal(got, c0)


def test_sgemm_bad_args(dev):
    with pytest.raises(bofhip.BofError):
        bofhip.sgemm("X", "N", "N", 1, 1, 1, 1.0, 0, 1, 0, 1, 0.0, 0, 1)


@pytest.mark.parametrize("ord_,ta,tb", [("R", "N", "N"), ("R", "T", "N"), ("C", "N", "T"),
                                        ("C", "T", "T")])
def test_gemm_resident_matches_flash_oracle(dev, ord_, ta, tb):
    """Tile DAG with tail-merge (640x600x500, tile 256: m 256+256+128, k 256+344,
    n 256+244 -- SURVEY App. D-1 example) against the restated flash::gemm."""
    m, k, n, blk = 640, 600, 500, 256
    alpha, beta = 0.5, 2.0
    rng = np.random.default_rng(7)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, alpha, beta, a, b, c0.copy(), 0, 0, 0, blk)
    da, db, dc = to_dev(a), to_dev(b), to_dev(c0)
    opts = bofhip.default_options(gemm_blk=blk, n_streams=3)
    bofhip.gemm_resident(ord_, ta, tb, m, n, k, alpha, beta, ptr(da), ptr(db), ptr(dc), 0, 0, 0,
                         opts, stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy()
    assert np.array_equal(got, ref), rel_err(got, ref)
    whole = orc.sgemm(ord_, ta, tb, m, n, k, alpha, a, sa[1], b, sb[1], beta, c0.copy(), sc[1])
    assert np.array_equal(got, whole)   # the default arithmetic IS the in-memory driver's single call (gemm_run.sh comparison)
    # bof_options.gemm_chain = 1: the reference's tile DAG, task by task on the forked streams, one rounding per
    # k-block -- bit-equal to the tile-by-tile restatement of flash::gemm, within rounding of the single chain
    chained = orc.flash_gemm(ord_, ta, tb, m, n, k, alpha, beta, a, b, c0.copy(), 0, 0, 0, blk, chain=1)
    dc1 = to_dev(c0)
    bofhip.gemm_resident(ord_, ta, tb, m, n, k, alpha, beta, ptr(da), ptr(db), ptr(dc1), 0, 0, 0,
                         bofhip.default_options(gemm_blk=blk, n_streams=3, gemm_chain=1), stream())
    torch.cuda.synchronize()
    got1 = dc1.cpu().numpy()
    assert np.array_equal(got1, chained), rel_err(got1, chained)
    assert rel_err(got1, whole) < TOL


@pytest.mark.parametrize("ord_,ta,tb", list(__import__("itertools").product("RC", "NT", "NT")))
def test_gemm_resident_pretransposed_operands(dev, ord_, ta, tb):
    """Shapes on which bof_gemm_resident replaces x-major operands by k-major copies (>= 4 tiles
    of reuse per operand tile, k % 32 == 0, m, n >= 2048): all 8 layouts, padded leading
    dimensions, bit-exact against the restated flash::gemm."""
    m, k, n, blk = 2304, 512, 2048, 512
    alpha, beta = 0.5, 2.0
    rng = np.random.default_rng(13)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    lda, ldb, ldc = sa[1] + 8, sb[1] + 4, sc[1] + 12
    a = rng.uniform(-1, 1, (sa[0], lda)).astype(np.float32)
    b = rng.uniform(-1, 1, (sb[0], ldb)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (sc[0], ldc)).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, alpha, beta, a, b, c0.copy(), lda, ldb, ldc, blk)
    da, db, dc = to_dev(a), to_dev(b), to_dev(c0)
    opts = bofhip.default_options(gemm_blk=blk, n_streams=2)
    bofhip.gemm_resident(ord_, ta, tb, m, n, k, alpha, beta, ptr(da), ptr(db), ptr(dc), lda, ldb, ldc,
                         opts, stream())
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)
    assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(db.cpu().numpy(), b)  # inputs untouched


def test_gemm_4096_generator_known_answer(dev):
    """cfg1 known answer: dense_create 's' inputs, 4096^3,
    C[0,0:4] = [81850, 100270, 73670, 92090]; full check against the closed form
    (C[i,j] depends on (i mod 5, j mod 10) because 4096 = 6 mod 10)."""
    n = 4096
    a = torch.empty(n * n, dtype=torch.float32, device=dev)
    c = torch.empty(n * n, dtype=torch.float32, device=dev)
    bofhip.gen_dense(ptr(a), 0, n * n, "s", 0, stream())
    bofhip.gemm_resident("R", "N", "N", n, n, n, 1.0, 0.0, ptr(a), ptr(a), ptr(c), 0, 0, 0,
                         bofhip.default_options(gemm_blk=1024), stream())
    torch.cuda.synchronize()
    C = c.view(n, n)
    assert C[0, :4].tolist() == [81850.0, 100270.0, 73670.0, 92090.0]
    # closed form: rows i and i+5 equal, columns j and j+10 equal
    idx = torch.arange(n, device=dev)
    pat = C[:5, :10]
    assert torch.equal(C, pat[idx % 5][:, idx % 10])
    A64 = (np.arange(5 * n, dtype=np.int64) % 10).reshape(5, n).astype(np.float64)
    B64 = ((np.arange(n)[:, None] * n + np.arange(10)[None, :]) % 10).astype(np.float64)
    assert np.array_equal(pat.cpu().numpy().astype(np.float64), A64 @ B64)


@pytest.fixture(scope="module")
def golden_big():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return np.load(os.path.join(root, "tests", "golden", "mkl_golden_big.npz"))


def test_gen_dense_u_matches_numpy_restatement(dev):
    """The inputs of mkl_golden_big.npz are regenerated, not stored: device generator == numpy."""
    from gen_u import dense_u
    for first, count, seed in [(0, 1 << 20, 11), (123456789, 300000, 12), (2 ** 33 + 5, 4096, 13)]:
        t = torch.empty(count, dtype=torch.float32, device=dev)
        bofhip.gen_dense(ptr(t), first, count, "u", seed, stream())
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), dense_u(first, count, seed))


@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
def test_big_kernels_vs_mkl_random(dev, golden_big, ord_, ta, tb):
    """4096 x 2048 x 1024, uniform [-1,1) inputs, alpha = 0.5, beta = 2: the 256x256 MFMA kernels
    (LDS-DMA staging for k-major x k-major operands, register staging otherwise) and the level-2
    tile DAG with 1024-tiles (one task per accumulate chain) and 256-tiles (512 tasks, chains of
    4) against cblas_sgemm (MKL 2021.4) sub-blocks, 1e-4 relative (BASELINE north_star)."""
    meta = str(golden_big["meta"][1]).split()
    m, n, k = int(meta[1]), int(meta[2]), int(meta[3])
    alpha, beta = float(meta[5]), float(meta[7])
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = torch.empty(sa[0] * sa[1], dtype=torch.float32, device=dev)
    b = torch.empty(sb[0] * sb[1], dtype=torch.float32, device=dev)
    c0 = torch.empty(sc[0] * sc[1], dtype=torch.float32, device=dev)
    bofhip.gen_dense(ptr(a), 0, a.numel(), "u", 11, stream())
    bofhip.gen_dense(ptr(b), 0, b.numel(), "u", 12, stream())
    bofhip.gen_dense(ptr(c0), 0, c0.numel(), "u", 13, stream())
    want = golden_big[f"{ord_}{ta}{tb}"]
    blocks = golden_big["blocks"]

    def check(c, what):
        C = c.view(sc).cpu().numpy()
        L = C if ord_ == "R" else C.T
        for (r, q), ref in zip(blocks, want):
            assert rel_err(L[r:r + 64, q:q + 64], ref) < TOL, (what, int(r), int(q))

    c = c0.clone()
    bofhip.sgemm(ord_, ta, tb, m, n, k, alpha, ptr(a), sa[1], ptr(b), sb[1], beta, ptr(c), sc[1], stream())
    torch.cuda.synchronize()
    check(c, "bof_sgemm")
    for blk in (1024, 256):
        c = c0.clone()
        bofhip.gemm_resident(ord_, ta, tb, m, n, k, alpha, beta, ptr(a), ptr(b), ptr(c), 0, 0, 0,
                             bofhip.default_options(gemm_blk=blk), stream())
        torch.cuda.synchronize()
        check(c, f"bof_gemm_resident blk={blk}")


@pytest.mark.parametrize("ta,tb,beta", [("T", "N", 0.0), ("N", "N", 1.5), ("N", "T", 0.0)])
def test_sgemm_big_ragged_strips_on_the_side_stream(dev, ta, tb, beta):
    """A launch whose 256-aligned interior has >= 1024 tiles runs its ragged strips (guarded 128 x 128 kernel) on a
    stream of their own BESIDE the interior kernel, forked in front of it and joined behind it (gemm_f32_mfma.hip,
    launch_modes).  8292 x 8244 x 512: interior 32 x 32 tiles, a 100-row bottom strip, a 52-column right strip; the
    strips, the corner, the rows / columns next to them and a band of the interior against the oracle bit for bit;
    then work queued on the SAME stream right behind the call (it must see the strips: the join), and the call
    again with BOF_GEMM_STRIP_STREAM=0 (strips behind the interior, as until round 5) -- same bits."""
    m, n, k = 8292, 8244, 512
    rng = np.random.default_rng(77)
    sa, sb, sc = stored_shapes("R", ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    da, db = to_dev(a), to_dev(b)

    def rows_of(r0, r1):          # the oracle on C rows [r0, r1)
        asub = a[r0:r1] if ta == "N" else np.ascontiguousarray(a[:, r0:r1])
        c = c0[r0:r1].copy()
        orc.sgemm("R", ta, tb, r1 - r0, n, k, 0.5, asub, asub.shape[1], b, sb[1], beta, c, n)
        return c

    def cols_of(c0_, c1_):        # the oracle on C columns [c0_, c1_)
        bsub = np.ascontiguousarray(b[:, c0_:c1_]) if tb == "N" else b[c0_:c1_]
        c = np.ascontiguousarray(c0[:, c0_:c1_])
        orc.sgemm("R", ta, tb, m, c1_ - c0_, k, 0.5, a, sa[1], bsub, bsub.shape[1], beta, c, c1_ - c0_)
        return c
    want_rows = {(0, 64): rows_of(0, 64), (8128, m): rows_of(8128, m)}
    want_cols = {(8128, n): cols_of(8128, n)}
    for knob in ("1", "0"):
        os.environ["BOF_GEMM_STRIP_STREAM"] = knob
        try:
            dc = to_dev(c0)
            tail = torch.zeros(1, dtype=torch.float32, device=dev)
            bofhip.sgemm("R", ta, tb, m, n, k, 0.5, ptr(da), sa[1], ptr(db), sb[1], beta, ptr(dc), n, stream())
            tail += dc[m - 1, n - 1]          # queued on the same stream right behind the call: sees the corner of the strips
            torch.cuda.synchronize()
            got = dc.cpu().numpy()
        finally:
            os.environ.pop("BOF_GEMM_STRIP_STREAM", None)
        for (r0, r1), w in want_rows.items():
            assert np.array_equal(got[r0:r1], w), (knob, r0, r1)
        for (c0_, c1_), w in want_cols.items():
            assert np.array_equal(got[:, c0_:c1_], w), (knob, c0_, c1_)
        assert float(tail.item()) == float(got[m - 1, n - 1])


# ---- round 6: the LDS-DMA tile kernel with progress counters instead of s_barrier (sgemm_tile256_dma2_kernel<EP, 0, 1, 1>) ----
def _both_syncs(monkeypatch, run):
    """run() with the round-6 default kernel and with the s_barrier kernel of rounds 2-5 ($BOF_GEMM_DMA2_SYNC=0, read at
    every launch): the two C must be bit-identical."""
    out = []
    for sync in ("1", "0"):
        monkeypatch.setenv("BOF_GEMM_DMA2_SYNC", sync)
        out.append(run())
    monkeypatch.delenv("BOF_GEMM_DMA2_SYNC")
    return out


@pytest.mark.parametrize("m,n,k", [(2048, 4096, 512), (4096, 2048, 576), (2304, 4096, 1024), (2048, 4352, 4160),
                                   (2100, 4200, 640), (4096, 4096, 64 * 37), (8192, 2048, 128 * 5)])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (0.5, 2.0)])
def test_dma2_counter_kernel_equals_barrier_kernel(dev, monkeypatch, m, n, k, alpha, beta):
    """'T','N' operands (k-major x k-major: the LDS-DMA kernel) on shapes with >= 128 interior tiles and K % 64 == 0,
    incl. ragged edges (strips on the guarded kernel) and slab counts that are odd multiples of the slab pair: the
    kernel whose four waves synchronise through LDS counters must store exactly what the barrier kernel stores, and
    both the k-ordered fmaf chain of the oracle (checked on sampled rows in float64-free form: torch fp32 matmul is NOT
    the reference here, the oracle's chain is -- so a 96-row band goes through orc.sgemm)."""
    g = torch.Generator(device="cpu").manual_seed(m * 7 + n * 3 + k)
    a = torch.rand(k, m, generator=g) * 2 - 1          # 'T': stored [k][m]
    b = torch.rand(k, n, generator=g) * 2 - 1          # 'N': stored [k][n]
    c0 = torch.rand(m, n, generator=g) * 2 - 1
    da, db = a.cuda(), b.cuda()

    def run():
        dc = c0.cuda()
        bofhip.sgemm("R", "T", "N", m, n, k, alpha, da.data_ptr(), m, db.data_ptr(), n, beta, dc.data_ptr(), n, stream())
        torch.cuda.synchronize()
        return dc.cpu()
    new, old = _both_syncs(monkeypatch, run)
    assert torch.equal(new.view(torch.int32), old.view(torch.int32))
    rows = 96
    band = orc.sgemm("R", "T", "N", rows, n, k, alpha, np.ascontiguousarray(a.numpy()[:, :rows]), rows, b.numpy(), n, beta,
                     c0.numpy()[:rows].copy(), n)
    assert np.array_equal(new.numpy()[:rows], band)


def test_dma2_counter_kernel_kmeans_epilogue(dev, monkeypatch):
    """The Rank1x2 instantiation (flash::kmeans' task: the product + two rank-1 terms in the tile store,
    include/tasks/kmeans_task.h:53-82) of the counter kernel against the barrier kernel, K = 512."""
    m, n, k = 2048, 4096, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    a = (torch.rand(k, m, generator=g) * 2 - 1).cuda()
    b = (torch.rand(k, n, generator=g) * 2 - 1).cuda()
    u = torch.rand(m, generator=g).cuda()
    v = torch.rand(n, generator=g).cuda()
    ones = torch.ones(max(m, n)).cuda()

    def run():
        dc = torch.zeros(m, n).cuda()
        bofhip.skmeans_task("R", "T", "N", m, n, k, -2.0, a.data_ptr(), m, b.data_ptr(), n, 0.0, dc.data_ptr(), n,
                            u.data_ptr(), v.data_ptr(), ones.data_ptr(), stream())
        torch.cuda.synchronize()
        return dc.cpu()
    new, old = _both_syncs(monkeypatch, run)
    assert torch.equal(new.view(torch.int32), old.view(torch.int32))
    assert torch.isfinite(new).all() and float(new.abs().max()) > 1.0


@pytest.mark.parametrize("beta", [0.0, 2.0])
def test_dma2_counter_kernel_in_the_panel_pipeline(dev, tmp_path, monkeypatch, beta):
    """The <ChainEpi> instantiation where the product uses it: flash::gemm on files through the row panels with
    2048-tiles (ramp launches 2048 x 4096 x 2048 = 128 tiles each, raw sums handed from k-block to k-block, the last
    panel in row slices): the C file with the counter kernel == the C file with the barrier kernel == one bof_sgemm
    over the whole matrices (the default arithmetic does not depend on the cut)."""
    from test_gpu_flash import Files
    monkeypatch.setenv("BOF_PANEL_RAMP_K", "1")          # one k-block per ramp launch: the chain hands raw sums on
    n = 4096
    rng = np.random.default_rng(41)
    a = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    da, db, dc = to_dev(a), to_dev(b), to_dev(c0)
    bofhip.sgemm("R", "N", "N", n, n, n, 0.5, ptr(da), n, ptr(db), n, beta, ptr(dc), n, stream())
    torch.cuda.synchronize()
    whole = dc.cpu().numpy()
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        def run():
            c0.tofile(F.paths["c"])
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
            bofhip.flash_gemm("R", "N", "N", n, n, n, 0.5, beta, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                              bofhip.default_options(gemm_blk=2048, gemm_path=2, panel_group=1, io_chunk_mib=4, panel_kmajor=3))
            mix = bofhip.flash_last_launch_mix()
            assert mix["chain_k_ranges"] == 2 and mix["whole_k_row_slices"] >= 2, mix
            return F.read("c", np.float32, (n, n))
        # default: A's row-major panels straight through the x-major DMA kernel (dmax), no k-major copy of A
        dmax = run()
        # $BOF_GEMM_DMAX=0: k-major copies of A's panels + the k-major DMA kernel, with counters and with s_barrier
        monkeypatch.setenv("BOF_GEMM_DMAX", "0")
        new, old = _both_syncs(monkeypatch, run)
        monkeypatch.delenv("BOF_GEMM_DMAX")
        assert np.array_equal(new.view(np.uint32), old.view(np.uint32))
        assert np.array_equal(new.view(np.uint32), whole.view(np.uint32))
        assert np.array_equal(dmax.view(np.uint32), whole.view(np.uint32))
    finally:
        F.close()


@pytest.mark.parametrize("m,n,k", [(2048, 4096, 512), (4096, 2048, 576), (2304, 4096, 1024), (2048, 4352, 4160),
                                   (2100, 4200, 640), (4096, 4096, 64 * 37), (8192, 2048, 128 * 5)])
@pytest.mark.parametrize("alpha,beta,pad", [(1.0, 0.0, 0), (0.5, 2.0, 4)])
@pytest.mark.parametrize("ta,tb", [("N", "N"), ("N", "T"), ("T", "T")])
def test_dmax_kernel_equals_register_staged_kernel(dev, monkeypatch, ta, tb, m, n, k, alpha, beta, pad):
    """Every layout with an x-major operand ('N','N' = the reference's own: A row-major; 'N','T'; 'T','T') through
    sgemm_tile256_dmax_kernel -- the x-major operand's rows by XOR-swizzled LDS-DMA, no k-major copy -- against the
    register-staged kernels of rounds 1-5 ($BOF_GEMM_DMAX=0) bit for bit, padded leading dimensions and ragged edges
    included, and a 96-row band against the oracle's k-ordered fmaf chain."""
    g = torch.Generator(device="cpu").manual_seed(m * 5 + n * 11 + k + ord(ta) + 3 * ord(tb))
    sa = (m, k) if ta == "N" else (k, m)
    sb = (k, n) if tb == "N" else (n, k)
    lda, ldb = sa[1] + pad, sb[1] + 2 * pad
    a = torch.rand(sa[0], lda, generator=g) * 2 - 1
    b = torch.rand(sb[0], ldb, generator=g) * 2 - 1
    c0 = torch.rand(m, n, generator=g) * 2 - 1
    da, db = a.cuda(), b.cuda()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("BOF_GEMM_DMAX", flag)
        dc = c0.cuda()
        bofhip.sgemm("R", ta, tb, m, n, k, alpha, da.data_ptr(), lda, db.data_ptr(), ldb, beta, dc.data_ptr(), n, stream())
        torch.cuda.synchronize()
        outs.append(dc.cpu())
    monkeypatch.delenv("BOF_GEMM_DMAX")
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    rows = 96
    a_band = a.numpy()[:rows].copy() if ta == "N" else np.ascontiguousarray(a.numpy()[:, :rows])
    band = orc.sgemm("R", ta, tb, rows, n, k, alpha, a_band, lda if ta == "N" else rows, b.numpy(), ldb, beta,
                     c0.numpy()[:rows].copy(), n)
    assert np.array_equal(outs[0].numpy()[:rows], band)


def test_dmax_kernel_column_major_and_kmeans_epilogue(dev, monkeypatch):
    """Column-major 'N','N' runs as the row-major product of the swapped operands (again x-major x k-major: dmax), and
    the Rank1x2 instantiation (flash::kmeans' task) of the same kernel: both against $BOF_GEMM_DMAX=0."""
    m, n, k = 4096, 2048, 512
    g = torch.Generator(device="cpu").manual_seed(9)
    a = (torch.rand(k, m, generator=g) * 2 - 1).cuda()        # column-major m x k: stored [k][m]
    b = (torch.rand(n, k, generator=g) * 2 - 1).cuda()        # column-major k x n: stored [n][k]
    u = torch.rand(m, generator=g).cuda()
    v = torch.rand(n, generator=g).cuda()
    ones = torch.ones(max(m, n)).cuda()
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("BOF_GEMM_DMAX", flag)
        c1 = torch.zeros(n, m).cuda()                         # column-major m x n: stored [n][m]
        bofhip.sgemm("C", "N", "N", m, n, k, 1.0, a.data_ptr(), m, b.data_ptr(), k, 0.0, c1.data_ptr(), m, stream())
        c2 = torch.zeros(n, m).cuda()
        bofhip.skmeans_task("C", "N", "N", m, n, k, -2.0, a.data_ptr(), m, b.data_ptr(), k, 0.0, c2.data_ptr(), m,
                            u.data_ptr(), v.data_ptr(), ones.data_ptr(), stream())
        torch.cuda.synchronize()
        res[flag] = (c1.cpu(), c2.cpu())
    monkeypatch.delenv("BOF_GEMM_DMAX")
    assert torch.equal(res["1"][0].view(torch.int32), res["0"][0].view(torch.int32))
    assert torch.equal(res["1"][1].view(torch.int32), res["0"][1].view(torch.int32))
    ref = (b.cpu().double() @ a.cpu().double()).float()      # [n][m] = (A B)^T
    assert (res["1"][0] - ref).abs().max() / ref.abs().max() < 1e-5


@pytest.mark.parametrize("ord_,ta,tb", [("R", "N", "N"), ("R", "T", "N"), ("R", "N", "T"), ("C", "N", "N"), ("C", "T", "T")])
@pytest.mark.parametrize("m,n,k", [(2048, 4096, 1024 + 4), (4096, 2048, 2048 + 37), (2304, 4352, 1600 + 32), (2100, 4200, 1100)])
def test_ragged_k_runs_as_two_launches_of_one_chain(dev, monkeypatch, ord_, ta, tb, m, n, k):
    """K % 64 != 0 on a product big enough for the LDS-DMA kernels (round 6, sgemm_rm_ksplit): K - K % 64 through the
    DMA kernel with its raw sums stored in C, the last < 64 k through the guarded kernel starting from them -- one
    chain, so the result equals the single launch of the register-staged kernel ($BOF_GEMM_KSPLIT=0) bit for bit,
    and a 64-row band equals the oracle's k-ordered chain."""
    g = torch.Generator(device="cpu").manual_seed(m + n + k)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = torch.rand(*sa, generator=g) * 2 - 1
    b = torch.rand(*sb, generator=g) * 2 - 1
    da, db = a.cuda(), b.cuda()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("BOF_GEMM_KSPLIT", flag)
        dc = torch.full(sc, float("nan")).cuda()            # beta == 0: whatever C held must not matter
        bofhip.sgemm(ord_, ta, tb, m, n, k, 0.75, da.data_ptr(), sa[1], db.data_ptr(), sb[1], 0.0, dc.data_ptr(), sc[1], stream())
        torch.cuda.synchronize()
        outs.append(dc.cpu())
    monkeypatch.delenv("BOF_GEMM_KSPLIT")
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    ref = orc.sgemm(ord_, ta, tb, m, n, k, 0.75, a.numpy(), sa[1], b.numpy(), sb[1], 0.0, np.zeros(sc, np.float32), sc[1]) \
        if m * n * k < 2.5e10 else None
    if ref is not None:
        assert np.array_equal(outs[0].numpy(), ref)


@pytest.mark.parametrize("beta", [0.0, 2.0])
def test_ragged_k_in_the_panel_pipeline(dev, tmp_path, monkeypatch, beta):
    """flash::gemm on files with K = 4100 and 2048-tiles: the k tiles are 2048 + 2052 (merged tail), so the second ramp
    launch and the whole-K launches have K % 64 = 4 -- each runs as two launches of its chain.  Against one bof_sgemm over
    the whole matrices and against the unsplit run."""
    from test_gpu_flash import Files
    monkeypatch.setenv("BOF_PANEL_RAMP_K", "1")
    m, n, k = 4096, 4096, 4100
    rng = np.random.default_rng(43)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        res = []
        for flag in ("1", "0"):
            monkeypatch.setenv("BOF_GEMM_KSPLIT", flag)
            c0.tofile(F.paths["c"])
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
            bofhip.flash_gemm("R", "N", "N", m, n, k, 0.5, beta, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                              bofhip.default_options(gemm_blk=2048, gemm_path=2, panel_group=1, io_chunk_mib=4, verify=1))
            res.append(F.read("c", np.float32, (m, n)))
        monkeypatch.delenv("BOF_GEMM_KSPLIT")
        assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))
        da, db, dc = to_dev(a), to_dev(b), to_dev(c0)
        monkeypatch.setenv("BOF_GEMM_KSPLIT", "0")
        bofhip.sgemm("R", "N", "N", m, n, k, 0.5, ptr(da), k, ptr(db), n, beta, ptr(dc), n, stream())
        torch.cuda.synchronize()
        assert np.array_equal(res[0].view(np.uint32), dc.cpu().numpy().view(np.uint32))
    finally:
        F.close()
