"""-m gpu parity of flash::kmeans (KMeansTask fused into the GEMM store) at the three
levels of the C ABI: against the real-MKL fixtures of the task's three cblas_sgemm calls (1e-4) and
against the oracle's restatement (bit-exact: same k-ordered chains, same two roundings per update)."""
import itertools
import os

import numpy as np
import pytest
import torch

import bofhip
import orc
from gpu_util import ptr, rel_err, stream, to_dev
from test_gpu_flash import Files, stored_shapes
from test_oracle import _kmeans_cases

pytestmark = pytest.mark.gpu


def run_task(ord_, ta, tb, m, n, k, alpha, beta, a, lda, b, ldb, c, ldc, cl, pl, ones):
    da, db, dc, dcl, dpl, do = (to_dev(x) for x in (a, b, c, cl, pl, ones))
    bofhip.skmeans_task(ord_, ta, tb, m, n, k, alpha, ptr(da), lda, ptr(db), ldb, beta, ptr(dc), ldc, ptr(dcl),
                        ptr(dpl), ptr(do), stream())
    torch.cuda.synchronize()
    return dc.cpu().numpy()


def test_kmeans_task_vs_mkl_golden(dev):
    """bof_skmeans_task against KMeansTask::execute's three cblas_sgemm calls made into MKL, and bit for
    bit against the oracle on the same inputs."""
    for key, ta, tb, m, n, k, lda, ldb, ldc, alpha, beta, d in _kmeans_cases():
        got = run_task("C", ta, tb, m, n, k, alpha, beta, d["a"], lda, d["b"], ldb, d["c0"], ldc, d["cl"], d["pl"],
                       d["ones"])
        assert rel_err(got, d["c"]) < 1e-4, key
        ref = orc.skmeans_task("C", ta, tb, m, n, k, alpha, d["a"], lda, d["b"], ldb, beta, d["c0"].copy(), ldc,
                               d["cl"], d["pl"], d["ones"])
        assert np.array_equal(got, ref), key


@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
@pytest.mark.parametrize("m,n,k,alpha,beta", [
    (300, 200, 100, -2.0, 0.0),       # ragged: guarded 128 x 128 kernel
    (4096, 2304, 64, -2.0, 0.5),      # 256 x 256 kernels + right strip, K = 2 slabs
    (4352, 2048, 96, 1.5, 0.0),       # K % 64 != 0: register-staging kernel, bottom strip
    (2304, 4096, 80, -2.0, 1.0),      # K % 32 != 0: guarded last slab
])
def test_kmeans_task_vs_oracle_all_layouts(dev, ord_, ta, tb, m, n, k, alpha, beta):
    """Every kernel shape behind bof_skmeans_task, all 8 layouts, non-constant `ones` (so a swapped
    factor or index shows), bit-exact against the oracle."""
    rng = np.random.default_rng(m + n + k)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    cl = rng.uniform(0, 8, m).astype(np.float32)
    pl = rng.uniform(0, 8, n).astype(np.float32)
    ones = rng.uniform(0.5, 1.5, max(m, n)).astype(np.float32)
    got = run_task(ord_, ta, tb, m, n, k, alpha, beta, a, sa[1], b, sb[1], c0, sc[1], cl, pl, ones)
    ref = orc.skmeans_task(ord_, ta, tb, m, n, k, alpha, a, sa[1], b, sb[1], beta, c0.copy(), sc[1], cl, pl, ones)
    assert np.array_equal(got, ref.reshape(sc))


@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
@pytest.mark.parametrize("m,n,k,alpha,beta,wgs", [(2048, 2048, 256, -2.0, 0.0, 8), (1024, 4096, 96, 1.5, 0.5, 8),
                                                  (2304, 1792, 128, 1.0, 1.0, 24), (1280, 1536, 320, 0.5, 0.0, 256)])
def test_short_k_persistent_kernel_vs_oracle(dev, monkeypatch, ord_, ta, tb, m, n, k, alpha, beta, wgs):
    """K < 512 with many aligned tiles: sgemm_tile256_p1w3_kernel (one workgroup per CU walks a run of
    256 x 256 tiles, the hand-scheduled slab pipeline continues across tile boundaries; K % 64 == 0, K >= 128),
    all 8 layouts, with the kmeans store (non-constant `ones`) and as a plain sgemm; bit-exact against the
    oracle's k-ordered chains.  The tile threshold is lowered and the number of workgroups cut down so that
    small problems give runs of several tiles (8 workgroups x 8 tiles; 24 x 2-3 of unequal length; 256
    workgroups for 30 tiles: most with none); K = 96 takes the 128 x 128 kernel instead."""
    monkeypatch.setenv("BOF_GEMM_PERSIST_MIN_TILES", "1")
    monkeypatch.setenv("BOF_GEMM_PERSIST_WGS", str(wgs))
    rng = np.random.default_rng(m + n + k)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    cl = rng.uniform(0, 8, m).astype(np.float32)
    pl = rng.uniform(0, 8, n).astype(np.float32)
    ones = rng.uniform(0.5, 1.5, max(m, n)).astype(np.float32)
    got = run_task(ord_, ta, tb, m, n, k, alpha, beta, a, sa[1], b, sb[1], c0, sc[1], cl, pl, ones)
    ref = orc.skmeans_task(ord_, ta, tb, m, n, k, alpha, a, sa[1], b, sb[1], beta, c0.copy(), sc[1], cl, pl, ones)
    assert np.array_equal(got, ref.reshape(sc))
    da, db, dc = to_dev(a), to_dev(b), to_dev(c0)
    bofhip.sgemm(ord_, ta, tb, m, n, k, alpha, ptr(da), sa[1], ptr(db), sb[1], beta, ptr(dc), sc[1], stream())
    torch.cuda.synchronize()
    ref = orc.sgemm(ord_, ta, tb, m, n, k, alpha, a, sa[1], b, sb[1], beta, c0.copy(), sc[1])
    assert np.array_equal(dc.cpu().numpy(), ref.reshape(sc))


@pytest.mark.parametrize("ord_,ta,tb", [("C", "T", "N"), ("C", "N", "T"), ("R", "N", "T"), ("R", "T", "N")])
def test_kmeans_resident_vs_flash_oracle(dev, ord_, ta, tb):
    """Tile DAG (level 2): tail-merged tiles, k spanning two blocks (the updates are then added twice,
    as the reference does), per-tile slices of the norm vectors."""
    m, n, k, blk = 640, 500, 600, 256
    rng = np.random.default_rng(3)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    cl = rng.uniform(0, 8, m).astype(np.float32)
    pl = rng.uniform(0, 8, n).astype(np.float32)
    ones = rng.uniform(0.5, 1.5, 512).astype(np.float32)   # largest tile edge: 640 - 256 = 384
    ref = orc.flash_kmeans(ord_, ta, tb, m, n, k, -2.0, 0.5, a, b, c0.copy(), 0, 0, 0, blk, cl, pl, ones)
    da, db, dc, dcl, dpl, do = (to_dev(x) for x in (a, b, c0, cl, pl, ones))
    bofhip.kmeans_resident(ord_, ta, tb, m, n, k, -2.0, 0.5, ptr(da), ptr(db), ptr(dc), 0, 0, 0, ptr(dcl), ptr(dpl),
                           ptr(do), bofhip.default_options(gemm_blk=blk, n_streams=3), stream())
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("ord_,ta,tb", [("C", "T", "N"), ("R", "N", "T"), ("C", "N", "N")])
def test_flash_kmeans_files(dev, tmp_path, ord_, ta, tb, path):
    """Level 3: A, B, C as files, the norm vectors in host memory, through the tile cache (1) and the
    row-panel pipeline (2); bit-exact against the oracle's flash::kmeans."""
    m, n, k, blk = 640, 500, 300, 256
    rng = np.random.default_rng(17)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    cl = rng.uniform(0, 8, m).astype(np.float32)
    pl = rng.uniform(0, 8, n).astype(np.float32)
    ones = np.ones(max(m, n), np.float32)
    ref = orc.flash_kmeans(ord_, ta, tb, m, n, k, -2.0, 0.0, a, b, c0.copy(), 0, 0, 0, blk, cl, pl, ones)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=2, n_io_threads=3, pinned_slots=4, gemm_path=path,
                                      io_chunk_mib=1)
        bofhip.flash_kmeans(ord_, ta, tb, m, n, k, -2.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                            cl.ctypes.data, pl.ctypes.data, ones.ctypes.data, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
    finally:
        F.close()


def test_kmeans_driver_shape_distances(dev):
    """The reference driver's call (drivers/kmeans.cpp:37-39): dist = -2 centers^T points + |c|^2 + |p|^2,
    column-major ncenters x npoints -- checked against float64 squared distances, and the nearest centre
    of every point against the float64 answer."""
    ncenters, npoints, dim = 256, 20000, 96
    rng = np.random.default_rng(9)
    centers = rng.normal(0, 4, (ncenters, dim)).astype(np.float32)      # stored as the driver's file: centre after centre
    assign = rng.integers(0, ncenters, npoints)
    points = (centers[assign] + rng.normal(0, 0.2, (npoints, dim))).astype(np.float32)
    cl = (centers.astype(np.float64) ** 2).sum(1).astype(np.float32)
    pl = (points.astype(np.float64) ** 2).sum(1).astype(np.float32)
    ones = np.ones(max(ncenters, npoints), np.float32)
    dist = np.zeros((npoints, ncenters), np.float32)                    # column-major ncenters x npoints
    dc, dp, dd, dcl, dpl, do = (to_dev(x) for x in (centers, points, dist, cl, pl, ones))
    bofhip.kmeans_resident("C", "T", "N", ncenters, npoints, dim, -2.0, 0.0, ptr(dc), ptr(dp), ptr(dd), dim, dim,
                           ncenters, ptr(dcl), ptr(dpl), ptr(do), bofhip.default_options(), stream())
    torch.cuda.synchronize()
    got = dd.cpu().numpy().astype(np.float64)
    want = ((points.astype(np.float64)[:, None, :] - centers.astype(np.float64)[None, :, :]) ** 2).sum(2)
    assert np.abs(got - want).max() < 1e-4 * want.max()
    assert np.array_equal(got.argmin(1), assign) and np.array_equal(want.argmin(1), assign)


def test_kmeans_degenerate(dev):
    """k == 0 / empty C: nothing to do, C stays as it is (the reference's tiler divides by zero there)."""
    c = to_dev(np.full((4, 4), 7.0, np.float32))
    v = to_dev(np.ones(8, np.float32))
    bofhip.kmeans_resident("C", "T", "N", 4, 4, 0, 1.0, 0.0, ptr(c), ptr(c), ptr(c), 1, 1, 4, ptr(v), ptr(v), ptr(v),
                           bofhip.default_options(), stream())
    bofhip.kmeans_resident("C", "T", "N", 0, 4, 4, 1.0, 0.0, ptr(c), ptr(c), ptr(c), 4, 4, 1, ptr(v), ptr(v), ptr(v),
                           bofhip.default_options(), stream())
    torch.cuda.synchronize()
    assert np.all(c.cpu().numpy() == 7.0)


def test_kmeans_driver_binary(dev, tmp_path):
    """bin/kmeans_driver with the reference's argv (drivers/kmeans.cpp:198-201): one Lloyd iteration on
    files through the C++ flash::kmeans; the rewritten centres file against a float64 iteration."""
    from test_gpu_flash import run_driver
    ncenters, npoints, dim = 40, 5000, 48
    rng = np.random.default_rng(21)
    true = rng.normal(0, 5, (ncenters, dim))
    assign = rng.integers(0, ncenters, npoints)
    points = (true[assign] + rng.normal(0, 0.3, (npoints, dim))).astype(np.float32)
    centers = (true + rng.normal(0, 0.5, (ncenters, dim))).astype(np.float32)
    pp, cp = str(tmp_path / "points.bin"), str(tmp_path / "centers.bin")
    points.tofile(pp)
    centers.tofile(cp)
    out = run_driver("kmeans_driver", [pp, cp, npoints, dim, ncenters], {"BOF_GEMM_BLK_SIZE": "1024"})
    assert "flash::kmeans() returned with 0" in out
    d = ((points.astype(np.float64)[:, None, :] - centers.astype(np.float64)[None, :, :]) ** 2).sum(2)
    near = d.argmin(1)
    want = np.stack([points[near == c].astype(np.float64).mean(0) if np.any(near == c) else np.zeros(dim)
                     for c in range(ncenters)])
    got = np.fromfile(cp, np.float32).reshape(ncenters, dim)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())


def test_flash_kmeans_files_large_equals_resident(dev, tmp_path):
    """512 centres x 262144 points x 128 dims from files (64 tile tasks of 4096 points, a 512 MiB distance
    file) through the default level-3 path: bit for bit the level-2 result, and squared distances against
    float64 on a sample."""
    ncenters, npoints, dim = 512, 262144, 128
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    centers = torch.randn(ncenters, dim, device="cuda", generator=g)
    points = torch.randn(npoints, dim, device="cuda", generator=g)
    cl = (centers.double() ** 2).sum(1).float()
    pl = (points.double() ** 2).sum(1).float()
    ones = torch.ones(4096 + 128, device="cuda")
    dist = torch.empty(npoints, ncenters, device="cuda")
    bofhip.kmeans_resident("C", "T", "N", ncenters, npoints, dim, -2.0, 0.0, ptr(centers), ptr(points), ptr(dist), dim,
                           dim, ncenters, ptr(cl), ptr(pl), ptr(ones), bofhip.default_options(), stream())
    torch.cuda.synchronize()
    want = ((points[:1024].double()[:, None, :] - centers.double()[None, :, :]) ** 2).sum(2)
    assert float((dist[:1024].double() - want).abs().max() / want.max()) < 1e-5
    F = Files(tmp_path, a=centers.cpu().numpy(), b=points.cpu().numpy(), c=np.zeros((npoints, ncenters), np.float32))
    try:
        cl_h, pl_h, ones_h = cl.cpu().numpy(), pl.cpu().numpy(), ones.cpu().numpy()
        bofhip.flash_kmeans("C", "T", "N", ncenters, npoints, dim, -2.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), dim,
                            dim, ncenters, cl_h.ctypes.data, pl_h.ctypes.data, ones_h.ctypes.data,
                            bofhip.default_options())
        st = bofhip.flash_last_stats()
        assert st["tasks"] == 64
        got = F.read("c", np.float32, (npoints, ncenters))
        assert np.array_equal(got, dist.cpu().numpy())
    finally:
        F.close()


@pytest.mark.parametrize("nproc", [1, 2])
def test_flash_kmeans_point_sharded_files(dev, tmp_path, nproc):
    """Multi-GPU form: the distance matrix shards by points (tile-aligned slices of the points file, of
    p_l2sq and of the dist file), no collective; two ranks share cuda:0 here.  The dist file equals the
    oracle's flash::kmeans bit for bit whatever the number of ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ncenters, npoints, dim, blk = 96, 1500, 40, 256
    rng = np.random.default_rng(77)
    centers = rng.uniform(-1, 1, (ncenters, dim)).astype(np.float32)
    points = rng.uniform(-1, 1, (npoints, dim)).astype(np.float32)
    cl = (centers.astype(np.float64) ** 2).sum(1).astype(np.float32)
    pl = (points.astype(np.float64) ** 2).sum(1).astype(np.float32)
    pc, pp, pd = (str(tmp_path / f) for f in ("centers", "points", "dist"))
    centers.tofile(pc); points.tofile(pp); np.zeros((npoints, ncenters), np.float32).tofile(pd)
    tool = os.path.join(root, "tools", "dist_file_kmeans.py")
    cmd = [sys.executable, tool] if nproc == 1 else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
         "127.0.0.1", "--master-port", "29583", tool]
    r = subprocess.run(cmd + [pc, pp, pd, str(ncenters), str(npoints), str(dim), str(blk)], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, BOF_BENCH_ONE_GPU="1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(recs) == nproc and sum(x["points"] for x in recs) == npoints
    # every rank's slice starts on a tile boundary, so its tiles are tiles of the whole problem -- except that
    # a rank's LAST tile is cut where the next rank begins, which only matters for tail-merging; the oracle is
    # therefore evaluated per rank slice, as the sharded call defines it
    ref = np.zeros((npoints, ncenters), np.float32)
    p0 = 0
    for x in sorted(recs, key=lambda v: v["rank"]):
        cnt = x["points"]
        if cnt:
            part = np.zeros((cnt, ncenters), np.float32)
            orc.flash_kmeans("C", "T", "N", ncenters, cnt, dim, -2.0, 0.0, centers, points[p0:p0 + cnt].copy(), part, dim,
                             dim, ncenters, blk, cl, pl[p0:p0 + cnt].copy(), np.ones(max(ncenters, blk + 127), np.float32))
            ref[p0:p0 + cnt] = part
        p0 += cnt
    got = np.fromfile(pd, np.float32).reshape(npoints, ncenters)
    assert np.array_equal(got, ref)
    want = ((points.astype(np.float64)[:, None, :] - centers.astype(np.float64)[None, :, :]) ** 2).sum(2)
    assert np.abs(got - want).max() < 1e-5 * want.max()


def test_reference_kmeans_driver_unchanged(dev, tmp_path):
    """The reference's OWN drivers/kmeans.cpp, compiled unchanged against blas-on-flash_amd/include
    (oracle/Makefile; its cblas_sdot / isamin / saxpy calls on the mapped files resolve to the plain host
    loops of bof_host_blas1.h) and linked against the product libraries: one Lloyd iteration on files, the
    distance matrix through flash::kmeans on the GPU.  The rewritten centres against a float64 iteration."""
    from test_gpu_flash import run_ref_driver
    ncenters, npoints, dim = 32, 3000, 24
    rng = np.random.default_rng(33)
    true = rng.normal(0, 5, (ncenters, dim))
    assign = rng.integers(0, ncenters, npoints)
    points = (true[assign] + rng.normal(0, 0.3, (npoints, dim))).astype(np.float32)
    centers = (true + rng.normal(0, 0.5, (ncenters, dim))).astype(np.float32)
    pp, cp = str(tmp_path / "points.bin"), str(tmp_path / "centers.bin")
    points.tofile(pp)
    centers.tofile(cp)
    run_ref_driver("ref_kmeans_driver", [pp, cp, npoints, dim, ncenters], {"BOF_GEMM_BLK_SIZE": "1024"})
    d = ((points.astype(np.float64)[:, None, :] - centers.astype(np.float64)[None, :, :]) ** 2).sum(2)
    near = d.argmin(1)
    want = np.stack([points[near == c].astype(np.float64).mean(0) if np.any(near == c) else np.zeros(dim)
                     for c in range(ncenters)])
    got = np.fromfile(cp, np.float32).reshape(ncenters, dim)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())
