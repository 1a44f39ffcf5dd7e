"""world_size-2 gloo tests (CPU) of the multi-GPU sharding plumbing: row-block
shards are disjoint and cover the output, and the CSRGEMV 'T' partial-sum
all-reduce reproduces the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bof_dist  # noqa: E402
import orc  # noqa: E402


def test_row_shard_covers_and_aligns():
    for m, world, align in [(65536, 8, 4096), (32768, 8, 4096), (1000, 3, 128), (5, 8, 1), (640, 2, 256)]:
        spans = [bof_dist.row_shard(m, world, r, align) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == m
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0 and a0 <= a1
        assert all(s[0] % align == 0 for s in spans)
    assert bof_dist.row_shard(65536, 8, 3, 4096) == (3 * 8192, 4 * 8192)      # cfg4: 8192 rows each
    assert bof_dist.gemm_shard_args(65536, 65536, 65536, 0, 0, 8, 2, 4096) == (8192, 2 * 8192 * 65536, 2 * 8192 * 65536)


def test_csr_row_shard_balances_nnz():
    rng = np.random.default_rng(0)
    ia = np.concatenate([[0], np.cumsum(rng.integers(0, 50, 10000))]).astype(np.int64)
    spans = [bof_dist.csr_row_shard(ia, 4, r) for r in range(4)]
    assert spans[0][0] == 0 and spans[-1][1] == 10000
    nnz = [ia[b] - ia[a] for a, b in spans]
    assert max(nnz) - min(nnz) < 100
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0


def test_csrmm_rank_cuts_fall_on_pages_of_the_c_file(monkeypatch):
    """Two processes must never share a page of an O_DIRECT C file (one may keep O_DIRECT for its row blocks while its
    neighbour goes through the page cache: csrc/flash_csr.cpp, c_blocks_aligned): flash_csrmm_row_sharded cuts the
    ranks' row ranges where r0 * k * 4 is a multiple of 4096, whatever k."""
    import bofhip
    rng = np.random.default_rng(1)
    m = 50000
    ia = np.concatenate([[0], np.cumsum(rng.integers(0, 20, m))]).astype(np.int64)
    calls = []
    monkeypatch.setattr(bofhip, "flash_csrmm", lambda *a: calls.append(a))
    for k in (1, 3, 7, 8, 24, 100, 128, 130, 1000, 1024):
        cuts = set()
        for rank in range(3):
            monkeypatch.setattr(bof_dist, "_world_rank", lambda group=None, r=rank: (3, r))
            r0, r1 = bof_dist.flash_csrmm_row_sharded(m, 100, k, 1.0, 0.0, 3, 4, 5, "R", 6, 7, ia)
            cuts.update((r0, r1))
            if r1 > r0:
                fc = calls[-1][-2]                    # the C file pointer handed to the call
                assert fc.foffset == r0 * k * 4
        assert 0 in cuts and m in cuts
        for r in cuts - {0, m}:
            assert (r * k * 4) % 4096 == 0, (k, r)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    x = (np.arange(m) % 10).astype(np.float32)
    r0, r1 = bof_dist.csr_row_shard(ia, world, rank, align=128)
    # this rank's partial y_g = A_g^T x_g (on the GPU box: bof_flash_csrgemv 'T' on the slab)
    part = np.zeros(n, np.float32)
    # shard = offset `ia` pointer; val/ja stay whole-file (absolute offsets, as flash_ptr + r0)
    orc.flash_csrgemv("T", r1 - r0, n, val, ia[r0:r1 + 1], ja, x[r0:r1], part, 1000, 5000)
    y = torch.from_numpy(part)
    # the reduce-scatter + all-gather form must give the same vector (gloo builds without
    # reduce_scatter are skipped on that leg); n + 1 elements exercise the padding path
    y2 = torch.cat([y.clone(), torch.full((1,), float(rank + 1))])
    try:
        bof_dist.allreduce_partial(y2, algo="rs_ag")
        rs_ok = True
    except (RuntimeError, NotImplementedError):
        rs_ok = False
    bof_dist.allreduce_partial(y)
    if rs_ok:
        assert torch.equal(y2[:-1], y) and float(y2[-1]) == world * (world + 1) / 2
    np.save(os.path.join(out_dir, f"rs_{rank}.npy"), np.array([1 if rs_ok else 0]))
    # 'N' needs no collective: disjoint slices
    xn = (np.arange(n) % 10).astype(np.float32)
    yn = np.zeros(r1 - r0, np.float32)
    orc.flash_csrgemv("N", r1 - r0, n, val, ia[r0:r1 + 1], ja, xn, yn, 1000, 5000)
    np.save(os.path.join(out_dir, f"T_{rank}.npy"), y.numpy())
    np.save(os.path.join(out_dir, f"N_{rank}.npy"), np.concatenate([[r0, r1], yn]))
    dist.barrier()
    dist.destroy_process_group()


def test_csrgemv_two_rank_reduce(tmp_path, golden):
    import hashlib
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
    for r in range(world):
        assert int(np.load(tmp_path / f"rs_{r}.npy")[0]) == 1     # reduce-scatter + all-gather leg ran
        yT = np.load(tmp_path / f"T_{r}.npy")
        assert hashlib.sha256(yT.tobytes()).hexdigest() == want["gen_csrgemv_T"]
    parts = [np.load(tmp_path / f"N_{r}.npy") for r in range(world)]
    parts.sort(key=lambda p: p[0])
    assert parts[0][1] == parts[1][0]
    yN = np.concatenate([p[2:] for p in parts]).astype(np.float32)
    assert hashlib.sha256(yN.tobytes()).hexdigest() == want["gen_csrgemv_N"]


# ---- node-shared staging ring of the one-process-per-GPU GEMM (share_world > 1): host code only ---------
def _ring_worker(rank, world, name, n_chunks, chunk_bytes, n_slots, timeout_s, skip, q):
    import bofhip
    if rank in skip:                       # a rank that never shows up
        q.put((rank, None))
        return
    q.put((rank, int(bofhip.lib().bof_share_selftest(name.encode(), rank, world, n_chunks, chunk_bytes, n_slots,
                                                     timeout_s))))


def _run_ring(world, n_chunks, chunk_bytes, n_slots, timeout_s, skip=()):
    import bofhip
    name = f"/bof_test_{os.getpid()}_{world}_{n_slots}_{len(skip)}"
    bofhip.lib().bof_share_cleanup(name.encode())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, name, n_chunks, chunk_bytes, n_slots, timeout_s, skip, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    bofhip.lib().bof_share_cleanup(name.encode())
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(name[1:])]
    return res


@pytest.mark.parametrize("world,n_slots", [(2, 1), (2, 8), (3, 4), (4, 64)])
def test_share_ring_delivers_every_chunk_once_to_every_peer(world, n_slots):
    """What bof_dist.flash_gemm_row_sharded relies on for "B once per node": chunk c is produced by rank
    c % world and taken by all others through a ring of n_slots reused slots (a slot is refilled only after
    every peer has taken its chunk); content checked chunk by chunk; no GPU involved."""
    n_chunks = 150
    res = _run_ring(world, n_chunks, 256 << 10, n_slots, 30.0)
    for r in range(world):
        own = len(range(r, n_chunks, world))
        assert res[r] == n_chunks - own, res


def test_share_ring_missing_peer_times_out():
    """A rank that never delivers: the others give up after the timeout with -ETIMEDOUT (or -EIO once a
    peer has marked the ring failed) instead of waiting forever."""
    res = _run_ring(3, 20, 64 << 10, 4, 1.5, skip=(1,))
    assert res[1] is None
    assert res[0] in (-110, -5) and res[2] in (-110, -5), res


# ---- the whole one-process-per-GPU file GEMM on CPU: real ranks (gloo), the C library linked against the mock HIP
# runtime of tests/native/mock_hip.cpp instead of libamdhip64 (test infrastructure only: nothing of it ships) ------
def _build_mock_library(out_dir):
    import subprocess
    csrc = os.path.join(ROOT, "blas-on-flash_amd", "csrc")
    so = os.path.join(out_dir, "libbof_hip_mock.so")
    srcs = [os.path.join(csrc, f) for f in ("plan.cpp", "fileio.cpp", "uring_io.cpp", "flash_support.cpp", "flash_runtime.cpp",
                                             "flash_csr.cpp", "flash_gemm_panels.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I",
           os.path.join(ROOT, "include"), "-I", csrc] + srcs + ["-x", "c++", os.path.join(csrc, "c_api.hip"), "-x", "none",
                                                               os.path.join(ROOT, "tests", "native", "mock_hip.cpp"), "-o", so,
                                                               "-lpthread", "-ldl", "-lrt"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return so


def _use_mock_library(so):
    import ctypes as C
    import bofhip
    L = C.CDLL(so)
    for name, res, args in bofhip.SYMBOLS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    bofhip._lib = L


def _file_gemm_worker(rank, world, port, so, out_dir, m, n, k, blk):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MOCK_HIP_DEVICES"] = "1"
    os.environ["MOCK_HIP_ASYNC"] = "1"
    _use_mock_library(so)
    import bofhip
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fds = [os.open(os.path.join(out_dir, f"{x}.bin"), os.O_RDWR) for x in "ABC"]
    try:
        opts = bofhip.default_options(gemm_blk=blk, io_chunk_mib=1, n_io_threads=3, use_odirect=0)
        st = bof_dist.flash_gemm_row_sharded(m, n, k, 1.0, 0.0, fds[0], fds[1], fds[2], opts=opts)
        np.save(os.path.join(out_dir, f"stats_{rank}.npy"),
                np.array([st["rows"], st["bytes_read"], st["bytes_written"], st["bytes_peer"]], np.int64))
    finally:
        for fd in fds:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
    dist.barrier()
    dist.destroy_process_group()


def _ineligible_rank_worker(rank, world, port, so, out_dir, m, n, k, blk, through_bof_dist):
    """rank 1's call cannot take the row-panel path (gemm_path = 1).  through_bof_dist: the ranks agree beforehand and
    fall back together; else every rank passes the share options to the library itself and must get an error, soon."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MOCK_HIP_DEVICES"] = "1"
    os.environ["MOCK_HIP_ASYNC"] = "1"
    _use_mock_library(so)
    import bofhip
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fds = [os.open(os.path.join(out_dir, f"{x}.bin"), os.O_RDWR) for x in "ABC"]
    t0 = time.time()
    msg, extra = "no error", ""
    try:
        kw = dict(gemm_blk=blk, io_chunk_mib=1, n_io_threads=3, use_odirect=0, gemm_path=1 if rank == 1 else 0)
        if through_bof_dist:
            st = bof_dist.flash_gemm_row_sharded(m, n, k, 1.0, 0.0, fds[0], fds[1], fds[2], opts=bofhip.default_options(**kw))
            extra = f"{st['bytes_peer']} {st.get('b_once', '')}"
        else:
            r0, r1 = bof_dist.row_shard(m, world, rank, blk)
            name = f"/bof_test_{port}"
            bofhip.flash_gemm("R", "N", "N", r1 - r0, n, k, 1.0, 0.0, bofhip.FPtr(fds[0], r0 * k * 4), bofhip.FPtr(fds[1], 0),
                              bofhip.FPtr(fds[2], r0 * n * 4), k, n, n,
                              bofhip.default_options(share_world=world, share_rank=rank, share_name=name, **kw))
    except bofhip.BofError as e:
        msg = str(e)
    finally:
        for fd in fds:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
    with open(os.path.join(out_dir, f"outcome_{rank}.txt"), "w") as f:
        f.write(f"{time.time() - t0:.1f}\n{msg}\n{extra}\n")
    dist.barrier()
    if not through_bof_dist and rank == 0:
        bofhip.lib().bof_share_cleanup(f"/bof_test_{port}".encode())
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def mock_lib(tmp_path_factory):
    return _build_mock_library(str(tmp_path_factory.mktemp("mocklib")))


@pytest.mark.parametrize("through_bof_dist", [True, False])
def test_row_sharded_rank_outside_the_panel_path(tmp_path, mock_lib, through_bof_dist):
    """with B shared through the staging ring, a rank whose call is not eligible for the panel path used to
    fall back to the tile cache silently, read B itself and succeed, while its peers waited BOF_SHARE_TIMEOUT_S
    (120 s) for panels it never published and the successful rank then hung in the barrier.
    Through bof_dist the ranks now agree BEFORE the call (every member of a share group must be able to take the
    panel path, else the whole group reads B itself): every rank succeeds, C is exact, nothing comes from a peer.
    Handed the share options directly, the library does not fall back: the odd rank raises the ring group's failure
    word and returns BOF_EINVAL, the peers' waits end at once -- every rank gets an error within seconds."""
    world, blk, m, n, k = 3, 128, 128 * 6, 300, 128 * 4
    rng = np.random.default_rng(5)
    a = rng.integers(-3, 4, (m, k)).astype(np.float32)
    b = rng.integers(-3, 4, (k, n)).astype(np.float32)
    a.tofile(tmp_path / "A.bin")
    b.tofile(tmp_path / "B.bin")
    np.zeros((m, n), np.float32).tofile(tmp_path / "C.bin")
    before = set(os.listdir("/dev/shm"))
    mp.spawn(_ineligible_rank_worker, args=(world, _free_port(), mock_lib, str(tmp_path), m, n, k, blk, through_bof_dist),
             nprocs=world, join=True)
    outcomes = [open(tmp_path / f"outcome_{r}.txt").read().split("\n") for r in range(world)]
    for r, (secs, msg, extra, *_) in enumerate(outcomes):
        assert float(secs) < 30, (r, secs)                       # not the 120 s timeout
        if through_bof_dist:
            assert msg == "no error", (r, msg)
            assert extra.startswith("0 ") and "cannot take the row-panel path" in extra, (r, extra)
        else:
            assert "not eligible" in msg or "failed" in msg or "I/O pipeline" in msg, (r, msg)
    if through_bof_dist:
        got = np.fromfile(tmp_path / "C.bin", np.float32).reshape(m, n)
        assert np.array_equal(got, (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32))
    else:
        assert "not eligible" in outcomes[1][1]
    assert set(os.listdir("/dev/shm")) - before == set()


def _kmeans_csr_worker(rank, world, port, so, out_dir, shp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MOCK_HIP_DEVICES"] = "1"
    os.environ["MOCK_HIP_ASYNC"] = "1"
    _use_mock_library(so)
    import bofhip
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ncent, npts, dim, m, n, k = shp
    names = ["centers", "points", "dist", "val", "ia", "ja", "b", "c"]
    fd = {x: os.open(os.path.join(out_dir, f"{x}.bin"), os.O_RDWR) for x in names}
    try:
        cl, pl = np.load(os.path.join(out_dir, "cl.npy")), np.load(os.path.join(out_dir, "pl.npy"))
        opts = bofhip.default_options(gemm_blk=128, io_chunk_mib=1, n_io_threads=2, use_odirect=0, max_nnzs=700, csrmm_rblk=300)
        bof_dist.flash_kmeans_point_sharded(ncent, npts, dim, fd["centers"], fd["points"], fd["dist"], cl, pl, opts=opts)
        ia = np.fromfile(os.path.join(out_dir, "ia.bin"), np.int64)
        bof_dist.flash_csrmm_row_sharded(m, n, k, 2.0, 0.0, fd["val"], fd["ia"], fd["ja"], "R", fd["b"], fd["c"], ia, opts=opts)
        x = np.load(os.path.join(out_dir, "x.npy"))
        xt = np.load(os.path.join(out_dir, "xt.npy"))
        yn = np.zeros(m, np.float32)
        r0, r1 = bof_dist.flash_csrgemv_row_sharded("N", m, n, fd["val"], fd["ia"], fd["ja"], x, yn, ia, opts=opts)
        yt = np.zeros(n, np.float32)
        bof_dist.flash_csrgemv_row_sharded("T", m, n, fd["val"], fd["ia"], fd["ja"], xt, yt, ia, opts=opts)   # the one all-reduce
        np.save(os.path.join(out_dir, f"yn_{rank}.npy"), np.concatenate([[r0, r1], yn[r0:r1]]))
        np.save(os.path.join(out_dir, f"yt_{rank}.npy"), yt)
    finally:
        for f in fd.values():
            bofhip.lib().bof_file_forget(f)
            os.close(f)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_kmeans_csrmm_csrgemv_ranks_on_mock_devices(tmp_path, mock_lib):
    """The other sharded entry points of bof_dist with real ranks on the CPU box (same mock-linked library):
    flash::kmeans by points, csrmm by nnz-balanced row ranges, csrgemv 'N' (disjoint slices) and 'T' (full-length
    partials + the one all-reduce, over gloo here).  Integer data, exact against numpy."""
    world = 2
    ncent, npts, dim, m, n, k = 200, 128 * 5, 96, 1500, 900, 24
    rng = np.random.default_rng(7)
    centers = rng.integers(-3, 4, (ncent, dim)).astype(np.float32)
    points = rng.integers(-3, 4, (npts, dim)).astype(np.float32)
    cl = rng.integers(0, 9, ncent).astype(np.float32)
    pl = rng.integers(0, 9, npts).astype(np.float32)
    nnz_row = rng.integers(0, 13, m)
    ia = np.concatenate([[0], np.cumsum(nnz_row)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in nnz_row]).astype(np.int64)
    val = rng.integers(1, 10, int(ia[-1])).astype(np.float32)
    b = rng.integers(0, 7, (n, k)).astype(np.float32)
    x = rng.integers(0, 10, n).astype(np.float32)
    xt = rng.integers(0, 10, m).astype(np.float32)
    for name, arr in (("centers", centers), ("points", points), ("dist", np.zeros((npts, ncent), np.float32)), ("val", val),
                      ("ia", ia), ("ja", ja), ("b", b), ("c", np.zeros((m, k), np.float32))):
        arr.tofile(tmp_path / f"{name}.bin")
    for name, arr in (("cl", cl), ("pl", pl), ("x", x), ("xt", xt)):
        np.save(tmp_path / f"{name}.npy", arr)
    mp.spawn(_kmeans_csr_worker, args=(world, _free_port(), mock_lib, str(tmp_path), (ncent, npts, dim, m, n, k)), nprocs=world,
             join=True)
    # kmeans: dist[p, c] = -2 <c, p> + |c|^2 + |p|^2 (one k block: the two rank-1 terms once)
    want = -2.0 * (points.astype(np.float64) @ centers.astype(np.float64).T) + cl[None, :] + pl[:, None]
    assert np.array_equal(np.fromfile(tmp_path / "dist.bin", np.float32).reshape(npts, ncent), want.astype(np.float32))
    import scipy.sparse as sp
    A = sp.csr_matrix((val.astype(np.float64), ja, ia), shape=(m, n))
    assert np.array_equal(np.fromfile(tmp_path / "c.bin", np.float32).reshape(m, k), (2.0 * (A @ b.astype(np.float64))).astype(np.float32))
    parts = sorted((np.load(tmp_path / f"yn_{r}.npy") for r in range(world)), key=lambda p: p[0])
    assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == m
    assert np.array_equal(np.concatenate([p[2:] for p in parts]).astype(np.float32), (A @ x.astype(np.float64)).astype(np.float32))
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"yt_{r}.npy"), (A.T @ xt.astype(np.float64)).astype(np.float32))

