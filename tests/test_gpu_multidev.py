"""-m gpu: the IN-PROCESS multi-device level 3 (bof_options.devices / $BOF_DEVICES): C row panels
(gemm / kmeans) or nnz-balanced row blocks (csrmm / csrgemv) dealt to the devices of the list, every
operand all devices need read from its file once and fanned out (reference: N workers inside one
process behind flash::gemm, src/scheduler/scheduler.cpp:9-16; the mutex-guarded vector add of
csrgemv 'T', include/tasks/csrgemv_task.h:169-176, becomes a device-to-device segment sum).

The device list names device 0 several times, so a 1-GPU box runs every code path of the sharded
call: per-device pipelines, shared-panel fan-out, per-device write-back, the partial-sum reduce.
The bar: files / vectors bit-equal to the single-device call and to the oracle, and the byte
counters show every matrix read ONCE whatever the number of devices."""
import hashlib
import itertools
import os
import subprocess

import numpy as np
import pytest

import bofhip
import orc
from test_gpu_flash import Files, stored_shapes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "blas-on-flash_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.parametrize("ndev", [2, 3])
@pytest.mark.parametrize("path", [1, 2, 3])
@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
def test_flash_gemm_devices_all_layouts(dev, tmp_path, ord_, ta, tb, path, ndev):
    """640 x 600 x 500, tile 128 (5 x 4 x 3 tiles, merged tails in k and n), unaligned leading dims,
    beta != 0, all 8 layouts; tile cache (1), row panels (2), row panels with k-major copies forced
    (3).  The C panels are dealt 3+2 / 2+2+1 (row-major: 5 panels along m) or 2+1 / 1+1+1
    (column-major: 3 panels along n)."""
    m, k, n, blk = 640, 600, 500, 128
    rng = np.random.default_rng(5)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=3, n_io_threads=3, pinned_slots=4,
                                      gemm_path=min(path, 2), io_chunk_mib=1, panel_kmajor=3 if path == 3 else 1,
                                      devices=[0] * ndev)
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
        st = bofhip.flash_last_stats()
        per = bofhip.flash_last_device_stats()
        assert len(per) == ndev
        assert st["tasks"] == 5 * 4 * 3 and sum(p["tasks"] for p in per) == st["tasks"]
        assert st["bytes_written"] == 4 * c0.size and sum(p["bytes_written"] for p in per) == st["bytes_written"]
        if path >= 2:
            # the panel path reads A, B and C exactly once in total: the shared operand is NOT re-read per device
            assert st["bytes_read"] == 4 * (a.size + b.size + c0.size)
            shared = b.size if ord_ == "R" else a.size        # the operand without the C panel dimension
            shared += (a.size if ord_ == "R" and ta == "T" else 0) + (b.size if ord_ == "C" and tb == "T" else 0)
            assert st["bytes_h2d"] == st["bytes_read"] + 4 * shared * (ndev - 1)   # ... but copied to every device
        else:
            assert st["bytes_read"] >= 4 * (a.size + b.size + c0.size)
    finally:
        F.close()


@pytest.mark.parametrize("ndev", [2, 3])
@pytest.mark.parametrize("path", [0, 2])
@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
def test_flash_gemm_peer_bcast(dev, tmp_path, ord_, ta, tb, path, ndev):
    """SURVEY 8f-4, B-panel broadcast device to device (bof_options.peer_bcast; no analogue in the reference, which
    reads every shared key once per process: include/tasks/csrmm_task.h:129-235).  A panel every device needs
    crosses PCIe ONCE, to its home device (panel p -> device p % ndev), and reaches the others from the home's
    HBM with hipMemcpyPeerAsync behind the home copy's event.  640 x 600 x 500, 128-tiles, beta != 0, all 8
    layouts, auto path and forced row panels: C bit-equal to the oracle; every shared byte is counted once in
    bytes_h2d and (ndev - 1) times in bytes_p2p, on the devices that are NOT its home."""
    m, k, n, blk = 640, 600, 500, 128
    rng = np.random.default_rng(41)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=3, n_io_threads=3, pinned_slots=4, gemm_path=path,
                                      io_chunk_mib=1, devices=[0] * ndev, peer_bcast=1, verify=1)
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
        st = bofhip.flash_last_stats()
        per = bofhip.flash_last_device_stats()
        assert len(per) == ndev and st["verify_checks"] > 0
        shared = b.size if ord_ == "R" else a.size        # the operand without the C panel dimension
        shared += (a.size if ord_ == "R" and ta == "T" else 0) + (b.size if ord_ == "C" and tb == "T" else 0)
        assert st["bytes_read"] == 4 * (a.size + b.size + c0.size)
        assert st["bytes_h2d"] == st["bytes_read"]                      # the host feeds every byte once ...
        assert st["bytes_p2p"] == 4 * shared * (ndev - 1)               # ... the peers get theirs device to device
        assert sum(p["bytes_p2p"] for p in per) == st["bytes_p2p"] and all(p["bytes_p2p"] > 0 for p in per)
        # the same call without the broadcast: same bits, no device-to-device byte
        c0.tofile(F.paths["c"])
        os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
        opts.peer_bcast = 2
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
        assert bofhip.flash_last_stats()["bytes_p2p"] == 0
    finally:
        F.close()


def test_flash_gemm_peer_bcast_at_scale(dev, tmp_path):
    """6144 x 5120 x 4096 with 1024-tiles from O_DIRECT files, three devices, several 4 MiB chunks per shared panel
    (each chunk: home copy, event, two peer copies): C equals the single-device call bit for bit."""
    m, n, k, blk = 6144, 5120, 4096, 1024
    rng = np.random.default_rng(43)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        res = []
        for devices, bc in ((None, 0), ([0, 0, 0], 1)):
            kw = dict(gemm_blk=blk, n_io_threads=6, pinned_slots=6, gemm_path=2, io_chunk_mib=4, peer_bcast=bc)
            if devices:
                kw["devices"] = devices
            bofhip.flash_gemm("R", "N", "N", m, n, k, 0.5, 1.5, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                              bofhip.default_options(**kw))
            st = bofhip.flash_last_stats()
            assert st["bytes_read"] == 4 * (a.size + b.size + c0.size) and st["bytes_written"] == 4 * c0.size
            if devices:
                assert st["bytes_p2p"] == 4 * b.size * 2 and st["bytes_h2d"] == st["bytes_read"]
            res.append(F.read("c", np.float32, (m, n)).copy())
            c0.tofile(F.paths["c"])
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
        assert np.array_equal(res[0], res[1])
    finally:
        F.close()


@pytest.mark.parametrize("ord_,ta,tb,beta", [("R", "N", "N", 0.0), ("R", "T", "N", 1.5), ("C", "N", "T", 0.0),
                                             ("C", "T", "T", 1.5)])
def test_flash_gemm_devices_ring_reuse(dev, tmp_path, ord_, ta, tb, beta):
    """Two devices, each with the smallest budget its slab's plan accepts: streamed-operand and C
    ring slots are reused on both devices while the shared operand's panels arrive once for both."""
    m, k, n, blk = 2200, 900, 1024, 128
    if ord_ == "C":
        m, n = n, m
    rng = np.random.default_rng(8)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.75, beta, a, b, c0.copy(), 0, 0, 0, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        # budget of one slab (half of the C panel dimension) with group 2: resident operands one slot per
        # panel, rings of 4 and 5 slots; a little slack for the shared operand's full width
        half_m, half_n = (m // 2 // blk * blk + blk, n) if ord_ == "R" else (m, n // 2 // blk * blk + blk)
        plan = bofhip.flash_gemm_panel_plan(ord_, ta, tb, half_m, half_n, k, blk, 1 << 40, group=2)
        budget = int(plan["need_bytes"] * 1.6)
        opts = bofhip.default_options(gemm_blk=blk, n_streams=2, n_io_threads=4, pinned_slots=4, gemm_path=2,
                                      io_chunk_mib=1, hbm_budget=budget, panel_group=2, panel_kmajor=1,
                                      devices=[0, 0])
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.75, beta, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
        st = bofhip.flash_last_stats()
        assert st["bytes_read"] == 4 * (a.size + b.size + (c0.size if beta else 0))
    finally:
        F.close()


@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
def test_flash_gemm_devices_equals_single_at_scale(dev, tmp_path, ord_, ta, tb):
    """6144 x 5120 x 4096 with 1024-tiles (6 / 5 C panels of 16-24 MiB, several 4 MiB chunks each, k chains of
    4), beta != 0, uniform-random data, O_DIRECT: the C file of the three-device call equals the C file of the
    single-device call bit for bit, in all 8 layouts (including those whose A / B is paneled along k and so
    shared with a column offset per device), and both agree with float64 on sampled rows."""
    m, n, k, blk = 6144, 5120, 4096, 1024
    rng = np.random.default_rng(12)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        res = []
        for devices in (None, [0, 0, 0]):
            kw = dict(gemm_blk=blk, n_io_threads=6, pinned_slots=6, gemm_path=2, io_chunk_mib=4)
            if devices:
                kw["devices"] = devices
            bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 1.5, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                              bofhip.default_options(**kw))
            st = bofhip.flash_last_stats()
            assert st["bytes_read"] == 4 * (a.size + b.size + c0.size) and st["bytes_written"] == 4 * c0.size
            res.append(F.read("c", np.float32, sc).copy())
            c0.tofile(F.paths["c"])
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
        assert np.array_equal(res[0], res[1])
        A = a.astype(np.float64) if (ta == "T") == (ord_ == "C") else a.T.astype(np.float64)     # logical m x k
        B = b.astype(np.float64) if (tb == "T") == (ord_ == "C") else b.T.astype(np.float64)     # logical k x n
        rows = [0, 1023, 1024, 3071, m - 1]
        want = 0.5 * (A[rows] @ B) + 1.5 * (c0 if ord_ == "R" else c0.T)[rows].astype(np.float64)
        got = (res[1] if ord_ == "R" else res[1].T)[rows]
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-4
    finally:
        F.close()


@pytest.mark.parametrize("ord_,ta,tb", [("C", "T", "N"), ("R", "N", "T")])
@pytest.mark.parametrize("path", [1, 2])
def test_flash_kmeans_devices(dev, tmp_path, ord_, ta, tb, path):
    """flash::kmeans over two devices: every device gets its own copy of the norm vectors and adds the
    slices of ITS tiles (row / column offsets of the slab)."""
    m, n, k, blk = 640, 500, 300, 128
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
                                      io_chunk_mib=1, devices=[0, 0])
        bofhip.flash_kmeans(ord_, ta, tb, m, n, k, -2.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                            cl.ctypes.data, pl.ctypes.data, ones.ctypes.data, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
    finally:
        F.close()


@pytest.mark.parametrize("ndev", [2, 3])
@pytest.mark.parametrize("ord_b,k,alpha,beta", [("R", 128, 1.0, 0.0), ("R", 1030, 0.5, 2.0), ("C", 128, 1.0, 0.0),
                                               ("C", 1030, 0.5, 2.0)])
def test_flash_csrmm_devices(dev, tmp_path, ord_b, k, alpha, beta, ndev):
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    b = orc.dense_fill(n, k, "s")
    rng = np.random.default_rng(k)
    c0 = rng.integers(0, 5, (m, k)).astype(np.float32) if beta else np.zeros((m, k), np.float32)
    if ord_b == "C":
        b, c0 = np.ascontiguousarray(b.T), np.ascontiguousarray(c0.T)
    ref = orc.flash_csrmm(ord_b, m, n, k, alpha, beta, val, ia, ja, b, c0.copy(), 1000, 5000, 1024)
    F = Files(tmp_path, val=val, ja=ja, ia=ia, b=b, c=c0)
    try:
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000, n_io_threads=2, devices=[0] * ndev)
        bofhip.flash_csrmm("N", m, n, k, alpha, beta, F.fptr("val"), F.fptr("ia"), F.fptr("ja"),
                           ord_b, F.fptr("b"), F.fptr("c"), opts)
        assert np.array_equal(F.read("c", np.float32, c0.shape), ref)
        st = bofhip.flash_last_stats()
        per = bofhip.flash_last_device_stats()
        assert len(per) == ndev and all(p["tasks"] > 0 for p in per)
        assert st["bytes_written"] == 4 * c0.size
        # B is read once; its bytes cross PCIe once per device
        assert st["bytes_h2d"] >= 4 * b.size * ndev
        csr = 12 * ja.size + 8 * ia.size
        slack = 2 * 1024 * sum(p["tasks"] for p in per)     # sector widening of the segments
        assert 4 * b.size + csr + (4 * c0.size if beta else 0) <= st["bytes_read"] <= \
            4 * b.size + csr + (4 * c0.size if beta else 0) + slack
    finally:
        F.close()


@pytest.mark.parametrize("ord_b", ["R", "C"])
@pytest.mark.parametrize("direct", [False, True])
def test_flash_csrmm_devices_large_b(dev, tmp_path, ord_b, direct):
    """A B operand of 400 MB and few non-zeros: the row blocks are in HBM long before B is.  Every device's
    pipeline has to wait for the ONE feed of B (read once, fanned out) on the host first -- the event the
    feed records behind its last copy does not exist for the device until it has been recorded (a pipeline
    that only queued a wait for it ran ahead of B at cfg3 size: caught by the full-size run, then here)."""
    m, n, k = 8192, 400_000, 256
    val, ja, ia = orc.sparse_create(m, n, 2.5e-5)
    rng = np.random.default_rng(3)
    b = rng.integers(0, 9, (n, k)).astype(np.float32)
    c0 = np.zeros((m, k), np.float32)
    if ord_b == "C":
        b, c0 = np.ascontiguousarray(b.T), np.ascontiguousarray(c0.T)
    ref = orc.flash_csrmm(ord_b, m, n, k, 1.0, 0.0, val, ia, ja, b, c0.copy(), 1000, 5000, 1024)
    F = Files(tmp_path, direct=direct, val=val, ja=ja, ia=ia, b=b, c=c0)
    try:
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000, n_io_threads=4, use_odirect=int(direct),
                                      devices=[0, 0, 0])
        for _ in range(3):
            bofhip.flash_csrmm("N", m, n, k, 1.0, 0.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), ord_b, F.fptr("b"),
                               F.fptr("c"), opts)
            assert np.array_equal(F.read("c", np.float32, c0.shape), ref)
            c0.tofile(F.paths["c"])
        x = (np.arange(n) % 10).astype(np.float32)
        y = np.full(m, -1.0, np.float32)
        bofhip.flash_csrgemv("N", m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), x.ctypes.data, y.ctypes.data, opts)
        assert np.array_equal(y, orc.flash_csrgemv("N", m, n, val, ia, ja, x, np.zeros(m, np.float32), 1000, 5000))
    finally:
        F.close()


@pytest.mark.parametrize("ord_b", ["R", "C"])
def test_flash_csrmm_inmem_devices(dev, tmp_path, ord_b):
    m, n, k = 4096, 2048, 136
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    rng = np.random.default_rng(1)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    if ord_b == "C":
        b, c0 = np.ascontiguousarray(b.T), np.ascontiguousarray(c0.T)
    ref = orc.flash_csrmm(ord_b, m, n, k, 0.5, 2.0, val, ia, ja, b, c0.copy(), 1000, 5000, 1024)
    F = Files(tmp_path, val=val, ja=ja, ia=ia)
    try:
        c = c0.copy()
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000, devices=[0, 0, 0])
        bofhip.flash_csrmm_inmem("N", m, n, k, 0.5, 2.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"),
                                 ord_b, b.ctypes.data, c.ctypes.data, opts)
        assert np.array_equal(c, ref)
    finally:
        F.close()


@pytest.mark.parametrize("ndev", [2, 3])
@pytest.mark.parametrize("trans", ["N", "T"])
def test_flash_csrgemv_devices(dev, tmp_path, golden, trans, ndev):
    """'N': disjoint y slices; 'T': full-length partials per device, summed segment by segment on the
    devices (sum_partials_kernel) -- the generator's integer data makes the result exact, so the
    reference's hash must come out."""
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    x = (np.arange(n if trans == "N" else m) % 10).astype(np.float32)
    y = np.full(m if trans == "N" else n, 3.0, np.float32)
    F = Files(tmp_path, val=val, ja=ja, ia=ia)
    try:
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000, devices=[0] * ndev)
        bofhip.flash_csrgemv(trans, m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), x.ctypes.data,
                             y.ctypes.data, opts)
        want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
        assert hashlib.sha256(y.tobytes()).hexdigest() == want["gen_csrgemv_" + trans]
        assert len(bofhip.flash_last_device_stats()) == ndev
    finally:
        F.close()


def test_flash_csr_devices_ragged_random(dev, tmp_path):
    """Random values, ragged rows with a run of empty rows, more devices than some shards have blocks for."""
    rng = np.random.default_rng(9)
    m, n, k = 3000, 777, 96
    counts = rng.integers(0, 12, m)
    counts[100:400] = 0
    ia = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts]).astype(np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    ref = orc.flash_csrmm("R", m, n, k, 0.5, 2.0, val, ia, ja, b, c0.copy(), 200, 300, 64)
    F = Files(tmp_path, direct=False, val=val, ja=ja, ia=ia, b=b, c=c0)
    try:
        opts = bofhip.default_options(max_nnzs=300, csrmm_rblk=200, csrmm_cblk=64, use_odirect=0, devices=[0, 0, 0, 0])
        bofhip.flash_csrmm("N", m, n, k, 0.5, 2.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), "R",
                           F.fptr("b"), F.fptr("c"), opts)
        assert np.array_equal(F.read("c", np.float32, (m, k)), ref)
        x = rng.uniform(-1, 1, n).astype(np.float32)
        y = np.zeros(m, np.float32)
        bofhip.flash_csrgemv("N", m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), x.ctypes.data, y.ctypes.data, opts)
        assert np.array_equal(y, orc.flash_csrgemv("N", m, n, val, ia, ja, x, np.zeros(m, np.float32), 200, 300))
        xt = rng.uniform(-1, 1, m).astype(np.float32)
        yt = np.zeros(n, np.float32)
        bofhip.flash_csrgemv("T", m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), xt.ctypes.data, yt.ctypes.data, opts)
        rt = orc.flash_csrgemv("T", m, n, val, ia, ja, xt, np.zeros(n, np.float32), 200, 300)
        assert np.abs(yt - rt).max() / np.abs(rt).max() < 1e-4   # atomics + segment sums: order-dependent rounding
    finally:
        F.close()


def test_devices_bad_list(dev, tmp_path):
    a = np.zeros((256, 256), np.float32)
    F = Files(tmp_path, a=a, b=a, c=a)
    try:
        with pytest.raises(bofhip.BofError, match="not one of"):
            bofhip.flash_gemm("R", "N", "N", 256, 256, 256, 1.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                              bofhip.default_options(gemm_blk=128, devices=[0, 63]))
    finally:
        F.close()


def test_devices_budget_error_leaves_c_untouched(dev, tmp_path):
    """A budget one device's slab fits and another's does not (the slab with the merged tail has tiles
    of 243 rows): BOF_ENOMEM must come BEFORE any slab writes C (found by tests/test_gpu_fuzz.py: slab 0
    had been written when slab 1 reported the budget)."""
    m, n, k, blk, ldc = 371, 353, 112, 128, 1024
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (k, 384)).astype(np.float32)          # 'T': A stored k x m, lda 384
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, ldc)).astype(np.float32)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, gemm_path=1, hbm_budget=983040, devices=[0, 0], use_odirect=0)
        with pytest.raises(bofhip.BofError, match="budget"):
            bofhip.flash_gemm("R", "T", "N", m, n, k, 2.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 384, n, ldc, opts)
        assert np.array_equal(F.read("c", np.float32, c0.shape), c0)
        opts.hbm_budget = 0
        bofhip.flash_gemm("R", "T", "N", m, n, k, 2.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 384, n, ldc, opts)
        ref = orc.flash_gemm("R", "T", "N", m, n, k, 2.0, 0.0, a, b, c0.copy(), 384, n, ldc, blk)
        assert np.array_equal(F.read("c", np.float32, c0.shape), ref)
    finally:
        F.close()


def _run(binary, args, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_reference_drivers_shard_over_bof_devices(dev, tmp_path, golden):
    """The UNCHANGED reference driver sources (oracle/_ref/ref_*_driver: drivers/gemm.cpp, csrmm.cpp,
    csrgemv.cpp compiled against our headers) with BOF_DEVICES=0,0: flash_setup hands the list to every
    kernel call, the outputs equal the single-device ones."""
    if not os.path.exists(os.path.join(REF, "ref_gemm_driver")):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    m, k, n, blk = 640, 600, 500, 128
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    ref = orc.flash_gemm("R", "N", "N", m, n, k, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, blk)
    pa, pb, pc = (str(tmp_path / x) for x in ("A", "B", "C"))
    a.tofile(pa); b.tofile(pb); c0.tofile(pc)
    env = {"BOF_DEVICES": "0,0", "BOF_GEMM_BLK_SIZE": str(blk), "BOF_TRACE": "1"}
    out = _run(os.path.join(REF, "ref_gemm_driver"), [pa, pb, pc, m, k, n, 0.5, 2.0, "N", "N", "R", k, n, n], env)
    assert "gemm() took" in out
    assert np.array_equal(np.fromfile(pc, np.float32).reshape(m, n), ref)

    mm, nn, kk = 4096, 2048, 128
    val, ja, ia = orc.sparse_create(mm, nn, 0.01)
    bm = orc.dense_fill(nn, kk, "s")
    cref = orc.flash_csrmm("R", mm, nn, kk, 1.0, 0.0, val, ia, ja, bm, np.zeros((mm, kk), np.float32), 1000, 5000, 1024)
    paths = {x: str(tmp_path / x) for x in ("val", "ja", "ia", "bm", "cm")}
    val.tofile(paths["val"]); ja.tofile(paths["ja"]); ia.tofile(paths["ia"]); bm.tofile(paths["bm"])
    np.zeros((mm, kk), np.float32).tofile(paths["cm"])
    env = {"BOF_DEVICES": "0,0,0", "BOF_MAX_NNZS": "5000", "BOF_CSRMM_RBLK_SIZE": "1000"}
    _run(os.path.join(REF, "ref_csrmm_driver"), [paths["val"], paths["ja"], paths["ia"], paths["bm"], paths["cm"],
                                                 mm, nn, kk, 1.0, 0.0, "N", "R"], env)
    assert np.array_equal(np.fromfile(paths["cm"], np.float32).reshape(mm, kk), cref)
    want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
    for trans in "NT":
        x = (np.arange(nn if trans == "N" else mm) % 10).astype(np.float32)
        px, py = str(tmp_path / f"x{trans}"), str(tmp_path / f"y{trans}")
        x.tofile(px)
        np.zeros(mm if trans == "N" else nn, np.float32).tofile(py)
        _run(os.path.join(REF, "ref_csrgemv_driver"), [paths["val"], paths["ja"], paths["ia"], px, py, mm, nn, trans], env)
        assert hashlib.sha256(np.fromfile(py, np.float32).tobytes()).hexdigest() == want["gen_csrgemv_" + trans]
