"""-m gpu parity of the FILE-resident path (level 3: reader -> pinned ring -> HBM
tile cache -> kernels -> write-back) through the C ABI and through the C++
drivers, against the oracle's restated flash::gemm/csrmm/csrgemv (bit-exact)."""
import fcntl
import itertools
import os
import subprocess

import numpy as np
import pytest

import bofhip
import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "blas-on-flash_amd", "bin")


def stored_shapes(ord_, ta, tb, m, n, k):
    a = (m, k) if (ta == "T") == (ord_ == "C") else (k, m)
    b = (k, n) if (tb == "T") == (ord_ == "C") else (n, k)
    c = (m, n) if ord_ == "R" else (n, m)
    return a, b, c


def open_rw(path, direct=True):
    if direct:
        try:
            return os.open(path, os.O_RDWR | os.O_DIRECT)
        except OSError:
            pass
    return os.open(path, os.O_RDWR)


class Files:
    def __init__(self, tmp_path, direct=True, **arrays):
        self.paths, self.fds = {}, {}
        for name, arr in arrays.items():
            p = str(tmp_path / f"{name}.bin")
            np.ascontiguousarray(arr).tofile(p)
            self.paths[name] = p
            self.fds[name] = open_rw(p, direct)

    def fptr(self, name, byte_off=0):
        return bofhip.FPtr(self.fds[name], byte_off)

    def read(self, name, dtype, shape):
        return np.fromfile(self.paths[name], dtype).reshape(shape)

    def close(self):
        for fd in self.fds.values():
            os.close(fd)


@pytest.mark.parametrize("path", [1, 2, 3])
@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
def test_flash_gemm_layouts_unaligned(dev, tmp_path, monkeypatch, ord_, ta, tb, path):
    """SURVEY 8c trust-matrix shape: 640x600x500 with unaligned leading dims, tile
    256 (separate tail, merged tail), alpha=0.5, beta=2, random C, all 8 layouts; through the
    tile cache (path 1), through the row-panel pipeline (path 2) and through the row-panel pipeline
    with the k-major copies of k-contiguous operand panels forced on (3: BOF_PANEL_KMAJOR=2)."""
    monkeypatch.setenv("BOF_PANEL_KMAJOR", "2" if path == 3 else "0")
    path = min(path, 2)
    m, k, n, blk = 640, 600, 500, 256
    rng = np.random.default_rng(11)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=3, n_io_threads=3, pinned_slots=4,
                                      gemm_path=path, io_chunk_mib=1)
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, F.fptr("a"), F.fptr("b"), F.fptr("c"),
                          0, 0, 0, opts)
        got = F.read("c", np.float32, sc)
        assert np.array_equal(got, ref)
        st = bofhip.flash_last_stats()
        assert st["tasks"] == 12
        assert st["bytes_read"] == 4 * (a.size + b.size + c0.size)   # every tile read once
        assert st["bytes_written"] == 4 * c0.size                    # C written once
    finally:
        F.close()


@pytest.mark.parametrize("path", [1, 2, 3])
@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
def test_flash_gemm_reference_chain(dev, tmp_path, ord_, ta, tb, path):
    """bof_options.gemm_chain = 1: the reference's task arithmetic, one rounding per k-block -- C = alpha*A_l*B_l + C
    for l > 0 (src/blas/gemm.cpp:122-127, include/tasks/gemm_task.h:87-90) -- bit for bit what the tile-by-tile
    restatement of flash::gemm computes (oracle/bof_oracle.c: orc_flash_gemm).  640 x 600 x 500 with 128-tiles
    (5 x 4 x 3 tiles: chains of 4 with a merged k tail), beta != 0, all 8 layouts, tile cache / row panels / row panels with
    k-major copies."""
    m, k, n, blk = 640, 600, 500, 128
    rng = np.random.default_rng(23)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, blk, chain=1)
    whole = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, blk)
    assert not np.array_equal(ref, whole)           # the two arithmetics do differ on random data (by rounding) ...
    assert np.abs(ref - whole).max() / np.abs(whole).max() < 1e-5     # ... and only by rounding
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=3, n_io_threads=3, pinned_slots=4, gemm_path=min(path, 2),
                                      io_chunk_mib=1, panel_kmajor=3 if path == 3 else 1, gemm_chain=1, verify=1)
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
        st = bofhip.flash_last_stats()
        assert st["tasks"] == 5 * 4 * 3 and st["bytes_read"] == 4 * (a.size + b.size + c0.size)
    finally:
        F.close()


@pytest.mark.parametrize("kw", [dict(gemm_path=1, devices=[0, 0]), dict(gemm_path=1, devices=[0, 0, 0], hbm_budget=3 * 40 * 128 * 128 * 4),
                                dict(gemm_path=2, devices=[0, 0, 0]), dict(gemm_path=2, devices=[0, 0], peer_bcast=1),
                                dict(gemm_path=2, panel_group=1, panel_streams=1), dict(gemm_path=2, panel_group=3, panel_streams=2),
                                dict(gemm_path=1, hbm_budget=40 * 128 * 128 * 4), dict(gemm_path=0)],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()).replace(" ", ""))
@pytest.mark.parametrize("ord_,ta,tb,beta", [("R", "N", "N", 2.0), ("R", "T", "N", 0.0), ("C", "N", "T", 2.0), ("C", "T", "T", 0.0)])
def test_flash_gemm_reference_chain_every_path(dev, tmp_path, ord_, ta, tb, beta, kw):
    """bof_options.gemm_chain = 1 (the reference's flash::gemm bits: one rounding per k-block, src/blas/gemm.cpp:122-127)
    on EVERY way the library can run the call: tile cache on two / three devices and under an eviction
    budget, row panels on three devices, with the device-to-device broadcast, with ramp groups of 1 and 3 and one / two
    compute streams, and the chooser -- each bit-equal to the tile-by-tile oracle, and NOT equal to the single chain."""
    m, k, n, blk = 640, 600, 500, 128
    rng = np.random.default_rng(31)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, beta, a, b, c0.copy(), 0, 0, 0, blk, chain=1)
    whole = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, beta, a, b, c0.copy(), 0, 0, 0, blk)
    assert not np.array_equal(ref, whole)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=3, n_io_threads=3, pinned_slots=4, io_chunk_mib=1,
                                      gemm_chain=1, verify=1, **kw)
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, beta, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
        assert bofhip.flash_last_stats()["tasks"] == 5 * 4 * 3
    finally:
        F.close()


@pytest.mark.parametrize("ord_,ta,tb,beta", [("R", "N", "N", 0.0), ("R", "N", "N", 2.0), ("R", "T", "T", 2.0),
                                             ("C", "N", "T", 0.0), ("C", "T", "N", 2.0)])
def test_flash_gemm_result_does_not_depend_on_the_cut(dev, tmp_path, ord_, ta, tb, beta):
    """The default arithmetic (bof_options.gemm_chain = 0): ONE k-ordered chain per output element over the whole K,
    whatever the tiler, the path, the budget or the device list cut the product into -- k-ranges that run as
    separate launches hand their raw accumulators on (gemm_f32_mfma.hip, ChainEpi).  So every configuration writes
    the SAME C file, the one drivers/in_mem_gemm.cpp:63-70 computes with its single cblas_sgemm call: equal to the
    oracle's in-memory gemm AND to one bof_sgemm over the whole resident matrices, bit for bit."""
    import torch
    m, k, n = 1100, 1500, 900
    rng = np.random.default_rng(29)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.in_mem_gemm(ord_, ta, tb, m, n, k, 0.5, beta, a, b, c0.copy())
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c0))
    bofhip.sgemm(ord_, ta, tb, m, n, k, 0.5, da.data_ptr(), sa[1], db.data_ptr(), sb[1], beta, dc.data_ptr(), sc[1],
                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        tile = 256 * 256 * 4
        for kw in (dict(gemm_blk=128, gemm_path=1), dict(gemm_blk=256, gemm_path=1, hbm_budget=14 * tile),
                   dict(gemm_blk=512, gemm_path=1), dict(gemm_blk=128, gemm_path=2, panel_group=1),
                   dict(gemm_blk=256, gemm_path=2, panel_group=3, panel_kmajor=3), dict(gemm_blk=4096, gemm_path=2),
                   dict(gemm_blk=256, gemm_path=2, devices=[0, 0, 0]), dict(gemm_blk=128, gemm_path=1, devices=[0, 0])):
            bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, beta, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                              bofhip.default_options(n_io_threads=3, pinned_slots=4, io_chunk_mib=1, verify=1, **kw))
            assert np.array_equal(F.read("c", np.float32, sc), ref), kw
            if "hbm_budget" not in kw and not ("devices" in kw and kw["gemm_path"] == 1):
                # (a budget of 14 tiles re-reads operands it had to evict; tile-cache slabs of several devices each read B)
                assert bofhip.flash_last_stats()["bytes_read"] == 4 * (a.size + b.size + (c0.size if beta else 0)), kw
            c0.tofile(F.paths["c"])
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
        # level 2: the tile DAG over resident matrices is the same single chain
        dc2 = torch.from_numpy(c0).cuda()
        bofhip.gemm_resident(ord_, ta, tb, m, n, k, 0.5, beta, da.data_ptr(), db.data_ptr(), dc2.data_ptr(), 0, 0, 0,
                             bofhip.default_options(gemm_blk=256), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(dc2.cpu().numpy(), ref)
    finally:
        F.close()


@pytest.mark.parametrize("ord_,ta,tb", list(itertools.product("RC", "NT", "NT")))
def test_flash_gemm_mixed_alignment_stress(dev, tmp_path, ord_, ta, tb):
    """Aligned leading dimension (640 floats = 5 sectors) + a tail-merged tile whose width is not a
    multiple of 128 (600 = 256 + 344): some tile regions of every file are sector aligned and some
    are not, and neighbouring C tiles share 4 KiB pages.  The reference serialises such writes
    (src/scheduler/io_executor.cpp:28-156); here every request of the call on such a file goes
    through ONE descriptor mode (the buffered twin), never O_DIRECT and buffered side by side.
    beta != 0, 4 writer threads, repeated: any lost update shows up as a mismatch."""
    m = n = k = 600
    ld, blk = 640, 256
    rng = np.random.default_rng(31)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)

    def padded(shape):
        x = rng.uniform(-1, 1, (shape[0], ld)).astype(np.float32)
        return x
    a, b, c0 = padded(sa), padded(sb), padded(sc)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, a, b, c0.copy(), ld, ld, ld, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=4, n_io_threads=8, pinned_slots=6)
        for rep in range(12):
            bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, 2.0, F.fptr("a"), F.fptr("b"), F.fptr("c"),
                              ld, ld, ld, opts)
            got = F.read("c", np.float32, c0.shape)
            assert np.array_equal(got, ref), rep        # incl. the padding columns: untouched
            c0.tofile(F.paths["c"])                     # restore C for the next round
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
    finally:
        F.close()


@pytest.mark.parametrize("direct_mode", ["widened", "twin"])
@pytest.mark.parametrize("ord_,ta,tb,beta,hdr", [("R", "N", "N", 0.0, 0), ("C", "T", "N", 1.5, 0),
                                                 ("R", "N", "T", 2.0, 52), ("C", "N", "N", 0.0, 1000)])
def test_flash_gemm_unaligned_keeps_odirect(dev, tmp_path, monkeypatch, ord_, ta, tb, beta, hdr, direct_mode):
    """The paper's unaligned case (Fig. 5 right: 31000-edge matrices) in small: edge 1550 with 256-tiles --
    rows of 6200 bytes, panels of 256 rows are sector aligned, the merged last panel (270 rows) ends at the
    file's unaligned end -- and the same behind an unaligned file offset (hdr bytes of header: NOTHING is
    aligned then).  O_DIRECT is kept (reference: flash_file_handle.cpp:462-506, 558-716): reads fetch the
    sector-aligned superset, writes send whole pages direct and the partial edge pages through the page
    cache; `twin` = the whole file through the buffered twin (BOF_UNALIGNED_DIRECT=0).  Bit-equal to the
    oracle, header and every byte outside C's extents untouched, I/O counted once."""
    monkeypatch.setenv("BOF_UNALIGNED_DIRECT", "1" if direct_mode == "widened" else "0")
    m = n = k = 1550
    blk = 256
    rng = np.random.default_rng(hdr + 1)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.5, beta, a, b, c0.copy(), 0, 0, 0, blk)
    head = rng.integers(0, 255, hdr, dtype=np.uint8).tobytes()
    tail = rng.integers(0, 255, 777, dtype=np.uint8).tobytes()
    blobs = {nm: np.frombuffer(head + x.tobytes() + tail, np.uint8) for nm, x in (("a", a), ("b", b), ("c", c0))}
    F = Files(tmp_path, **blobs)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_io_threads=4, pinned_slots=4, gemm_path=2, io_chunk_mib=1)
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, beta, F.fptr("a", hdr), F.fptr("b", hdr), F.fptr("c", hdr),
                          0, 0, 0, opts)
        raw = F.read("c", np.uint8, (-1,)).tobytes()
        assert raw[:hdr] == head and raw[hdr + c0.nbytes:] == tail
        got = np.frombuffer(raw[hdr:hdr + c0.nbytes], np.float32).reshape(sc)
        assert np.array_equal(got, ref)
        st = bofhip.flash_last_stats()
        assert st["bytes_read"] == 4 * (a.size + b.size + (c0.size if beta else 0)) and st["bytes_written"] == 4 * c0.size
        # two devices: the boundary between their slabs is a panel boundary like any other
        np.frombuffer(head + c0.tobytes() + tail, np.uint8).tofile(F.paths["c"])
        os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
        opts2 = bofhip.default_options(gemm_blk=blk, n_io_threads=4, pinned_slots=4, gemm_path=2, io_chunk_mib=1,
                                       devices=[0, 0])
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.5, beta, F.fptr("a", hdr), F.fptr("b", hdr), F.fptr("c", hdr),
                          0, 0, 0, opts2)
        raw = F.read("c", np.uint8, (-1,)).tobytes()
        assert raw[:hdr] == head and raw[hdr + c0.nbytes:] == tail
        assert np.array_equal(np.frombuffer(raw[hdr:hdr + c0.nbytes], np.float32).reshape(sc), ref)
    finally:
        F.close()


@pytest.mark.parametrize("group", [1, 2, 3])
@pytest.mark.parametrize("ord_,ta,tb,beta", [("R", "N", "N", 0.0), ("R", "N", "N", 1.5), ("R", "T", "N", 1.5),
                                             ("C", "N", "T", 0.0), ("C", "T", "T", 1.5), ("R", "N", "T", 0.0)])
def test_flash_gemm_panels_ring_reuse(dev, tmp_path, monkeypatch, ord_, ta, tb, beta, group):
    """Panel pipeline with an HBM budget that holds the resident operand(s) plus the minimum
    rings (2*group panels of the streamed operand, 2*group+1 of C): every ring slot is reused several
    times, so the write-after-read (operand panels) and write-back-before-refill (C panels, read
    again when beta != 0) orderings are all exercised.  Tail-merged last panels in m and k.
    `group` = C panels of the first (ramp) group; the later panels go one at a time."""
    monkeypatch.setenv("BOF_PANEL_GROUP", str(group))
    m, k, n, blk = 1100, 900, 1024, 128          # m: 8 panels + merged tail (76), k: 7 + separate... 900 = 7*128+4 -> merged
    rng = np.random.default_rng(7)
    sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
    a = rng.uniform(-1, 1, sa).astype(np.float32)
    b = rng.uniform(-1, 1, sb).astype(np.float32)
    c0 = rng.uniform(-1, 1, sc).astype(np.float32)
    ref = orc.flash_gemm(ord_, ta, tb, m, n, k, 0.75, beta, a, b, c0.copy(), 0, 0, 0, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        # the smallest budget the plan accepts: resident operands hold one slot per panel, the streamed
        # one 2*group slots, C 2*group + 1 (every slot is an allocation of its own, rounded to 2 MiB)
        big = bofhip.flash_gemm_panel_plan(ord_, ta, tb, m, n, k, blk, 1 << 40, group=group)
        assert big["eligible"]
        npc = big["n_panels"][2]
        slots = [big["n_slots"][0], big["n_slots"][1], min(npc, 2 * group + 1)]
        budget = sum(slots[x] * big["slot_bytes"][x] for x in range(3))
        assert not bofhip.flash_gemm_panel_plan(ord_, ta, tb, m, n, k, blk, budget - 1, group=group)["eligible"]
        if beta:      # ... plus the raw accumulator panels of the ramp group's chains (C still holds the caller's values)
            assert big["acc_bytes"] == group * big["slot_bytes"][2]
            budget += big["acc_bytes"]
        opts = bofhip.default_options(gemm_blk=blk, n_streams=3, n_io_threads=4, pinned_slots=3, gemm_path=2,
                                      io_chunk_mib=1, hbm_budget=budget)
        bofhip.flash_gemm(ord_, ta, tb, m, n, k, 0.75, beta, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, sc), ref)
        st = bofhip.flash_last_stats()
        nt = bofhip.gemm_plan(ord_, ta, tb, m, n, k, beta, 0, 0, 0, blk)[0]
        assert st["tasks"] == len(nt)
        assert st["bytes_read"] == 4 * (a.size + b.size + (c0.size if beta else 0))   # everything once
        assert st["bytes_written"] == 4 * c0.size
        # big sequential requests: a handful per panel instead of one per tile row
        assert st["read_ops"] + st["write_ops"] < 3 * (st["bytes_read"] + st["bytes_written"]) / (1 << 20) + 64
    finally:
        F.close()


def test_flash_gemm_io_uring_engine_subprocess(dev):
    """The io_uring engine (BOF_IO_ENGINE is read once per process): the panel-ring and the 8-layout
    file tests again in a child process whose aligned O_DIRECT requests go through io_uring with the
    pinned staging slots registered as fixed buffers."""
    import sys
    env = dict(os.environ, BOF_IO_ENGINE="uring")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k",
                        "panels_ring_reuse or layouts_unaligned or small_budget"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:]


def test_flash_gemm_host_confirmed_handovers_subprocess(dev):
    """BOF_HOST_HANDOVER=1 (read once per process): the host-confirmed hand-overs of round 5 (the consuming thread
    waits on the host for the producer's event before it submits dependent work; off by default again since round 6,
    whose stand-alone stresses showed the events it was built against to be workgroups running under a wrong ID, not
    hand-overs) stay a supported A/B configuration: the panel-ring, the 8-layout and the reference-chain file tests
    again in a child process with it."""
    import sys
    env = dict(os.environ, BOF_HOST_HANDOVER="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k",
                        "panels_ring_reuse or layouts_unaligned or reference_chain"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:]


@pytest.mark.parametrize("path", [1, 2])
def test_flash_gemm_unaligned_foffset_small_dims(dev, tmp_path, path):
    """flash_ptr + a byte offset that is not sector aligned (12-byte headers), dimensions smaller
    than / not multiples of the tile: every request of every file goes through the buffered twin
    (one descriptor mode per file per call); both paths, beta != 0, column-major too."""
    hdr = 12
    for ord_, ta, tb, (m, n, k) in [("R", "N", "N", (300, 100, 200)), ("C", "T", "N", (129, 257, 64)),
                                   ("R", "N", "T", (1, 1, 1))]:
        blk = 128
        rng = np.random.default_rng(m)
        sa, sb, sc = stored_shapes(ord_, ta, tb, m, n, k)
        a = rng.uniform(-1, 1, sa).astype(np.float32)
        b = rng.uniform(-1, 1, sb).astype(np.float32)
        c0 = rng.uniform(-1, 1, sc).astype(np.float32)
        ref = orc.flash_gemm(ord_, ta, tb, m, n, k, -0.5, 1.25, a, b, c0.copy(), 0, 0, 0, blk)
        pad = np.full(hdr // 4, 7.0, np.float32)
        sub = tmp_path / f"{ord_}{ta}{tb}{path}"
        sub.mkdir()
        F = Files(sub, a=np.concatenate([pad, a.ravel()]), b=np.concatenate([pad, b.ravel()]),
                  c=np.concatenate([pad, c0.ravel(), pad]))
        try:
            bofhip.flash_gemm(ord_, ta, tb, m, n, k, -0.5, 1.25, F.fptr("a", hdr), F.fptr("b", hdr), F.fptr("c", hdr),
                              0, 0, 0, bofhip.default_options(gemm_blk=blk, gemm_path=path, io_chunk_mib=1))
            got = F.read("c", np.float32, (-1,))
            assert np.array_equal(got[:3], pad) and np.array_equal(got[-3:], pad)      # neighbours untouched
            assert np.array_equal(got[3:-3].reshape(sc), ref), (ord_, ta, tb)
        finally:
            F.close()


def test_flash_gemm_panels_not_eligible_falls_back(dev, tmp_path):
    """ldc > n (the gaps between C's rows are not ours to rewrite) and tiny budgets go to the tile
    cache; gemm_path = 2 makes that an error instead of a silent change of path."""
    m, k, n, blk = 512, 384, 256, 128
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n + 128)).astype(np.float32)
    ref = orc.flash_gemm("R", "N", "N", m, n, k, 1.0, 1.0, a, b, c0.copy(), 0, 0, n + 128, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        bofhip.flash_gemm("R", "N", "N", m, n, k, 1.0, 1.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, n + 128,
                          bofhip.default_options(gemm_blk=blk))
        assert np.array_equal(F.read("c", np.float32, c0.shape), ref)
        with pytest.raises(bofhip.BofError):
            bofhip.flash_gemm("R", "N", "N", m, n, k, 1.0, 1.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, n + 128,
                              bofhip.default_options(gemm_blk=blk, gemm_path=2))
    finally:
        F.close()


def test_flash_gemm_aligned_multitile_small_budget(dev, tmp_path):
    """1024^3 with 256-tiles (4x4x4 = 64 tasks), HBM budget of 12 tile slots: forces the
    C super-block ordering and farthest-next-use eviction; C written once, never re-read."""
    n, blk = 1024, 256
    a = orc.dense_fill(n, n, "s")
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    c0 = np.zeros((n, n), np.float32)
    ref = orc.flash_gemm("R", "N", "N", n, n, n, 1.0, 0.0, a, b, c0.copy(), 0, 0, 0, blk)
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, hbm_budget=12 * blk * blk * 4, n_streams=2)
        bofhip.flash_gemm("R", "N", "N", n, n, n, 1.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"),
                          0, 0, 0, opts)
        assert np.array_equal(F.read("c", np.float32, (n, n)), ref)
        st = bofhip.flash_last_stats()
        assert st["tasks"] == 64 and st["bytes_written"] == 4 * n * n
        # beta == 0: C never read; A and B re-read at most (#super-block passes) times
        assert 2 * 4 * n * n <= st["bytes_read"] <= 6 * 4 * n * n
        # the real pipeline moves exactly the bytes its dry-run scheduler predicts
        sim = bofhip.flash_gemm_simulate("R", "N", "N", n, n, n, 0.0, blk, 12, lookahead=2 * opts.pinned_slots)
        assert (st["bytes_read"], st["bytes_written"], st["tasks"]) == \
            (sim["bytes_read"], sim["bytes_written"], sim["tasks"])
    finally:
        F.close()
    # with the whole working set resident every tile is read exactly once
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        bofhip.flash_gemm("R", "N", "N", n, n, n, 1.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"),
                          0, 0, 0, bofhip.default_options(gemm_blk=blk))
        assert np.array_equal(F.read("c", np.float32, (n, n)), ref)
        assert bofhip.flash_last_stats()["bytes_read"] == 2 * 4 * n * n
    finally:
        F.close()


def test_flash_gemm_foffset_and_row_shard(dev, tmp_path):
    """flash_ptr + offset: two 'ranks' each compute a row slab of C in place (the
    multi-GPU sharding of SURVEY 8e run sequentially on one GPU), with a 4 KiB header
    in front of every matrix (foffset != 0)."""
    import bof_dist
    m, k, n, blk, hdr = 512, 384, 256, 128, 4096
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    ref = orc.flash_gemm("R", "N", "N", m, n, k, 1.0, 1.0, a, b, c0.copy(), 0, 0, 0, blk)
    pad = np.zeros(hdr // 4, np.float32)
    F = Files(tmp_path, a=np.concatenate([pad, a.ravel()]), b=np.concatenate([pad, b.ravel()]),
              c=np.concatenate([pad, c0.ravel()]))
    try:
        opts = bofhip.default_options(gemm_blk=blk)
        for rank in range(2):
            ml, offa, offc = bof_dist.gemm_shard_args(m, n, k, 0, 0, 2, rank, blk)
            bofhip.flash_gemm("R", "N", "N", ml, n, k, 1.0, 1.0, F.fptr("a", hdr + 4 * offa),
                              F.fptr("b", hdr), F.fptr("c", hdr + 4 * offc), 0, 0, 0, opts)
        got = F.read("c", np.float32, (-1,))[hdr // 4:].reshape(m, n)
        assert np.array_equal(got, ref)
    finally:
        F.close()


@pytest.mark.parametrize("ord_b,k,alpha,beta", [("R", 128, 1.0, 0.0), ("R", 1030, 0.5, 2.0),
                                               ("C", 128, 1.0, 0.0), ("C", 1030, 0.5, 2.0)])
def test_flash_csrmm(dev, tmp_path, golden, ord_b, k, alpha, beta):
    """SURVEY 8c csrmm cases incl. ('C', k=1030) which the reference gets wrong
    (App. B-15: in-place 1-based conversion of a shared cached buffer)."""
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
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000, n_io_threads=2)
        bofhip.flash_csrmm("N", m, n, k, alpha, beta, F.fptr("val"), F.fptr("ia"), F.fptr("ja"),
                           ord_b, F.fptr("b"), F.fptr("c"), opts)
        got = F.read("c", np.float32, c0.shape)
        assert np.array_equal(got, ref)
        assert np.array_equal(F.read("ja", np.int64, (-1,)), ja)     # CSR files untouched
        assert np.array_equal(F.read("ia", np.int64, (-1,)), ia)
        # reference error behaviour: -1 for unrecognised flags (csrmm.cpp:433-448)
        with pytest.raises(bofhip.BofError):
            bofhip.flash_csrmm("N", m, n, k, alpha, beta, F.fptr("val"), F.fptr("ia"),
                               F.fptr("ja"), "X", F.fptr("b"), F.fptr("c"), opts)
        with pytest.raises(bofhip.BofError):
            bofhip.flash_csrmm("Q", m, n, k, alpha, beta, F.fptr("val"), F.fptr("ia"),
                               F.fptr("ja"), ord_b, F.fptr("b"), F.fptr("c"), opts)
    finally:
        F.close()


@pytest.mark.parametrize("ord_b,k", [("R", 128), ("R", 96), ("C", 128), ("C", 72)])
def test_flash_csrmm_mixed_alignment_stress(dev, tmp_path, ord_b, k):
    """Row blocks of 300 rows: some C block regions are sector aligned, most are not, and
    neighbouring blocks share pages; C is read (beta != 0) and written by a pool of threads.  Every
    request of the call on the C file must go through one descriptor mode; repeated so that a lost
    update would show."""
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    rng = np.random.default_rng(k)
    b = rng.integers(0, 7, (n, k)).astype(np.float32)
    c0 = rng.integers(0, 5, (m, k)).astype(np.float32)
    if ord_b == "C":
        b, c0 = np.ascontiguousarray(b.T), np.ascontiguousarray(c0.T)
    ref = orc.flash_csrmm(ord_b, m, n, k, 0.5, 2.0, val, ia, ja, b, c0.copy(), 300, 5000, 1024)
    F = Files(tmp_path, val=val, ja=ja, ia=ia, b=b, c=c0)
    try:
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=300, n_io_threads=8, pinned_slots=8)
        for rep in range(12):
            bofhip.flash_csrmm("N", m, n, k, 0.5, 2.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), ord_b,
                               F.fptr("b"), F.fptr("c"), opts)
            assert np.array_equal(F.read("c", np.float32, c0.shape), ref), rep
            c0.tofile(F.paths["c"])
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
    finally:
        F.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("beta,head", [(0.0, 0), (2.0, 0), (2.0, 1000)])
def test_flash_csrmm_unaligned_c_keeps_odirect(dev, tmp_path, monkeypatch, devices, beta, head):
    """k = 100: 400-byte C rows, row blocks of 300 rows -- hardly any block of the C file starts or ends on a sector,
    neighbouring blocks (of one device, of two devices) share pages.  The C file stays on O_DIRECT (VERDICT r5 item 5;
    the reference: sector read-modify-write with neighbour ordering, src/file_handles/flash_file_handle.cpp:558-716,
    src/scheduler/io_executor.cpp:28-156): whole pages direct from the pinned buffer, the partial edge pages through
    the page cache, old contents (beta != 0) as the sector-aligned superset; NO row block goes through the buffered
    twin.  Repeated, bit-exact against the oracle every time; $BOF_UNALIGNED_DIRECT=0 restores the twin."""
    m, n, k = 4096, 2048, 100
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    rng = np.random.default_rng(100 + head)
    b = rng.integers(0, 7, (n, k)).astype(np.float32)
    c0 = rng.integers(0, 5, (m, k)).astype(np.float32)
    ref = orc.flash_csrmm("R", m, n, k, 0.5, beta, val, ia, ja, b, c0.copy(), 300, 5000, 1024)
    pad = np.full(head // 4, -7.0, np.float32)
    F = Files(tmp_path, val=val, ja=ja, ia=ia, b=b, c=np.concatenate([pad, c0.ravel()]))
    try:
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=300, n_io_threads=8, pinned_slots=8, devices=devices)
        direct = bool(fcntl.fcntl(F.fds["c"], fcntl.F_GETFL) & os.O_DIRECT)
        for rep in range(6):
            bofhip.flash_csrmm("N", m, n, k, 0.5, beta, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), "R",
                               F.fptr("b"), F.fptr("c", head), opts)
            mode, twin = bofhip.flash_last_c_file()
            got = np.fromfile(F.paths["c"], np.float32)
            assert np.array_equal(got[:pad.size], pad) and np.array_equal(got[pad.size:].reshape(m, k), ref), rep
            if direct:
                assert mode == 2 and twin == 0, (mode, twin)
            np.concatenate([pad, c0.ravel()]).tofile(F.paths["c"])
            os.posix_fadvise(F.fds["c"], 0, 0, os.POSIX_FADV_DONTNEED)
        monkeypatch.setenv("BOF_UNALIGNED_DIRECT", "0")
        bofhip.flash_csrmm("N", m, n, k, 0.5, beta, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), "R",
                           F.fptr("b"), F.fptr("c", head), opts)
        mode, twin = bofhip.flash_last_c_file()
        assert np.array_equal(np.fromfile(F.paths["c"], np.float32)[pad.size:].reshape(m, k), ref)
        if direct:
            assert mode == 0 and twin == m * k * 4, (mode, twin)
    finally:
        F.close()


@pytest.mark.parametrize("ord_b", ["R", "C"])
def test_flash_csrmm_inmem_bc(dev, tmp_path, ord_b):
    """csrmm overload with B and C in host memory (include/flash_blas.h:43-46).
    The reference returns -1 for 'R' even after doing the work; both layouts succeed here."""
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
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000)
        bofhip.flash_csrmm_inmem("N", m, n, k, 0.5, 2.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"),
                                 ord_b, b.ctypes.data, c.ctypes.data, opts)
        assert np.array_equal(c, ref)
        with pytest.raises(bofhip.BofError):
            bofhip.flash_csrmm_inmem("Q", m, n, k, 0.5, 2.0, F.fptr("val"), F.fptr("ia"),
                                     F.fptr("ja"), ord_b, b.ctypes.data, c.ctypes.data, opts)
        # trans_a = 'T' with host B (m x k) / C (n x k): rejected by the reference, computed here
        bt = rng.uniform(-1, 1, (m, k)).astype(np.float32)
        ct0 = rng.uniform(-1, 1, (n, k)).astype(np.float32)
        ref_t = orc.scsrmm_t(m, n, k, 0.5, val, ia, ja, bt, k, 2.0, ct0.copy(), k)
        if ord_b == "C":
            bt, ct = np.ascontiguousarray(bt.T), np.ascontiguousarray(ct0.T)
        else:
            ct = ct0.copy()
        bofhip.flash_csrmm_inmem("T", m, n, k, 0.5, 2.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"),
                                 ord_b, bt.ctypes.data, ct.ctypes.data, opts)
        assert np.array_equal(ct.T if ord_b == "C" else ct, ref_t)
    finally:
        F.close()


@pytest.mark.parametrize("direct,budget", [(True, 0), (False, 0), (True, 2 << 20), (False, 3 << 20)])
def test_flash_csrcsc(dev, tmp_path, golden_tr, direct, budget):
    """flash::csrcsc on files: generator matrix pinned by the mkl_csrcsc hashes,
    a ragged random matrix with empty rows/columns against the oracle, inputs left untouched."""
    import hashlib
    want = {t.split()[1]: t.split()[2] for t in golden_tr["meta"] if t.startswith("exact")}
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    nnz = ja.size
    F = Files(tmp_path, direct, val=val, ja=ja, ia=ia, val_tr=np.zeros(nnz, np.float32),
              ja_tr=np.zeros(nnz, np.int64), ia_tr=np.zeros(n + 1, np.int64))
    try:
        # budget > 0 forces the out-of-core scheme: row blocks transposed in HBM into temporary
        # files, then column blocks merged (several blocks of each kind at these sizes)
        bofhip.flash_csrcsc(m, n, F.fptr("ia"), F.fptr("ja"), F.fptr("val"), F.fptr("ia_tr"),
                            F.fptr("ja_tr"), F.fptr("val_tr"),
                            bofhip.default_options(n_io_threads=3, hbm_budget=budget))
        st = bofhip.flash_last_stats()
        if budget == 0:
            assert st["bytes_read"] == nnz * 12 + (m + 1) * 8 and st["bytes_written"] == nnz * 12 + (n + 1) * 8
        else:   # every entry goes through the temporary files once: 2x reads, 2x writes
            assert st["bytes_read"] == 2 * nnz * 12 + (m + 1) * 8
            assert st["bytes_written"] == 2 * nnz * 12 + (n + 1) * 8 and st["tasks"] >= 4
        h = lambda name, dt: hashlib.sha256(F.read(name, dt, (-1,)).tobytes()).hexdigest()
        assert h("val_tr", np.float32) == want["gen_tr_val"]
        assert h("ja_tr", np.int64) == want["gen_tr_ja"]
        assert h("ia_tr", np.int64) == want["gen_tr_ia"]
        assert np.array_equal(F.read("ja", np.int64, (-1,)), ja)
        assert np.array_equal(F.read("val", np.float32, (-1,)), val)
    finally:
        F.close()
    rng = np.random.default_rng(9)
    m, n = 3000, 70000
    counts = rng.integers(0, 40, m); counts[0] = 0; counts[m - 1] = 0; counts[77] = 900
    ia = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ja = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts]).astype(np.int64)
    val = rng.uniform(-1, 1, ja.size).astype(np.float32)
    wv, wi, wj = orc.csrcsc(m, n, val, ia, ja)
    sub = tmp_path / "r"; sub.mkdir()
    F = Files(sub, direct, val=val, ja=ja, ia=ia, val_tr=np.zeros(ja.size, np.float32),
              ja_tr=np.zeros(ja.size, np.int64), ia_tr=np.zeros(n + 1, np.int64))
    try:
        bofhip.flash_csrcsc(m, n, F.fptr("ia"), F.fptr("ja"), F.fptr("val"), F.fptr("ia_tr"),
                            F.fptr("ja_tr"), F.fptr("val_tr"),
                            bofhip.default_options(hbm_budget=(3 << 20) if budget else 0))
        if budget:
            assert bofhip.flash_last_stats()["tasks"] >= 4
        assert np.array_equal(F.read("ia_tr", np.int64, (-1,)), wi)
        assert np.array_equal(F.read("ja_tr", np.int64, (-1,)), wj)
        assert np.array_equal(F.read("val_tr", np.float32, (-1,)), wv)
    finally:
        F.close()


@pytest.mark.parametrize("budget", [0, 4 << 20])
@pytest.mark.parametrize("ord_b,k,alpha,beta", [("R", 128, 1.0, 0.0), ("R", 1030, 0.5, 2.0),
                                               ("C", 128, 0.5, 2.0), ("C", 1030, 1.0, 0.0)])
def test_flash_csrmm_trans(dev, tmp_path, golden_tr, ord_b, k, alpha, beta, budget):
    """csrmm trans_a='T' on files: C[n x k] = alpha A^T B[m x k] + beta C (the reference's path
    is broken; the oracle is mkl_scsrmm('T')-pinned)."""
    import hashlib
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    b = orc.dense_fill(m, k, "s")
    rng = np.random.default_rng(k)
    c0 = rng.integers(0, 5, (n, k)).astype(np.float32) if beta else np.zeros((n, k), np.float32)
    ref = orc.scsrmm_t(m, n, k, alpha, val, ia, ja, b, k, beta, c0.copy(), k)
    if k == 128 and alpha == 1.0 and beta == 0.0:
        want = {t.split()[1]: t.split()[2] for t in golden_tr["meta"] if t.startswith("exact")}
        assert hashlib.sha256(ref.tobytes()).hexdigest() == want["gen_csrmmT_c"]
    bs, cs = (np.ascontiguousarray(b.T), np.ascontiguousarray(c0.T)) if ord_b == "C" else (b, c0)
    F = Files(tmp_path, val=val, ja=ja, ia=ia, b=bs, c=cs)
    try:
        # budget > 0: A^T does not "fit" -> out-of-core transposition into temporary files, then
        # the ordinary file pipeline of the 'N' case on them
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=300, n_io_threads=2, hbm_budget=budget)
        bofhip.flash_csrmm("T", m, n, k, alpha, beta, F.fptr("val"), F.fptr("ia"), F.fptr("ja"),
                           ord_b, F.fptr("b"), F.fptr("c"), opts)
        got = F.read("c", np.float32, cs.shape)
        assert np.array_equal(got.T if ord_b == "C" else got, ref)
        assert np.array_equal(F.read("ja", np.int64, (-1,)), ja)
    finally:
        F.close()


@pytest.mark.parametrize("trans", ["N", "T"])
def test_flash_csrgemv(dev, tmp_path, golden, trans):
    import hashlib
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    x = (np.arange(n if trans == "N" else m) % 10).astype(np.float32)
    y = np.full(m if trans == "N" else n, 3.0, np.float32)
    F = Files(tmp_path, val=val, ja=ja, ia=ia)
    try:
        opts = bofhip.default_options(max_nnzs=5000, csrmm_rblk=1000)
        bofhip.flash_csrgemv(trans, m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), x.ctypes.data,
                             y.ctypes.data, opts)
        want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
        assert hashlib.sha256(y.tobytes()).hexdigest() == want["gen_csrgemv_" + trans]
    finally:
        F.close()


def test_flash_csr_ragged_random(dev, tmp_path):
    """Random values, ragged rows incl. empty rows and empty blocks; buffered descriptors."""
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
        opts = bofhip.default_options(max_nnzs=300, csrmm_rblk=200, csrmm_cblk=64, use_odirect=0)
        bofhip.flash_csrmm("N", m, n, k, 0.5, 2.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), "R",
                           F.fptr("b"), F.fptr("c"), opts)
        assert np.array_equal(F.read("c", np.float32, (m, k)), ref)
        x = rng.uniform(-1, 1, n).astype(np.float32)
        y = np.zeros(m, np.float32)
        bofhip.flash_csrgemv("N", m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), x.ctypes.data,
                             y.ctypes.data, opts)
        assert np.array_equal(y, orc.flash_csrgemv("N", m, n, val, ia, ja, x, np.zeros(m, np.float32), 200, 300))
        xt = rng.uniform(-1, 1, m).astype(np.float32)
        yt = np.zeros(n, np.float32)
        bofhip.flash_csrgemv("T", m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), xt.ctypes.data,
                             yt.ctypes.data, opts)
        rt = orc.flash_csrgemv("T", m, n, val, ia, ja, xt, np.zeros(n, np.float32), 200, 300)
        assert np.abs(yt - rt).max() / np.abs(rt).max() < 1e-4   # atomics: order-dependent rounding
    finally:
        F.close()


# ---- the C++ API through the drivers (same argv as the reference's drivers) ----------------
def run_driver(name, args, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([os.path.join(BIN, name)] + [str(a) for a in args], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_gemm_driver(dev, tmp_path):
    m, k, n = 384, 256, 512
    rng = np.random.default_rng(4)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)       # 'T': stored n x k
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    ref = orc.flash_gemm("R", "N", "T", m, n, k, 1.5, 0.5, a, b, c0.copy(), 0, 0, 0, 128)
    pa, pb, pc = (str(tmp_path / f) for f in ("A", "B", "C"))
    a.tofile(pa); b.tofile(pb); c0.tofile(pc)
    out = run_driver("gemm_driver", [pa, pb, pc, m, k, n, 1.5, 0.5, "N", "T", "R", k, k, n],
                     {"BOF_GEMM_BLK_SIZE": "128"})
    assert "gemm() took" in out and "returned with 0" in out
    assert np.array_equal(np.fromfile(pc, np.float32).reshape(m, n), ref)


REF_BIN = os.path.join(ROOT, "oracle", "_ref")


def run_ref_driver(name, args, env_extra):
    exe = os.path.join(REF_BIN, name)
    if not os.access(exe, os.X_OK):
        pytest.skip("oracle/_ref/%s is built only where the reference tree exists (`make -C oracle ref`)" % name)
    env = dict(os.environ, **env_extra)
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_reference_driver_binaries_unchanged(dev, tmp_path, golden):
    """The reference's OWN drivers/gemm.cpp, csrmm.cpp, csrmm_pmem.cpp, csrgemv.cpp -- compiled unchanged against
    blas-on-flash_amd/include, linked against libflashblas.so + libbof_hip.so (oracle/Makefile) --
    run with the reference's argv on files: the drop-in boundary end to end."""
    import hashlib
    m, k, n = 384, 256, 512
    rng = np.random.default_rng(4)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k)).astype(np.float32)       # 'T': stored n x k
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    ref = orc.flash_gemm("R", "N", "T", m, n, k, 1.5, 0.5, a, b, c0.copy(), 0, 0, 0, 128)
    pa, pb, pc = (str(tmp_path / f) for f in ("A", "B", "C"))
    a.tofile(pa); b.tofile(pb); c0.tofile(pc)
    out = run_ref_driver("ref_gemm_driver", [pa, pb, pc, m, k, n, 1.5, 0.5, "N", "T", "R", k, k, n],
                         {"BOF_GEMM_BLK_SIZE": "128"})
    assert "gemm() took" in out
    assert np.array_equal(np.fromfile(pc, np.float32).reshape(m, n), ref)
    m, n, k = 4096, 2048, 128
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    p = {x: str(tmp_path / x) for x in ("csr", "col", "off", "B2", "C2", "x", "y")}
    val.tofile(p["csr"]); ja.tofile(p["col"]); ia.tofile(p["off"]); orc.dense_fill(n, k, "s").tofile(p["B2"])
    np.zeros((m, k), np.float32).tofile(p["C2"])
    env = {"BOF_MAX_NNZS": "5000", "BOF_CSRMM_RBLK_SIZE": "1000"}
    out = run_ref_driver("ref_csrmm_driver", [p["csr"], p["col"], p["off"], p["B2"], p["C2"], m, n, k, 1.0, 0.0, "N", "R"], env)
    assert "csrmm() took" in out
    want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
    assert hashlib.sha256(np.fromfile(p["C2"], np.float32).tobytes()).hexdigest() == want["gen_csrmm_c"]
    # drivers/csrmm_pmem.cpp: the overload with B and C in host memory, no flash_setup either.
    # It runs to the end; its own write-back of C opens the file with std::ios::binary alone, which
    # libstdc++ refuses, so the file keeps its contents -- only completion can be checked here (the
    # overload's results are checked through the C ABI in test_flash_csrmm_inmem_bc)
    out = run_ref_driver("ref_csrmm_pmem_driver", [p["csr"], p["col"], p["off"], p["B2"], p["C2"], m, n, k, 1.0, 0.0, "N", "R"], env)
    assert "Finished csrmm" in out
    # the reference's csrgemv driver never calls flash_setup (it hangs with the reference library,
    # SURVEY App. B-1); here it runs as shipped
    for trans in "NT":
        (np.arange(n if trans == "N" else m) % 10).astype(np.float32).tofile(p["x"])
        np.zeros(m if trans == "N" else n, np.float32).tofile(p["y"])
        run_ref_driver("ref_csrgemv_driver", [p["csr"], p["col"], p["off"], p["x"], p["y"], m, n, trans], env)
        assert hashlib.sha256(np.fromfile(p["y"], np.float32).tobytes()).hexdigest() == want["gen_csrgemv_" + trans]


def test_csrmm_and_csrgemv_drivers(dev, tmp_path, golden):
    import hashlib
    m, n, k = 4096, 2048, 128
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    b = orc.dense_fill(n, k, "s")
    p = {x: str(tmp_path / x) for x in ("csr", "col", "off", "B", "C", "x", "y")}
    val.tofile(p["csr"]); ja.tofile(p["col"]); ia.tofile(p["off"]); b.tofile(p["B"])
    np.zeros((m, k), np.float32).tofile(p["C"])
    env = {"BOF_MAX_NNZS": "5000", "BOF_CSRMM_RBLK_SIZE": "1000"}
    out = run_driver("csrmm_driver", [p["csr"], p["col"], p["off"], p["B"], p["C"], m, n, k, 1.0, 0.0,
                                      "N", "R"], env)
    assert "csrmm() took" in out
    want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
    assert hashlib.sha256(np.fromfile(p["C"], np.float32).tobytes()).hexdigest() == want["gen_csrmm_c"]
    for trans in "NT":
        (np.arange(n if trans == "N" else m) % 10).astype(np.float32).tofile(p["x"])
        np.zeros(m if trans == "N" else n, np.float32).tofile(p["y"])
        run_driver("csrgemv_driver", [p["csr"], p["col"], p["off"], p["x"], p["y"], m, n, trans], env)
        assert hashlib.sha256(np.fromfile(p["y"], np.float32).tobytes()).hexdigest() == \
            want["gen_csrgemv_" + trans]


def test_csrcsc_driver_and_csrmm_driver_trans(dev, tmp_path, golden_tr):
    """The C++ boundary for the transposition row: csrcsc_driver (reference argv,
    drivers/csrcsc.cpp:17-28) and csrmm_driver with trans_a = T."""
    import hashlib
    want = {t.split()[1]: t.split()[2] for t in golden_tr["meta"] if t.startswith("exact")}
    m, n, k = 4096, 2048, 128
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    p = {x: str(tmp_path / x) for x in ("csr", "col", "off", "tcsr", "tcol", "toff", "B", "C")}
    val.tofile(p["csr"]); ja.tofile(p["col"]); ia.tofile(p["off"])
    np.zeros(ja.size, np.float32).tofile(p["tcsr"]); np.zeros(ja.size, np.int64).tofile(p["tcol"])
    np.zeros(n + 1, np.int64).tofile(p["toff"])
    out = run_driver("csrcsc_driver", [p["csr"], p["col"], p["off"], p["tcsr"], p["tcol"], p["toff"], m, n], {})
    assert "csrcsc() took" in out
    h = lambda f, dt: hashlib.sha256(np.fromfile(p[f], dt).tobytes()).hexdigest()
    assert (h("tcsr", np.float32), h("tcol", np.int64), h("toff", np.int64)) == \
        (want["gen_tr_val"], want["gen_tr_ja"], want["gen_tr_ia"])
    orc.dense_fill(m, k, "s").tofile(p["B"])
    np.zeros((n, k), np.float32).tofile(p["C"])
    out = run_driver("csrmm_driver", [p["csr"], p["col"], p["off"], p["B"], p["C"], m, n, k, 1.0, 0.0,
                                      "T", "R"], {"BOF_MAX_NNZS": "5000", "BOF_CSRMM_RBLK_SIZE": "300"})
    assert "csrmm() took" in out
    assert h("C", np.float32) == want["gen_csrmmT_c"]
    # transposed matrix fed back through the 'N' path gives the same product
    np.zeros((n, k), np.float32).tofile(p["C"])
    run_driver("csrmm_driver", [p["tcsr"], p["tcol"], p["toff"], p["B"], p["C"], n, m, k, 1.0, 0.0,
                                "N", "R"], {})
    assert h("C", np.float32) == want["gen_csrmmT_c"]


@pytest.mark.parametrize("b_once", ["0", "1"])
@pytest.mark.parametrize("padded", [False, True])
@pytest.mark.parametrize("nproc", [1, 2, 3])
def test_flash_gemm_row_sharded_files(dev, tmp_path, nproc, b_once, padded):
    """Multi-GPU file path, one process per GPU: every rank runs the level-3
    pipeline on its row slab (flash_ptr + offset), no data-path collective.  b_once = 1 (the default
    of bof_dist.flash_gemm_row_sharded): B's panels are read from the file once per NODE -- panel l by
    rank l % world, published in a node-shared staging ring, taken from there by the others --
    so the ranks' bytes_read add up to A + B + C, and bytes_peer to (world - 1) x B.  padded: leading
    dimensions with gaps send the call to the tile cache, which reads its tiles itself.  Either way
    the C file equals the restated flash::gemm bit for bit.  nproc > 1 shares cuda:0 between the
    ranks (single-GPU box; gloo for the barriers)."""
    import json
    import sys
    m, k, n, blk = 1100, 600, 500, 128
    lda, ldb, ldc = (k + 8, n + 4, n + 12) if padded else (k, n, n)
    alpha, beta = 0.5, 2.0
    rng = np.random.default_rng(23)
    a = rng.uniform(-1, 1, (m, lda)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, ldb)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, ldc)).astype(np.float32)
    ref = orc.flash_gemm("R", "N", "N", m, n, k, alpha, beta, a, b, c0.copy(), lda, ldb, ldc, blk)
    pa, pb, pc = (str(tmp_path / f) for f in ("A", "B", "C"))
    a.tofile(pa); b.tofile(pb); c0.tofile(pc)
    tool = os.path.join(ROOT, "tools", "dist_file_gemm.py")
    args = [pa, pb, pc, m, n, k, alpha, beta, lda, ldb, ldc, blk]
    if nproc == 1:
        cmd = [sys.executable, tool]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(29577 + int(b_once) + 2 * int(padded)), tool]
    r = subprocess.run(cmd + [str(x) for x in args], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, BOF_BENCH_ONE_GPU="1", BOF_B_ONCE=b_once, BOF_IO_CHUNK_MIB="1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(recs) == nproc
    assert sum(x["rows"] for x in recs) == m
    got = np.fromfile(pc, np.float32).reshape(m, ldc)
    assert np.array_equal(got, ref)
    if not padded:
        rd = sum(x["bytes_read"] for x in recs)
        peer = sum(x["bytes_peer"] for x in recs)
        if b_once == "1":
            assert rd == 4 * (a.size + b.size + c0.size) and peer == (nproc - 1) * 4 * b.size
        else:
            assert rd == 4 * (a.size + nproc * b.size + c0.size) and peer == 0
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("bof_2957")]     # staging removed


@pytest.mark.parametrize("nproc", [1, 2])
def test_flash_csr_row_sharded_files(dev, tmp_path, golden, nproc):
    """csrmm / csrgemv 'N' / csrgemv 'T' on files, rows sharded over the ranks by non-zeros; only
    csrgemv 'T' uses a collective (all-reduce of the partial vectors).  Results pinned by the MKL
    hashes of the generator matrix (identical whatever the number of ranks)."""
    import hashlib
    import json
    import sys
    m, n, k = 4096, 2048, 128
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    val.tofile(tmp_path / "A.csr"); ja.tofile(tmp_path / "A.col"); ia.tofile(tmp_path / "A.off")
    orc.dense_fill(n, k, "s").tofile(tmp_path / "B.bin")
    np.zeros((m, k), np.float32).tofile(tmp_path / "C.bin")
    (np.arange(max(m, n)) % 10).astype(np.float32).tofile(tmp_path / "x.bin")
    tool = os.path.join(ROOT, "tools", "dist_file_csr.py")
    cmd = [sys.executable, tool] if nproc == 1 else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
         "127.0.0.1", "--master-port", "29579", tool]
    r = subprocess.run(cmd + [str(tmp_path), str(m), str(n), str(k)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, BOF_BENCH_ONE_GPU="1", BOF_MAX_NNZS="5000", BOF_CSRMM_RBLK_SIZE="1000"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(recs) == nproc and recs[0]["rows"][0] == 0 and recs[-1]["rows"][1] == m
    want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
    h = lambda f: hashlib.sha256(np.fromfile(tmp_path / f, np.float32).tobytes()).hexdigest()
    assert h("C.bin") == want["gen_csrmm_c"]
    assert h("yN.bin") == want["gen_csrgemv_N"]
    assert h("yT.bin") == want["gen_csrgemv_T"]


@pytest.mark.parametrize("path", [1, 2])
def test_flash_gemm_buffered_descriptors(dev, tmp_path, path):
    """Files opened WITHOUT O_DIRECT and just written (so in the page cache): reads are preads, the 1 MiB
    write-back requests of the panel path take the shared-mapping route of fileio.cpp (pages resident),
    the tile path's row-sized ones pwrite.  Bit-exact against the oracle either way, beta != 0."""
    m, k, n, blk = 1024, 768, 1024, 256
    rng = np.random.default_rng(41)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    ref = orc.flash_gemm("R", "N", "N", m, n, k, 0.5, 1.5, a, b, c0.copy(), 0, 0, 0, blk)
    F = Files(tmp_path, direct=False, a=a, b=b, c=c0)
    try:
        opts = bofhip.default_options(gemm_blk=blk, n_streams=2, n_io_threads=4, pinned_slots=4, gemm_path=path,
                                      io_chunk_mib=1, use_odirect=0)
        bofhip.flash_gemm("R", "N", "N", m, n, k, 0.5, 1.5, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0, opts)
        for fd in F.fds.values():
            bofhip.lib().bof_file_forget(fd)
        assert np.array_equal(F.read("c", np.float32, (m, n)), ref)
    finally:
        F.close()


@pytest.mark.parametrize("m,n,k", [(3, 2000, 3), (5, 1500, 130), (2, 4000, 64)])
def test_flash_gemm_tilecache_short_wide_tiles(dev, tmp_path, m, n, k):
    """Tile cache with tiles that are short and wide (m of a few rows): one ROW of a 16-tile row group is wider than a
    whole tile, so the pinned staging slots must be sized by the group row -- sized by the tile they were overrun by the
    first chunk (the segmentation faults of the round-4 fuzz: flush_wgroup -> hipMemcpy2DAsync past the pinned block)."""
    rng = np.random.default_rng(m * 1000 + k)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    want = orc.flash_gemm("R", "N", "N", m, n, k, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, 128)
    paths = [str(tmp_path / x) for x in "ABC"]
    for x, p in zip((a, b, c0), paths):
        x.tofile(p)
    for rep in range(3):                       # fresh pinned blocks the first time, cached ones later
        c0.tofile(paths[2])
        fds = [os.open(p, os.O_RDWR) for p in paths]
        try:
            bofhip.lib().bof_flash_release()   # exact-size staging blocks, no slack from the block cache
            bofhip.flash_gemm("R", "N", "N", m, n, k, 0.5, 2.0, bofhip.FPtr(fds[0], 0), bofhip.FPtr(fds[1], 0),
                              bofhip.FPtr(fds[2], 0), 0, 0, 0,
                              bofhip.default_options(gemm_blk=128, gemm_path=1, use_odirect=0, io_chunk_mib=1))
        finally:
            for fd in fds:
                bofhip.lib().bof_file_forget(fd)
                os.close(fd)
        assert np.array_equal(np.fromfile(paths[2], np.float32).reshape(m, n), want)
