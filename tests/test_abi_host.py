"""CPU tests of the product library's host side: the C ABI loads and exports every
symbol include/bof_hip.h declares, the tilers agree with the oracle, the file
reader honours the StrideInfo contract, and -- on a box without a GPU -- every
compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bofhip
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bof_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bof_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = bofhip.lib()
    names = declared_symbols()
    assert len(names) >= 30
    bound = {s[0] for s in bofhip.SYMBOLS}
    for n in names:
        assert hasattr(L, n), f"{n} declared in bof_hip.h but not exported"
        assert n in bound, f"{n} has no ctypes signature in bofhip.SYMBOLS"
    assert L.bof_abi_version() == 5


def test_header_cites_reference_for_each_level():
    src = open(HEADER).read()
    for cite in ("include/tasks/gemm_task.h:87-90", "include/tasks/csrmm_task.h:226-228",
                 "include/tasks/csrgemv_task.h:74", "src/blas/gemm.cpp:27-202",
                 "include/flash_blas.h:14-18", "include/pointers/pointer.h:15-18"):
        assert cite in src


def test_default_options_match_reference_tunables():
    o = bofhip.default_options()
    assert (o.gemm_blk, o.max_nnzs, o.csrmm_rblk, o.csrmm_cblk) == (4096, 10_000_000, 131072, 1024)
    assert o.n_io_threads == 8 and o.n_streams == 4 and o.use_odirect == 1


@pytest.mark.parametrize("args", [
    ("R", "N", "N", 640, 500, 600, 2.0, 0, 0, 0, 256),
    ("C", "T", "N", 640, 500, 600, 0.0, 700, 800, 900, 256),
    ("R", "T", "T", 100, 50, 4096, 0.0, 0, 0, 0, 128),
    ("C", "N", "T", 300, 257, 129, 1.0, 0, 0, 0, 128),
    ("R", "N", "T", 4096 + 127, 4096 + 128, 4096, 0.5, 0, 0, 0, 4096),
    ("R", "N", "N", 32768, 32768, 32768, 0.0, 0, 0, 0, 4096),
    ("C", "N", "N", 1, 1, 1, 0.0, 0, 0, 0, 4096),
])
def test_gemm_plan_matches_oracle(args):
    a, nb = bofhip.gemm_plan(*args)
    b, nb2 = orc.gemm_plan(*args)
    assert nb == nb2 and len(a) == len(b)
    for x, y in zip(a, b):
        for f, _ in bofhip.GemmTask._fields_:
            vx, vy = getattr(x, f), getattr(y, f)
            if hasattr(vx, "__len__"):
                assert list(vx) == list(vy), f
            else:
                assert vx == vy, f


def test_gemm_plan_bad_args():
    with pytest.raises(bofhip.BofError):
        bofhip.gemm_plan("R", "N", "N", 10, 10, 10, 0.0, 0, 0, 0, 0)


def test_csr_blocks_match_oracle():
    rng = np.random.default_rng(0)
    for m, mx in [(1000, 30), (5000, 3), (128, 10), (129, 10), (1, 5)]:
        ia = np.concatenate([[0], np.cumsum(rng.integers(0, mx, m))]).astype(np.int64)
        for (mn, mr, nnz) in [(128, 131072, 10_000_000), (128, 300, 500), (16, 64, 100), (1, 7, 1)]:
            a = bofhip.csr_blocks(ia, m, mn, mr, nnz)
            b = orc.csr_blocks(ia, m, mn, mr, nnz)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert a[1].sum() == m
    m = 10_000_000
    ia = np.arange(m + 1, dtype=np.int64) * 100
    st, sz = bofhip.csr_blocks(ia, m)
    assert len(sz) == 100 and np.all(sz[:99] == 100001) and sz[99] == 99901


# ---- file reader: StrideInfo contract (reference misc/flash_file_handle_test.cpp) ---------
def _open(path, direct):
    flags = os.O_RDWR | (os.O_DIRECT if direct else 0)
    try:
        return os.open(path, flags)
    except OSError:
        pytest.skip("O_DIRECT not supported on this file system")


def _aligned(nbytes, align=4096):
    raw = np.empty(nbytes + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes]


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("use_aio", [0, 1])
def test_file_sread_swrite_iota(tmp_path, direct, use_aio):
    n = 1 << 20                                         # 8 MiB iota(uint64) file
    ref = np.arange(n, dtype=np.uint64)
    path = str(tmp_path / "iota.bin")
    ref.tofile(path)
    fd = _open(path, direct)
    L = bofhip.lib()
    rng = np.random.default_rng(1)
    view = ref.view(np.uint8)
    try:
        for trial in range(40):
            aligned = trial % 2 == 0
            if aligned:                                  # tile-like: sector-aligned everything
                ln = 512 * int(rng.integers(1, 16))
                stride = ln + 512 * int(rng.integers(0, 8))
                ns = int(rng.integers(1, 64))
                off = 512 * int(rng.integers(0, 64))
            else:                                        # CSR-segment-like: arbitrary bytes
                ln = int(rng.integers(1, 5000))
                stride = ln + int(rng.integers(0, 3000))
                ns = int(rng.integers(1, 20))
                off = int(rng.integers(0, 10000))
            buf = _aligned(ns * ln)
            buf[:] = 0xEE
            assert L.bof_file_sread(fd, off, stride, ns, ln, buf.ctypes.data, use_aio) == 0
            want = np.concatenate([view[off + s * stride: off + s * stride + ln] for s in range(ns)])
            assert np.array_equal(buf, want), (trial, aligned)
        # strided write then read back through a plain file read
        for trial, (off, stride, ns, ln) in enumerate([(4096, 8192, 32, 4096), (1000, 777, 9, 333),
                                                       (512 * 3, 512 * 5, 17, 512 * 2)]):
            data = _aligned(ns * ln)
            data[:] = rng.integers(0, 255, ns * ln, dtype=np.uint8)
            assert L.bof_file_swrite(fd, off, stride, ns, ln, data.ctypes.data, use_aio) == 0
            for s in range(ns):
                view[off + s * stride: off + s * stride + ln] = data[s * ln:(s + 1) * ln]
            os.fsync(fd)
            disk = np.fromfile(path, np.uint8)
            assert np.array_equal(disk, view), trial
        # reading past EOF is an error, not a short read
        buf = _aligned(4096)
        assert L.bof_file_sread(fd, n * 8 - 512, 0, 1, 4096, buf.ctypes.data, use_aio) != 0
        assert b"bof_file_sread" in L.bof_last_error()
    finally:
        os.close(fd)


def test_file_io_bad_fd():
    buf = _aligned(512)
    assert bofhip.lib().bof_file_sread(-1, 0, 0, 1, 512, buf.ctypes.data, 1) != 0


# ---- no GPU => loud failure, never a CPU fallback -------------------------------------------
@pytest.mark.skipif(bofhip.lib().bof_device_count() > 0, reason="only meaningful without a GPU")
def test_compute_entry_points_fail_without_gpu(tmp_path):
    with pytest.raises(bofhip.BofError):
        bofhip.require_device()
    L = bofhip.lib()
    a = np.zeros(4, np.float32)
    p = a.ctypes.data
    assert L.bof_sgemm(b"R", b"N", b"N", 2, 2, 2, 1.0, p, 2, p, 2, 0.0, p, 2, None) != 0
    assert L.bof_gemm_resident(b"R", b"N", b"N", 2, 2, 2, 1.0, 0.0, p, p, p, 0, 0, 0, None, None) != 0
    path = str(tmp_path / "m.bin")
    np.zeros(16, np.float32).tofile(path)
    fd = os.open(path, os.O_RDWR)
    try:
        f = bofhip.FPtr(fd, 0)
        rc = L.bof_flash_gemm(b"R", b"N", b"N", 2, 2, 2, 1.0, 0.0, f, f, f, 0, 0, 0, None)
        assert rc == -4 and b"no HIP device" in L.bof_last_error()       # BOF_ENODEV
        rc = L.bof_flash_csrgemv(b"N", 2, 2, f, f, f, p, p, None)
        assert rc == -4
    finally:
        os.close(fd)


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "blas-on-flash_amd")
    for dp, _, files in os.walk(pkg):
        if os.sep + "build" in dp or os.sep + "lib" in dp or os.sep + "bin" in dp:
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "liboracle" not in txt and "bof_oracle.h" not in txt and "import orc" not in txt, fn


# ---- HBM tile-cache schedule (dry run of the level-3 GEMM scheduler; no GPU needed) -----------
def test_flash_gemm_schedule_compulsory_io_when_everything_fits():
    """cfg2: 32768^3 / 4096-tile = 192 tiles; with >= 192 slots (12 GiB of the 288 GB HBM) every
    A/B tile is read exactly once, C is never read (beta = 0) and written exactly once -- the
    reference reaches this only while all C tiles fit its 8 GiB DRAM budget."""
    n, blk = 32768, 4096
    s = bofhip.flash_gemm_simulate("R", "N", "N", n, n, n, 0.0, blk, 192)
    assert s["tasks"] == 512 and s["tile_misses"] == 192
    assert s["bytes_read"] == 2 * n * n * 4 and s["bytes_written"] == n * n * 4
    s1 = bofhip.flash_gemm_simulate("R", "N", "N", n, n, n, 1.0, blk, 192)
    assert s1["bytes_read"] == 3 * n * n * 4 and s1["bytes_written"] == n * n * 4   # beta != 0: C read once
    # cfg4 per-GPU slab: 8192 x 65536 x 65536 -> A 32 + B 256 + C 32 tiles
    s = bofhip.flash_gemm_simulate("R", "N", "N", 8192, 65536, 65536, 0.0, blk, 320)
    assert s["tasks"] == 512 and s["bytes_read"] == 4 * (8192 * 65536 + 65536 * 65536)


def test_flash_gemm_schedule_under_pressure():
    """Smaller budgets: C is still written exactly once and never re-read (accumulators are
    pinned for their chain), re-reads of A/B grow monotonically as the budget shrinks and stay
    far below the reference's hash-order eviction (4.0x reads / 2.3x writes at its cfg4 analogue)."""
    n, blk = 65536, 4096
    comp = 2 * n * n * 4
    prev = None
    for slots in (768, 300, 128, 64, 32, 12):
        s = bofhip.flash_gemm_simulate("R", "N", "N", n, n, n, 0.0, blk, slots)
        assert s["tasks"] == 4096
        assert s["bytes_written"] == n * n * 4
        assert s["bytes_read"] >= comp
        if prev is not None:
            assert s["bytes_read"] >= prev
        prev = s["bytes_read"]
        if slots >= 128:      # 8 GiB, the reference's default PROGRAM_BUDGET
            assert s["bytes_read"] <= 2.0 * comp
    # ragged shapes with tail-merge and all layouts keep the invariants
    for (o, ta, tb) in [("R", "N", "N"), ("C", "T", "N"), ("R", "T", "T")]:
        s = bofhip.flash_gemm_simulate(o, ta, tb, 640, 500, 600, 2.0, 256, 8)
        assert s["tasks"] == 12 and s["bytes_written"] == 640 * 500 * 4
        assert s["bytes_read"] >= 4 * (640 * 600 + 600 * 500 + 640 * 500)
    with pytest.raises(bofhip.BofError):
        bofhip.flash_gemm_simulate("R", "N", "N", 1024, 1024, 1024, 0.0, 256, 3)


def test_flash_gemm_panel_plan_paths():
    """Which path bof_flash_gemm takes (pure host logic, include/bof_hip.h bof_flash_gemm_panel_plan):
    row panels whenever B, two A panels and three C panels fit the budget and C's rows are contiguous;
    the tile cache otherwise."""
    GiB = 1 << 30
    # cfg2 (32768^3, 4096-tile): B 4 GiB resident, A ring of 2 x 512 MiB, C ring: fits the reference's 8 GiB
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 32768, 32768, 32768, 4096, 8 * GiB)
    assert p["eligible"] and p["why"] == 0 and p["streamed"] == 0 and p["resident"][:2] == [0, 1]
    assert p["n_panels"] == [8, 8, 8] and p["n_slots"][0] == 2 and p["n_slots"][1] == 8 and 3 <= p["n_slots"][2] <= 8
    assert p["slot_bytes"][0] == 512 << 20 and p["need_bytes"] <= 8 * GiB and p["groups"] == 8
    # a larger budget deepens the C ring only up to its cap
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 32768, 32768, 32768, 4096, 200 * GiB)
    assert p["eligible"] and p["n_slots"][2] == 6            # the C ring is capped at 2*group + 4 slots
    # cfg4 size on one GPU: 16 GiB of B cannot live in an 8 GiB budget -> tile cache
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 65536, 65536, 65536, 4096, 8 * GiB)
    assert not p["eligible"] and p["why"] == 5 and p["need_bytes"] > 16 * GiB
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 65536, 65536, 65536, 4096, 200 * GiB)
    assert p["eligible"] and p["n_panels"] == [16, 16, 16] and p["slot_bytes"][1] == 1 << 30
    # a rank's slab of cfg4 (8192 x 65536 x 65536): 2 C panels, both A panels in the ring
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 8192, 65536, 65536, 4096, 200 * GiB)
    assert p["eligible"] and p["n_panels"] == [2, 16, 2] and p["streamed"] == -1 and p["resident"][2] == 1
    # A 'T' is stored k x m: paneled along k, so it is resident like B
    p = bofhip.flash_gemm_panel_plan("R", "T", "N", 32768, 32768, 32768, 4096, 200 * GiB)
    assert p["eligible"] and p["resident"][:2] == [1, 1] and p["streamed"] == -1
    # column-major C is paneled along n: B is the streamed operand, A the resident one
    p = bofhip.flash_gemm_panel_plan("C", "N", "N", 32768, 32768, 32768, 4096, 200 * GiB)
    assert p["eligible"] and p["streamed"] == 1 and p["resident"][:2] == [1, 0]
    # gaps between C's rows are not ours to rewrite; a narrow view of a wide operand is mostly gaps
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 600, 600, 600, 256, GiB, ldc=640)
    assert not p["eligible"] and p["why"] == 4
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 600, 600, 600, 256, GiB, lda=2048)
    assert not p["eligible"] and p["why"] == 3
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 600, 600, 600, 256, GiB, ldb=599)
    assert not p["eligible"] and p["why"] == 2
    # tail-merged last panel: 600 = 256 + 344 rows -> the slot is sized for 344 rows
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 600, 600, 600, 256, GiB)
    assert p["eligible"] and p["n_panels"] == [2, 2, 2] and p["slot_bytes"][0] == 2 << 20
    p = bofhip.flash_gemm_panel_plan("R", "N", "N", 600, 600, 600, 256, GiB, group=2)
    assert p["eligible"] and p["groups"] == 1


def test_panel_plan_invariants_property():
    """bof_flash_gemm_panel_plan on random problems (pure host code): an eligible plan fits its budget, its
    slot counts are consistent with the resident / streamed flags, the group arithmetic holds, and a plan
    that fits a budget also fits a larger one."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    @settings(max_examples=300, deadline=None)
    @given(st.sampled_from("RC"), st.sampled_from("NT"), st.sampled_from("NT"),
           st.integers(1, 40000), st.integers(1, 40000), st.integers(1, 40000),
           st.sampled_from([128, 256, 1000, 4096]), st.integers(1, 1 << 36), st.integers(1, 9))
    def check(ord_, ta, tb, m, n, k, blk, budget, group):
        p = bofhip.flash_gemm_panel_plan(ord_, ta, tb, m, n, k, blk, budget, group=group)
        if not p["eligible"]:
            assert p["why"] in (1, 2, 3, 4, 5)
            return
        npan, nsl = p["n_panels"], p["n_slots"]
        assert p["need_bytes"] <= budget
        assert p["need_bytes"] == sum(nsl[x] * p["slot_bytes"][x] for x in range(3))
        for x in range(3):
            assert 1 <= nsl[x] <= npan[x]
            if p["resident"][x]:
                assert nsl[x] == npan[x]
        assert p["streamed"] in (-1, 0, 1)
        if p["streamed"] >= 0:
            assert not p["resident"][p["streamed"]] and nsl[p["streamed"]] < npan[p["streamed"]]
        assert 1 <= p["first_group"] <= min(group, npan[2])
        assert p["groups"] == 1 + npan[2] - p["first_group"]
        assert nsl[2] >= min(npan[2], 2 * p["first_group"] + 1)
        bigger = bofhip.flash_gemm_panel_plan(ord_, ta, tb, m, n, k, blk, 2 * budget, group=group)
        assert bigger["eligible"] and bigger["need_bytes"] >= p["need_bytes"]

    check()
