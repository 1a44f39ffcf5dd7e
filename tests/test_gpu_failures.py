"""-m gpu: failure injection on level 3.  The reference retries an I/O five times and then dies loudly
(src/file_handles/flash_file_handle.cpp:28-76: GLOG_FATAL -> exit(-1)).  The C ABI instead returns
BOF_EIO / BOF_ENOMEM / BOF_EINVAL with a message (the C++ veneer turns that into the reference's
fatal exit) -- and it must RETURN: every reader / writer / flusher / dispatcher thread woken and
joined, no wait left without its wake-up, whatever stage failed.  Every case runs under a 60 s
timeout, checks the error code and text, that the process has as many threads afterwards as
before, and that the next call in the same process succeeds.  Row-panel path, tile cache, the
in-process multi-device form of both, and the CSR pipeline."""
import os

import numpy as np
import pytest

import bofhip
import orc
from test_gpu_flash import Files

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(60)]

M = K = N = 512
BLK = 128


def n_threads():
    return len(os.listdir("/proc/self/task"))


def gemm_inputs(seed=1):
    rng = np.random.default_rng(seed)
    return tuple(rng.uniform(-1, 1, (512, 512)).astype(np.float32) for _ in range(3))


def gemm_opts(path, devices=None, **kw):
    base = dict(gemm_blk=BLK, n_streams=2, n_io_threads=4, pinned_slots=4, gemm_path=path, io_chunk_mib=1)
    base.update(kw)
    if devices:
        base["devices"] = devices
    return bofhip.default_options(**base)


def good_call(tmp_path, path, devices=None):
    """a healthy call right after the failure: same process, same cached rings / slots"""
    a, b, c0 = gemm_inputs(2)
    sub = tmp_path / "after"
    sub.mkdir(exist_ok=True)
    F = Files(sub, a=a, b=b, c=c0)
    try:
        bofhip.flash_gemm("R", "N", "N", M, N, K, 0.5, 2.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                          gemm_opts(path, devices))
        ref = orc.flash_gemm("R", "N", "N", M, N, K, 0.5, 2.0, a, b, c0.copy(), 0, 0, 0, BLK)
        assert np.array_equal(F.read("c", np.float32, (M, N)), ref)
    finally:
        F.close()


def _settled_threads(ceiling):
    # (a joined thread's /proc entry can outlive pthread_join by a moment: the kernel wakes the joiner
    # before it releases the task)
    import time
    t_end = time.time() + 2.0
    while n_threads() > ceiling and time.time() < t_end:
        time.sleep(0.01)
    return n_threads()


def expect_failure(call, code, text):
    """The call must fail with `code` / `text` -- twice, and the second failure must not leave more threads
    behind than the first: a pipeline thread that is not joined on the error path would add one per call (the
    HIP runtime may start a helper thread of its own the first time a path is taken, which is why the first
    call is not compared with the count before it)."""
    counts = []
    for _ in range(2):
        with pytest.raises(bofhip.BofError) as ei:
            call()
        msg = str(ei.value)
        assert f"rc={code}" in msg, msg
        assert text.lower() in msg.lower(), msg
        counts.append(_settled_threads(counts[0] if counts else 0))
    assert counts[1] <= counts[0], f"threads of the failed call are still alive: {counts}"


CASES = [(1, None), (2, None), (1, [0, 0]), (2, [0, 0, 0])]


@pytest.mark.parametrize("path,devices", CASES)
@pytest.mark.parametrize("direct", [True, False])
def test_gemm_truncated_operand(dev, tmp_path, path, devices, direct):
    """A (or B) ends in the middle of a panel: the reader's request comes back short."""
    good_call(tmp_path, path, devices)      # the runtime's own helper threads exist before threads are counted
    a, b, c0 = gemm_inputs()
    for victim in ("a", "b"):
        sub = tmp_path / victim
        sub.mkdir()
        F = Files(sub, direct=direct, a=a, b=b, c=c0)
        try:
            os.truncate(F.paths[victim], (M * K * 4) // 2 + 4096)
            expect_failure(lambda: bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), F.fptr("b"),
                                                     F.fptr("c"), 0, 0, 0, gemm_opts(path, devices, use_odirect=int(direct))),
                           -3, "I/O pipeline failed")
        finally:
            F.close()
    good_call(tmp_path, path, devices)


@pytest.mark.parametrize("path,devices", CASES)
def test_gemm_c_read_only_and_no_space(dev, tmp_path, path, devices):
    """C opened O_RDONLY: every write-back request fails (EBADF).  C = /dev/full: every write-back request
    fails with ENOSPC, the full-disk case of a write-back.  beta = 0, so nothing is read from C."""
    good_call(tmp_path, path, devices)
    a, b, c0 = gemm_inputs()
    F = Files(tmp_path, a=a, b=b, c=c0)
    ro = os.open(F.paths["c"], os.O_RDONLY)
    full = os.open("/dev/full", os.O_RDWR)
    try:
        expect_failure(lambda: bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), F.fptr("b"),
                                                 bofhip.FPtr(ro, 0), 0, 0, 0, gemm_opts(path, devices)),
                       -3, "Bad file descriptor")
        assert np.array_equal(F.read("c", np.float32, (M, N)), c0)          # untouched
        expect_failure(lambda: bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), F.fptr("b"),
                                                 bofhip.FPtr(full, 0), 0, 0, 0, gemm_opts(path, devices)),
                       -3, "No space left on device")
    finally:
        bofhip.lib().bof_file_forget(ro)
        os.close(ro)
        os.close(full)
        F.close()
    good_call(tmp_path, path, devices)


@pytest.mark.parametrize("path,devices", CASES)
def test_gemm_closed_descriptor(dev, tmp_path, path, devices):
    good_call(tmp_path, path, devices)
    a, b, c0 = gemm_inputs()
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        dead = os.open(F.paths["b"], os.O_RDWR)
        os.close(dead)
        expect_failure(lambda: bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), bofhip.FPtr(dead, 0),
                                                 F.fptr("c"), 0, 0, 0, gemm_opts(path, devices)), -3, "Bad file descriptor")
        with pytest.raises(bofhip.BofError, match="rc=-1"):
            bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), bofhip.FPtr(-1, 0), F.fptr("c"), 0, 0, 0,
                              gemm_opts(path, devices))
    finally:
        F.close()
    good_call(tmp_path, path, devices)


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_gemm_budget_too_small(dev, tmp_path, devices):
    """hbm_budget below one panel (panel path demanded: BOF_ENOMEM, nothing started) and below six tile
    slots (tile cache: BOF_ENOMEM); with gemm_path = 0 the first falls back to the second."""
    good_call(tmp_path, 0, devices)
    a, b, c0 = gemm_inputs()
    F = Files(tmp_path, a=a, b=b, c=c0)
    try:
        one_panel = BLK * N * 4
        expect_failure(lambda: bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"),
                                                 0, 0, 0, gemm_opts(2, devices, hbm_budget=one_panel // 2)), -5, "not eligible")
        expect_failure(lambda: bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"),
                                                 0, 0, 0, gemm_opts(0, devices, hbm_budget=3 * BLK * BLK * 4)), -5, "tile slots")
        assert np.array_equal(F.read("c", np.float32, (M, N)), c0)
    finally:
        F.close()
    good_call(tmp_path, 0, devices)


def csr_inputs():
    m, n, k = 4096, 2048, 128
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    b = orc.dense_fill(n, k, "s")
    return m, n, k, val, ja, ia, b


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_csr_failures(dev, tmp_path, golden, devices):
    import hashlib
    m, n, k, val, ja, ia, b = csr_inputs()
    c0 = np.zeros((m, k), np.float32)
    kw = dict(max_nnzs=5000, csrmm_rblk=1000, n_io_threads=4)
    if devices:
        kw["devices"] = devices
    opts = bofhip.default_options(**kw)

    def csrmm(F, fc=None):
        bofhip.flash_csrmm("N", m, n, k, 1.0, 0.0, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), "R", F.fptr("b"),
                           fc if fc is not None else F.fptr("c"), opts)
    warm = tmp_path / "warm"
    warm.mkdir()
    W = Files(warm, val=val, ja=ja, ia=ia, b=b, c=c0)
    try:
        csrmm(W)          # the runtime's own helper threads exist before threads are counted
    finally:
        W.close()
    for victim, text in (("ja", "I/O pipeline failed"), ("val", "I/O pipeline failed"), ("ia", "reading ia failed"),
                         ("b", "failed")):
        sub = tmp_path / victim
        sub.mkdir()
        F = Files(sub, val=val, ja=ja, ia=ia, b=b, c=c0)
        try:
            os.truncate(F.paths[victim], os.path.getsize(F.paths[victim]) // 2 // 512 * 512)
            expect_failure(lambda: csrmm(F), -3, text)
        finally:
            F.close()
    F = Files(tmp_path, val=val, ja=ja, ia=ia, b=b, c=c0)
    ro = os.open(F.paths["c"], os.O_RDONLY)
    full = os.open("/dev/full", os.O_RDWR)
    try:
        expect_failure(lambda: csrmm(F, bofhip.FPtr(ro, 0)), -3, "Bad file descriptor")
        expect_failure(lambda: csrmm(F, bofhip.FPtr(full, 0)), -3, "No space left on device")
        # csrgemv with a truncated index file, then healthy calls of both in the same process
        x = (np.arange(n) % 10).astype(np.float32)
        y = np.zeros(m, np.float32)
        os.truncate(F.paths["ja"], ja.nbytes // 2)
        expect_failure(lambda: bofhip.flash_csrgemv("N", m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), x.ctypes.data,
                                                    y.ctypes.data, opts), -3, "I/O pipeline failed")
        ja.tofile(F.paths["ja"])
        csrmm(F)
        want = {t.split()[1]: t.split()[2] for t in golden["meta"] if t.startswith("exact")}
        assert hashlib.sha256(F.read("c", np.float32, (m, k)).tobytes()).hexdigest() == want["gen_csrmm_c"]
        bofhip.flash_csrgemv("N", m, n, F.fptr("val"), F.fptr("ia"), F.fptr("ja"), x.ctypes.data, y.ctypes.data, opts)
        assert hashlib.sha256(y.tobytes()).hexdigest() == want["gen_csrgemv_N"]
    finally:
        bofhip.lib().bof_file_forget(ro)
        os.close(ro)
        os.close(full)
        F.close()


@pytest.mark.parametrize("budget_mib", [0, 4])
def test_csrcsc_output_no_space(dev, tmp_path, golden_tr, budget_mib):
    """flash::csrcsc whose output files cannot take the data (ENOSPC on the write of A^T), resident
    transposition and the out-of-core one (row blocks spilled to temporary files, column-block merge)."""
    m, n = 4096, 2048
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    F = Files(tmp_path, val=val, ja=ja, ia=ia, vt=np.zeros_like(val), jt=np.zeros_like(ja), it=np.zeros(n + 1, np.int64))
    full = os.open("/dev/full", os.O_RDWR)
    try:
        opts = bofhip.default_options(hbm_budget=budget_mib << 20, use_odirect=0)
        expect_failure(lambda: bofhip.flash_csrcsc(m, n, F.fptr("ia"), F.fptr("ja"), F.fptr("val"), F.fptr("it"),
                                                   F.fptr("jt"), bofhip.FPtr(full, 0), opts), -3, "No space left")
        bofhip.flash_csrcsc(m, n, F.fptr("ia"), F.fptr("ja"), F.fptr("val"), F.fptr("it"), F.fptr("jt"), F.fptr("vt"), opts)
        vt, _, jt = orc.csrcsc(m, n, val, ia, ja)
        assert np.array_equal(F.read("vt", np.float32, (-1,)), vt) and np.array_equal(F.read("jt", np.int64, (-1,)), jt)
    finally:
        os.close(full)
        F.close()


def test_stall_watchdog_fails_a_call_that_stops_moving(dev, tmp_path, monkeypatch, capfd):
    """A call that stops making progress must not wait forever: rank 0 of a two-rank shared-B call whose
    peer never shows up parks its reader on panel 1 of B (the peer's to publish).  The peer time-out is set
    far away (60 s); the stall watchdog (BOF_STALL_TIMEOUT_S = 2) notices that no byte moves and no task
    starts, says so on stderr and fails the call -- BOF_EIO with the time-out text, all threads joined, next
    call fine."""
    good_call(tmp_path, 2)
    monkeypatch.setenv("BOF_SHARE_TIMEOUT_S", "60")
    monkeypatch.setenv("BOF_STALL_TIMEOUT_S", "2")
    a, b, c0 = gemm_inputs()
    F = Files(tmp_path, a=a, b=b, c=c0)
    name = f"/bof_test_stall_{os.getpid()}"
    try:
        import time
        t0 = time.time()
        def call():
            bofhip.lib().bof_share_cleanup(name.encode())       # the failed attempt left its "gave up" marks in the ring
            bofhip.flash_gemm("R", "N", "N", M, N, K, 1.0, 0.0, F.fptr("a"), F.fptr("b"), F.fptr("c"), 0, 0, 0,
                              gemm_opts(2, share_world=2, share_rank=0, share_name=name))
        expect_failure(call, -3, "timed out")
        assert time.time() - t0 < 30
        assert "no progress for" in capfd.readouterr().err
    finally:
        bofhip.lib().bof_share_cleanup(name.encode())
        F.close()
    monkeypatch.delenv("BOF_STALL_TIMEOUT_S")
    good_call(tmp_path, 2)

