"""-m gpu: BOF_VERIFY (hand-over checksums, include/bof_hip.h "Instrumentation") on the real device: clean calls of
both level-3 GEMM paths pass every comparison and still produce the oracle's C bit for bit; a word damaged between
two hand-over points ($BOF_VERIFY_INJECT, a self-test hook of the instrumentation) fails the call with BOF_EVERIFY
and the message names the two points.  The CPU twin of this file is tests/test_verify_mock.py."""
import os

import numpy as np
import pytest

import bofhip
import orc

pytestmark = pytest.mark.gpu


def run(tmp_path, path, beta, inject=0, devices=None, budget=0, kmeans=False, direct=False):
    m, n, k, blk = 640, 600, 500, 128
    rng = np.random.default_rng(9)
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    want = orc.flash_gemm("R", "N", "N", m, n, k, 0.5, beta, a, b, c0.copy(), 0, 0, 0, blk)
    paths = [str(tmp_path / x) for x in "ABC"]
    for x, p in zip((a, b, c0), paths):
        x.tofile(p)
    fds = [os.open(p, os.O_RDWR | (os.O_DIRECT if direct else 0)) for p in paths]
    os.environ["BOF_VERIFY_INJECT"] = str(inject)
    err = None
    try:
        kw = dict(gemm_blk=blk, gemm_path=path, use_odirect=1 if direct else 0, io_chunk_mib=1, verify=1, hbm_budget=budget)
        if devices:
            kw["devices"] = devices
        bofhip.flash_gemm("R", "N", "N", m, n, k, 0.5, beta, bofhip.FPtr(fds[0], 0), bofhip.FPtr(fds[1], 0), bofhip.FPtr(fds[2], 0),
                          0, 0, 0, bofhip.default_options(**kw))
    except bofhip.BofError as e:
        err = str(e)
    finally:
        os.environ.pop("BOF_VERIFY_INJECT", None)
        for fd in fds:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
    got = np.fromfile(paths[2], np.float32).reshape(m, n)
    return err, bofhip.flash_last_stats(), bool(np.array_equal(got, want))


@pytest.mark.parametrize("path,budget,devices", [(2, 0, None), (2, 0, [0, 0, 0]), (1, 0, None), (1, 14 * 256 * 256 * 4, None),
                                                 (1, 0, [0, 0, 0])])
@pytest.mark.parametrize("beta", [0.0, 2.0])
def test_verify_clean_calls(dev, tmp_path, path, budget, devices, beta):
    err, st, exact = run(tmp_path, path, beta, devices=devices, budget=budget)
    assert err is None, err
    assert exact
    assert st["verify_checks"] >= 30, st


def test_verify_clean_call_odirect(dev, tmp_path):
    err, st, exact = run(tmp_path, 0, 0.0, direct=True)
    assert err is None and exact and st["verify_checks"] >= 30


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("inject,needle", [(1, "after the file read vs"), (2, "after D2H vs the file after the write"),
                                           (3, "64 sampled outputs recomputed vs stored")])
def test_verify_names_a_damaged_word(dev, tmp_path, path, inject, needle, capfd):
    err, st, exact = run(tmp_path, path, 0.0, inject=inject)
    assert err is not None and "BOF_VERIFY mismatch" in err and needle in err, err
    assert "rc=-6" in err or "-6" in err          # BOF_EVERIFY
    assert not exact                              # the damage was real: C differs from the oracle's
    assert "[bof events]" in capfd.readouterr().err


def run_csrmm(tmp_path, ord_b, k, beta, inject=0, devices=None, n_streams=0):
    """flash::csrmm on files with the launch receipts of round 6 on (bof_options.verify): every workgroup of every
    csrmm launch must have run exactly once (profiles/r6/incident_csrmm)."""
    m, n = 2500, 1500
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    b = orc.dense_fill(n, k, "s")
    c0 = np.random.default_rng(k).integers(0, 5, (m, k)).astype(np.float32)
    if ord_b == "C":
        b, c0 = np.ascontiguousarray(b.T), np.ascontiguousarray(c0.T)
    want = orc.flash_csrmm(ord_b, m, n, k, 0.5, beta, val, ia, ja, b, c0.copy(), 300, 5000, 1024)
    arrs = dict(val=val, ia=ia, ja=ja, b=b, c=c0)
    paths = {nm: str(tmp_path / nm) for nm in arrs}
    for nm, x in arrs.items():
        x.tofile(paths[nm])
    fds = {nm: os.open(p, os.O_RDWR) for nm, p in paths.items()}
    os.environ["BOF_VERIFY_INJECT"] = str(inject)
    err = None
    try:
        kw = dict(max_nnzs=5000, csrmm_rblk=300, n_io_threads=2, use_odirect=0, verify=1)
        if devices:
            kw["devices"] = devices
        if n_streams:
            kw["n_streams"] = n_streams
        bofhip.flash_csrmm("N", m, n, k, 0.5, beta, *(bofhip.FPtr(fds[x], 0) for x in ("val", "ia", "ja")), ord_b,
                           bofhip.FPtr(fds["b"], 0), bofhip.FPtr(fds["c"], 0), bofhip.default_options(**kw))
    except bofhip.BofError as e:
        err = str(e)
    finally:
        os.environ.pop("BOF_VERIFY_INJECT", None)
        for fd in fds.values():
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
    got = np.fromfile(paths["c"], np.float32).reshape(c0.shape)
    return err, bofhip.flash_last_stats(), bool(np.array_equal(got, want))


@pytest.mark.parametrize("ord_b,k,devices,n_streams", [("R", 128, None, 0), ("R", 200, [0, 0], 3), ("R", 37, None, 1),
                                                       ("C", 72, None, 0)])
@pytest.mark.parametrize("beta", [0.0, 2.0])
def test_csrmm_launch_receipts_clean(dev, tmp_path, ord_b, k, devices, n_streams, beta):
    err, st, exact = run_csrmm(tmp_path, ord_b, k, beta, devices=devices, n_streams=n_streams)
    assert err is None, err
    assert exact
    assert st["verify_checks"] >= 9, st              # one receipt per csrmm launch: >= one per row block


def test_csrmm_launch_receipts_catch_a_launch_that_ran_twice(dev, tmp_path):
    """$BOF_VERIFY_INJECT=4: the call's first csrmm launch is submitted twice -- what the workgroups of one XCD did in
    the incident's dumps, for a whole launch; beta != 0, so the second pass also damages C."""
    err, st, exact = run_csrmm(tmp_path, "R", 128, 2.0, inject=4)
    assert err is not None and "workgroup receipts" in err and ("-6" in err), err
    assert not exact


def run_csrgemv(tmp_path, trans, inject=0, devices=None):
    """flash::csrgemv on files with the launch receipts on: y against the oracle's fmaf chain ('N') / exact integer
    sums ('T')."""
    m, n = 5000, 3000
    val, ja, ia = orc.sparse_create(m, n, 0.01)
    x = (np.arange(n if trans == "N" else m) % 10).astype(np.float32)
    y = np.full(m if trans == "N" else n, 3.0, np.float32)
    dense = np.zeros((m, n), np.float64)
    for i in range(m):
        dense[i, ja[ia[i]:ia[i + 1]]] = val[ia[i]:ia[i + 1]]
    want = (dense @ x if trans == "N" else dense.T @ x).astype(np.float32)       # integer data: exact in any order
    arrs = dict(val=val, ia=ia, ja=ja)
    paths = {nm: str(tmp_path / nm) for nm in arrs}
    for nm, a in arrs.items():
        a.tofile(paths[nm])
    fds = {nm: os.open(p, os.O_RDWR) for nm, p in paths.items()}
    os.environ["BOF_VERIFY_INJECT"] = str(inject)
    err = None
    try:
        kw = dict(max_nnzs=5000, csrmm_rblk=700, n_io_threads=2, use_odirect=0, verify=1)
        if devices:
            kw["devices"] = devices
        bofhip.flash_csrgemv(trans, m, n, *(bofhip.FPtr(fds[q], 0) for q in ("val", "ia", "ja")), x.ctypes.data, y.ctypes.data,
                             bofhip.default_options(**kw))
    except bofhip.BofError as e:
        err = str(e)
    finally:
        os.environ.pop("BOF_VERIFY_INJECT", None)
        for fd in fds.values():
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
    return err, bofhip.flash_last_stats(), bool(np.array_equal(y, want))


@pytest.mark.parametrize("trans", ["N", "T"])
@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_csrgemv_launch_receipts_clean(dev, tmp_path, trans, devices):
    err, st, exact = run_csrgemv(tmp_path, trans, devices=devices)
    assert err is None, err
    assert exact
    assert st["verify_checks"] >= 8, st              # one receipt per csrgemv launch = per row block


@pytest.mark.parametrize("trans", ["N", "T"])
def test_csrgemv_launch_receipts_catch_a_launch_that_ran_twice(dev, tmp_path, trans):
    """$BOF_VERIFY_INJECT=4: the call's first launch is submitted twice.  'T' adds that block's products twice (a wrong
    y); 'N' stores the same values again -- only the receipt can tell."""
    err, st, exact = run_csrgemv(tmp_path, trans, inject=4)
    assert err is not None and "flash csrgemv" in err and "workgroup receipts" in err and "-6" in err, err
