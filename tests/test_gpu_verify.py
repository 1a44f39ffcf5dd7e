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
