"""BOF_VERIFY (hand-over checksums of the level-3 GEMM pipelines, include/bof_hip.h "Instrumentation") on the
CPU box: the product's host code linked against the mock HIP runtime (tests/native/mock_hip.cpp, asynchronous
streams).  A clean call compares every panel / tile at every hand-over and passes; a word damaged between two
hand-over points ($BOF_VERIFY_INJECT) fails the call with BOF_EVERIFY and names the pair of points."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
from test_dist_gloo import _use_mock_library
_use_mock_library(SO)
import bofhip
m, n, k, blk = 640, 600, 500, 128
rng = np.random.default_rng(3)
a = rng.integers(-3, 4, (m, k)).astype(np.float32); b = rng.integers(-3, 4, (k, n)).astype(np.float32)
c0 = rng.integers(-3, 4, (m, n)).astype(np.float32)
for name, x in (("A", a), ("B", b), ("C", c0)): x.tofile(os.path.join(DIR, name))
fds = [os.open(os.path.join(DIR, x), os.O_RDWR) for x in "ABC"]
out = {}
try:
    bofhip.flash_gemm("R", "N", "N", m, n, k, 1.0, BETA, bofhip.FPtr(fds[0], 0), bofhip.FPtr(fds[1], 0), bofhip.FPtr(fds[2], 0),
                      0, 0, 0, bofhip.default_options(gemm_blk=blk, gemm_path=PATH, use_odirect=0, io_chunk_mib=1, verify=VERIFY,
                                                      hbm_budget=BUDGET, devices=DEVS))
    out["rc"] = 0
except bofhip.BofError as e:
    out["rc"] = 1; out["err"] = str(e)
out["stats"] = bofhip.flash_last_stats()
got = np.fromfile(os.path.join(DIR, "C"), np.float32).reshape(m, n)
out["exact"] = bool(np.array_equal(got, (a.astype(np.float64) @ b.astype(np.float64) + BETA * c0).astype(np.float32)))
print("RESULT " + json.dumps(out), flush=True)
'''


@pytest.fixture(scope="module")
def mock_lib(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dist_gloo import _build_mock_library
    return _build_mock_library(str(tmp_path_factory.mktemp("mocklib_verify")))


def run_child(tmp_path, so, path, beta, budget, devs, inject, verify=1, extra_env=None):
    import json
    d = tmp_path / f"p{path}_i{inject}_{len(devs)}_{budget}_{verify}_{len(extra_env or {})}"
    d.mkdir()
    code = (f"ROOT={ROOT!r}\nSO={so!r}\nDIR={str(d)!r}\nPATH={path}\nBETA={beta}\nBUDGET={budget}\nDEVS={devs!r}\nVERIFY={verify}\n" + CHILD)
    env = dict(os.environ, MOCK_HIP_DEVICES="4", MOCK_HIP_ASYNC="1", MOCK_HIP_JITTER_US="100", BOF_VERIFY_INJECT=str(inject))
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(line[-1][7:]), r.stderr


@pytest.mark.parametrize("path,budget,devs", [(2, 0, [0]), (2, 0, [0, 1, 2]), (1, 0, [1]), (1, 14 * 256 * 256 * 4, [0]),
                                              (1, 0, [0, 0, 0])])
@pytest.mark.parametrize("beta", [0.0, 2.0])
def test_clean_call_passes_every_hand_over_check(tmp_path, mock_lib, path, budget, devs, beta):
    out, err = run_child(tmp_path, mock_lib, path, beta, budget, devs, 0)
    assert out["rc"] == 0, out.get("err", "") + err[-2000:]
    assert out["exact"]
    # A and B panels / tiles in (x2 or x3 points each), every C panel / tile out (3 points)
    assert out["stats"]["verify_checks"] >= 30, out["stats"]


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("inject,needle", [(1, "after the file read vs"), (2, "after D2H vs the file after the write"),
                                           (3, "64 sampled outputs recomputed vs stored")])
def test_damaged_word_is_caught_and_named(tmp_path, mock_lib, path, inject, needle):
    out, err = run_child(tmp_path, mock_lib, path, 0.0, 0, [0], inject)
    assert out["rc"] == 1, "a damaged word / a dropped launch went unnoticed"
    assert "BOF_VERIFY mismatch" in out["err"] and needle in out["err"], out["err"]
    assert "[bof events]" in err            # the event ring came with it


K_ZERO_CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
from test_dist_gloo import _use_mock_library
_use_mock_library(SO)
import bofhip
m, n, blk = 300, 280, 128
rng = np.random.default_rng(4)
c0 = rng.integers(-3, 4, (m, n)).astype(np.float32)
c0[2, 3] = np.nan
for name, x in (("A", np.zeros(8, np.float32)), ("B", np.zeros(8, np.float32)), ("C", c0)): x.tofile(os.path.join(DIR, name))
fds = [os.open(os.path.join(DIR, x), os.O_RDWR) for x in "ABC"]
out = {"rc": []}
for path in (0, 1, 2):
    for beta in (0.0, 2.0):
        try:
            bofhip.flash_gemm("R", "N", "N", m, n, 0, 1.5, beta, bofhip.FPtr(fds[0], 0), bofhip.FPtr(fds[1], 0),
                              bofhip.FPtr(fds[2], 0), 0, 0, 0, bofhip.default_options(gemm_blk=blk, gemm_path=path, use_odirect=0))
            st = bofhip.flash_last_stats()
            out["rc"].append([0, st["bytes_written"], st["bytes_read"], st["tasks"]])
        except bofhip.BofError as e:
            out["rc"].append([1, str(e)])
got = np.fromfile(os.path.join(DIR, "C"), np.uint32)
out["untouched"] = bool(np.array_equal(got, c0.view(np.uint32).ravel()))
print("RESULT " + json.dumps(out), flush=True)
'''


def test_k_zero_on_files_returns_ok_and_leaves_c_alone(tmp_path, mock_lib):
    """flash::gemm with k == 0: the reference's tiler has zero k-blocks, creates no task and returns 0 -- C is not
    touched, not even by beta (src/blas/gemm.cpp:69-75, 83-129, 176-200); both paths and the chooser."""
    import json
    code = f"ROOT={ROOT!r}\nSO={mock_lib!r}\nDIR={str(tmp_path)!r}\n" + K_ZERO_CHILD
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MOCK_HIP_DEVICES="1"))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(line[-1][7:])
    assert out["rc"] == [[0, 0, 0, 0]] * 6, out
    assert out["untouched"]


@pytest.mark.parametrize("devs", [[0], [0, 1, 2]])
@pytest.mark.parametrize("beta", [0.0, 2.0])
@pytest.mark.parametrize("verify", [2, 1])
@pytest.mark.parametrize("every_panel", ["1", "0"])
def test_whole_k_panels_in_row_slices(tmp_path, mock_lib, devs, beta, verify, every_panel):
    """Round 6: the LAST C panel of a slab ($BOF_PANEL_SLICES_ALL=1: every C panel that runs as one launch over the
    whole K) is multiplied in row slices, each copied out and written while the next is still being multiplied
    ($BOF_PANEL_SLICES, here 3 slices of multiples of 32 rows on 128-row panels).  Asynchronous, jittered mock streams: a chunk that left before its slice's launch had run would
    carry the old C.  With the hand-over checks off (the flusher then waits slice by slice) and on."""
    out, err = run_child(tmp_path, mock_lib, 2, beta, 0, devs, 0, verify=verify,
                         extra_env={"BOF_PANEL_SLICES": "3", "BOF_PANEL_SLICE_ROWS": "32", "BOF_PANEL_GROUP": "1",
                                    "BOF_PANEL_SLICES_ALL": every_panel})
    assert out["rc"] == 0, out.get("err", "") + err[-2000:]
    assert out["exact"]
    assert out["stats"]["tasks"] == 5 * 4 * 3          # 640 x 600 x 500 in 128-tiles (merged tails): a slice is not a task


CSRMM_CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
from test_dist_gloo import _use_mock_library
_use_mock_library(SO)
import bofhip
m, n, k = 700, 400, 24
rng = np.random.default_rng(5)
dense = (rng.random((m, n)) < 0.02) * rng.integers(1, 4, (m, n))
ia = np.concatenate([[0], np.cumsum((dense != 0).sum(1))]).astype(np.int64)
ja = np.nonzero(dense)[1].astype(np.int64); val = dense[dense != 0].astype(np.float32)
b = rng.integers(-3, 4, (n, k)).astype(np.float32); c0 = rng.integers(-3, 4, (m, k)).astype(np.float32)
arrs = dict(val=val, ia=ia, ja=ja, b=b, c=c0)
for nm, x in arrs.items(): x.tofile(os.path.join(DIR, nm))
fds = {nm: os.open(os.path.join(DIR, nm), os.O_RDWR) for nm in arrs}
out = {}
try:
    bofhip.flash_csrmm("N", m, n, k, 1.0, 2.0, *(bofhip.FPtr(fds[x], 0) for x in ("val", "ia", "ja")), "R",
                       bofhip.FPtr(fds["b"], 0), bofhip.FPtr(fds["c"], 0),
                       bofhip.default_options(max_nnzs=600, csrmm_rblk=90, n_io_threads=2, use_odirect=0, verify=1, devices=DEVS))
    out["rc"] = 0
except bofhip.BofError as e:
    out["rc"] = 1; out["err"] = str(e)
out["stats"] = bofhip.flash_last_stats()
got = np.fromfile(os.path.join(DIR, "c"), np.float32).reshape(m, k)
out["exact"] = bool(np.array_equal(got, (dense.astype(np.float64) @ b + 2.0 * c0).astype(np.float32)))
print("RESULT " + json.dumps(out), flush=True)
'''


@pytest.mark.parametrize("devs,inject", [([0], 0), ([0, 1, 2], 0), ([0], 4)])
def test_csrmm_launch_receipts(tmp_path, mock_lib, devs, inject):
    """Round 6 (profiles/r6/incident_csrmm): under bof_options.verify every csrmm launch carries a receipt -- each
    workgroup counts itself, a checker behind the launch compares with 1.  $BOF_VERIFY_INJECT=4 submits the call's first
    launch twice (what the workgroups of one XCD did in the incident): BOF_EVERIFY, and with beta != 0 a wrong C."""
    import json
    code = f"ROOT={ROOT!r}\nSO={mock_lib!r}\nDIR={str(tmp_path)!r}\nDEVS={devs!r}\n" + CSRMM_CHILD
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MOCK_HIP_DEVICES="4", MOCK_HIP_ASYNC="1", MOCK_HIP_JITTER_US="100",
                                BOF_VERIFY_INJECT=str(inject), BOF_CRASH_TRACE="1"))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    # (RESULT is flushed as soon as it is known: a crash while the mock runtime's worker threads are torn down at exit
    #  does not cost the verdict; one before it shows its exit code and, with BOF_CRASH_TRACE, its native stack)
    assert line, f"child exit code {r.returncode}\n" + r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(line[-1][7:])
    if inject == 0:
        assert out["rc"] == 0 and out["exact"], out
        assert out["stats"]["verify_checks"] >= 8, out["stats"]
    else:
        assert out["rc"] == 1 and "workgroup receipts" in out["err"], out
        assert not out["exact"]


CSRGEMV_CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
from test_dist_gloo import _use_mock_library
_use_mock_library(SO)
import bofhip
m, n = 900, 500
rng = np.random.default_rng(6)
dense = (rng.random((m, n)) < 0.02) * rng.integers(1, 4, (m, n))
ia = np.concatenate([[0], np.cumsum((dense != 0).sum(1))]).astype(np.int64)
ja = np.nonzero(dense)[1].astype(np.int64); val = dense[dense != 0].astype(np.float32)
arrs = dict(val=val, ia=ia, ja=ja)
for nm, a in arrs.items(): a.tofile(os.path.join(DIR, nm))
fds = {nm: os.open(os.path.join(DIR, nm), os.O_RDWR) for nm in arrs}
out = {}
for trans in ("N", "T"):
    x = rng.integers(-3, 4, n if trans == "N" else m).astype(np.float32)
    y = np.zeros(m if trans == "N" else n, np.float32)
    try:
        bofhip.flash_csrgemv(trans, m, n, *(bofhip.FPtr(fds[q], 0) for q in ("val", "ia", "ja")), x.ctypes.data, y.ctypes.data,
                             bofhip.default_options(max_nnzs=700, csrmm_rblk=90, n_io_threads=2, use_odirect=0, verify=1, devices=DEVS))
        out[trans] = {"rc": 0}
    except bofhip.BofError as e:
        out[trans] = {"rc": 1, "err": str(e)}
    out[trans]["checks"] = bofhip.flash_last_stats()["verify_checks"]
    out[trans]["exact"] = bool(np.array_equal(y, ((dense if trans == "N" else dense.T).astype(np.float64) @ x).astype(np.float32)))
print("RESULT " + json.dumps(out), flush=True)
'''


@pytest.mark.parametrize("devs,inject", [([0], 0), ([0, 1, 2], 0), ([0], 4)])
def test_csrgemv_launch_receipts(tmp_path, mock_lib, devs, inject):
    """The csrgemv twin of test_csrmm_launch_receipts: one receipt per row-block launch of either kernel."""
    import json
    code = f"ROOT={ROOT!r}\nSO={mock_lib!r}\nDIR={str(tmp_path)!r}\nDEVS={devs!r}\n" + CSRGEMV_CHILD
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MOCK_HIP_DEVICES="4", MOCK_HIP_ASYNC="1", MOCK_HIP_JITTER_US="100",
                                BOF_VERIFY_INJECT=str(inject), BOF_CRASH_TRACE="1"))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    # (RESULT is flushed as soon as it is known: a crash while the mock runtime's worker threads are torn down at exit
    #  does not cost the verdict; one before it shows its exit code and, with BOF_CRASH_TRACE, its native stack)
    assert line, f"child exit code {r.returncode}\n" + r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(line[-1][7:])
    for trans in ("N", "T"):
        o = out[trans]
        if inject == 0:
            assert o["rc"] == 0 and o["exact"] and o["checks"] >= 5, out
        else:
            assert o["rc"] == 1 and "flash csrgemv" in o["err"] and "workgroup receipts" in o["err"], out


@pytest.mark.parametrize("devs,beta,flush", [([0], 0.0, "1"), ([0, 1], 2.0, "1"), ([0], 2.0, "0")])
def test_ramp_group_panels_leave_one_by_one(tmp_path, mock_lib, devs, beta, flush):
    """Round 6, second session: every C panel of the ramp group carries an event behind the LAST launch of its chain
    and leaves when that has run ($BOF_PANEL_RAMP_FLUSH=1, the default; 0 = the whole group behind its last kernel).
    Hand-over checks off (with them on the flusher waits for the group), asynchronous jittered mock streams, a ramp
    group of 3 of the 5 C panels, one k-block per launch: a panel that left before its last k-block had been added
    would carry a partial sum into the file."""
    out, err = run_child(tmp_path, mock_lib, 2, beta, 0, devs, 0, verify=2,
                         extra_env={"BOF_PANEL_GROUP": "3", "BOF_PANEL_RAMP_K": "1", "BOF_PANEL_RAMP_FLUSH": flush,
                                    "MOCK_HIP_JITTER_US": "400"})
    assert out["rc"] == 0, out.get("err", "") + err[-2000:]
    assert out["exact"]
