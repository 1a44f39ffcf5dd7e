"""CPU-side checks of bench.py's bookkeeping (no GPU): the per-launch algorithmic bytes of the row-panel schedule,
the merging of the I/O ceilings ("the probe moves, never the run") and the raised-ceiling rule of a leg's roofline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_per_launch_of_cfg2():
    n, blk = 32768, 4096
    # G = 4 ramp panels x 8 k-blocks + 4 whole-K launches
    per = bench._alg_bytes_per_launch(n, blk, 36.0)
    ramp_first = 4 * (blk * n + blk * n + blk * n)                # A panel, B panel l, C written
    whole = 4 * (blk * n + n * n + blk * n)
    # every launch moves at least its operands once; the average sits between the ramp launch and the whole-K one
    assert ramp_first < per < whole
    # one k-block, one launch: the whole product's SURVEY 8(d) bytes
    assert bench._alg_bytes_per_launch(4096, 4096, 1.0) == 4 * 3 * 4096 * 4096
    # all panels in the ramp (G = 8): 64 launches
    assert bench._alg_bytes_per_launch(n, blk, 64.0) < per


def test_merge_ceilings_keeps_the_best_probe():
    base = {"disk_read_GBps": 14.0, "disk_write_GBps": 10.0, "disk_read_GBps_while_writing": 9.0,
            "disk_write_GBps_while_reading": 5.0, "pcie_h2d_GBps": 56.0}
    before = {"disk_read_GBps": 21.0, "disk_write_GBps": 12.0, "disk_read_GBps_while_writing": 11.0,
              "disk_write_GBps_while_reading": 8.0}
    after = {"disk_read_GBps": 20.0, "disk_write_GBps": 13.0, "disk_read_GBps_while_writing": 10.0,
             "disk_write_GBps_while_reading": 7.0}
    m = bench.merge_ceilings(base, before, after, {})
    assert (m["disk_read_GBps"], m["disk_write_GBps"]) == (21.0, 13.0)
    assert (m["disk_read_GBps_while_writing"], m["disk_write_GBps_while_reading"]) == (11.0, 8.0)   # the best PAIR, kept together
    assert m["pcie_h2d_GBps"] == 56.0 and base["disk_read_GBps"] == 14.0                           # input untouched


def test_a_leg_that_beats_its_bound_raises_the_ceiling_and_says_so():
    ceil = {"disk_read_GBps": 10.0, "disk_write_GBps": 10.0, "pcie_h2d_GBps": 50.0, "pcie_d2h_GBps": 50.0}
    leg = {"seconds": 0.5, "gflops": 1.0, "stats": {"bytes_read": 8e9, "bytes_written": 0, "bytes_h2d": 8e9, "bytes_d2h": 0}}
    r = bench.roofline_e2e(leg, ceil, 2e9, 0.001, "odirect")
    assert r["bound"] == "disk_read" and r["frac"] == 1.0 and "probe_raised" in r and r["t_bound_s"] == 0.5
    leg["seconds"] = 1.0
    r = bench.roofline_e2e(leg, ceil, 2e9, 0.001, "odirect")
    assert r["frac"] == 0.8 and "probe_raised" not in r
