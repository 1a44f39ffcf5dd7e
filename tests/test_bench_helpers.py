"""CPU-side checks of bench.py's bookkeeping (no GPU): the per-launch algorithmic bytes of the row-panel schedule,
the merging of the I/O ceilings ("the probe moves, never the run") and the raised-ceiling rule of a leg's roofline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_per_launch_of_cfg2():
    n, blk = 32768, 4096
    # the launch mix as the library counts it (bof_flash_last_launch_mix): G = 4 ramp panels x 8 k-blocks, 3 whole-K
    # panels, the last panel in 4 row slices
    mix = {"chain_k_ranges": 32, "whole_k_panels": 3, "whole_k_row_slices": 4}
    per = bench._alg_bytes_per_launch(n, blk, mix, 39.0)
    ramp_first = 4 * (blk * n + blk * n + blk * n)                # A panel, B panel l, C written
    whole = 4 * (blk * n + n * n + blk * n)
    # every launch moves at least its operands once; the average sits between the ramp launch and the whole-K one
    assert ramp_first < per < whole
    # one k-block, one launch: the whole product's SURVEY 8(d) bytes
    assert bench._alg_bytes_per_launch(4096, 4096, {}, 1.0) == 4 * 3 * 4096 * 4096
    # all panels in the ramp (G = 8): 64 launches
    assert bench._alg_bytes_per_launch(n, blk, {"chain_k_ranges": 64, "whole_k_panels": 0, "whole_k_row_slices": 0}, 64.0) < per
    # slices re-read B: the sliced panel's bytes exceed a whole panel's
    unsliced = bench._alg_bytes_per_launch(n, blk, {"chain_k_ranges": 32, "whole_k_panels": 4, "whole_k_row_slices": 0}, 36.0)
    assert per * 39 > unsliced * 36


def test_disk_time_bound_mixes_only_where_mixing_pays():
    R, W = 8 * 2**30, 4 * 2**30
    # this pool's disks: 15.4 / 21 + 3.7 / 16.5 = 0.96 <= 1 -> reads and writes add up
    t, model = bench.disk_time_bound(R, W, {"disk_read_GBps": 21.0, "disk_write_GBps": 16.5,
                                            "disk_read_GBps_while_writing": 15.4, "disk_write_GBps_while_reading": 3.7})
    assert model == "reads + writes" and abs(t - (R / 21e9 + W / 16.5e9)) < 1e-12
    # no mixed probe at all: the same
    assert bench.disk_time_bound(R, W, {"disk_read_GBps": 21.0, "disk_write_GBps": 16.5})[1] == "reads + writes"
    # a disk that gains from mixing (12 / 18 + 8 / 14 = 1.24): mixed while both have bytes, the rest alone
    ceil = {"disk_read_GBps": 18.0, "disk_write_GBps": 14.0, "disk_read_GBps_while_writing": 12.0,
            "disk_write_GBps_while_reading": 8.0}
    t, model = bench.disk_time_bound(R, W, ceil)
    tau = min(R / 12e9, W / 8e9)
    assert model == "mixed, then the rest"
    assert abs(t - (tau + (R - 12e9 * tau) / 18e9 + (W - 8e9 * tau) / 14e9)) < 1e-12
    assert max(R / 18e9, W / 14e9) < t < R / 18e9 + W / 14e9


def test_mixed_window_rates_count_only_the_common_window():
    # reads: 10 chunks, one every 0.1 s from t = 0; writes: 20 chunks, one every 0.2 s: the reads end at 1.0, the writes at
    # 4.0 -- over their own durations 10 and 5 chunks per second, inside the common window (0 .. 1.0) 10 and 5
    rs = [0.0] + [0.1 * (i + 1) for i in range(10)]
    ws = [0.0] + [0.2 * (i + 1) for i in range(5)] + [1.0 + 0.2 * (i + 1) for i in range(15)]      # faster once alone
    out = bench.mixed_window_rates(rs, ws, 10**9, {"r": 10.0, "w": 20 / 4.0})
    assert out["mixed_window_s"] == 1.0
    assert out["disk_read_GBps_while_writing"] == 10.0 and out["disk_write_GBps_while_reading"] == 5.0
    assert out["mixed_pass_own_duration_GBps"] == [10.0, 5.0]
    # passes that barely overlap say nothing about mixing
    assert "disk_read_GBps_while_writing" not in bench.mixed_window_rates([0.0, 0.01], [0.0, 5.0], 10**9, {"r": 1.0, "w": 1.0})


def test_row_panel_disk_bound_is_tighter_than_the_agnostic_one():
    n, blk = 32768, 4096
    ceil = {"disk_read_GBps": 18.0, "disk_write_GBps": 14.0, "disk_read_GBps_while_writing": 12.0,
            "disk_write_GBps_while_reading": 8.0}
    rp = bench.row_panel_disk_bound(n, blk, ceil)
    agnostic = bench.disk_time_bound(8.0 * n * n, 4.0 * n * n, ceil)[0]
    first = 4.0 * (n * n + blk * n)
    assert rp > agnostic                                   # B + one A panel cannot overlap with any write
    assert abs(rp - (first / 18e9 + bench.disk_time_bound(4.0 * (n * n - blk * n), 4.0 * n * n, ceil)[0])) < 1e-9
    # a disk that gains nothing from mixing: both bounds are reads + writes
    flat = {"disk_read_GBps": 21.0, "disk_write_GBps": 16.5, "disk_read_GBps_while_writing": 15.4, "disk_write_GBps_while_reading": 3.7}
    assert abs(bench.row_panel_disk_bound(n, blk, flat) - bench.disk_time_bound(8.0 * n * n, 4.0 * n * n, flat)[0]) < 1e-9
    assert bench.row_panel_disk_bound(n, blk, {}) is None


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
    assert r["frac_raw"] == 1.6          # the unclamped fraction names the mis-probed ceiling
    leg["seconds"] = 1.0
    r = bench.roofline_e2e(leg, ceil, 2e9, 0.001, "odirect")
    assert r["frac"] == 0.8 and "probe_raised" not in r and r["frac_raw"] == 0.8


def test_step_timeline_phases_from_an_event_ring():
    """tools/r6/step_timeline.py on a synthetic ring: 4 chunk reads (two before the first write begins), 2 chunk writes."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "r6"))
    import step_timeline as T
    ev = ["0.000  t1   bof_flash_gemm (panels) begin      8 8 8",
          "1.000  t2   panel chunk read begin             0 0 0", "11.000  t2   panel chunk read end               0 0 0",
          "11.500  t2   panel chunk read begin             0 0 1", "21.000  t2   panel chunk read end               0 0 1",
          "21.500  t3   panel in HBM (ready recorded)      0 0 0", "21.600  t1   launch                             0 1 0",
          "22.000  t2   panel chunk read begin             1 0 0", "32.000  t2   panel chunk read end               1 0 0",
          "30.000  t4   C chunk D2H complete, write begin  0 0 0", "45.000  t4   C chunk write end                  0 0 0",
          "33.000  t2   panel chunk read begin             1 0 1", "43.000  t2   panel chunk read end               1 0 1",
          "46.000  t4   C chunk D2H complete, write begin  0 1 32", "60.000  t4   C chunk write end                  0 1 32"]
    txt = T.summarize(T.parse(ev), 0.060, win=20.0, chunk=1 << 20)
    assert "4 chunk reads, 2 chunk writes, 1 launches" in txt
    assert "phase 1, reads alone (0-30 ms)" in txt and "phase 3, writes alone (43-60 ms)" in txt
    assert "A0@22" in txt and "0:1@22" in txt
