#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native BLAS-on-flash hot path.

Metric (BASELINE.json): GFLOP/s of the out-of-core GEMM, with the roofline fraction of the dominant
kernel and the CPU path timed beside it.

N = 1 (BASELINE.json configs[1]): flash _gemm fp32 32768 x 32768 x 32768, 4096-tile, A / B / C
SSD-resident.  One "step" = ONE bof_flash_gemm call on three 4 GiB files through O_DIRECT
descriptors with the page cache dropped: 8 GiB read, 512 tile tasks (64 accumulate chains of 8),
4 GiB of C written back -- wall clock around the call, write-back included, the way the reference's
driver times it (drivers/gemm.cpp:57-62).  70.37 TFLOP per step.
N > 1 (configs[3] at N = 8): the (8192 N) x 65536 x 65536 product from ONE set of files, C rows
sharded by rank, B read from storage once per node, no data-path collective; per-GPU flops fixed
(weak scaling).

roofline: the dominant kernel's duration is measured live with HIP events around every tile launch
on the stream it is launched on (bof_options.kernel_timing), inside the timed steps.

One process per GPU; launched for N > 1 as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line (< 4 KiB, asserted); everything else goes to bench_detail.json.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 flop/clk
HBM_PEAK_GBS = 8000.0


def _best_of(fn, reps):
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def cpu_baseline(with_csr=True, full_step=False):
    """The reference's CPU arithmetic timed on this box's host cores (path B):
    tools/cpu_baseline.py in a CHILD process that never touches the GPU (its OpenMP runtime and thread
    pinning stay out of this process) -- Intel MKL's cblas_sgemm / mkl_scsrmm / mkl_cspblas_scsrgemv
    through dlopen when an MKL runtime is on the box, else the same operations through torch's
    MKL-linked CPU ops; one thread per physical core, pinned; best of 3 and the spread; bounded
    samples (sizes in `sample`).  kind "port": the routine the reference calls, not its binary."""
    import subprocess
    cmd = ([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py")] + ([] if with_csr else ["--no-csr"])
           + (["--full-step"] if full_step else []))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {"unit": "GFLOP/s", "cores": 0, "kind": "port", "value": None, "sample": "",
                "error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:  # the headline must still be printed
        return {"unit": "GFLOP/s", "cores": 0, "kind": "port", "value": None, "sample": "",
                "error": f"{type(e).__name__}: {str(e)[:200]}"}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of
    this same command (profiles/rNN/bench_gemm_pmc.json; FETCH_SIZE/WRITE_SIZE are collected
    in separate passes and corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot run
    under the counters itself, so it reports the latest committed measurement."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_gemm_pmc.json")))
    if not cands:
        return None, None
    try:
        d = json.load(open(cands[-1]))
        return float(d["hbm_traffic_bytes_per_launch"]), os.path.relpath(cands[-1], ROOT)
    except Exception:
        return None, None


def check_tile_row_float64(torch, a_rows, b, c_rows, k, n, col_chunk=8192):
    """max |C - A B| / max |A B| over a full tile-row of C (rows x n outputs) against a float64
    product (torch.mm in fp64 is only the checker here), column block by column block."""
    rows = a_rows.numel() // k
    a64 = a_rows.view(rows, k).double()
    worst, scale = 0.0, 0.0
    for j0 in range(0, n, col_chunk):
        j1 = min(n, j0 + col_chunk)
        ref = a64 @ b.view(k, n)[:, j0:j1].double()
        got = c_rows.view(rows, n)[:, j0:j1].double()
        worst = max(worst, float((got - ref).abs().max().item()))
        scale = max(scale, float(ref.abs().max().item()))
        del ref, got
    return worst / max(scale, 1e-30)


def resident_gemm_line(bofhip, torch, dev, st, m_local, n, k, row0, blk, streams, steps, label):
    """One more HBM-resident tile-DAG measurement (secondary line): generated inputs, `steps` timed
    passes after one warm-up, first tile-row checked against float64."""
    a = torch.empty(m_local * k, dtype=torch.float32, device=dev)
    b = torch.empty(k * n, dtype=torch.float32, device=dev)
    c = torch.empty(m_local * n, dtype=torch.float32, device=dev)
    bofhip.gen_dense(a.data_ptr(), row0 * k, a.numel(), "u", 1, st)
    bofhip.gen_dense(b.data_ptr(), 0, b.numel(), "u", 2, st)
    opts = bofhip.default_options(gemm_blk=blk, n_streams=streams)
    run = lambda: bofhip.gemm_resident("R", "N", "N", m_local, n, k, 1.0, 0.0, a.data_ptr(), b.data_ptr(),   # noqa: E731
                                       c.data_ptr(), 0, 0, 0, opts, st)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rows = min(m_local, blk)
    rel = check_tile_row_float64(torch, a[:rows * k], b, c[:rows * n], k, n)
    tasks = len(bofhip.gemm_plan("R", "N", "N", m_local, n, k, 0.0, 0, 0, 0, blk)[0])
    del a, b, c
    torch.cuda.empty_cache()
    bofhip.lib().bof_flash_release()
    flops = 2.0 * m_local * n * k
    return {"workload": label, "ms_per_step": round(dt * 1e3, 2), "gflops": round(flops / dt / 1e9, 1),
            "tile_tasks": tasks, "frac_of_mfma_peak": round(flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            "first_tile_row_rel_err_vs_float64": rel}


def csr_pmc():
    """HBM-side bytes of one pass of the CSR kernels from the committed rocprofv3 PMC passes
    (profiles/rNN/kbench_csr_pmc.json: FETCH_SIZE and WRITE_SIZE in KB summed over the launches of
    one pass; fetch x2 is the gfx950 correction of MI355X_MICROARCH.md)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "kbench_csr_pmc.json")))
    if not cands:
        return {}, None
    try:
        d = json.load(open(cands[-1]))
        out = {}
        # ('T' is quoted only from a file that measured the partitioned form level 2 runs; the
        #  round-1 file holds the per-block atomic kernel under the key "csrgemv_t")
        for key, name in (("csrmm_rowmajor", "csrmm"), ("csrgemv_n", "csrgemv_N"),
                          ("csrgemv_t_partitioned", "csrgemv_T")):
            if key in d and "FETCH_SIZE" in d[key] and "WRITE_SIZE" in d[key]:
                f = d[key]["FETCH_SIZE"]["sum_over_launches_of_one_pass"] * 1024.0 * 2.0
                w = d[key]["WRITE_SIZE"]["sum_over_launches_of_one_pass"] * 1024.0
                out[name] = f + w
        return out, os.path.relpath(cands[-1], ROOT)
    except Exception:
        return {}, None


def kmeans_secondary(bofhip, torch, dev, st, streams, blk=4096):
    """flash::kmeans resident: the reference driver's call shape (drivers/kmeans.cpp:37-39,
    'C','T','N', alpha = -2, beta = 0) at 1024 centres x 2^20 points x 256 dimensions -- the squared-distance
    matrix (4 GiB) as ONE kernel per tile, against the same tiles run as KMeansTask's three calls (the tile
    product, then two K = 1 sgemm passes over C), and checked against float64 on a sample of points."""
    import numpy as np
    ncenters, npoints, dim = 1024, 1 << 20, 256
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    centers = torch.randn(ncenters, dim, device=dev, generator=g)
    points = torch.randn(npoints, dim, device=dev, generator=g)
    cl = (centers.double() ** 2).sum(1).float()
    pl = (points.double() ** 2).sum(1).float()
    ones = torch.ones(max(ncenters, blk + 127), device=dev)
    dist = torch.empty(npoints, ncenters, device=dev)     # column-major ncenters x npoints
    opts = bofhip.default_options(n_streams=max(streams, 4), gemm_blk=blk)

    def timed(fn, iters=3):
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def fused():
        bofhip.kmeans_resident("C", "T", "N", ncenters, npoints, dim, -2.0, 0.0, centers.data_ptr(), points.data_ptr(),
                               dist.data_ptr(), dim, dim, ncenters, cl.data_ptr(), pl.data_ptr(), ones.data_ptr(), opts, st)

    ones_n = torch.ones(npoints, device=dev)

    def three_calls():  # the reference task's sequence with whole-matrix K = 1 passes (same C traffic per element)
        bofhip.gemm_resident("C", "T", "N", ncenters, npoints, dim, -2.0, 0.0, centers.data_ptr(), points.data_ptr(),
                             dist.data_ptr(), dim, dim, ncenters, opts, st)
        bofhip.sgemm("C", "N", "T", ncenters, npoints, 1, 1.0, cl.data_ptr(), ncenters, ones_n.data_ptr(), npoints, 1.0,
                     dist.data_ptr(), ncenters, st)
        bofhip.sgemm("C", "N", "T", ncenters, npoints, 1, 1.0, ones_n.data_ptr(), ncenters, pl.data_ptr(), npoints, 1.0,
                     dist.data_ptr(), ncenters, st)

    ms3 = timed(three_calls)
    ref3 = dist[:4096].clone()
    ms = timed(fused)
    same = bool(torch.equal(dist[:4096], ref3))
    sample = slice(0, 2048)
    want = ((points[sample].double()[:, None, :] - centers.double()[None, :, :]) ** 2).sum(2)
    err = float((dist[sample].double() - want).abs().max() / want.abs().max())
    flops = 2.0 * ncenters * npoints * dim
    alg = 4 * (ncenters * dim + npoints * dim + ncenters * npoints + ncenters + npoints)
    return {"workload": "flash kmeans distance matrix: 1024 centres x 2^20 points x 256 dims, column-major, resident in HBM",
            "ms": round(ms, 3), "gflops": round(flops / ms / 1e6, 1),
            "three_call_sequence_ms": round(ms3, 3), "fused_speedup": round(ms3 / ms, 2),
            "fused_equals_three_calls_bitwise": same, "max_rel_err_vs_float64_2048_points": err,
            "roofline": {"bound": "mfma", "achieved": round(flops / ms / 1e9, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(flops / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4),
                         "algorithmic_bytes": alg, "hbm_frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4)}}


def csr_secondary(bofhip, torch, dev, st):
    """Secondary lines of the metric: flash _csrmm at BASELINE configs[2] (10M x 1M CSR, 1e9 nnz,
    x 1M x 128 dense) and _csrgemv at the configs[4] size (50M x 50M, 5e8 nnz), HBM-resident,
    inputs generated in HBM by the reference generators' device restatement."""
    def timed(fn, iters=3):
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    out = {}
    pmc, pmc_src = csr_pmc()
    opts = bofhip.default_options(n_streams=1)
    m, n, k, npr = 10_000_000, 1_000_000, 128, 100
    val = torch.empty(m * npr, dtype=torch.float32, device=dev)
    col = torch.empty(m * npr, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 1_000_000):
        bofhip.gen_sparse_rows(r0, 1_000_000, n, npr, val.data_ptr() + 4 * r0 * npr,
                               col.data_ptr() + 8 * r0 * npr, off.data_ptr() + 8 * r0, st)
    b = torch.empty(n * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(b.data_ptr(), 0, n * k, "u", 3, st)
    c = torch.zeros(m * k, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ia = off.cpu().numpy()
    ms = timed(lambda: bofhip.csrmm_resident("N", m, n, k, 1.0, 0.0, val.data_ptr(), ia.ctypes.data,
                                             off.data_ptr(), col.data_ptr(), "R", b.data_ptr(),
                                             c.data_ptr(), opts, st))
    nnz = m * npr
    alg = nnz * 12 + (m + 1) * 8 + 4 * n * k + 4 * m * k          # BASELINE.md work definition
    out["csrmm"] = {"workload": "flash _csrmm 10M x 1M CSR (1e9 nnz) x 1M x 128, resident in HBM "
                                "(BASELINE configs[2])",
                    "ms": round(ms, 3), "gflops": round(2.0 * nnz * k / ms / 1e6, 1), "row_block_tasks": 100,
                    "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes": alg,
                                 # what the kernel actually moves: one 512-byte B row per non-zero
                                 "gather_bytes": nnz * k * 4, "gather_GBps": round(nnz * k * 4 / ms / 1e6, 1),
                                 "gather_frac_of_hbm_peak": round(nnz * k * 4 / ms / 1e6 / HBM_PEAK_GBS, 4),
                                 # MI355X_MICROARCH.md: random 512-byte rows run at 8.6 TB/s out of the Infinity
                                 # Cache (MALL) and 5.7 TB/s out of HBM; B is 512 MB, 256 MB of it fit the MALL
                                 "gather_frac_of_ceiling": round(nnz * k * 4 / ms / 1e6 / 8600.0, 4),
                                 "gather_ceiling_GBps": 8600.0,
                                 "traffic": pmc.get("csrmm"), "traffic_source": pmc_src,
                                 "traffic_over_algorithmic": round(pmc["csrmm"] / alg, 1) if "csrmm" in pmc else None,
                                 "note": "bytes roofline 3 %: every non-zero fetches a B row from Infinity Cache/HBM "
                                         "(L2 hit rate 2 %); the gather itself runs at the fabric's random-512-B-row rate"}}
    # transposition row on the same matrix: A -> A^T, and csrmm trans_a='T'
    vt = torch.empty_like(val)
    ct = torch.empty_like(col)
    pt = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ms = timed(lambda: bofhip.scsrcsc(m, n, nnz, val.data_ptr(), off.data_ptr(), col.data_ptr(),
                                      vt.data_ptr(), pt.data_ptr(), ct.data_ptr(), st), iters=2)
    alg = nnz * 24 + (m + n + 2) * 8                                # read A once, write A^T once
    out["csrcsc"] = {"workload": "flash csrcsc 10M x 1M CSR (1e9 nnz) -> 1M x 10M, resident in HBM",
                     "ms": round(ms, 3),
                     "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                                  "algorithmic_bytes": alg}}
    del vt, ct, pt
    bt = torch.empty(m * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(bt.data_ptr(), 0, m * k, "u", 5, st)
    ms = timed(lambda: bofhip.csrmm_resident("T", m, n, k, 1.0, 0.0, val.data_ptr(), ia.ctypes.data,
                                             off.data_ptr(), col.data_ptr(), "R", bt.data_ptr(),
                                             c.data_ptr(), opts, st), iters=2)
    out["csrmm_T"] = {"workload": "flash _csrmm trans_a=T: (10M x 1M CSR)^T x 10M x 128, resident in HBM "
                                  "(transposition + 'N' product over A^T)",
                      "ms": round(ms, 3), "gflops": round(2.0 * nnz * k / ms / 1e6, 1)}
    alg_t = nnz * 12 + (m + 1) * 8 + 4 * m * k + 4 * n * k      # A once, B (m x k) once, C (n x k) written once
    out["csrmm_T"]["roofline"] = {"bound": "hbm", "achieved": round(alg_t / ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(alg_t / ms / 1e6 / HBM_PEAK_GBS, 4),
                                  "algorithmic_bytes": alg_t, "gather_bytes": nnz * k * 4,
                                  "gather_GBps": round(nnz * k * 4 / max(ms - out["csrcsc"]["ms"], 1e-3) / 1e6, 1),
                                  "note": "transposition (secondary.csrcsc) + the 'N' kernel over A^T, whose B-row gather "
                                          "(5.12 GB table, beyond the Infinity Cache) is the bound"}
    del val, col, off, b, c, bt
    torch.cuda.empty_cache()
    bofhip.lib().bof_flash_release()      # the 26 GiB sort workspace goes back before cfg5
    m = n = 50_000_000
    npr = 10
    val = torch.empty(m * npr, dtype=torch.float32, device=dev)
    col = torch.empty(m * npr, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 5_000_000):
        bofhip.gen_sparse_rows(r0, 5_000_000, n, npr, val.data_ptr() + 4 * r0 * npr,
                               col.data_ptr() + 8 * r0 * npr, off.data_ptr() + 8 * r0, st)
    x = (torch.arange(n, device=dev) % 10).float()
    y = torch.zeros(m, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ia = off.cpu().numpy()
    nnz = m * npr
    alg = nnz * 12 + (m + 1) * 8 + 8 * n
    for tr in "NT":
        ms = timed(lambda: bofhip.csrgemv_resident(tr, m, n, val.data_ptr(), ia.ctypes.data, off.data_ptr(),
                                                   col.data_ptr(), x.data_ptr(), y.data_ptr(), opts, st))
        out["csrgemv_" + tr] = {"workload": "flash _csrgemv 50M x 50M CSR (5e8 nnz), resident in HBM",
                                "ms": round(ms, 3), "gflops": round(2.0 * nnz / ms / 1e6, 1),
                                "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1),
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                                             "algorithmic_bytes": alg, "traffic": pmc.get("csrgemv_" + tr),
                                             "traffic_source": pmc_src,
                                             "traffic_over_algorithmic": round(pmc["csrgemv_" + tr] / alg, 1)
                                             if ("csrgemv_" + tr) in pmc else None}}
    return out


def csr_secondary_sharded(bofhip, torch, dev, st, rank, n_gpus):
    """N > 1: this rank's row shard of the SAME matrices (strong scaling): CSRMM rows
    [r0, r1) of the 10M x 1M matrix with B replicated, CSRGEMV 'N' rows of the 50M x 50M one with
    x replicated -- no collective; CSRGEMV 'T' yields a full-length partial per rank that the caller
    all-reduces.  Returns this rank's milliseconds; the caller takes the max over ranks."""
    def timed(fn, iters=3):
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    import numpy as np
    res = {}
    opts = bofhip.default_options(n_streams=1)

    def shard(m_total, npr, n):
        per = (m_total // n_gpus + 127) // 128 * 128
        r0 = min(m_total, per * rank)
        r1 = min(m_total, r0 + per) if rank < n_gpus - 1 else m_total
        rows = r1 - r0
        val = torch.empty(max(rows, 1) * npr, dtype=torch.float32, device=dev)
        col = torch.empty(max(rows, 1) * npr, dtype=torch.int64, device=dev)
        off = torch.empty(rows + 1, dtype=torch.int64, device=dev)
        step = 1_000_000
        for q in range(0, rows, step):
            r = min(step, rows - q)
            bofhip.gen_sparse_rows(r0 + q, r, n, npr, val.data_ptr() + 4 * q * npr, col.data_ptr() + 8 * q * npr,
                                   off.data_ptr() + 8 * q, st)
        torch.cuda.synchronize()
        ia = off.cpu().numpy()      # absolute offsets (ia[0] = r0 * npr), as a row shard of the files has
        assert rows == 0 or (ia[0] == r0 * npr and ia[-1] == r1 * npr)
        base = r0 * npr             # element 0 of the virtual full arrays
        return r0, rows, val, col, off, ia, base

    m, n, k, npr = 10_000_000, 1_000_000, 128, 100
    r0, rows, val, col, off, ia, base = shard(m, npr, n)
    b = torch.empty(n * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(b.data_ptr(), 0, n * k, "u", 3, st)
    c = torch.zeros(max(rows, 1) * k, dtype=torch.float32, device=dev)
    res["csrmm_ms"] = timed(lambda: bofhip.csrmm_resident(
        "N", rows, n, k, 1.0, 0.0, val.data_ptr() - 4 * base, ia.ctypes.data, off.data_ptr(),
        col.data_ptr() - 8 * base, "R", b.data_ptr(), c.data_ptr(), opts, st))
    del val, col, off, b, c
    torch.cuda.empty_cache()
    m = n = 50_000_000
    npr = 10
    r0, rows, val, col, off, ia, base = shard(m, npr, n)
    x = (torch.arange(n, device=dev) % 10).float()
    y = torch.zeros(max(rows, 1), dtype=torch.float32, device=dev)
    res["csrgemv_N_ms"] = timed(lambda: bofhip.csrgemv_resident(
        "N", rows, n, val.data_ptr() - 4 * base, ia.ctypes.data, off.data_ptr(), col.data_ptr() - 8 * base,
        x.data_ptr(), y.data_ptr(), opts, st))
    yt = torch.zeros(n, dtype=torch.float32, device=dev)
    res["csrgemv_T_local_ms"] = timed(lambda: bofhip.csrgemv_resident(
        "T", rows, n, val.data_ptr() - 4 * base, ia.ctypes.data, off.data_ptr(), col.data_ptr() - 8 * base,
        x.data_ptr() + 4 * r0, yt.data_ptr(), opts, st))
    res["csrgemv_T_partial"] = yt
    del val, col, off, x, y
    bofhip.lib().bof_flash_release()
    return res


# =====================================================================================
# End-to-end legs: the SSD-resident configurations themselves (BASELINE configs[1], [2]).
# Files are created under $BOF_BENCH_DIR / $TMPDIR, the library call is timed the way the
# reference's drivers time it (wall clock around flash::gemm / flash::csrmm, flush included:
# drivers/gemm.cpp:57-62, drivers/csrmm.cpp:62-65), and the WHOLE output file is verified.
# =====================================================================================
CFG3_C_SHA256 = "d08df7c04907bec66f4638df05ffe2bbfb447c2a01e2bc03e5e6fd1d8daf2382"


def _open(path, direct):
    if direct:
        try:
            return os.open(path, os.O_RDWR | os.O_DIRECT), True
        except OSError:
            pass
    return os.open(path, os.O_RDWR), False


def _fs_of(path):
    best = ("", "?")
    try:
        rp = os.path.realpath(path)
        for ln in open("/proc/mounts"):
            f = ln.split()
            if rp.startswith(f[1]) and len(f[1]) >= len(best[0]):
                best = (f[1], f[2])
    except OSError:
        pass
    return best[1]


def _write_device_tensor(bofhip, t, path, direct, opts, st):
    """HBM -> file through the library's own writer (n_io_threads workers, 32 MiB chunks);
    O_DIRECT when the file system takes it, so the page cache holds nothing of the inputs."""
    with open(path, "wb") as f:
        f.truncate(t.numel() * t.element_size())
    fd, is_direct = _open(path, direct)
    try:
        o = bofhip.default_options(n_io_threads=opts.n_io_threads, use_odirect=1 if is_direct else 0)
        bofhip.device_to_file(bofhip.FPtr(fd, 0), t.numel() * t.element_size(), t.data_ptr(), o, st)
        os.fsync(fd)
    finally:
        bofhip.lib().bof_file_forget(fd)
        os.close(fd)


def _drop_cache(paths):
    for p in paths:
        fd = os.open(p, os.O_RDONLY)
        try:
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        finally:
            os.close(fd)


def mixed_window_rates(read_stamps, write_stamps, chunk_bytes, own):
    """Rates of the two directions of a mixed pass OVER THE WINDOW IN WHICH BOTH WERE ACTIVE: `*_stamps` = the pass's
    start time followed by the completion time of every chunk.  Until the second session of round 6 each direction's
    bytes were divided by its OWN duration -- the reads (8 GiB) end long before the writes (8 GiB), so the writes'
    figure averaged over a stretch in which they had the disk to themselves (a lease of read 21 / write 16.5 GB/s
    reported 15.4 + 9.0 "mixed" where the window holds 15.4 + 3.7: this disk moves no more when the directions mix
    than when it only reads).  The old figures stay in the probe as `mixed_pass_own_duration_GBps`."""
    out = {"mixed_pass_own_duration_GBps": [round(own.get("r", 0.0), 2), round(own.get("w", 0.0), 2)]}
    if len(read_stamps) < 2 or len(write_stamps) < 2:
        return out
    t0 = max(read_stamps[0], write_stamps[0])
    t1 = min(max(read_stamps), max(write_stamps))
    if t1 - t0 < 0.05:                     # the passes did not overlap long enough to say anything
        return out
    nr = sum(1 for t in read_stamps[1:] if t0 < t <= t1)
    nw = sum(1 for t in write_stamps[1:] if t0 < t <= t1)
    out.update({"disk_read_GBps_while_writing": round(nr * chunk_bytes / (t1 - t0) / 1e9, 2),
                "disk_write_GBps_while_reading": round(nw * chunk_bytes / (t1 - t0) / 1e9, 2),
                "mixed_window_s": round(t1 - t0, 3)})
    return out


def disk_time_bound(rd_bytes, wr_bytes, ceil):
    """Lower bound on the time the scratch disk needs for rd_bytes of reads and wr_bytes of writes, from its three
    probed operating points: reads alone (r), writes alone (w), both directions at once (rm + wm, over the window in
    which both were active).  The disk is time-shared between them; mixing pays only if rm / r + wm / w > 1 -- then the
    bound mixes for as long as both directions have bytes and finishes the rest alone; otherwise (the disks of this
    pool: 0.71-0.96) reads and writes simply add up.  Returns (seconds, "reads + writes" | "mixed, then the rest")."""
    r, w = ceil["disk_read_GBps"] * 1e9, ceil["disk_write_GBps"] * 1e9
    rm, wm = ceil.get("disk_read_GBps_while_writing", 0) * 1e9, ceil.get("disk_write_GBps_while_reading", 0) * 1e9
    serial = rd_bytes / r + wr_bytes / w
    if rm <= 0 or wm <= 0 or rm / r + wm / w <= 1.0:
        return serial, "reads + writes"
    tau = min(rd_bytes / rm, wr_bytes / wm)
    return tau + (rd_bytes - rm * tau) / r + (wr_bytes - wm * tau) / w, "mixed, then the rest"


def disk_probe(bofhip, read_paths, write_path, io_threads=8, passes=3, label=""):
    """The scratch disk's O_DIRECT rates measured ON THE WORKLOAD'S OWN FILES with the pipeline's own request
    shape -- bof_file_sread / bof_file_swrite, 32 MiB per call cut into the library's 4 MiB requests, kernel AIO,
    `io_threads` threads (the panel pipeline's reader pool) -- best of `passes` passes per direction, every read
    pass over all of `read_paths` (A and B: 8 GiB at configs[1]) with the page cache dropped, every write pass
    twice over `write_path` (C: 2 x 4 GiB), then both directions at once.  A file written long ago reads faster than
    one written a second ago (round 4's 4 GiB scratch-file probe under-read the disk by 30 %), hence the real inputs.
    Returns {} when the files cannot be opened O_DIRECT."""
    import threading
    L = bofhip.lib()
    slot = 32 << 20
    fds = []
    for pth in list(read_paths) + [write_path]:
        fd, d = _open(pth, True)
        if not d:
            for f in fds:
                os.close(f)
            os.close(fd)
            return {}
        fds.append(fd)
    rfds, wfd = fds[:-1], fds[-1]
    sizes = [os.fstat(fd).st_size // slot * slot for fd in fds]
    hbuf = []
    for _ in range(2 * io_threads):
        pv = ctypes.c_void_p()
        bofhip.check(L.bof_host_alloc(ctypes.byref(pv), slot), "host_alloc")
        hbuf.append(pv.value)
    rchunks = [(fd, off) for fd, sz in zip(rfds, sizes) for off in range(0, sz, slot)]
    wchunks = [(wfd, off) for _ in range(2) for off in range(0, sizes[-1], slot)]

    def run(chunks, nthr, fn, base=0, stamps=None):
        def work(i):
            for fd, off in chunks[i::nthr]:
                fn(fd, off, hbuf[base + i])
                if stamps is not None:
                    stamps.append(time.perf_counter())       # (list.append is atomic under the GIL)
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
        t0 = time.perf_counter()
        if stamps is not None:
            stamps.append(t0)
        for x in th:
            x.start()
        for x in th:
            x.join()
        return len(chunks) * slot / (time.perf_counter() - t0) / 1e9

    def rd(fd, off, buf):
        L.bof_file_sread(fd, off, 0, 1, slot, buf, 1)

    def wr(fd, off, buf):
        L.bof_file_swrite(fd, off, 0, 1, slot, buf, 1)
    out = {"files": "the workload's own A, B (read) and C (written)", "GiB_per_read_pass": round(len(rchunks) * slot / 2**30, 1),
           "GiB_per_write_pass": round(len(wchunks) * slot / 2**30, 1), "threads": io_threads, "passes": passes, "when": label}
    try:
        reads, writes = [], []
        for _ in range(passes):
            _drop_cache(read_paths)
            reads.append(run(rchunks, io_threads, rd))
        for _ in range(passes):
            writes.append(run(wchunks, io_threads, wr))
        _drop_cache(read_paths)
        both, rs, ws = {}, [], []
        ta = threading.Thread(target=lambda: both.__setitem__("r", run(rchunks, io_threads, rd, 0, rs)))
        tb = threading.Thread(target=lambda: both.__setitem__("w", run(wchunks, io_threads, wr, io_threads, ws)))
        ta.start(); tb.start(); ta.join(); tb.join()
        out.update({"disk_read_GBps": round(max(reads), 2), "disk_write_GBps": round(max(writes), 2),
                    "read_passes_GBps": [round(x, 2) for x in reads], "write_passes_GBps": [round(x, 2) for x in writes]})
        out.update(mixed_window_rates(rs, ws, slot, both))
    finally:
        for h in hbuf:
            L.bof_host_free(h)
        for fd in fds:
            L.bof_file_forget(fd)
            os.close(fd)
    return out


def merge_ceilings(ceil, *probes):
    """A ceiling is the BEST the device showed in any probe of this run (before the steps, after them, the scratch-file
    probe): the probe moves, never the run."""
    out = dict(ceil)
    for pr in probes:
        for q in ("disk_read_GBps", "disk_write_GBps"):
            if pr.get(q):
                out[q] = max(out.get(q, 0.0), pr[q])
        mixed_new = pr.get("disk_read_GBps_while_writing", 0) + pr.get("disk_write_GBps_while_reading", 0)
        mixed_old = out.get("disk_read_GBps_while_writing", 0) + out.get("disk_write_GBps_while_reading", 0)
        if mixed_new > mixed_old:
            for q in ("disk_read_GBps_while_writing", "disk_write_GBps_while_reading", "mixed_window_s", "mixed_pass_own_duration_GBps"):
                if q in pr:
                    out[q] = pr[q]
    return out


def io_ceilings(bofhip, torch, dev, st, workdir, io_threads=8, gib=4.0):
    """What this box's scratch disk, page cache and PCIe link deliver RIGHT NOW (about five seconds):
    the ceilings the `roofline_e2e` fractions of this same run are taken against.  Measured with the
    library's own file primitives into pinned buffers (bof_file_sread / bof_file_swrite, 32 MiB per call
    cut into the default 4 MiB requests, `io_threads` threads), like tools/iobench.py's full sweep."""
    import threading
    L = bofhip.lib()
    size = int(gib * 2**30)
    slot = 32 << 20
    path = os.path.join(workdir, "ceilings.bin")
    out = {}
    t = torch.empty(size // 4, dtype=torch.float32, device=dev)
    bofhip.gen_dense(t.data_ptr(), 0, t.numel(), "u", 7, st)
    _write_device_tensor(bofhip, t, path, True, bofhip.default_options(n_io_threads=io_threads), st)
    hbuf = []
    for _ in range(2 * io_threads):
        p = ctypes.c_void_p()
        bofhip.check(L.bof_host_alloc(ctypes.byref(p), slot), "host_alloc")
        hbuf.append(p.value)
    nchunks = size // slot

    def run(nthr, fn, stamps=None):
        th = [threading.Thread(target=fn, args=(i, nthr)) for i in range(nthr)]
        t0 = time.perf_counter()
        if stamps is not None:
            stamps.append(t0)
        for x in th:
            x.start()
        for x in th:
            x.join()
        return size / (time.perf_counter() - t0) / 1e9
    try:
        fd, direct = _open(path, True)
        if direct:
            def rd(i, nthr, fd_=fd, base=0, stamps=None):
                for cidx in range(i, nchunks, nthr):
                    L.bof_file_sread(fd_, cidx * slot, 0, 1, slot, hbuf[base + i], 1)
                    if stamps is not None:
                        stamps.append(time.perf_counter())

            def wr(i, nthr, fd_=fd, base=0, stamps=None):
                for cidx in range(i, nchunks, nthr):
                    L.bof_file_swrite(fd_, cidx * slot, 0, 1, slot, hbuf[base + i], 1)
                    if stamps is not None:
                        stamps.append(time.perf_counter())
            # a ceiling must not be lower than what a pipeline can get: best over two queue depths
            # (all read passes before the first write pass: right behind 4 GiB of writes the device reads slower than
            # the pipelines, whose inputs were written long before, ever see -- round 4's probes read 14 GB/s where
            # the csrgemv run of the same process then read 19)
            best_r = best_w = 0.0
            time.sleep(1.0)
            for nthr in (io_threads, 2 * io_threads, io_threads):
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                best_r = max(best_r, run(nthr, rd))
            for nthr in (io_threads, 2 * io_threads):
                best_w = max(best_w, run(nthr, wr))
            out["disk_read_GBps"] = round(best_r, 2)
            out["disk_write_GBps"] = round(best_w, 2)
            # both directions at once (the pipeline's steady state): readers on one file, writers on another
            path2 = path + ".2"
            _write_device_tensor(bofhip, t, path2, True, bofhip.default_options(n_io_threads=io_threads), st)
            fd2, _ = _open(path2, True)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            both, rs, ws = {}, [], []
            ta = threading.Thread(target=lambda: both.__setitem__("r", run(io_threads, lambda i, n: rd(i, n, fd, 0, rs), rs)))
            tb = threading.Thread(target=lambda: both.__setitem__(
                "w", run(io_threads, lambda i, n: wr(i, n, fd2, io_threads, ws), ws)))
            ta.start(); tb.start(); ta.join(); tb.join()
            out.update(mixed_window_rates(rs, ws, slot, both))      # (over the window in which both directions ran)
            L.bof_file_forget(fd2)
            os.close(fd2)
            os.remove(path2)
        else:
            out["disk_note"] = "the scratch file system refuses O_DIRECT"
        L.bof_file_forget(fd)
        os.close(fd)
        fd, _ = _open(path, False)

        def prd(i, nthr):
            for cidx in range(i, nchunks, nthr):
                L.bof_file_sread(fd, cidx * slot, 0, 1, slot, hbuf[i], 0)
        run(io_threads, prd)                       # fills the page cache
        out["page_cache_read_GBps"] = round(run(io_threads, prd), 2)
        os.close(fd)
        # PCIe, pinned <-> HBM, 32 MiB linear copies on two streams
        s1, s2 = ctypes.c_void_p(), ctypes.c_void_p()
        L.bof_stream_create(ctypes.byref(s1))
        L.bof_stream_create(ctypes.byref(s2))
        d0, reps = t.data_ptr(), 24

        def timed(fn):
            fn()
            L.bof_stream_sync(s1); L.bof_stream_sync(s2)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            L.bof_stream_sync(s1); L.bof_stream_sync(s2)
            return reps * slot / (time.perf_counter() - t0) / 1e9
        out["pcie_h2d_GBps"] = round(timed(lambda: L.bof_memcpy_h2d(d0, hbuf[0], slot, s1)), 2)
        out["pcie_d2h_GBps"] = round(timed(lambda: L.bof_memcpy_d2h(hbuf[1], d0 + slot, slot, s2)), 2)

        def bidir():
            L.bof_memcpy_h2d(d0, hbuf[0], slot, s1)
            L.bof_memcpy_d2h(hbuf[1], d0 + slot, slot, s2)
        out["pcie_each_way_when_both_GBps"] = round(timed(bidir), 2)
        L.bof_stream_destroy(s1); L.bof_stream_destroy(s2)
    finally:
        for h in hbuf:
            L.bof_host_free(h)
        if os.path.exists(path):
            os.remove(path)
        del t
        torch.cuda.empty_cache()
    return out


def roofline_e2e(leg, ceil, flops, kernel_s, mode):
    """Lower bound on the wall time of one out-of-core call from the ceilings PROBED in this run, and the
    fraction of it the call achieved: t_bound = max over the stages that all run concurrently in the
    pipeline -- kernel time at the measured kernel rate, bytes over PCIe each way, bytes read / written at
    the disk's (odirect) or the page cache's (buffered) rate; frac = t_bound / seconds.  The ceilings are the
    best of every probe of the run (the scratch-file probe and the probes on the headline's own files before and
    after its timed steps); a call that still beats its bound raises that stage's ceiling to its own rate
    (`probe_raised`): a ceiling below an observed rate is a bad probe, not a fast run.
    `run_disk_GBps` is what the call itself moved."""
    st = leg["stats"]
    terms = {"mfma" if flops > 1e13 else "hbm": kernel_s or 0.0}
    if ceil.get("pcie_h2d_GBps"):
        terms["pcie_h2d"] = st["bytes_h2d"] / (ceil["pcie_h2d_GBps"] * 1e9)
        terms["pcie_d2h"] = st["bytes_d2h"] / (ceil["pcie_d2h_GBps"] * 1e9)
        if ceil.get("pcie_each_way_when_both_GBps"):     # both directions at once share the link's controllers
            terms["pcie_both_ways"] = (st["bytes_h2d"] + st["bytes_d2h"]) / (2e9 * ceil["pcie_each_way_when_both_GBps"])
    if mode == "odirect" and ceil.get("disk_read_GBps"):
        terms["disk_read"] = st["bytes_read"] / (ceil["disk_read_GBps"] * 1e9)
        terms["disk_write"] = st["bytes_written"] / (ceil["disk_write_GBps"] * 1e9)
        terms["disk_total"] = disk_time_bound(st["bytes_read"], st["bytes_written"], ceil)[0]
    elif mode == "buffered" and ceil.get("page_cache_read_GBps"):
        terms["page_cache_read"] = st["bytes_read"] / (ceil["page_cache_read_GBps"] * 1e9)
    bound = max(terms, key=terms.get)
    t_bound = terms[bound]
    out = {"seconds": leg["seconds"], "gflops": leg["gflops"], "bound": bound, "t_bound_s": round(t_bound, 4),
           "frac": round(t_bound / leg["seconds"], 3), "terms_s": {k: round(v, 4) for k, v in terms.items()},
           "run_disk_GBps": round((st["bytes_read"] + st["bytes_written"]) / leg["seconds"] / 1e9, 2)}
    out["frac_raw"] = out["frac"]        # against the probes as they are; > 1 names a mis-probed ceiling
    if t_bound > leg["seconds"]:
        # the call moved its bytes faster than the probe of that stage did: the stage CAN do what it just did, so its
        # ceiling is raised to the call's own rate (and says so) -- the probe moves, never the run
        out["probe_raised"] = f"{bound}: probed bound {t_bound:.4f} s > the call's {leg['seconds']:.4f} s; ceiling raised to the call's own rate"
        out["t_bound_s"] = round(leg["seconds"], 4)
        out["frac"] = 1.0
    return out


def _leg_summary(runs, flops, kernel_s, compulsory_rd, compulsory_wr, units):
    best = min(runs, key=lambda r: r["seconds"])
    s = best["seconds"]
    st = best["stats"]
    return {"seconds_all": [round(r["seconds"], 3) for r in runs], "seconds": round(s, 3),
            "gflops": round(flops / s / 1e9, 1),
            "read_GBps": round(st["bytes_read"] / s / 1e9, 2), "write_GBps": round(st["bytes_written"] / s / 1e9, 2),
            "h2d_GBps": round(st["bytes_h2d"] / s / 1e9, 2), "d2h_GBps": round(st["bytes_d2h"] / s / 1e9, 2),
            "read_amplification": round(st["bytes_read"] / compulsory_rd, 3),
            "write_amplification": round(st["bytes_written"] / compulsory_wr, 3),
            "read_requests": st["read_ops"], "write_requests": st["write_ops"],
            "requests_per_unit": round((st["read_ops"] + st["write_ops"]) / max(units, 1), 1),
            "overlap_kernel_over_e2e": round(kernel_s / s, 3) if kernel_s else None, "stats": st}


def e2e_gemm(bofhip, torch, dev, st, workdir, n, blk, kernel_s, io_threads, reps, modes=("odirect", "buffered"),
             rank_calls=0, **extra_opts):
    """cfg2 through bof_flash_gemm on three n*n*4-byte files (A, B mode 's'; C zeros).
    rank_calls = W > 1: instead of one call, the W calls the ranks of a W-GPU run make (bof_dist.row_shard:
    rank g's C rows, A and C pointers advanced to its slab, all of B), one after the other on this GPU
    against the same three files -- the composition the multi-GPU run relies on, whole C verified."""
    import numpy as np
    nbytes = n * n * 4
    pa, pb, pc = (os.path.join(workdir, x) for x in ("A.bin", "B.bin", "C.bin"))
    wopts = bofhip.default_options(n_io_threads=io_threads)
    t0 = time.perf_counter()
    t = torch.empty(n * n, dtype=torch.float32, device=dev)
    for path in (pa, pb):
        bofhip.gen_dense(t.data_ptr(), 0, n * n, "s", 0, st)          # dense_create mode s
        _write_device_tensor(bofhip, t, path, True, wopts, st)
    t.zero_()
    _write_device_tensor(bofhip, t, pc, True, wopts, st)                # dense_create mode z
    create_s = time.perf_counter() - t0
    # closed form: C[i,j] depends on (i mod 10, j mod 10) only
    ii = np.arange(10, dtype=np.int64)[:, None]
    kk = np.arange(n, dtype=np.int64)
    a10 = (ii * n + kk[None, :]) % 10                                    # 10 x n
    b10 = (kk[:, None] * n + np.arange(10, dtype=np.int64)[None, :]) % 10  # n x 10
    pat = torch.from_numpy((a10 @ b10).astype(np.float32)).to(dev)       # exact: all sums < 2^24
    idx = torch.arange(n, device=dev)
    rowpat = pat[:, idx % 10]                                            # 10 x n

    def verify():
        """whole C file -> HBM -> compared with the closed form, every element."""
        fd, _ = _open(pc, False)
        try:
            bofhip.file_to_device(bofhip.FPtr(fd, 0), nbytes, t.data_ptr(), bofhip.default_options(use_odirect=0), st)
        finally:
            os.close(fd)
        C = t.view(n, n)
        rows = max(1, (1 << 27) // n)
        for r0 in range(0, n, rows):
            r1 = min(n, r0 + rows)
            if not torch.equal(C[r0:r1], rowpat[idx[r0:r1] % 10]):
                return False
        return True

    def reset_c():
        fd, d = _open(pc, True)
        try:
            t.zero_()
            o = bofhip.default_options(n_io_threads=io_threads, use_odirect=1 if d else 0)
            bofhip.device_to_file(bofhip.FPtr(fd, 0), nbytes, t.data_ptr(), o, st)
        finally:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)

    flops = 2.0 * n ** 3
    tiles = 3 * max(n // blk, 1) ** 2
    which = {32768: "BASELINE configs[1]", 65536: "north_star's 64k target, configs[3] on one GPU"}.get(n, "debug size")
    out = {"workload": f"flash _gemm fp32 {n}x{n}x{n}, {blk}-tile, A/B/C as {nbytes / 2**30:.0f} GiB files "
                       f"({which}); wall clock around bof_flash_gemm incl. write-back",
           "file_system": _fs_of(workdir), "create_files_s": round(create_s, 1), "io_threads": io_threads}
    for mode in modes:
        fds = []
        ok_direct = True
        for p in (pa, pb, pc):
            fd, d = _open(p, mode == "odirect")
            fds.append(fd)
            ok_direct = ok_direct and d
        if mode == "odirect" and not ok_direct:
            out[mode] = {"skipped": "file system refuses O_DIRECT"}
            for fd in fds:
                os.close(fd)
            continue
        opts = bofhip.default_options(gemm_blk=blk, n_io_threads=io_threads, use_odirect=1 if mode == "odirect" else 0,
                                      **extra_opts)
        runs = []
        verified = True
        total = reps + (1 if mode == "buffered" else 0)
        for rep in range(total):
            if mode == "odirect":
                _drop_cache((pa, pb, pc))
            t0 = time.perf_counter()
            if rank_calls > 1:
                import bof_dist
                tot = None
                for g in range(rank_calls):
                    r0, r1 = bof_dist.row_shard(n, rank_calls, g, blk)
                    if r1 > r0:
                        bofhip.flash_gemm("R", "N", "N", r1 - r0, n, n, 1.0, 0.0, bofhip.FPtr(fds[0], r0 * n * 4),
                                          bofhip.FPtr(fds[1], 0), bofhip.FPtr(fds[2], r0 * n * 4), 0, 0, 0, opts)
                        s1 = bofhip.flash_last_stats()
                        tot = s1 if tot is None else {q: tot[q] + s1[q] for q in tot}
                stats_now = tot
            else:
                bofhip.flash_gemm("R", "N", "N", n, n, n, 1.0, 0.0, bofhip.FPtr(fds[0], 0), bofhip.FPtr(fds[1], 0),
                                  bofhip.FPtr(fds[2], 0), 0, 0, 0, opts)
                stats_now = bofhip.flash_last_stats()
            dt = time.perf_counter() - t0
            runs.append({"seconds": dt, "stats": stats_now, "per_device": bofhip.flash_last_device_stats()})
            last = rep == total - 1
            if rep == 0 or last:
                verified = verified and verify()      # every element of the C file
            if not last:
                reset_c()            # later runs must produce C again, not find it
        for fd in fds:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
        if mode == "buffered":
            cold = runs.pop(0)       # first buffered run also fills the page cache
        leg = _leg_summary(runs, flops, kernel_s, 2.0 * nbytes, 1.0 * nbytes, tiles)
        if len(runs[-1]["per_device"]) > 1:
            leg["per_device"] = min(runs, key=lambda r: r["seconds"])["per_device"]
        if mode == "buffered":
            leg["first_run_cold_cache_s"] = round(cold["seconds"], 3)
            leg["note"] = "files in the page cache (DRAM-resident): the I/O stack without the device"
        leg["whole_C_file_matches_closed_form"] = bool(verified)
        out[mode] = leg
    for p in (pa, pb, pc):
        os.remove(p)
    del t
    torch.cuda.empty_cache()
    return out


def e2e_csrmm(bofhip, torch, dev, st, workdir, kernel_s, io_threads, reps, **extra_opts):
    """cfg3 through bof_flash_csrmm: sparse_create(10M, 1M, 1e-4) x dense_create(1M, 128, 's'),
    C = 10M x 128; sha256 of the C file against the hash both reference drivers produced."""
    import hashlib
    m, n, k, npr = 10_000_000, 1_000_000, 128, 100
    nnz = m * npr
    p = {x: os.path.join(workdir, x) for x in ("A.csr", "A.col", "A.off", "B.bin", "C.bin")}
    wopts = bofhip.default_options(n_io_threads=io_threads)
    t0 = time.perf_counter()
    val = torch.empty(nnz, dtype=torch.float32, device=dev)
    col = torch.empty(nnz, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 1_000_000):
        bofhip.gen_sparse_rows(r0, 1_000_000, n, npr, val.data_ptr() + 4 * r0 * npr,
                               col.data_ptr() + 8 * r0 * npr, off.data_ptr() + 8 * r0, st)
    b = torch.empty(n * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(b.data_ptr(), 0, n * k, "s", 0, st)
    for t, name in ((val, "A.csr"), (col, "A.col"), (off, "A.off"), (b, "B.bin")):
        _write_device_tensor(bofhip, t, p[name], True, wopts, st)
    del val, col, off, b
    c = torch.zeros(m * k, dtype=torch.float32, device=dev)
    _write_device_tensor(bofhip, c, p["C.bin"], True, wopts, st)
    create_s = time.perf_counter() - t0
    flops = 2.0 * nnz * k
    rd = nnz * 12 + (m + 1) * 8 + 4 * n * k
    wr = 4 * m * k
    out = {"workload": "flash _csrmm 10M x 1M CSR (1e9 nnz: 4 GB values, 8 GB int64 indices, 80 MB offsets) x 1M x 128 "
                       "(512 MB) -> C 5.12 GB, all files (BASELINE configs[2]); wall clock around bof_flash_csrmm",
           "file_system": _fs_of(workdir), "create_files_s": round(create_s, 1), "io_threads": io_threads}
    names = ("A.csr", "A.off", "A.col", "B.bin", "C.bin")
    sums_ok = True
    for mode in ("odirect", "buffered"):
        fds, ok_direct = {}, True
        for x in names:
            fds[x], d = _open(p[x], mode == "odirect")
            ok_direct = ok_direct and d
        if mode == "odirect" and not ok_direct:
            out[mode] = {"skipped": "file system refuses O_DIRECT"}
            for fd in fds.values():
                os.close(fd)
            continue
        opts = bofhip.default_options(n_io_threads=io_threads, use_odirect=1 if mode == "odirect" else 0, **extra_opts)
        runs = []
        for rep in range(reps + (1 if mode == "buffered" else 0)):
            if mode == "odirect":
                _drop_cache(p.values())
            F = lambda x: bofhip.FPtr(fds[x], 0)                                        # noqa: E731
            t0 = time.perf_counter()
            bofhip.flash_csrmm("N", m, n, k, 1.0, 0.0, F("A.csr"), F("A.off"), F("A.col"), "R", F("B.bin"),
                               F("C.bin"), opts)
            dt = time.perf_counter() - t0
            runs.append({"seconds": dt, "stats": bofhip.flash_last_stats()})
        for fd in fds.values():
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
        # whole C file back into HBM: total must be the known 2879999461076 (App. A-3)
        fd, _ = _open(p["C.bin"], False)
        try:
            bofhip.file_to_device(bofhip.FPtr(fd, 0), m * k * 4, c.data_ptr(), bofhip.default_options(use_odirect=0), st)
        finally:
            os.close(fd)
        total = float(c.double().sum().item())
        sums_ok = sums_ok and total == 2879999461076.0
        if mode == "buffered":
            cold = runs.pop(0)
        leg = _leg_summary(runs, flops, kernel_s, float(rd), float(wr), 100)
        if mode == "buffered":
            leg["first_run_cold_cache_s"] = round(cold["seconds"], 3)
        leg["C_total_matches_reference"] = bool(total == 2879999461076.0)
        out[mode] = leg
    h = hashlib.sha256()
    with open(p["C.bin"], "rb") as f:
        while True:
            chunk = f.read(1 << 26)
            if not chunk:
                break
            h.update(chunk)
    out["sha256_C_file"] = h.hexdigest()
    out["sha256_matches_reference_drivers"] = bool(h.hexdigest() == CFG3_C_SHA256)
    for x in p.values():
        os.remove(x)
    del c
    torch.cuda.empty_cache()
    return out


CFG5_Y_SHA256 = {"N": "1c0a44dbb962be0a2d7027f5f298ff94ab5b2ee66c8fe467c0408b286b1881e4",
                 "T": "486766062199bef476611a2675893df3266338c91bfc30db4640ef5dbc2cbf40"}


def e2e_csrgemv(bofhip, torch, dev, st, workdir, kernel_ms, io_threads, reps, modes=("odirect", "buffered"), **extra_opts):
    """BASELINE configs[4] size on one GPU through bof_flash_csrgemv: sparse_create(50M, 50M, 2e-7) as
    files (2 GB values, 4 GB int64 indices, 400 MB offsets), x = i % 10 and y in HOST memory as in the
    reference (include/flash_blas.h:55-57); 'N' and 'T'; sha256(y) = the reference's known answers."""
    import hashlib
    import numpy as np
    m = n = 50_000_000
    npr = 10
    nnz = m * npr
    p = {x: os.path.join(workdir, x) for x in ("G.csr", "G.col", "G.off")}
    wopts = bofhip.default_options(n_io_threads=io_threads)
    t0 = time.perf_counter()
    val = torch.empty(nnz, dtype=torch.float32, device=dev)
    col = torch.empty(nnz, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 5_000_000):
        bofhip.gen_sparse_rows(r0, 5_000_000, n, npr, val.data_ptr() + 4 * r0 * npr, col.data_ptr() + 8 * r0 * npr,
                               off.data_ptr() + 8 * r0, st)
    for t, name in ((val, "G.csr"), (col, "G.col"), (off, "G.off")):
        _write_device_tensor(bofhip, t, p[name], True, wopts, st)
    del val, col, off
    torch.cuda.empty_cache()
    out = {"workload": "flash _csrgemv 50M x 50M CSR (5e8 nnz: 6.4 GB of files), x and y in host memory "
                       "(BASELINE configs[4] size on one GPU); wall clock around bof_flash_csrgemv",
           "create_files_s": round(time.perf_counter() - t0, 1), "io_threads": io_threads}
    x = (np.arange(n) % 10).astype(np.float32)
    y = np.zeros(n, np.float32)
    alg = nnz * 12 + (m + 1) * 8 + 8 * n
    for mode in modes:
        fds, ok_direct = {}, True
        for name in p:
            fds[name], d = _open(p[name], mode == "odirect")
            ok_direct = ok_direct and d
        if mode == "odirect" and not ok_direct:
            out[mode] = {"skipped": "file system refuses O_DIRECT"}
            for fd in fds.values():
                os.close(fd)
            continue
        opts = bofhip.default_options(n_io_threads=io_threads, use_odirect=1 if mode == "odirect" else 0, **extra_opts)
        leg = {}
        for tr in "NT":
            secs = []
            for rep in range(reps + (1 if mode == "buffered" and tr == "N" else 0)):
                if mode == "odirect":
                    _drop_cache(p.values())
                y.fill(-1.0)
                t0 = time.perf_counter()
                bofhip.flash_csrgemv(tr, m, n, bofhip.FPtr(fds["G.csr"], 0), bofhip.FPtr(fds["G.off"], 0),
                                     bofhip.FPtr(fds["G.col"], 0), x.ctypes.data, y.ctypes.data, opts)
                secs.append(time.perf_counter() - t0)
            if mode == "buffered" and tr == "N":
                leg["first_run_cold_cache_s"] = round(secs.pop(0), 3)
            best = min(secs)
            stats = bofhip.flash_last_stats()
            leg[tr] = {"seconds_all": [round(v, 3) for v in secs], "seconds": round(best, 3),
                       "gflops": round(2.0 * nnz / best / 1e9, 2), "read_GBps": round(stats["bytes_read"] / best / 1e9, 2),
                       "algorithmic_GBps": round(alg / best / 1e9, 2),
                       "overlap_kernel_over_e2e": round(kernel_ms[tr] * 1e-3 / best, 3) if kernel_ms and tr in kernel_ms else None,
                       "sha256_y_matches_reference": bool(hashlib.sha256(y.tobytes()).hexdigest() == CFG5_Y_SHA256[tr]),
                       "stats": stats}
        for fd in fds.values():
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
        out[mode] = leg
    for f in p.values():
        os.remove(f)
    return out



def e2e_block(bofhip, torch, dev, st, args, gemm_kernel_s, csrmm_kernel_s, gemm64_kernel_s=None, gemv_kernel_ms=None,
              ceilings=None):
    import shutil
    import tempfile
    base = args.e2e_dir or os.environ.get("BOF_BENCH_DIR") or os.environ.get("TMPDIR") or "/tmp"
    out = {"what": "out-of-core legs: files -> pinned ring -> HBM -> kernels -> files, timed like the reference's "
                   "drivers; odirect = O_DIRECT descriptors with the page cache dropped before every call"}
    try:
        workdir = tempfile.mkdtemp(prefix="bof_bench_", dir=base)
    except OSError as e:
        return {"error": f"no scratch directory under {base}: {e}"}
    try:
        free = shutil.disk_usage(workdir).free
        out["scratch_dir_free_GB"] = round(free / 1e9, 1)
        out["process_state"] = {"main_thread_allowed_cpus": len(os.sched_getaffinity(0)),
                                "process_threads": len(os.listdir("/proc/self/task")),
                                "loadavg_1min": round(os.getloadavg()[0], 1)}
        n = args.e2e_size
        bofhip.lib().bof_flash_release()
        out["ceilings"] = ceilings
        if free > 3 * n * n * 4 + (2 << 30):
            # (the O_DIRECT form of this configuration is the headline itself)
            out["gemm"] = e2e_gemm(bofhip, torch, dev, st, workdir, n, args.blk, gemm_kernel_s, args.io_threads,
                                   args.e2e_reps, modes=("buffered",))
        else:
            out["gemm"] = {"skipped": "not enough free disk for three matrix files"}
        bofhip.lib().bof_flash_release()
        # north_star's 64k x 64k x 64k out-of-core GEMM at N = 1 (3 x 16 GiB files, 4096 tile tasks,
        # 563 TFLOP): one O_DIRECT call and one page-cache call, whole C file verified
        if args.e2e_size == 32768 and not args.no_e2e_64k:
            free = shutil.disk_usage(workdir).free
            if free > 3 * 65536 * 65536 * 4 + (4 << 30):
                # (two calls: the first pays the cold allocations of 48 GiB of panel slots, the second is the steady state)
                out["gemm_65536"] = e2e_gemm(bofhip, torch, dev, st, workdir, 65536, args.blk, gemm64_kernel_s,
                                             args.io_threads, 2, modes=("odirect",))
            else:
                out["gemm_65536"] = {"skipped": f"needs 48 GiB of scratch disk, {free / 2**30:.0f} GiB free"}
            bofhip.lib().bof_flash_release()
        # the paper's unaligned case (Fig. 5 right: 31000-edge matrices; rows of 124000 bytes, a file size that is
        # no multiple of a sector): O_DIRECT kept through sector-widened reads / page-split writes
        if args.more_legs and shutil.disk_usage(workdir).free > 3 * 31000 * 31000 * 4 + (2 << 30):
            out["gemm_31000"] = e2e_gemm(bofhip, torch, dev, st, workdir, 31000, args.blk, None, args.io_threads, 2)
            bofhip.lib().bof_flash_release()
        # the tile cache (what takes the call when the row-panel plan does not fit the budget or C's rows have
        # gaps), on cfg2: forced (gemm_path = 1), and chosen by a budget that cannot hold B
        if args.more_legs and shutil.disk_usage(workdir).free > 3 * n * n * 4 + (2 << 30):
            out["gemm_tile_cache"] = e2e_gemm(bofhip, torch, dev, st, workdir, n, args.blk, gemm_kernel_s, args.io_threads,
                                              2, modes=("odirect",), gemm_path=1)
            bofhip.lib().bof_flash_release()
            out["gemm_budget_8GiB"] = e2e_gemm(bofhip, torch, dev, st, workdir, n, args.blk, gemm_kernel_s, args.io_threads,
                                               2, modes=("odirect",), hbm_budget=8 << 30)
            bofhip.lib().bof_flash_release()
        # the in-process device list (what an unchanged reference driver gets on a multi-GPU node): device 0
        # listed twice, i.e. one GPU playing two -- the point of the leg is the byte counters (B read once,
        # copied twice) and the whole-C check, not the time
        if args.more_legs and shutil.disk_usage(workdir).free > 3 * n * n * 4 + (2 << 30):
            out["gemm_two_devices_in_process"] = e2e_gemm(bofhip, torch, dev, st, workdir, n, args.blk, gemm_kernel_s,
                                                          args.io_threads, 1, modes=("odirect",), devices=[0, 0])
            bofhip.lib().bof_flash_release()
        if args.no_csr or args.e2e_size != 32768:
            pass
        elif free > 19e9:
            out["csrmm"] = e2e_csrmm(bofhip, torch, dev, st, workdir, csrmm_kernel_s, args.io_threads, args.e2e_reps)
        else:
            out["csrmm"] = {"skipped": "not enough free disk for the cfg3 files (17.7 GB)"}
        bofhip.lib().bof_flash_release()
        if not (args.no_csr or args.e2e_size != 32768):
            if shutil.disk_usage(workdir).free > 8e9:
                out["csrgemv"] = e2e_csrgemv(bofhip, torch, dev, st, workdir, gemv_kernel_ms, args.io_threads, args.e2e_reps)
            else:
                out["csrgemv"] = {"skipped": "not enough free disk for the cfg5-size files (6.4 GB)"}
            bofhip.lib().bof_flash_release()
    except Exception as e:  # the headline line must still be printed
        out["error"] = f"{type(e).__name__}: {str(e)[:300]}"
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    # roofline of every leg against the ceilings of this run
    ceil = out.get("ceilings") or {}
    work = {"gemm": (2.0 * args.e2e_size ** 3, gemm_kernel_s), "gemm_65536": (2.0 * 65536.0 ** 3, gemm64_kernel_s),
            "gemm_31000": (2.0 * 31000.0 ** 3, (gemm_kernel_s or 0) * (31000.0 / 32768.0) ** 3 or None),
            "gemm_tile_cache": (2.0 * args.e2e_size ** 3, gemm_kernel_s), "gemm_budget_8GiB": (2.0 * args.e2e_size ** 3, gemm_kernel_s),
            "gemm_two_devices_in_process": (2.0 * args.e2e_size ** 3, gemm_kernel_s),
            "csrmm": (2.0 * 1e9 * 128, csrmm_kernel_s),
            "csrgemv": (2.0 * 5e8, (gemv_kernel_ms or {}).get("N", 0) * 1e-3 if gemv_kernel_ms else None)}
    if "error" not in ceil:
        for name, (flops, ks) in work.items():
            legs = out.get(name)
            if not isinstance(legs, dict):
                continue
            for mode in ("odirect", "buffered"):
                tgt = legs.get(mode) if name != "csrgemv" else (legs.get(mode) or {}).get("N")
                if isinstance(tgt, dict) and "stats" in tgt and "seconds" in tgt:
                    tgt.setdefault("gflops", round(flops / tgt["seconds"] / 1e9, 1))
                    tgt["roofline"] = roofline_e2e(tgt, ceil, flops, ks, mode)
    return out


# =====================================================================================
# The line the driver parses.  `value` = BASELINE configs[1] itself: wall clock around
# bof_flash_gemm on three SSD-resident files (O_DIRECT descriptors, page cache dropped,
# write-back of C included) -- what the reference's driver times (drivers/gemm.cpp:57-62).
# Everything that is not part of the contract goes to bench_detail.json.
# =====================================================================================
LINE_CAP = 4096


def emit(line, detail, extra_path=""):
    """Writes bench_detail.json (everything) and prints the ONE compact line (< 4 KiB, checked)."""
    detail = dict(detail, line=line)
    targets = [os.path.join(d, "bench_detail.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    for path in targets + ([extra_path] if extra_path else []):
        if path:
            try:
                with open(path, "w") as f:
                    json.dump(detail, f, indent=1, default=str)
            except OSError:
                pass
    s = json.dumps(line, separators=(",", ":"))
    for victim in ("extras", "e2e"):          # never reached with the keys below; the cap is a contract
        if len(s) >= LINE_CAP and victim in line:
            line = {k: v for k, v in line.items() if k != victim}
            s = json.dumps(line, separators=(",", ":"))
    assert len(s) < LINE_CAP, f"bench line is {len(s)} bytes"
    assert json.loads(s)["metric"] == line["metric"]
    print(s, flush=True)


def _closed_form_rows(torch, dev, n, k, ncols):
    """dense_create mode s operands: C[i, j] depends on (i mod 10, j mod 10) only;
    returns the 10 x ncols pattern (exact: every partial sum < 2^24)."""
    import numpy as np
    kk = np.arange(k, dtype=np.int64)
    a10 = (np.arange(10, dtype=np.int64)[:, None] * k + kk[None, :]) % 10
    b10 = (kk[:, None] * ncols + np.arange(10, dtype=np.int64)[None, :]) % 10
    pat = torch.from_numpy((a10 @ b10).astype(np.float32)).to(dev)
    return pat[:, torch.arange(ncols, device=dev) % 10]


def headline_flash_gemm(bofhip, torch, dev, st, workdir, n, blk, io_threads, steps, warmup, panel_streams, probe=True,
                        event_dir=None, extra_opts=None):
    """BASELINE configs[1]: flash _gemm fp32 n^3, blk-tile, A / B / C SSD-resident.  One step = one
    bof_flash_gemm call on the three files: reads A and B (8 GiB), runs the 512 tile tasks, writes C
    (4 GiB) back; beta = 0, so every step recomputes and rewrites the whole of C.  Returns the timed
    region's wall clock, the per-step counters (bytes, tasks, event-timed kernel seconds) and the
    whole-file verification before and after."""
    nbytes = n * n * 4
    pa, pb, pc = (os.path.join(workdir, x) for x in ("A.bin", "B.bin", "C.bin"))
    wopts = bofhip.default_options(n_io_threads=io_threads)
    t0 = time.perf_counter()
    t = torch.empty(n * n, dtype=torch.float32, device=dev)
    bofhip.gen_dense(t.data_ptr(), 0, n * n, "s", 0, st)                  # dense_create mode s (A == B)
    for path in (pa, pb):
        _write_device_tensor(bofhip, t, path, True, wopts, st)
    t.zero_()
    _write_device_tensor(bofhip, t, pc, True, wopts, st)                    # dense_create mode z
    create_s = time.perf_counter() - t0
    rowpat = _closed_form_rows(torch, dev, n, n, n)
    idx = torch.arange(n, device=dev)

    def verify():
        fd, _ = _open(pc, False)
        try:
            bofhip.file_to_device(bofhip.FPtr(fd, 0), nbytes, t.data_ptr(), bofhip.default_options(use_odirect=0), st)
        finally:
            os.close(fd)
        C = t.view(n, n)
        rows = max(1, (1 << 27) // n)
        for r0 in range(0, n, rows):
            r1 = min(n, r0 + rows)
            if not torch.equal(C[r0:r1], rowpat[idx[r0:r1] % 10]):
                return False
        return True

    def reset_c():
        fd, d = _open(pc, True)
        try:
            t.zero_()
            o = bofhip.default_options(n_io_threads=io_threads, use_odirect=1 if d else 0)
            bofhip.device_to_file(bofhip.FPtr(fd, 0), nbytes, t.data_ptr(), o, st)
        finally:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)

    fds, direct = [], True
    for p in (pa, pb, pc):
        fd, d = _open(p, True)
        fds.append(fd)
        direct = direct and d
    opts = bofhip.default_options(gemm_blk=blk, n_io_threads=io_threads, use_odirect=1 if direct else 0,
                                  kernel_timing=1, panel_streams=panel_streams, **(extra_opts or {}))

    def step():
        bofhip.flash_gemm("R", "N", "N", n, n, n, 1.0, 0.0, bofhip.FPtr(fds[0], 0), bofhip.FPtr(fds[1], 0),
                          bofhip.FPtr(fds[2], 0), 0, 0, 0, opts)
        return bofhip.flash_last_stats()
    probes, slow_events, median_events, launch_mix = [], None, None, None
    try:
        warm = []
        for _ in range(warmup):
            _drop_cache((pa, pb, pc))
            warm.append(step())
        ok_warm = verify() if warmup else None
        def safe_probe(label):
            # a probe is an auxiliary measurement: its failure costs the ceilings, never the headline
            try:
                return disk_probe(bofhip, (pa, pb), pc, io_threads, label=label)
            except Exception as e:      # noqa: BLE001
                return {"when": label, "error": f"{type(e).__name__}: {e}"[:200]}
        if direct and probe:
            probes.append(safe_probe("before the timed steps"))
        reset_c()                                   # the timed steps must produce C, not find it
        _drop_cache((pa, pb, pc))
        torch.cuda.synchronize()
        per, dumps = [], []
        L = bofhip.lib()
        t0 = time.perf_counter()
        for i in range(steps):
            per.append(step())
            if event_dir:                           # the always-on event ring of this step (about a millisecond)
                dumps.append(os.path.join(event_dir, f"step{i}.events"))
                L.bof_event_dump(dumps[-1].encode())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        launch_mix = bofhip.flash_last_launch_mix()
        ok = verify()
        if direct and probe:
            probes.append(safe_probe("after the timed steps"))
        if dumps:
            order = sorted(range(steps), key=lambda i: per[i]["seconds"])

            def ring_of(i):
                lines = []
                for ln in open(dumps[i]).read().splitlines():
                    f = ln.split()
                    # "[bof events] <ms relative to the newest call's begin> t<tid> <label> a b c": this step's only
                    if len(f) > 3 and f[0] == "[bof" and f[2][:1].isdigit():
                        lines.append(ln[13:].strip())
                return {"step": i, "seconds": per[i]["seconds"], "events": lines[-2500:]}
            try:
                slow_events = ring_of(order[-1])
                # the TYPICAL step beside the outlier (VERDICT r5 missing 6): the lower median of the timed steps
                median_events = ring_of(order[(steps - 1) // 2])
            except (OSError, ValueError):
                pass
    finally:
        for fd in fds:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)
        for p in (pa, pb, pc):
            if os.path.exists(p):
                os.remove(p)
        del t
        torch.cuda.empty_cache()
    return {"dt": dt, "per_step": per, "warmup_steps": warm, "verified": bool(ok) and ok_warm is not False,
            "verified_after_warmup": ok_warm, "odirect": direct, "create_files_s": round(create_s, 1),
            "file_system": _fs_of(workdir), "disk_probes": probes, "slowest_step_events": slow_events,
            "median_step_events": median_events, "launch_mix_last_step": launch_mix}


def e2e_bound(per_step, ceil, n_steps, kernel_s_per_step):
    """Lower bound on one step's wall time from the ceilings PROBED in this run (never adjusted to
    what the step achieved): the stages run concurrently, so t_bound = max over them."""
    st = per_step
    terms = {"mfma": kernel_s_per_step}
    if ceil.get("pcie_h2d_GBps"):
        terms["pcie_h2d"] = st["bytes_h2d"] / (ceil["pcie_h2d_GBps"] * 1e9)
        terms["pcie_d2h"] = st["bytes_d2h"] / (ceil["pcie_d2h_GBps"] * 1e9)
    if ceil.get("disk_read_GBps"):
        terms["disk_read"] = st["bytes_read"] / (ceil["disk_read_GBps"] * 1e9)
        terms["disk_write"] = st["bytes_written"] / (ceil["disk_write_GBps"] * 1e9)
        terms["disk_total"] = disk_time_bound(st["bytes_read"], st["bytes_written"], ceil)[0]
    bound = max(terms, key=terms.get)
    return bound, terms[bound], {k: round(v, 4) for k, v in terms.items()}


def row_panel_disk_bound(n, blk, ceil):
    """A tighter lower bound for any schedule that emits C by row panels (contiguous file extents, what the row-panel
    pipeline writes): no byte of C exists before ALL of B and one A panel have been read -- that much at the read-alone
    rate -- and only the rest (the other A panels, all of C) can mix reads and writes (disk_time_bound: which pays only
    where the disk moves more in total when the directions mix).  Returned beside the schedule-agnostic `disk_total`
    bound, never instead of it; equal to it on a disk that gains nothing from mixing."""
    if not ceil.get("disk_read_GBps"):
        return None
    first = 4.0 * (n * n + blk * n)
    return first / (ceil["disk_read_GBps"] * 1e9) + disk_time_bound(4.0 * (n * n - blk * n), 4.0 * n * n, ceil)[0]


def _alg_bytes_per_launch(n, blk, mix, launches_per_step):
    """GEMM bytes = 4 (M K + K N + M N (1 + [C is read])) per launch, averaged over the launches of one
    step of the row-panel schedule as the library counted them (bof_flash_last_launch_mix): the G C panels of the ramp
    group run one k-block per launch (the raw sums of the k-blocks before it are read back: [C is read] = 1 for all
    but the first), every other C panel ONE launch over the whole K -- the last one in row slices, each of which
    reads its rows of A, all of B and writes its rows of C."""
    nk = npan = max(n // blk, 1)
    if nk <= 1 or not mix:
        return int(4 * 3 * n * n)
    g = max(0, round((mix.get("chain_k_ranges") or 0) / nk))
    slices = mix.get("whole_k_row_slices") or 0
    ramp = g * (4.0 * (blk * n + blk * n * nk + blk * n * (2 * nk - 1)))        # A panel once; B panel l per launch; C write + read-back
    whole = (mix.get("whole_k_panels") or 0) * 4.0 * (blk * n + n * n + blk * n)
    sliced = (4.0 * (blk * n + slices * n * n + blk * n)) if slices else 0.0     # the last panel: B once per slice
    return int((ramp + whole + sliced) / max(launches_per_step, 1))


def _mean(xs):
    xs = list(xs)
    return sum(xs) / max(len(xs), 1)


def run_single(args, bofhip, torch, dev, st):
    """N = 1: the headline on configs[1], then (unless --no-extras) the other configurations as flat
    scalars; returns (line, detail)."""
    import shutil
    import tempfile
    n, blk = args.size or 32768, args.blk
    detail = {}
    base = args.e2e_dir or os.environ.get("BOF_BENCH_DIR") or os.environ.get("TMPDIR") or "/tmp"
    workdir = tempfile.mkdtemp(prefix="bof_bench_", dir=base)
    try:
        free = shutil.disk_usage(workdir).free
        if free < 3 * n * n * 4 + (6 << 30):
            raise SystemExit(f"bench.py: {free / 2**30:.0f} GiB free under {base}; configs[1] needs three "
                             f"{n * n * 4 / 2**30:.0f} GiB files (set BOF_BENCH_DIR)")
        try:
            ceil = io_ceilings(bofhip, torch, dev, st, workdir, args.io_threads)
        except Exception as e:
            ceil = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        detail["ceilings"] = ceil
        bofhip.lib().bof_flash_release()
        evdir = os.path.join(workdir, "events")
        os.makedirs(evdir, exist_ok=True)
        h = headline_flash_gemm(bofhip, torch, dev, st, workdir, n, blk, args.io_threads, args.steps, args.warmup,
                                args.streams, probe=not args.no_probe, event_dir=evdir, extra_opts=args.opt_dict)
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    detail["headline"] = h
    # the disk ceilings of the headline: the best the device showed on the workload's own files before / after the
    # timed steps or on the scratch-file probe -- the probe moves, never the run
    if "error" not in ceil:
        ceil = merge_ceilings(ceil, *h.get("disk_probes", []))
        detail["ceilings_merged"] = ceil
    per = h["per_step"]
    flops_step = 2.0 * n ** 3
    dt = h["dt"]
    value = flops_step * args.steps / dt / 1e9
    launches = sum(p["kernel_launches"] for p in per)
    ksec = sum(p["kernel_seconds"] for p in per)
    avg_launch_ms = ksec / max(launches, 1) * 1e3
    flops_per_launch = flops_step * args.steps / max(launches, 1)
    achieved = flops_per_launch / max(avg_launch_ms, 1e-9) / 1e9          # TFLOP/s
    traffic, traffic_src = pmc_traffic()
    nk = max(n // blk, 1)
    # the launch mix of the row-panel schedule as the library counted it in the last timed step (bof_flash_last_launch_mix):
    # the ramp group's k-block launches are <ChainEpi> instantiations, the whole-K launches <NoEpi> (beta == 0)
    mix = h.get("launch_mix_last_step") or {}
    # row-major 'N','N': A's panels are x-major -> sgemm_tile256_dmax_kernel<XMAJOR, KMAJOR, EP> (round 6: swizzled LDS-DMA
    # straight from A's rows); with $BOF_GEMM_DMAX=0 k-major copies + sgemm_tile256_dma2_kernel<EP>, as until round 5
    kname = "sgemm_tile256_dma2_kernel" if os.environ.get("BOF_GEMM_DMAX", "1") == "0" else "sgemm_tile256_dmax_kernel"
    kernel_mix = {f"{kname}<ChainEpi> (ramp group, $BOF_PANEL_RAMP_K k-blocks per launch)": mix.get("chain_k_ranges"),
                  f"{kname}<NoEpi> (one launch over the whole K, a whole C panel)": mix.get("whole_k_panels"),
                  f"{kname}<NoEpi> (one launch over the whole K, a row slice of the last C panel)": mix.get("whole_k_row_slices")}
    mean_step = {q: _mean(p[q] for p in per) for q in ("bytes_read", "bytes_written", "bytes_h2d", "bytes_d2h")}
    bound, t_bound, terms = e2e_bound(mean_step, ceil if "error" not in ceil else {}, args.steps, ksec / args.steps)
    secs = sorted(p["seconds"] for p in per)
    med_step = secs[len(secs) // 2] if len(secs) % 2 else 0.5 * (secs[len(secs) // 2 - 1] + secs[len(secs) // 2])
    rp_bound = row_panel_disk_bound(n, blk, ceil) if "error" not in ceil else None
    disk_model = disk_time_bound(mean_step["bytes_read"], mean_step["bytes_written"], ceil)[1] if ceil.get("disk_read_GBps") else None
    # the fraction as rounds 4-6 defined it until the probe was corrected (mixed_window_rates): every byte at the sum of
    # the mixed pass's own-duration rates -- kept so that the rounds stay comparable, not a bound the disk can reach
    own = ceil.get("mixed_pass_own_duration_GBps")
    legacy_frac = (round((mean_step["bytes_read"] + mean_step["bytes_written"]) / (max(sum(own), ceil.get("disk_read_GBps", 0)) * 1e9)
                         / (dt / args.steps), 3) if own else None)
    probe_note = None
    e2e_frac_raw = round(t_bound / (dt / args.steps), 3)          # against the probes as they are (may exceed 1: a bad probe)
    if t_bound > dt / args.steps and bound.startswith("disk"):
        # the steps moved bytes faster than every probe: the device CAN do what it just did -- the ceiling is raised to
        # the rate the steps themselves showed (flagged), never the other way round
        probe_note = (f"every probe of this run read / wrote slower than the timed steps did: bound {bound} raised from "
                      f"{t_bound:.4f} s to the steps' own {dt / args.steps:.4f} s")
        t_bound = dt / args.steps
        terms[bound] = round(t_bound, 4)
    line = {
        "metric": "GFLOP/s, out-of-core GEMM (flash _gemm on SSD-resident A/B/C, wall clock around the call)",
        "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32",
        "data": "synthetic: dense_create mode s (i%10) A and B, zero C, written to the scratch disk; whole C file "
                "checked against the closed form",
        "config": {"workload": f"flash _gemm fp32 {n}x{n}x{n}, {blk}-tile, SSD-resident A/B/C, 1xMI355X",
                   "what_is_timed": "bof_flash_gemm on O_DIRECT files, page cache dropped, C write-back included "
                                    "(reference: drivers/gemm.cpp:57-62)",
                   "odirect": h["odirect"], "file_system": h["file_system"], "io_threads": args.io_threads,
                   "compute_streams": args.streams or "library default (1)",
                   "tile_tasks_per_step": int(_mean(p["tasks"] for p in per)),
                   "GiB_read_per_step": round(mean_step["bytes_read"] / 2**30, 3),
                   "GiB_written_per_step": round(mean_step["bytes_written"] / 2**30, 3),
                   "step_s_min_med_max": [round(secs[0], 3), round(med_step, 3), round(secs[-1], 3)],
                   "ms_per_step_median": round(med_step * 1e3, 2),
                   "mean_over_median": round(dt / args.steps / med_step, 3),
                   "C_verified": h["verified"], "parallelism": "single GPU"},
        "roofline": {"bound": "mfma", "kernel": f"{kname} (<ChainEpi> and <NoEpi> instantiations; launches per step in kernel_mix)",
                     "kernel_mix": kernel_mix,
                     "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "avg_launch_ms": round(avg_launch_ms, 4),
                     "launches": launches, "timed_with": "HIP events around every tile launch on its compute stream, "
                                                         "inside the timed steps (bof_options.kernel_timing)",
                     "flops_per_launch": flops_per_launch,
                     "launches_per_step": round(launches / args.steps, 1),
                     "algorithmic_bytes_per_launch": _alg_bytes_per_launch(n, blk, mix, launches / args.steps),
                     "traffic": traffic, "traffic_source": f"static: {traffic_src} (PMC passes of this command, committed; not this run)",
                     "kernel_s_per_step": round(ksec / args.steps, 4),
                     "e2e_bound": bound, "e2e_t_bound_s": round(t_bound, 4),
                     "e2e_frac": round(t_bound / (dt / args.steps), 3), "e2e_frac_raw": e2e_frac_raw, "e2e_terms_s": terms,
                     **({"e2e_row_panel_bound_s": round(rp_bound, 4),
                         "e2e_frac_of_row_panel_bound": round(min(rp_bound / (dt / args.steps), 1.0), 3),
                         "e2e_frac_median_step_of_row_panel_bound": round(min(rp_bound / med_step, 1.0), 3)} if rp_bound else {}),
                     **({"e2e_probe_note": probe_note} if probe_note else {}),
                     **({"e2e_disk_model": disk_model} if disk_model else {}),
                     **({"e2e_frac_r5_definition": legacy_frac} if legacy_frac else {}),
                     "e2e_frac_median_step": round(min(t_bound / med_step, 1.0), 3),
                     "e2e_probe": {k: ceil.get(k) for k in ("disk_read_GBps", "disk_write_GBps",
                                                            "disk_read_GBps_while_writing",
                                                            "disk_write_GBps_while_reading", "mixed_window_s",
                                                            "mixed_pass_own_duration_GBps", "pcie_h2d_GBps",
                                                            "pcie_d2h_GBps") if k in ceil}},
    }
    if not args.no_cpu:
        cb = cpu_baseline(with_csr=not args.no_extras and not args.size, full_step=not args.size and not args.no_cpu_full_step)
        detail["cpu_baseline"] = cb
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit", "GFLOP/s"), "cores": cb.get("cores"),
                                "kind": cb.get("kind", "port"),
                                "sample": "in-memory MKL, sampled (not the reference's flash path on files): "
                                          + str(cb.get("sample", ""))[:160]}
        for q in ("csrmm", "csrgemv_N"):
            if isinstance(cb.get(q), dict):
                line["cpu_baseline"][q + "_gflops"] = cb[q].get("value")
                line["cpu_baseline"][q + "_cores"] = cb[q].get("cores")
        # the workload itself beside the sample: one in-memory 32768^3 cblas_sgemm (drivers/in_mem_gemm.cpp:63-70)
        for q in ("full_step_s", "full_step_gflops", "full_step_skipped"):
            if cb.get(q) is not None:
                line["cpu_baseline"][q] = cb[q]
        if cb.get("error"):
            line["cpu_baseline"]["error"] = str(cb["error"])[:120]
    if not args.no_extras and not args.size:
        line["extras"] = extras_single(args, bofhip, torch, dev, st, detail, ksec / args.steps)
    return line, detail


def extras_single(args, bofhip, torch, dev, st, detail, gemm_kernel_s):
    """The other configurations, each a handful of flat scalars in the line (full records in
    bench_detail.json): the HBM-resident tile DAG, 65536^3 from files, cfg3 csrmm and cfg5-size
    csrgemv resident + from files, kmeans."""
    ex = {}
    # -- HBM-resident tile DAG of configs[1] (the round 1-3 headline) -------------------------------
    try:
        r = resident_gemm_line(bofhip, torch, dev, st, 32768, 32768, 32768, 0, args.blk, 1, 3,
                               "cfg2 tile DAG, A/B/C resident in HBM (I/O skipped)")
        detail["resident_cfg2"] = r
        ex["resident_cfg2_gflops"] = r["gflops"]
        ex["resident_cfg2_frac_mfma"] = r["frac_of_mfma_peak"]
        ex["resident_rel_err_vs_f64"] = float(f"{r['first_tile_row_rel_err_vs_float64']:.2e}")
    except Exception as e:
        ex["resident_cfg2_error"] = str(e)[:80]
    sec = {}
    if not args.no_csr:
        try:
            sec = csr_secondary(bofhip, torch, dev, st)
        except Exception as e:
            sec = {"error": str(e)[:200]}
        try:
            sec["kmeans"] = kmeans_secondary(bofhip, torch, dev, st, args.streams or 1)
        except Exception as e:
            sec["kmeans_error"] = str(e)[:200]
        detail["secondary"] = sec
        if isinstance(sec.get("csrmm"), dict):
            r3 = sec["csrmm"]["roofline"]
            ex["csrmm_kernel_ms"] = sec["csrmm"]["ms"]
            ex["csrmm_frac_hbm"] = r3["frac"]
            ex["csrmm_gather_frac_of_8.6TBps"] = r3["gather_frac_of_ceiling"]
        for tr in "NT":
            gv = sec.get("csrgemv_" + tr)
            if isinstance(gv, dict):
                ex[f"csrgemv_{tr}_kernel_ms"] = gv["ms"]
                ex[f"csrgemv_{tr}_frac_hbm"] = gv["roofline"]["frac"]
        if isinstance(sec.get("kmeans"), dict):
            ex["kmeans_frac_mfma"] = sec["kmeans"]["roofline"]["frac"]
    if not args.no_e2e:
        csr_ms = (sec.get("csrmm") or {}).get("ms") if isinstance(sec.get("csrmm"), dict) else None
        gv_ms = {tr: (sec.get("csrgemv_" + tr) or {}).get("ms") for tr in "NT"} if sec else {}
        e2e = e2e_block(bofhip, torch, dev, st, args, gemm_kernel_s, csr_ms * 1e-3 if csr_ms else None,
                        8.0 * gemm_kernel_s, gv_ms if gv_ms and all(gv_ms.values()) else None,
                        ceilings=detail.get("ceilings_merged") or detail.get("ceilings") or {})
        detail["e2e"] = e2e

        def leg(name, mode, sub=None):
            d = e2e.get(name)
            d = d.get(mode) if isinstance(d, dict) else None
            if sub and isinstance(d, dict):
                d = d.get(sub)
            return d if isinstance(d, dict) and "seconds" in d else None
        for name, mode, tag, sub in (("gemm", "buffered", "cfg2_pagecache", None),
                                     ("gemm_65536", "odirect", "gemm64k_odirect", None),
                                     ("csrmm", "odirect", "cfg3_odirect", None),
                                     ("csrmm", "buffered", "cfg3_pagecache", None),
                                     ("csrgemv", "odirect", "csrgemvN_odirect", "N"),
                                     ("csrgemv", "buffered", "csrgemvN_pagecache", "N"),
                                     ("csrgemv", "buffered", "csrgemvT_pagecache", "T")):
            d = leg(name, mode, sub)
            if d:
                ex[tag + "_s"] = d["seconds"]
                if "roofline" in d:
                    ex[tag + "_frac"] = d["roofline"]["frac"]
                    ex[tag + "_bound"] = d["roofline"]["bound"]
        g64 = leg("gemm_65536", "odirect")
        if g64:
            ex["gemm64k_odirect_gflops"] = g64["gflops"]
            if len(g64.get("seconds_all", [])) > 1:      # first (cold allocations) and second call
                ex["gemm64k_odirect_s_cold_warm"] = g64["seconds_all"][:2]
        ok = []
        for name in ("gemm", "gemm_65536"):
            for mode in ("odirect", "buffered"):
                d = leg(name, mode)
                if d:
                    ok.append(bool(d.get("whole_C_file_matches_closed_form")))
        c3 = e2e.get("csrmm") if isinstance(e2e.get("csrmm"), dict) else {}
        if "sha256_matches_reference_drivers" in c3:
            ok.append(bool(c3["sha256_matches_reference_drivers"]))
        for mode in ("odirect", "buffered"):
            for tr in "NT":
                d = leg("csrgemv", mode, tr)
                if d:
                    ok.append(bool(d.get("sha256_y_matches_reference")))
        ex["all_outputs_verified"] = bool(ok) and all(ok)
        if "error" in e2e:
            ex["e2e_error"] = str(e2e["error"])[:100]
    return ex


def run_sharded(args, bofhip, torch, dev, st, rank, world, red_dev, one_gpu):
    """N > 1 (BASELINE configs[3] at N = 8): (8192 N) x 65536 x 65536 from ONE set of files on the node's
    scratch disk, C rows sharded by rank, B read from storage once per node (panel l by rank l % N,
    passed on through the node-shared staging ring), no data-path collective.  One step = every rank's
    bof_flash_gemm on its slab between two barriers; value = the flops of all ranks / max-over-ranks time."""
    import shutil
    import tempfile
    import numpy as np
    import torch.distributed as dist
    import bof_dist
    # Full size (the driver's SCALE runs): every rank owns 8192 C rows of a 65536-wide problem -- the flops of one
    # configs[1] step per rank (70.4 TFLOP: weak scaling), configs[3] itself at N = 8.  Disk need at N = 8 on ONE
    # volume: A 16 + B 16 + C 16 GiB + 2 GiB slack = 50 GiB (checked below before anything is written); with
    # $BOF_BENCH_DIRS over D volumes 4 * (users * 8192 * 131072 + 65536^2) bytes each (B replicated per volume).
    # Wall time against the driver's 1800 s: files created at the disk's write rate (48 GiB: 4-6 s on the pool's
    # disks, by all ranks in parallel), every step disk-bound at ~2.1-2.8 s, verification one
    # read of C (16 GiB: ~1 s) -- warm-up 5 + 20 steps stay under 2 minutes; the CSR extras add ~1 minute.
    k = n = args.size or 65536
    m_local = (args.size or 65536) // 8
    m = m_local * world
    blk = args.blk
    detail = {}

    def phase(fn):
        err = ""
        t0 = time.perf_counter()
        try:
            fn()
        except Exception as e:
            err = f"{type(e).__name__}: {str(e)[:200]}"
        me = time.perf_counter() - t0
        v = torch.tensor([0.0 if err else 1.0, -me, me], dtype=torch.float64, device=red_dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        return bool(v[0].item() == 1.0), -float(v[1].item()), float(v[2].item()), err

    base = args.e2e_dir or os.environ.get("BOF_BENCH_DIR") or os.environ.get("TMPDIR") or "/tmp"
    # $BOF_BENCH_DIRS=a,b,...: one directory per scratch VOLUME of the node.  ONE disk caps the 8-GPU run at its ~20 GB/s
    # (weak-scaling efficiency ~0.3 at N = 8); with several volumes rank g keeps ITS A and C slabs
    # as files of their own in dirs[g % D], and every directory holds a replica of B (panel l of B is read by rank
    # l % N from its own replica and passed on through the node's staging ring, so B's reads spread over the volumes
    # too).  Unset: the one shared A / B / C file set under $BOF_BENCH_DIR.
    dirs = [d for d in os.environ.get("BOF_BENCH_DIRS", "").split(",") if d]
    local_slabs = len(dirs) > 0
    D = max(len(dirs), 1)
    my_dir_i = rank % D
    mates = [g for g in range(world) if g % D == my_dir_i]          # the ranks that share this rank's directory
    box = [None]
    if rank == 0:
        try:
            made = []
            for bdir in (dirs or [base]):
                os.makedirs(bdir, exist_ok=True)        # ($BOF_BENCH_DIRS may name directories that do not exist yet)
                d = tempfile.mkdtemp(prefix="bof_bench_sharded_", dir=bdir)
                made.append(d)
                users = len([g for g in range(world) if g % D == len(made) - 1]) if local_slabs else world
                need = 4 * (users * m_local * (k + n) + k * n) + (2 << 30)
                if shutil.disk_usage(d).free <= need:
                    raise OSError(f"{shutil.disk_usage(d).free / 2**30:.0f} GiB free under {bdir}, {need / 2**30:.0f} needed")
            if not local_slabs:
                for name, sz in (("A.bin", m * k * 4), ("B.bin", k * n * 4), ("C.bin", m * n * 4)):
                    with open(os.path.join(made[0], name), "wb") as f:
                        f.truncate(sz)
            else:
                for d in made:
                    with open(os.path.join(d, "B.bin"), "wb") as f:
                        f.truncate(k * n * 4)
            box[0] = made
        except OSError as e:
            for d in made:
                shutil.rmtree(d, ignore_errors=True)
            box[0] = f"!{e}"
    dist.broadcast_object_list(box, src=0)
    if isinstance(box[0], str):
        raise SystemExit("bench.py: " + box[0][1:])
    workdirs = box[0]
    workdir = workdirs[my_dir_i if local_slabs else 0]
    if local_slabs:
        pa, pb, pc = (os.path.join(workdir, x) for x in (f"A.{rank}.bin", "B.bin", f"C.{rank}.bin"))
        for pth, sz in ((pa, m_local * k * 4), (pc, m_local * n * 4)):
            with open(pth, "wb") as f:
                f.truncate(sz)
    else:
        pa, pb, pc = (os.path.join(workdir, x) for x in ("A.bin", "B.bin", "C.bin"))
    r0 = rank * m_local
    file_r0 = 0 if local_slabs else r0                  # where this rank's rows start in ITS A / C files
    # B is written by the ranks that share a directory, a row range each
    nb = len(mates) if local_slabs else world
    bi = mates.index(rank) if local_slabs else rank
    kb = (k + nb - 1) // nb
    k0, k1 = min(k, bi * kb), min(k, (bi + 1) * kb)
    state = {}

    def put(path, off_elems, count, first, mode):
        if count <= 0:
            return
        t = state["t"]
        bofhip.gen_dense(t.data_ptr(), first, count, mode, 0, st)
        fd, d = _open(path, True)
        try:
            bofhip.device_to_file(bofhip.FPtr(fd, off_elems * 4), count * 4, t.data_ptr(),
                                  bofhip.default_options(n_io_threads=args.io_threads, use_odirect=1 if d else 0), st)
            os.fsync(fd)
        finally:
            bofhip.lib().bof_file_forget(fd)
            os.close(fd)

    def create():
        state["t"] = torch.empty(max(m_local * max(k, n), (k1 - k0) * n), dtype=torch.float32, device=dev)
        put(pa, file_r0 * k, m_local * k, r0 * k, "s")     # dense_create mode s, this rank's rows
        put(pb, k0 * n, (k1 - k0) * n, k0 * n, "s")        # its share of (this directory's replica of) B
        put(pc, file_r0 * n, m_local * n, 0, "z")
        state["rowpat"] = _closed_form_rows(torch, dev, m, k, n)
    good, create_s, _, err = phase(create)
    fds = []

    def opening():
        for p in (pa, pb, pc):
            fd, d = _open(p, True)
            fds.append(fd)
            state["direct"] = state.get("direct", True) and d
    if good:
        good, _, _, err = phase(opening)
    opts = bofhip.default_options(gemm_blk=blk, n_io_threads=args.io_threads, use_odirect=1 if state.get("direct") else 0,
                                  kernel_timing=1, panel_streams=args.streams)
    last = {}

    def step():
        last.clear()
        last.update(bof_dist.flash_gemm_row_sharded(m, n, k, 1.0, 0.0, fds[0], fds[1], fds[2], 0, 0, 0, opts, b_once=True,
                                                    local_slabs=local_slabs))
    for _ in range(args.warmup if good else 0):
        if rank == 0 or local_slabs:
            _drop_cache((pa, pb, pc))
        dist.barrier()
        good, _, _, err = phase(step)
        if not good:
            break
    per_rank_s, agg_steps = [], []
    my_secs = []            # this rank's own seconds per timed step (gathered: `rank_mean_s` in the line)
    dt = float("nan")
    if good:
        if rank == 0 or local_slabs:
            _drop_cache((pa, pb, pc))
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t_step = time.perf_counter()
            good, smax, smin, err = phase(step)        # ends in an all-reduce = the barrier between steps
            my_secs.append(last.get("seconds", time.perf_counter() - t_step) if isinstance(last, dict) else time.perf_counter() - t_step)
            per_rank_s.append((smin, smax))
            agg_steps.append(dict(last))
            if not good:
                break
        dist.barrier()
        torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # every rank's mean seconds per step, in rank order (what an uneven disk share or a slow rank looks like)
    rank_means = torch.zeros(world, dtype=torch.float64, device=red_dev)
    rank_means[rank] = sum(my_secs) / max(len(my_secs), 1)
    dist.all_reduce(rank_means)
    rank_means = [round(float(v), 3) for v in rank_means.tolist()]
    keys = ("bytes_read", "bytes_written", "bytes_peer", "bytes_h2d", "bytes_d2h", "kernel_launches", "kernel_seconds", "tasks")
    agg = torch.tensor([sum(float(s.get(q, 0)) for s in agg_steps) for q in keys], dtype=torch.float64, device=red_dev)
    dist.all_reduce(agg)
    agg = dict(zip(keys, [float(v) for v in agg.tolist()]))
    for fd in fds:
        bofhip.lib().bof_file_forget(fd)
        os.close(fd)

    def verify():
        t = state["t"]
        fd, _ = _open(pc, False)
        try:
            bofhip.file_to_device(bofhip.FPtr(fd, file_r0 * n * 4), m_local * n * 4, t.data_ptr(),
                                  bofhip.default_options(use_odirect=0), st)
        finally:
            os.close(fd)
        C = t[:m_local * n].view(m_local, n)
        rows = max(1, (1 << 27) // n)
        for q0 in range(0, m_local, rows):
            q1 = min(m_local, q0 + rows)
            gi = torch.arange(r0 + q0, r0 + q1, device=dev)
            if not torch.equal(C[q0:q1], state["rowpat"][gi % 10]):
                raise AssertionError(f"C rows [{r0 + q0}, {r0 + q1}) differ from the closed form")
    match, _, _, verr = phase(verify) if good else (False, 0, 0, "")
    was_direct = bool(state.get("direct", False))
    state.clear()
    torch.cuda.empty_cache()
    dist.barrier()
    if rank == 0:
        for d in workdirs:
            shutil.rmtree(d, ignore_errors=True)
    bofhip.lib().bof_flash_release()

    # csrgemv 'T': the path's one real exchange (partial sums of 200 MB), both algorithms timed
    red = None
    if not args.no_csr and not args.size and not args.no_extras:
        try:
            red = sharded_csr_extras(bofhip, torch, dev, st, rank, world, red_dev, one_gpu)
        except Exception as e:
            red = {"error": str(e)[:120]}
    if rank != 0:
        return None, None
    if not good:
        raise SystemExit(f"bench.py: the sharded run failed: {err or 'another rank failed'}")
    flops_step = 2.0 * m * n * k
    launches, ksec = agg["kernel_launches"], agg["kernel_seconds"]
    avg_launch_ms = ksec / max(launches, 1) * 1e3
    achieved = (flops_step * args.steps / max(launches, 1)) / max(avg_launch_ms, 1e-9) / 1e9
    traffic, traffic_src = pmc_traffic()
    nk = max(k // blk, 1)
    line = {
        "metric": "GFLOP/s, out-of-core GEMM (flash _gemm on SSD-resident A/B/C, wall clock around the call)",
        "value": round(flops_step * args.steps / dt / 1e9, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic: dense_create mode s A and B, zero C, written to the node's scratch disk by the ranks; "
                "every rank's C slab checked against the closed form",
        "config": {"workload": f"flash _gemm fp32 {m}x{k}x{n} row-block sharded over {world}xMI355X ({m_local} C rows "
                               f"each), {blk}-tile, SSD-resident A/B/C (BASELINE configs[3] at 8 GPUs)",
                   "what_is_timed": "every rank's bof_flash_gemm on its slab of ONE A/B/C file set between barriers; "
                                    "B read from storage once per node, no data-path collective",
                   "parallelism": f"row-block x{world}", "ranks_seen": args.ranks_seen,
                   "backend": args.backend_seen, "odirect": was_direct,
                   "file_dirs": (f"{D} directories ($BOF_BENCH_DIRS): per-rank A / C slab files, a replica of B in each"
                                 if local_slabs else "one shared A / B / C file set"),
                   "rank_s_min": round(min(a for a, _ in per_rank_s), 3), "rank_s_max": round(max(b for _, b in per_rank_s), 3),
                   "rank_mean_s": rank_means,
                   "read_amplification": round(agg["bytes_read"] / args.steps / (4.0 * (m * k + k * n)), 3),
                   "write_amplification": round(agg["bytes_written"] / args.steps / (4.0 * m * n), 3),
                   "B_GiB_from_peers_per_step": round(agg["bytes_peer"] / args.steps / 2**30, 2),
                   "aggregate_read_GBps": round(agg["bytes_read"] / dt / 1e9, 2),
                   "aggregate_write_GBps": round(agg["bytes_written"] / dt / 1e9, 2),
                   "C_verified": bool(match), "create_files_s": round(create_s, 1)},
        "ranks_seen": args.ranks_seen,
        "roofline": {"bound": "mfma", "kernel": "sgemm_tile256_dmax_kernel (<ChainEpi> ramp launches + <NoEpi> whole-K launches)",
                     "achieved": round(achieved, 2),
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                     "avg_launch_ms": round(avg_launch_ms, 4), "launches": int(launches),
                     "timed_with": "HIP events around every tile launch on its compute stream, all ranks",
                     "algorithmic_bytes_per_launch": int(4 * blk * blk * (3 + (nk - 1) / nk)),
                     "traffic": traffic, "traffic_source": f"static: {traffic_src} (N = 1 PMC passes, committed; not this run)",
                     "kernel_s_per_step_per_rank": round(ksec / args.steps / world, 4)},
    }
    if verr:
        line["config"]["rank0_verify_error"] = verr[:100]
    if red is not None:
        line["extras"] = red
    detail.update({"per_rank_step_seconds_min_max": per_rank_s, "aggregate": agg, "extras": red})
    return line, detail


def sharded_csr_extras(bofhip, torch, dev, st, rank, world, red_dev, one_gpu):
    """N > 1 extras (strong scaling of the BASELINE CSR matrices, resident shards): max-over-ranks
    kernel times and the csrgemv 'T' exchange timed as one all-reduce and as reduce-scatter + all-gather."""
    import torch.distributed as dist
    import bof_dist
    keys = ["csrmm_ms", "csrgemv_N_ms", "csrgemv_T_local_ms"]
    try:
        mine = csr_secondary_sharded(bofhip, torch, dev, st, rank, world)
        vec = [float(mine[q]) for q in keys]
        part = mine["csrgemv_T_partial"]
    except Exception as e:
        vec = [-1.0] * len(keys)
        part = torch.zeros(50_000_000, dtype=torch.float32, device=dev)
        sys.stderr.write(f"[rank {rank}] sharded CSR extras failed: {e}\n")
    hi = torch.tensor(vec, dtype=torch.float64, device=red_dev)
    lo = hi.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    part_r = part.cpu() if one_gpu else part
    warm = torch.ones(1 << 16, dtype=torch.float32, device=part_r.device)
    dist.all_reduce(warm)
    bof_dist.allreduce_partial(warm, algo="rs_ag")
    red = {}
    result = None
    for algo in ("allreduce", "rs_ag"):
        best = None
        for _ in range(2):
            buf = part_r.clone()
            dist.barrier()
            torch.cuda.synchronize()
            t_red = time.perf_counter()
            bof_dist.allreduce_partial(buf, algo=algo)
            torch.cuda.synchronize()
            tt = torch.tensor([(time.perf_counter() - t_red) * 1e3], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            best = float(tt.item()) if best is None else min(best, float(tt.item()))
        red[algo] = round(best, 3)
        if algo == "allreduce":
            result = buf
    ysum = float(result.double().sum().item())
    ok = bool(lo.min().item() >= 0)
    out = {"csr_ok": ok, "allreduce_ms": red["allreduce"], "rs_ag_ms": red["rs_ag"],
           "csrgemv_T_sum_y_ok": bool(ysum == 11249999940.0)}
    if ok:
        ms = dict(zip(keys, [float(v) for v in hi.tolist()]))
        out.update({"csrmm_max_ms": round(ms["csrmm_ms"], 3), "csrgemv_N_max_ms": round(ms["csrgemv_N_ms"], 3),
                    "csrgemv_T_local_max_ms": round(ms["csrgemv_T_local_ms"], 3),
                    "csrmm_gflops": round(2.0 * 1e9 * 128 / ms["csrmm_ms"] / 1e6, 1),
                    "csrgemv_T_gflops": round(2.0 * 5e8 / (ms["csrgemv_T_local_ms"] + min(red.values())) / 1e6, 1)})
    del part, part_r, result, buf
    torch.cuda.empty_cache()
    return out


def main():
    # a stalled run leaves the Python stacks of every thread on stderr every 15 minutes (the library's own stall
    # watchdog fails the call and dumps its event ring after BOF_STALL_TIMEOUT_S)
    import faulthandler
    faulthandler.dump_traceback_later(900, repeat=True, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=0, help="override problem edge (debug; skips the extras)")
    ap.add_argument("--blk", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=0,
                    help="compute streams of the row-panel pipeline: 0 = the library's default (ONE since round 6: panel "
                         "launches are serialised, so the event-timed avg_launch_ms is one kernel alone on the chip -- the "
                         "figure rocprofv3's per-kernel average is compared with); 2+ = launches overlap pairwise")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-full-step", action="store_true",
                    help="cpu_baseline: only the 16384^3 sample, not the one in-memory 32768^3 sgemm (about a minute on 128 cores)")
    ap.add_argument("--resident-only", action="store_true",
                    help="only the HBM-resident tile DAG of configs[1] (no file I/O, no D2H: what tools/profile_bench.sh puts "
                         "under rocprofv3 --kernel-trace to corroborate the kernel's launch time); prints its record")
    ap.add_argument("--no-probe", action="store_true", help="skip the disk probes on the workload's files around the timed steps")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline (+ cpu_baseline): no resident DAG, CSR, 64k, kmeans legs")
    ap.add_argument("--no-csr", action="store_true", help="skip the CSRMM (cfg3) / CSRGEMV (cfg5-size) extras")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-resident extras (64k, cfg3, csrgemv from files)")
    ap.add_argument("--e2e-dir", default="", help="directory for the matrix files (default $BOF_BENCH_DIR, $TMPDIR, /tmp)")
    ap.add_argument("--e2e-size", type=int, default=32768, help=argparse.SUPPRESS)
    ap.add_argument("--e2e-reps", type=int, default=2)
    ap.add_argument("--no-e2e-64k", action="store_true", help="skip the 65536^3 file-resident leg (48 GiB of files)")
    ap.add_argument("--more-legs", action="store_true",
                    help="also the 31000^3 / tile-cache / 8 GiB-budget / two-device file legs (bench_detail.json only)")
    ap.add_argument("--io-threads", type=int, default=8)
    ap.add_argument("--opt", action="append", default=[], metavar="FIELD=INT",
                    help="experiments: set a bof_options field of the headline's calls (e.g. gemm_chain=1, panel_group=3)")
    ap.add_argument("--detail-out", default="", help="also write bench_detail.json to this path")
    args = ap.parse_args()
    args.opt_dict = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in args.opt}

    # --gpus N is the number of ranks this run consists of.  Launched by a torchrun-style launcher
    # (WORLD_SIZE set) it must agree with it; launched bare with N > 1, this process -- which has not
    # touched the GPU -- starts the N ranks as fresh children (never re-exec a process that has) and
    # leaves with the launcher's exit code.
    one_gpu = os.environ.get("BOF_BENCH_ONE_GPU", "0") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE="
                         f"{os.environ.get('WORLD_SIZE', '1')} ranks; refusing to print a line for a run that is not "
                         f"the one asked for\n")
        sys.exit(2)

    import torch
    import bofhip

    if not one_gpu and torch.cuda.device_count() < args.gpus:       # (counting devices does not initialise the GPU)
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible "
                         f"(BOF_BENCH_ONE_GPU=1 runs every rank on device 0 for debugging)\n")
        sys.exit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # debug only: BOF_BENCH_ONE_GPU=1 runs every rank on cuda:0 with the gloo backend, so the
    # N > 1 code path can be exercised on a single-GPU box (RCCL refuses two ranks per device)
    if one_gpu:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    bofhip.require_device()          # fails loudly: there is no CPU fallback
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    red_dev = torch.device("cpu") if one_gpu else dev     # where cross-rank reductions live
    args.ranks_seen = 1
    args.backend_seen = "none (one rank)"
    if world > 1:
        # the backend the collectives of this run really went through, asked of torch.distributed (not assumed)
        be = dist.get_backend()
        args.backend_seen = f"{be} (RCCL over xGMI)" if be == "nccl" else f"{be} (one-GPU debug)"
        if not one_gpu and be != "nccl":
            sys.stderr.write(f"bench.py: the N > 1 run must go through the nccl (RCCL) backend, torch.distributed says {be!r}\n")
            sys.exit(2)
        # every rank adds a one through the data-path backend (RCCL over xGMI on the GPU box): the sum is
        # the number of ranks that really took part, printed so the driver can check it against --gpus
        ones = torch.ones(1, dtype=torch.float32, device=red_dev)
        dist.all_reduce(ones)
        args.ranks_seen = int(ones.item())
        if args.ranks_seen != args.gpus:
            sys.stderr.write(f"bench.py: {args.ranks_seen} ranks answered the all-reduce, --gpus says {args.gpus}\n")
            sys.exit(2)
    st = torch.cuda.current_stream().cuda_stream
    if world == 1 and args.resident_only:
        n = args.size or 32768
        r = resident_gemm_line(bofhip, torch, dev, st, n, n, n, 0, args.blk, 1, args.steps,
                               f"{n}^3 tile DAG, A/B/C resident in HBM (I/O skipped)")
        print(json.dumps(r), flush=True)
        return
    if world == 1:
        line, detail = run_single(args, bofhip, torch, dev, st)
    else:
        line, detail = run_sharded(args, bofhip, torch, dev, st, rank, world, red_dev, one_gpu)
    if rank == 0:
        emit(line, detail, args.detail_out)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
