#!/usr/bin/env python3
"""Kernel micro-benchmarks (dev tool): one tile-level launch per measurement, HIP
events on the launch stream, interleaved rounds in one process."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402


def time_ms(fn, iters):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm(args):
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (m, n, k) in args.shapes:
        a = torch.empty(m * k, dtype=torch.float32, device=dev)
        b = torch.empty(k * n, dtype=torch.float32, device=dev)
        c = torch.zeros(m * n, dtype=torch.float32, device=dev)
        bofhip.gen_dense(a.data_ptr(), 0, a.numel(), args.data, 1, st)
        bofhip.gen_dense(b.data_ptr(), 0, b.numel(), args.data, 2, st)
        for (ta, tb) in args.layouts:
            for beta in args.betas:
                lda = k if ta == "N" else m
                ldb = n if tb == "N" else k
                f = lambda: bofhip.sgemm("R", ta, tb, m, n, k, 1.0, a.data_ptr(), lda, b.data_ptr(),
                                         ldb, beta, c.data_ptr(), n, st)
                best = min(time_ms(f, args.iters) for _ in range(args.rounds))
                print(f"gemm {m}x{n}x{k} {ta}{tb} beta={beta}: {best:.4f} ms  "
                      f"{2.0 * m * n * k / best / 1e9:.1f} TFLOP/s", flush=True)


def csr(args):
    """cfg3 CSRMM (10M x 1M, 1e9 nnz, k=128) and cfg5 CSRGEMV (50M, 5e8 nnz), resident."""
    import numpy as np
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    scale = args.scale
    m, n, k, npr = 10_000_000 // scale, 1_000_000, 128, 100
    if args.csr_shape:
        m, n, npr = (int(v) for v in args.csr_shape.split("x"))
    val = torch.empty(m * npr, dtype=torch.float32, device=dev)
    col = torch.empty(m * npr, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 1_000_000):
        r = min(1_000_000, m - r0)
        bofhip.gen_sparse_rows(r0, r, n, npr, val.data_ptr() + 4 * r0 * npr, col.data_ptr() + 8 * r0 * npr,
                               off.data_ptr() + 8 * r0, st)
    b = torch.empty(n * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(b.data_ptr(), 0, n * k, args.data, 3, st)
    c = torch.zeros(m * k, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ia = off.cpu().numpy()
    opts = bofhip.default_options(n_streams=args.streams)
    f = lambda: bofhip.csrmm_resident("N", m, n, k, 1.0, 0.0, val.data_ptr(), ia.ctypes.data, off.data_ptr(),
                                      col.data_ptr(), "R", b.data_ptr(), c.data_ptr(), opts, st)
    best = min(time_ms(f, 3) for _ in range(args.rounds))
    nnz = m * npr
    alg = nnz * 12 + (m + 1) * 8 + 4 * n * k + 4 * m * k
    print(f"csrmm {m}x{n} nnz={nnz} k={k}: {best:.3f} ms  {2.0 * nnz * k / best / 1e6:.1f} GFLOP/s  "
          f"algorithmic {alg / best / 1e6:.1f} GB/s  gather {nnz * k * 4 / best / 1e6:.1f} GB/s", flush=True)
    del val, col, off, b, c
    if args.csr_shape:
        return
    m = n = 50_000_000 // scale
    npr = 10
    val = torch.empty(m * npr, dtype=torch.float32, device=dev)
    col = torch.empty(m * npr, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 5_000_000):
        r = min(5_000_000, m - r0)
        bofhip.gen_sparse_rows(r0, r, n, npr, val.data_ptr() + 4 * r0 * npr, col.data_ptr() + 8 * r0 * npr,
                               off.data_ptr() + 8 * r0, st)
    x = (torch.arange(n, device=dev) % 10).float()
    y = torch.zeros(m, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ia = off.cpu().numpy()
    for tr in "NT":
        f = lambda: bofhip.csrgemv_resident(tr, m, n, val.data_ptr(), ia.ctypes.data, off.data_ptr(),
                                            col.data_ptr(), x.data_ptr(), y.data_ptr(), opts, st)
        best = min(time_ms(f, 3) for _ in range(args.rounds))
        nnz = m * npr
        alg = nnz * 12 + (m + 1) * 8 + 8 * n
        print(f"csrgemv {tr} {m}x{n} nnz={nnz}: {best:.3f} ms  {2.0 * nnz / best / 1e6:.1f} GFLOP/s  "
              f"algorithmic {alg / best / 1e6:.1f} GB/s", flush=True)


def csrcsc(args):
    """cfg3-size transposition (10M x 1M, 1e9 nnz) and the csrmm 'T' built on it."""
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    m, n, k, npr = 10_000_000 // args.scale, 1_000_000, 128, 100
    if args.csr_shape:
        m, n, npr = (int(v) for v in args.csr_shape.split("x"))
    nnz = m * npr
    val = torch.empty(nnz, dtype=torch.float32, device=dev)
    col = torch.empty(nnz, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 1_000_000):
        r = min(1_000_000, m - r0)
        bofhip.gen_sparse_rows(r0, r, n, npr, val.data_ptr() + 4 * r0 * npr, col.data_ptr() + 8 * r0 * npr,
                               off.data_ptr() + 8 * r0, st)
    vt = torch.empty_like(val); ct = torch.empty_like(col)
    pt = torch.empty(n + 1, dtype=torch.int64, device=dev)
    f = lambda: bofhip.scsrcsc(m, n, nnz, val.data_ptr(), off.data_ptr(), col.data_ptr(), vt.data_ptr(),
                               pt.data_ptr(), ct.data_ptr(), st)
    best = min(time_ms(f, 2) for _ in range(args.rounds))
    alg = nnz * 24 + (m + n + 2) * 8   # read A once, write A^T once
    print(f"csrcsc {m}x{n} nnz={nnz}: {best:.3f} ms  algorithmic {alg / best / 1e6:.1f} GB/s  "
          f"workspace {bofhip.lib().bof_csrcsc_workspace_bytes(n, nnz) / 2**30:.2f} GiB", flush=True)
    if args.csr_shape:
        return
    b = torch.empty(m * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(b.data_ptr(), 0, m * k, args.data, 3, st)
    c = torch.zeros(n * k, dtype=torch.float32, device=dev)
    ia = off.cpu().numpy()
    opts = bofhip.default_options(n_streams=args.streams)
    f = lambda: bofhip.csrmm_resident("T", m, n, k, 1.0, 0.0, val.data_ptr(), ia.ctypes.data, off.data_ptr(),
                                      col.data_ptr(), "R", b.data_ptr(), c.data_ptr(), opts, st)
    best = min(time_ms(f, 2) for _ in range(args.rounds))
    print(f"csrmm T {m}x{n} nnz={nnz} k={k}: {best:.3f} ms  {2.0 * nnz * k / best / 1e6:.1f} GFLOP/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--data", default="u")
    ap.add_argument("--shapes", default="4096x4096x4096")
    ap.add_argument("--layouts", default="NN,NT,TN,TT")
    ap.add_argument("--betas", default="0,1")
    ap.add_argument("--what", default="gemm")
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--csr-shape", default="", help="csrcsc mode: MxNxNNZ_PER_ROW")
    a = ap.parse_args()
    a.shapes = [tuple(int(x) for x in s.split("x")) for s in a.shapes.split(",")]
    a.layouts = [(s[0], s[1]) for s in a.layouts.split(",")]
    a.betas = [float(x) for x in a.betas.split(",")]
    if a.what == "gemm":
        gemm(a)
    elif a.what == "csrcsc":
        csrcsc(a)
    else:
        csr(a)
