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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--data", default="u")
    ap.add_argument("--shapes", default="4096x4096x4096")
    ap.add_argument("--layouts", default="NN,NT,TN,TT")
    ap.add_argument("--betas", default="0,1")
    a = ap.parse_args()
    a.shapes = [tuple(int(x) for x in s.split("x")) for s in a.shapes.split(",")]
    a.layouts = [(s[0], s[1]) for s in a.layouts.split(",")]
    a.betas = [float(x) for x in a.betas.split(",")]
    gemm(a)
