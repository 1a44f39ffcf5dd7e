#!/usr/bin/env python3
"""Random shapes through sgemm_tile256_dmax_kernel / sgemm_tile256_dma2_kernel (round 6) against the kernels of rounds
1-5 ($BOF_GEMM_DMAX=0, $BOF_GEMM_DMA2_SYNC=0), bit for bit: M, N around 8-40 tile rows / columns with ragged edges, K a
multiple of 64 (sometimes not: then both runs take the register-staged kernels and must still agree), all four layouts,
both orders, padded leading dimensions, alpha / beta drawn.  Usage: dmax_fuzz.py SECONDS [SEED]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n_cases, bad = time.time(), 0, 0
while time.time() - t0 < secs:
    tiles_m, tiles_n = rng.randint(8, 40), rng.randint(8, 40)
    if tiles_m * tiles_n < 128:
        tiles_n = (128 + tiles_m - 1) // tiles_m
    m = tiles_m * 256 + rng.choice([0, 0, 0, 1, 37, 128, 255])
    n = tiles_n * 256 + rng.choice([0, 0, 0, 4, 100, 129])
    k = 64 * rng.randint(8, 40) + rng.choice([0, 0, 0, 0, 32, 5])
    ordc = rng.choice("RRC")
    ta, tb = rng.choice("NT"), rng.choice("NT")
    alpha, beta = rng.choice([1.0, 0.5, -2.0]), rng.choice([0.0, 0.0, 1.0, 2.0])
    pa, pb, pc = rng.choice([0, 0, 4, 8]), rng.choice([0, 0, 4, 12]), rng.choice([0, 0, 8])
    # stored shapes: row-major op(A) is m x k; column-major stores the transposes
    if ordc == "R":
        sa = (m, k) if ta == "N" else (k, m)
        sb = (k, n) if tb == "N" else (n, k)
        sc = (m, n)
    else:
        sa = (k, m) if ta == "N" else (m, k)
        sb = (n, k) if tb == "N" else (k, n)
        sc = (n, m)
    lda, ldb, ldc = sa[1] + pa, sb[1] + pb, sc[1] + pc
    g = torch.Generator(device="cpu").manual_seed(rng.randint(0, 1 << 30))
    a = (torch.rand(sa[0], lda, generator=g) * 2 - 1).to(dev)
    b = (torch.rand(sb[0], ldb, generator=g) * 2 - 1).to(dev)
    c0 = (torch.rand(sc[0], ldc, generator=g) * 2 - 1).to(dev)
    outs = []
    for dmax, sync in (("1", "1"), ("0", "0")):
        os.environ["BOF_GEMM_DMAX"], os.environ["BOF_GEMM_DMA2_SYNC"] = dmax, sync
        c = c0.clone()
        bofhip.sgemm(ordc, ta, tb, m, n, k, alpha, a.data_ptr(), lda, b.data_ptr(), ldb, beta, c.data_ptr(), ldc, st)
        torch.cuda.synchronize()
        outs.append(c)
    nbad = int((outs[0].view(torch.int32) != outs[1].view(torch.int32)).sum().item())
    n_cases += 1
    if nbad:
        bad += 1
        print(f"MISMATCH {ordc} {ta}{tb} {m}x{n}x{k} alpha={alpha} beta={beta} pads={pa},{pb},{pc}: {nbad} words differ", flush=True)
    del a, b, c0, outs
print(f"dmax_fuzz: {n_cases} cases in {time.time() - t0:.0f} s, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
