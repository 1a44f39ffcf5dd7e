#!/usr/bin/env python3
"""Dev check of sgemm_tile256_dmax_kernel (x-major operands through XOR-swizzled LDS-DMA; $BOF_GEMM_DMAX=2: every layout
with an x-major operand, 1: 'N','N' only, 0: the register-staged kernels): bit-compare DMAX=2 with DMAX=0 on several
shapes and all layouts, then time both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def run(ta, tb, m, n, k, alpha, beta, pad=0):
    g = torch.Generator(device="cpu").manual_seed(m + 3 * n + 7 * k + ord(ta) + 2 * ord(tb))
    sa = (m, k) if ta == "N" else (k, m)
    sb = (k, n) if tb == "N" else (n, k)
    lda, ldb = sa[1] + pad, sb[1] + 2 * pad
    a = (torch.rand(sa[0], lda, generator=g) * 2 - 1).to(dev)
    b = (torch.rand(sb[0], ldb, generator=g) * 2 - 1).to(dev)
    c0 = (torch.rand(m, n, generator=g) * 2 - 1).to(dev)
    outs = []
    for flag in ("2", "0"):
        os.environ["BOF_GEMM_DMAX"] = flag
        c = c0.clone()
        bofhip.sgemm("R", ta, tb, m, n, k, alpha, a.data_ptr(), lda, b.data_ptr(), ldb, beta, c.data_ptr(), n, st)
        torch.cuda.synchronize()
        outs.append(c)
    nbad = int((outs[0].view(torch.int32) != outs[1].view(torch.int32)).sum().item())
    print(f"{ta}{tb} {m}x{n}x{k} alpha={alpha} beta={beta} pad={pad}: dmax == register-staged: {nbad == 0} ({nbad} words differ)", flush=True)
    return nbad == 0


def time_ms(fn, iters=8):
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


ok = True
for (ta, tb) in (("N", "N"), ("N", "T"), ("T", "T")):
    for shape in [(2048, 4096, 512, 1.0, 0.0), (4096, 2048, 576, 0.5, 2.0), (2100, 4200, 640, 1.0, 1.0), (2048, 4096, 512, 1.0, 0.0, 4),
                  (4096, 4096, 64 * 37, 1.0, 0.0)]:
        ok = run(ta, tb, *shape) and ok
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)
for (m, n, k) in [(4096, 4096, 4096), (4096, 32768, 4096), (4096, 32768, 32768)]:
    a = torch.empty(m * k, dtype=torch.float32, device=dev)
    b = torch.empty(k * n, dtype=torch.float32, device=dev)
    c = torch.zeros(m * n, dtype=torch.float32, device=dev)
    bofhip.gen_dense(a.data_ptr(), 0, a.numel(), "u", 1, st)
    bofhip.gen_dense(b.data_ptr(), 0, b.numel(), "u", 2, st)
    for (ta, tb) in (("N", "N"), ("N", "T"), ("T", "T"), ("T", "N")):
        for flag in ("2", "0"):
            if (ta, tb) == ("T", "N") and flag == "0":
                continue
            os.environ["BOF_GEMM_DMAX"] = flag
            lda = k if ta == "N" else m
            ldb = n if tb == "N" else k
            f = lambda: bofhip.sgemm("R", ta, tb, m, n, k, 1.0, a.data_ptr(), lda, b.data_ptr(), ldb, 0.0, c.data_ptr(), n, st)
            best = min(time_ms(f, 4 if k > 8192 else 10) for _ in range(3))
            what = "k-major DMA kernel" if (ta, tb) == ("T", "N") else ("dmax" if flag == "2" else "register-staged")
            print(f"{m}x{n}x{k} {ta}{tb} {what}: {best:.4f} ms  {2.0 * m * n * k / best / 1e9:.1f} TFLOP/s = {2.0 * m * n * k / best / 1e9 / 157.3:.4f}", flush=True)
