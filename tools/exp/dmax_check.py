#!/usr/bin/env python3
"""Dev check of sgemm_tile256_dmax_kernel ($BOF_GEMM_DMAX=1: 'N','N' with A through swizzled LDS-DMA): bit-compare with the
register-staged kernel ($BOF_GEMM_DMAX=0) on several shapes, then time both (and 'T','N' through the k-major DMA kernel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def run(m, n, k, alpha, beta, lda_pad=0, ldb_pad=0):
    g = torch.Generator(device="cpu").manual_seed(m + 3 * n + 7 * k)
    lda, ldb = k + lda_pad, n + ldb_pad
    a = (torch.rand(m, lda, generator=g) * 2 - 1).to(dev)
    b = (torch.rand(k, ldb, generator=g) * 2 - 1).to(dev)
    c0 = (torch.rand(m, n, generator=g) * 2 - 1).to(dev)
    outs = []
    for flag in (os.environ.get("DMAX_VARIANT", "1"), "0"):
        os.environ["BOF_GEMM_DMAX"] = flag
        c = c0.clone()
        bofhip.sgemm("R", "N", "N", m, n, k, alpha, a.data_ptr(), lda, b.data_ptr(), ldb, beta, c.data_ptr(), n, st)
        torch.cuda.synchronize()
        outs.append(c)
    same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    nbad = int((outs[0].view(torch.int32) != outs[1].view(torch.int32)).sum().item())
    print(f"{m}x{n}x{k} alpha={alpha} beta={beta} pads=({lda_pad},{ldb_pad}): dmax == 1w3: {same} ({nbad} words differ)", flush=True)
    if not same:
        d = (outs[0] - outs[1]).abs()
        idx = torch.nonzero(d > 0)[:5].tolist()
        print("   first differing positions:", idx, "max abs diff", float(d.max()), flush=True)
    return same


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
for shape in [(2048, 4096, 512, 1.0, 0.0), (4096, 2048, 576, 0.5, 2.0), (2304, 4096, 1024, 1.0, 0.0), (2100, 4200, 640, 1.0, 1.0),
              (4096, 4096, 64 * 37, 1.0, 0.0), (2048, 4096, 512, 1.0, 0.0, 4, 8), (4096, 4096, 4096, 1.0, 0.0)]:
    ok = run(*shape) and ok
print("ALL BIT-EQUAL" if ok else "MISMATCH", flush=True)
for (m, n, k) in [(4096, 4096, 4096), (4096, 32768, 4096), (4096, 32768, 32768)]:
    a = torch.empty(m * k, dtype=torch.float32, device=dev)
    b = torch.empty(k * n, dtype=torch.float32, device=dev)
    c = torch.zeros(m * n, dtype=torch.float32, device=dev)
    bofhip.gen_dense(a.data_ptr(), 0, a.numel(), "u", 1, st)
    bofhip.gen_dense(b.data_ptr(), 0, b.numel(), "u", 2, st)
    for name, flag, ta in (("NN dmax (b128 + pick)", "1", "N"), ("NN dmax (read2_b32 + xor)", "2", "N"), ("NN dmax (2 x b32, per-group bases)", "3", "N"), ("NN 1w3", "0", "N"), ("TN dma2", "0", "T")):
        os.environ["BOF_GEMM_DMAX"] = flag
        lda = k if ta == "N" else m
        f = lambda: bofhip.sgemm("R", ta, "N", m, n, k, 1.0, a.data_ptr(), lda, b.data_ptr(), n, 0.0, c.data_ptr(), n, st)
        best = min(time_ms(f, 4 if k > 8192 else 10) for _ in range(3))
        print(f"{m}x{n}x{k} {name}: {best:.4f} ms  {2.0 * m * n * k / best / 1e9:.1f} TFLOP/s = {2.0 * m * n * k / best / 1e9 / 157.3:.4f}", flush=True)
