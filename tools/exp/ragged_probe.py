#!/usr/bin/env python3
"""Where does the ragged 31000-edge GEMM (paper Fig. 5 right) lose its time?  One bof_sgemm per line, HIP events:
the 256-aligned interior alone with tight and with the ragged problem's leading dimensions, then the whole ragged
problem (interior on the 256 x 256 kernel + strips on the guarded 128 x 128 kernel)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402


def time_ms(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
K = 4096
big = 31040
a = torch.empty(big * big, dtype=torch.float32, device=dev)      # large enough for every A / B / C below
b = torch.empty(32768 * 32768, dtype=torch.float32, device=dev)   # (the whole-K launches read a 32768 x 32768 B)
c = torch.zeros(big * big, dtype=torch.float32, device=dev)
bofhip.gen_dense(a.data_ptr(), 0, a.numel(), "u", 1, st)
bofhip.gen_dense(b.data_ptr(), 0, b.numel(), "u", 2, st)
for label, ta, tb, m, n, lda, ldb, ldc in [
        ("interior 30976^2, tight lds, TN (k-major x k-major)", "T", "N", 30976, 30976, 30976, 30976, 30976),
        ("interior 30976^2, lds of the 31000 problem, TN", "T", "N", 30976, 30976, 31000, 31000, 31000),
        ("interior 30976^2, lds padded to 31040 (128 B rows), TN", "T", "N", 30976, 30976, 31040, 31040, 31040),
        ("whole 31000^2, TN", "T", "N", 31000, 31000, 31000, 31000, 31000),
        ("interior 30976^2, tight, NN", "N", "N", 30976, 30976, K, 30976, 30976),
        ("interior 30976^2, ldb/ldc 31000, NN", "N", "N", 30976, 30976, K, 31000, 31000),
        ("whole 31000^2, NN", "N", "N", 31000, 31000, K, 31000, 31000),
        ("C panel 4096+100 rows x 32768+100 (tail-merged panel of a ragged problem), TN", "T", "N", 4196, 30900, 4196, 30900, 30900),
        ("30720^2 (120 x 120 tiles), tight, TN", "T", "N", 30720, 30720, 30720, 30720, 30720),
        ("32768^2, tight, TN", "T", "N", 32768 - 2048, 32768 - 2048, 32768 - 2048, 32768 - 2048, 32768 - 2048)]:
    f = lambda: bofhip.sgemm("R", ta, tb, m, n, K, 1.0, a.data_ptr(), lda, b.data_ptr(), ldb, 0.0, c.data_ptr(), ldc, st)
    ms = time_ms(f)
    print(f"{label}: {ms:.3f} ms = {2.0 * m * n * K / ms / 1e9:.1f} TFLOP/s", flush=True)
# the whole-K shapes of the panel path
for label, m, n, k in [("C panel 4096 x 32768, K = 4096 (ramp launch)", 4096, 32768, 4096),
                       ("C panel 4096 x 32768, K = 32768 (whole-K launch)", 4096, 32768, 32768),
                       ("1024-row sub-panel x 32768, K = 32768", 1024, 32768, 32768),
                       ("2048-row sub-panel x 32768, K = 32768", 2048, 32768, 32768)]:
    f = lambda: bofhip.sgemm("R", "T", "N", m, n, k, 1.0, a.data_ptr(), m, b.data_ptr(), n, 0.0, c.data_ptr(), n, st)
    ms = time_ms(f, 3)
    print(f"{label} TN: {ms:.3f} ms = {2.0 * m * n * k / ms / 1e9:.1f} TFLOP/s", flush=True)
