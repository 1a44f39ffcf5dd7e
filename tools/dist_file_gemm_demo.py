#!/usr/bin/env python3
"""cfg2-size demo of the multi-rank file GEMM on whatever GPUs are visible: writes A, B (dense_create
mode 's') and a zero C, launches tools/dist_file_gemm.py with N ranks, checks C against the closed form.
usage: dist_file_gemm_demo.py DIR N_RANKS [n]   (BOF_BENCH_ONE_GPU=1 to share one GPU)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import numpy as np  # noqa: E402


def main():
    d, nproc = sys.argv[1], int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
    os.makedirs(d, exist_ok=True)
    pa, pb, pc = (os.path.join(d, x) for x in ("A.bin", "B.bin", "C.bin"))
    gen = ("import sys, torch; sys.path.insert(0, %r); import bofhip\n"
           "n = %d\n"
           "for p in (%r, %r):\n"
           "    t = torch.empty(n * n, dtype=torch.float32, device='cuda')\n"
           "    bofhip.gen_dense(t.data_ptr(), 0, t.numel(), 's', 0, torch.cuda.current_stream().cuda_stream)\n"
           "    torch.cuda.synchronize()\n"
           "    f = open(p, 'wb')\n"
           "    [f.write(t[i:i + (1 << 28)].cpu().numpy().tobytes()) for i in range(0, t.numel(), 1 << 28)]\n"
           "    f.close()\n" % (os.path.join(ROOT, "blas-on-flash_amd"), n, pa, pb))
    subprocess.run([sys.executable, "-c", gen], check=True)
    with open(pc, "wb") as f:
        f.truncate(n * n * 4)
    tool = os.path.join(ROOT, "tools", "dist_file_gemm.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", "29578", tool] if nproc > 1 else [sys.executable, tool]
    t0 = time.time()
    r = subprocess.run(cmd + [pa, pb, pc, str(n), str(n), str(n), "1.0", "0.0"], capture_output=True, text=True)
    wall = time.time() - t0
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    c = np.fromfile(pc, np.float32, count=n * 16).reshape(16, n)
    last = np.fromfile(pc, np.float32, count=n * 16, offset=(n - 16) * n * 4).reshape(16, n)
    a64 = lambda r0: ((np.arange(r0, r0 + 16)[:, None] * n + np.arange(n)[None, :]) % 10).astype(np.float64)
    b64 = ((np.arange(n)[:, None] * n + np.arange(10)[None, :]) % 10).astype(np.float64)
    ok = bool(np.array_equal(c.astype(np.float64), (a64(0) @ b64)[:, np.arange(n) % 10]) and
              np.array_equal(last.astype(np.float64), (a64(n - 16) @ b64)[:, np.arange(n) % 10]))
    print(json.dumps({"what": "row-sharded file GEMM, %d ranks" % nproc, "n": n, "rc": r.returncode,
                      "launcher_wall_s": round(wall, 2), "ranks": recs, "first_and_last_16_rows_exact": ok,
                      "b_bytes_read_total": sum(x["b_panel_rows"] for x in recs) * n * 4}))
    if r.returncode:
        print(r.stderr[-2000:])
    for p in (pa, pb, pc):
        os.remove(p)


if __name__ == "__main__":
    main()
