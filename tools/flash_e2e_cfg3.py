#!/usr/bin/env python3
"""cfg3 END TO END through the C++ driver: sparse_create(10M, 1M, 1e-4) x dense_create(1M,128,'s')
written as files (generated in HBM, byte-identical to the reference tools' output), then
`csrmm_driver <csr> <col> <off> <B> <C> 10000000 1000000 128 1 0 N R`, then sha256(C file) against
the hash the reference's own drivers produced."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blas-on-flash_amd"))
import torch  # noqa: E402
import bofhip  # noqa: E402

GOLD = "d08df7c04907bec66f4638df05ffe2bbfb447c2a01e2bc03e5e6fd1d8daf2382"


def dump(t, path):
    with open(path, "wb") as f:
        flat = t.view(-1)
        step = 1 << 27
        for i in range(0, flat.numel(), step):
            f.write(flat[i:i + step].cpu().numpy().tobytes())


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/bof_cfg3"
    scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    os.makedirs(d, exist_ok=True)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    m, n, k, npr = 10_000_000 // scale, 1_000_000, 128, 100
    t0 = time.time()
    val = torch.empty(m * npr, dtype=torch.float32, device=dev)
    col = torch.empty(m * npr, dtype=torch.int64, device=dev)
    off = torch.empty(m + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, m, 1_000_000):
        r = min(1_000_000, m - r0)
        bofhip.gen_sparse_rows(r0, r, n, npr, val.data_ptr() + 4 * r0 * npr, col.data_ptr() + 8 * r0 * npr,
                               off.data_ptr() + 8 * r0, st)
    b = torch.empty(n * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(b.data_ptr(), 0, n * k, "s", 0, st)
    torch.cuda.synchronize()
    p = {x: os.path.join(d, x) for x in ("A.csr", "A.col", "A.off", "B.bin", "C.bin")}
    dump(val, p["A.csr"]); dump(col, p["A.col"]); dump(off, p["A.off"]); dump(b, p["B.bin"])
    with open(p["C.bin"], "wb") as f:
        f.truncate(m * k * 4)
    del val, col, off, b
    torch.cuda.empty_cache()
    os.sync()
    print(f"inputs written in {time.time() - t0:.1f} s", flush=True)
    drv = os.path.join(ROOT, "blas-on-flash_amd", "bin", "csrmm_driver")
    for odirect in ("0", "1"):
        env = dict(os.environ, BOF_ODIRECT=odirect)
        t0 = time.time()
        r = subprocess.run([drv, p["A.csr"], p["A.col"], p["A.off"], p["B.bin"], p["C.bin"], str(m), str(n),
                            str(k), "1.0", "0.0", "N", "R"], capture_output=True, text=True, env=env)
        wall = time.time() - t0
        took = [ln for ln in r.stdout.splitlines() if "csrmm() took" in ln]
        secs = float(took[0].split("took")[1].split("\x1b")[0]) if took else float("nan")
        h = hashlib.sha256()
        with open(p["C.bin"], "rb") as f:
            while True:
                chunk = f.read(1 << 26)
                if not chunk:
                    break
                h.update(chunk)
        out = {"what": "csrmm_driver end to end (cfg3 files)", "scale": scale, "odirect": int(odirect),
               "rc": r.returncode, "csrmm_took_s": secs, "process_wall_s": round(wall, 2),
               "gflops": round(2.0 * m * npr * k / secs / 1e9, 1) if secs == secs else None,
               "sha256_C": h.hexdigest(), "matches_reference_hash": (h.hexdigest() == GOLD) if scale == 1 else None}
        print(json.dumps(out), flush=True)
    for f in p.values():
        os.remove(f)


if __name__ == "__main__":
    main()
