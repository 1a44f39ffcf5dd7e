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
    # ---- transposition row: csrcsc_driver on the same files, twice -----------
    def sha(path):
        h = hashlib.sha256()
        with open(path, "rb") as f:
            while True:
                chunk = f.read(1 << 26)
                if not chunk:
                    break
                h.update(chunk)
        return h.hexdigest()

    nnz = m * npr
    q = {x: os.path.join(d, x) for x in ("T.csr", "T.col", "T.off", "U.csr", "U.col", "U.off", "Bt.bin", "Ct.bin")}
    for name, size in (("T.csr", nnz * 4), ("T.col", nnz * 8), ("T.off", (n + 1) * 8), ("U.csr", nnz * 4),
                       ("U.col", nnz * 8), ("U.off", (m + 1) * 8), ("Ct.bin", n * k * 4)):
        with open(q[name], "wb") as f:
            f.truncate(size)
    orig = {x: sha(p[x]) for x in ("A.csr", "A.col", "A.off")}
    drv = os.path.join(ROOT, "blas-on-flash_amd", "bin", "csrcsc_driver")

    def took(out, tag):
        ln = [x for x in out.splitlines() if tag in x]
        return float(ln[0].split("took")[1].split("\x1b")[0]) if ln else float("nan")

    for odirect, budget in (("0", "0"), ("1", "0"), ("0", str(8 << 30))):
        # budget 8 GiB (the reference's PROGRAM_BUDGET) forces the out-of-core scheme: row blocks
        # transposed in HBM into temporary files, column blocks merged
        env = dict(os.environ, BOF_ODIRECT=odirect, BOF_HBM_BUDGET=budget)
        r1 = subprocess.run([drv, p["A.csr"], p["A.col"], p["A.off"], q["T.csr"], q["T.col"], q["T.off"], str(m), str(n)],
                            capture_output=True, text=True, env=env)
        r2 = subprocess.run([drv, q["T.csr"], q["T.col"], q["T.off"], q["U.csr"], q["U.col"], q["U.off"], str(n), str(m)],
                            capture_output=True, text=True, env=env)
        back = {"A.csr": sha(q["U.csr"]), "A.col": sha(q["U.col"]), "A.off": sha(q["U.off"])}
        print(json.dumps({"what": "csrcsc_driver end to end (cfg3 files), A -> A^T -> A", "odirect": int(odirect),
                          "hbm_budget": int(budget),
                          "rc": [r1.returncode, r2.returncode], "csrcsc_took_s": [took(r1.stdout, "csrcsc() took"),
                                                                                  took(r2.stdout, "csrcsc() took")],
                          "transpose_of_transpose_equals_input": back == orig,
                          "input_sha256_16": {x: orig[x][:16] for x in orig}}), flush=True)
    # csrmm 'T' on A  ==  csrmm 'N' on the A^T files (B' = dense_create(10M, 128, 's'))
    bt = torch.empty(m * k, dtype=torch.float32, device=dev)
    bofhip.gen_dense(bt.data_ptr(), 0, m * k, "s", 0, st)
    torch.cuda.synchronize()
    dump(bt, q["Bt.bin"])
    del bt
    drv = os.path.join(ROOT, "blas-on-flash_amd", "bin", "csrmm_driver")
    res = {}
    for tag, args in (("T_on_A", [p["A.csr"], p["A.col"], p["A.off"], q["Bt.bin"], q["Ct.bin"], str(m), str(n), str(k),
                                  "1.0", "0.0", "T", "R"]),
                      ("N_on_At", [q["T.csr"], q["T.col"], q["T.off"], q["Bt.bin"], q["Ct.bin"], str(n), str(m), str(k),
                                   "1.0", "0.0", "N", "R"])):
        with open(q["Ct.bin"], "wb") as f:
            f.truncate(n * k * 4)
        r = subprocess.run([drv] + args, capture_output=True, text=True, env=dict(os.environ, BOF_ODIRECT="0"))
        res[tag] = {"rc": r.returncode, "csrmm_took_s": took(r.stdout, "csrmm() took"), "sha256_C": sha(q["Ct.bin"])}
    print(json.dumps({"what": "csrmm_driver trans_a=T end to end (cfg3 files)", "runs": res,
                      "T_equals_N_on_transposed_files": res["T_on_A"]["sha256_C"] == res["N_on_At"]["sha256_C"]}),
          flush=True)
    for f in list(p.values()) + list(q.values()):
        os.remove(f)


if __name__ == "__main__":
    main()
