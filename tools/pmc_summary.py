#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs into one JSON: per counter, mean/min/max per launch
of the kernels whose name contains KERNEL.  usage: pmc_summary.py KERNEL out.json dir [dir ...]"""
import csv
import glob
import json
import os
import sys

kernel, out, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if kernel not in r["Kernel_Name"]:
                continue
            key = (r["Counter_Name"], r["Dispatch_Id"])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
        for (name, _), v in per_dispatch.items():
            acc.setdefault(name, []).append(v)
res = {"kernel": kernel, "raw": {n: {"mean_per_launch": sum(v) / len(v), "launches": len(v), "min": min(v),
                                     "max": max(v)} for n, v in sorted(acc.items())}}
raw = res["raw"]
res["command"] = ("rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                  "--no-cpu --no-csr  (one pass per counter group: FETCH_SIZE | WRITE_SIZE | SQ/GRBM)")
if "FETCH_SIZE" in raw and "WRITE_SIZE" in raw:
    fetch = raw["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * 2.0
    write = raw["WRITE_SIZE"]["mean_per_launch"] * 1024.0
    res["hbm_fetch_bytes_per_launch"] = fetch
    res["hbm_write_bytes_per_launch"] = write
    res["hbm_traffic_bytes_per_launch"] = fetch + write
    res["corrections"] = ("FETCH_SIZE x 1024 B x 2 (MI355X_MICROARCH.md HBM: gfx950 FETCH_SIZE reports exactly half "
                          "of a 16 B/lane coalesced stream, global_load and buffer_load ... lds alike); "
                          "WRITE_SIZE x 1024 B")
if "SQ_VALU_MFMA_BUSY_CYCLES" in raw and "GRBM_GUI_ACTIVE" in raw:
    res["mfma_busy_cycles_per_simd"] = raw["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / 1024.0
    res["gpu_cycles_per_launch"] = raw["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
    res["mfma_util_under_profiling"] = res["mfma_busy_cycles_per_simd"] / res["gpu_cycles_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("raw", "command", "corrections")}))
