#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs of `tools/kbench.py --what csr --rounds 1` (4 passes of
every call: 1 warm-up + 3 timed) into per-PASS sums per kernel family.
usage: pmc_csr_summary.py out.json passes dir [dir ...]"""
import csv
import glob
import json
import os
import sys

out, passes, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
FAMILIES = {
    "csrmm_rowmajor": ["csrmm_rowmajor_kernel"],
    "csrgemv_n": ["csrgemv_n_kernel"],
    # y = A^T x partitioned by column bin: everything that call launches
    "csrgemv_t_partitioned": ["tile_rows_kernel", "radix_hist_kernel", "scan_reduce_kernel", "scan_apply_kernel",
                              "radix_scatter_kernel", "offsets_by_search_kernel", "gemv_t_accumulate_kernel"],
}
acc = {f: {} for f in FAMILIES}
launches = {f: 0 for f in FAMILIES}
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            for fam, keys in FAMILIES.items():
                if any(k in r["Kernel_Name"] for k in keys):
                    acc[fam][r["Counter_Name"]] = acc[fam].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    seen.add((fam, r["Dispatch_Id"]))
        for fam, _ in seen:
            launches[fam] += 1
res = {}
for fam, counters in acc.items():
    if not counters:
        continue
    res[fam] = {name: {"sum_over_launches_of_one_pass": v / passes} for name, v in sorted(counters.items())}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        f = counters["FETCH_SIZE"] / passes * 1024.0 * 2.0
        w = counters["WRITE_SIZE"] / passes * 1024.0
        res[fam]["hbm_traffic_bytes_per_pass"] = f + w
        res[fam]["corrections"] = "FETCH_SIZE KB x 1024 x 2 (gfx950 half-count, MI355X_MICROARCH.md) + WRITE_SIZE KB x 1024"
    if "TCC_HIT_sum" in counters and "TCC_MISS_sum" in counters:
        res[fam]["l2_hit_rate"] = counters["TCC_HIT_sum"] / max(counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"], 1.0)
res["command"] = ("rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 tools/kbench.py --what csr --rounds 1 "
                  "(one pass per counter group: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum); sums divided by "
                  f"{passes} passes per call")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)} for k, v in res.items() if isinstance(v, dict)}))
