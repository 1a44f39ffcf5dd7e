#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one line per kernel and counter:
average per launch).  usage: pmc_by_kernel.py dir [dir ...] [--match substr]"""
import csv
import glob
import os
import sys

args = sys.argv[1:]
match = None
if "--match" in args:
    i = args.index("--match")
    match = args[i + 1]
    del args[i:i + 2]
acc = {}
for d in args:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if match and match not in name:
                continue
            key = (name[:110], r["Counter_Name"])
            s = acc.setdefault(key, [0.0, set()])
            s[0] += float(r["Counter_Value"])
            s[1].add(r["Dispatch_Id"])
for (name, ctr), (tot, ids) in sorted(acc.items()):
    avg = tot / max(len(ids), 1)
    extra = ""
    if ctr == "FETCH_SIZE":
        extra = f"  = {avg * 1024 * 2 / 1e9:.2f} GB/launch (KB x 1024 x 2: gfx950 half-count)"
    if ctr == "WRITE_SIZE":
        extra = f"  = {avg * 1024 / 1e9:.2f} GB/launch"
    print(f"{name:110s} {ctr:14s} launches={len(ids):4d} avg={avg:.4g}{extra}")
