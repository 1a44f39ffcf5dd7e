#!/usr/bin/env python3
"""Print the per-kernel rows of a rocprofv3 *_kernel_stats.csv (name filter optional)."""
import csv
import sys

path, keys = sys.argv[1], sys.argv[2:]
for r in csv.DictReader(open(path)):
    n = r["Name"]
    if not keys or any(k in n for k in keys):
        print("%-90s calls=%5s avg_ms=%9.3f total_ms=%9.3f" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                              float(r["TotalDurationNs"]) / 1e6))
