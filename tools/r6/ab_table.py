#!/usr/bin/env python3
"""Table of a tools/r6/ab_headline.sh run: per variant and round, the headline's seconds per step (mean / min /
median / max), the kernel fraction, the probed disk ceilings and e2e_frac."""
import glob
import json
import os
import sys

d = sys.argv[1]
rows = {}
for p in sorted(glob.glob(os.path.join(d, "*.line"))):
    name, rnd = os.path.basename(p).rsplit(".", 2)[:2]
    try:
        ln = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        rows.setdefault(name, []).append((rnd, None, f"no line ({type(e).__name__})"))
        continue
    rows.setdefault(name, []).append((rnd, ln, ""))
print("| variant | round | ms/step mean | min / med / max (s) | kernel frac | read / write / mixed r+w GB/s | e2e_frac | verified |")
print("|---|---|---|---|---|---|---|---|")
summary = []
for name, rs in rows.items():
    meds, means = [], []
    for rnd, ln, note in rs:
        if ln is None:
            print(f"| {name} | {rnd} | {note} | | | | | |")
            continue
        cfg, rf = ln.get("config", {}), ln.get("roofline", {})
        pr = rf.get("e2e_probe", {})
        mmm = cfg.get("step_s_min_med_max") or []
        if len(mmm) == 3:
            meds.append(mmm[1])
        means.append(ln["ms_per_step"])
        print(f"| {name} | {rnd} | {ln['ms_per_step']} | {' / '.join(str(x) for x in mmm)} | {rf.get('frac')} | "
              f"{pr.get('disk_read_GBps')} / {pr.get('disk_write_GBps')} / {pr.get('disk_read_GBps_while_writing')} + "
              f"{pr.get('disk_write_GBps_while_reading')} | {rf.get('e2e_frac')} | {cfg.get('C_verified')} |")
    if means:
        summary.append((name, sum(means) / len(means), min(means), sorted(meds)[len(meds) // 2] if meds else None))
print()
print("| variant | mean of ms/step over rounds | best round | median of medians (s) |")
print("|---|---|---|---|")
for name, m, b, md in summary:
    print(f"| {name} | {m:.1f} | {b:.1f} | {md} |")
