#!/usr/bin/env python3
"""Table of every tools/exp/handover_stress result under a directory tree (one JSON line per process and variant)."""
import glob
import json
import os
import sys

root = sys.argv[1]
rows = []
for p in sorted(glob.glob(os.path.join(root, "**", "*.json*"), recursive=True)):
    for ln in open(p):
        ln = ln.strip()
        if not ln.startswith("{") or '"handovers"' not in ln:
            continue
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        if "events" in d:
            rows.append((os.path.relpath(p, root), d))
tot = sum(d["handovers"] for _, d in rows if not d.get("no_wait"))
bad = sum(d["h2d_stream_check"]["wrong_words"] + d["compute_stream_check"]["wrong_words"] for _, d in rows if not d.get("no_wait"))
print(f"# Stand-alone hand-over stress (`tools/exp/handover_stress.hip`): {tot:,} hand-overs over {sum(1 for _, d in rows if not d.get('no_wait'))} "
      f"runs, {bad} wrong words (self-test with the wait removed excluded)\n")
print("| file | events | copy | check wgs | slots x KiB | pipelines | launcher thread | host confirm | readers / streams | seconds | hand-overs | per s | wrong words (h2d-stream check / compute-stream check) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, d in rows:
    print(f"| {name}{' (SELF-TEST: no wait)' if d.get('no_wait') else ''} | {d['events']} | {d['copy']} | {d['wgs']} | {d['slots']} x {d['slot_KiB']} | "
          f"{d.get('pipelines', 1)} | {d['launcher']} | {d['host_confirm']} | {d['readers']} / {d['streams']} | {d['seconds']:.0f} | "
          f"{d['handovers']:,} | {d['per_s']:.0f} | {d['h2d_stream_check']['wrong_words']} / {d['compute_stream_check']['wrong_words']} |")
