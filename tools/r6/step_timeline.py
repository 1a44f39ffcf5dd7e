#!/usr/bin/env python3
"""Timeline of ONE headline step from the event ring bench.py keeps (bench_detail.json: headline.median_step_events /
slowest_step_events): per 50 ms window the bytes the readers finished, the bytes the writers finished, the launches
dispatched; then the phases (first byte, last read, first C handed over, last write) and the rates inside them.
Usage: step_timeline.py bench_detail.json [median|slowest] [window_ms]"""
import json
import sys
from collections import defaultdict


def parse(events):
    out = []
    for ln in events:
        f = ln.split()
        try:
            t = float(f[0])
        except (ValueError, IndexError):
            continue
        out.append((t, f[1], " ".join(f[2:-3]), int(f[-3]), int(f[-2]), int(f[-1])))
    return out


def summarize(ev, seconds, win=50.0, chunk=32 << 20):
    rd_end = [e for e in ev if e[2] == "panel chunk read end"]
    rd_beg = [e for e in ev if e[2] == "panel chunk read begin"]
    wr_beg = [e for e in ev if e[2] == "C chunk D2H complete, write begin"]
    wr_end = [e for e in ev if e[2] == "C chunk write end"]
    launch = [e for e in ev if e[2] == "launch"]
    ready = [e for e in ev if e[2] == "panel in HBM (ready recorded)"]
    handed = [e for e in ev if e[2] == "C panel handed to the flusher"]
    rows = defaultdict(lambda: [0, 0, 0])
    for e in rd_end:
        rows[int(e[0] // win)][0] += chunk
    for e in wr_end:
        rows[int(e[0] // win)][1] += chunk
    for e in launch:
        rows[int(e[0] // win)][2] += 1
    lines = [f"step of {seconds:.3f} s; {len(rd_end)} chunk reads, {len(wr_end)} chunk writes, {len(launch)} launches",
             "| window (ms) | read GB/s | write GB/s | read + write | launches dispatched |", "|---|---|---|---|---|"]
    for w in range(0, int(seconds * 1e3 // win) + 1):
        r, wv, l = rows[w]
        lines.append(f"| {int(w * win)}-{int((w + 1) * win)} | {r / win / 1e6:.1f} | {wv / win / 1e6:.1f} | {(r + wv) / win / 1e6:.1f} | {l} |")
    if rd_end and wr_end:
        t_r0, t_r1 = rd_beg[0][0], rd_end[-1][0]
        t_w0, t_w1 = wr_beg[0][0], wr_end[-1][0]
        nb_r, nb_w = len(rd_end) * chunk, len(wr_end) * chunk
        rd_before = sum(chunk for e in rd_end if e[0] <= t_w0)
        rd_during = nb_r - rd_before
        wr_during = sum(chunk for e in wr_end if e[0] <= t_r1)
        wr_after = nb_w - wr_during
        lines += ["", f"reads {t_r0:.0f}-{t_r1:.0f} ms ({nb_r / 1e9:.2f} GB, {nb_r / (t_r1 - t_r0) / 1e6:.1f} GB/s overall); "
                      f"writes {t_w0:.0f}-{t_w1:.0f} ms ({nb_w / 1e9:.2f} GB, {nb_w / (t_w1 - t_w0) / 1e6:.1f} GB/s overall)",
                  f"phase 1, reads alone (0-{t_w0:.0f} ms): {rd_before / 1e9:.2f} GB at {rd_before / max(t_w0 - t_r0, 1e-9) / 1e6:.1f} GB/s",
                  f"phase 2, both ({t_w0:.0f}-{t_r1:.0f} ms): reads {rd_during / 1e9:.2f} GB at {rd_during / max(t_r1 - t_w0, 1e-9) / 1e6:.1f} GB/s + "
                  f"writes {wr_during / 1e9:.2f} GB at {wr_during / max(t_r1 - t_w0, 1e-9) / 1e6:.1f} GB/s",
                  f"phase 3, writes alone ({t_r1:.0f}-{t_w1:.0f} ms): {wr_after / 1e9:.2f} GB at {wr_after / max(t_w1 - t_r1, 1e-9) / 1e6:.1f} GB/s"]
    if ready:
        lines.append("panels in HBM (mat panel @ ms): " + " ".join(f"{'ABC'[e[3]]}{e[4]}@{e[0]:.0f}" for e in ready))
    if handed:
        lines.append("C panels handed to the flusher (@ ms): " + " ".join(f"C{e[3]}@{e[0]:.0f}" for e in handed))
    if launch:
        lines.append("launches dispatched (C panel:first k-row @ ms): " + " ".join(f"{e[3]}:{e[4] & ~1 if False else e[4]}@{e[0]:.0f}" for e in launch))
    return "\n".join(lines)


def main():
    d = json.load(open(sys.argv[1]))
    which = sys.argv[2] if len(sys.argv) > 2 else "median"
    win = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
    h = d["headline"]
    rec = h.get(f"{which}_step_events") or h.get("slowest_step_events")
    print(f"## {sys.argv[1]} -- {which} step (step {rec['step']})")
    print(summarize(parse(rec["events"]), rec["seconds"], win))


if __name__ == "__main__":
    main()
