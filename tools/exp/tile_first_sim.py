#!/usr/bin/env python3
"""What would C by TILES be worth in the headline (first row)?  The two-resource model of
panel_schedule_sim.py (one disk with pure and mixed rates, one GPU running its launch list in order), generalised to
launches with arbitrary operand sets and write-backs of arbitrary size, on two schedules of the 32768^3 / 4096 step:

  panels  today: B row panels stream in while the ramp group of G C panels runs k-block by k-block; no C panel is
          complete before all of B is in; later C panels are one launch each.
  tiles   A panels 0..G-1 first, then B by COLUMN blocks (strided reads): column block j completes G C tiles
          (whole-K launches of 4096 x 4096), which leave at once; after the last column block B is resident and the
          later C panels are one launch each, as today.

Rates: R / W alone, Rmix / Wmix while reads and writes are both pending (bench.py's disk probes)."""
import sys


def simulate(fetch, launches, R, W, Rmix, Wmix, tf):
    """fetch: [(key, bytes)] in order; launches: [(deps, flops, write_bytes)] in order."""
    dt = 0.00025
    t = 0.0
    fi, frem = 0, fetch[0][1]
    have = set()
    li, lend = 0, None          # launch in flight ends at lend
    wq, wrem = [], 0.0
    busy = 0.0
    first_c = None
    while fi < len(fetch) or li < len(launches) or lend is not None or wq:
        reading, writing = fi < len(fetch), bool(wq)
        if reading:
            frem -= (Rmix if writing else R) * dt
            if frem <= 0:
                have.add(fetch[fi][0]); fi += 1
                frem = fetch[fi][1] if fi < len(fetch) else 0
        if writing:
            wq[0] -= (Wmix if reading else W) * dt
            if wq[0] <= 0:
                wq.pop(0)
        if lend is not None and t >= lend:
            wb = launches[li][2]
            if wb:
                wq.append(float(wb))
                first_c = first_c if first_c is not None else t
            li += 1; lend = None
        if lend is None and li < len(launches) and all(d in have for d in launches[li][0]):
            d = launches[li][1] / tf
            lend = t + d; busy += d
        t += dt
        if t > 20:
            break
    return t, first_c, busy


def schedules(n, blk, G):
    Np = n // blk
    pb = blk * n * 4
    fl_tile = 2.0 * blk * blk * blk
    # panels (today)
    fetch = [(("A", p), pb) for p in range(G)]
    fetch = []
    seen = set()
    launches = []
    for l in range(Np):
        for p in range(G):
            for key in (("A", p), ("B", l)):
                if key not in seen:
                    seen.add(key); fetch.append((key, pb))
            launches.append(([("A", p), ("B", l)], fl_tile * Np, pb if l == Np - 1 else 0))
    for p in range(G, Np):
        fetch.append((("A", p), pb))
        launches.append(([("A", p)], fl_tile * Np * Np, pb))
    panels = (fetch, launches)
    # tiles
    fetch = [(("A", p), pb) for p in range(G)]
    launches = []
    for j in range(Np):
        fetch.append((("Bc", j), pb))
        for p in range(G):
            launches.append(([("A", p), ("Bc", j)], fl_tile * Np, blk * blk * 4))
    for p in range(G, Np):
        fetch.append((("A", p), pb))
        launches.append(([("A", p)], fl_tile * Np * Np, pb))
    tiles = (fetch, launches)
    return panels, tiles


if __name__ == "__main__":
    n, blk, tf = 32768, 4096, 148e12
    leases = {"fast (final steps-20 run)": (22.3e9, 16.7e9, 15.8e9, 9.3e9),
              "slow (last run)": (17.8e9, 14.2e9, 11.7e9, 7.3e9),
              "lease d's slowest step": (21e9, 13e9, 13e9, 3e9)}
    for name, (R, W, Rm, Wm) in leases.items():
        print(f"== {name}: read {R/1e9:.1f}, write {W/1e9:.1f}, mixed {Rm/1e9:.1f} + {Wm/1e9:.1f} GB/s; "
              f"all bytes at the mixed rate: {(3*n*n*4)/(Rm+Wm):.3f} s")
        for G in (2, 3, 4, 6, 8):
            (pf, pl), (tfe, tl) = schedules(n, blk, G)
            a = simulate(pf, pl, R, W, Rm, Wm, tf)
            b = simulate(tfe, tl, R, W, Rm, Wm, tf)
            print(f"  G {G}: panels {a[0]:.3f} s (first C at {a[1]:.3f})   tiles {b[0]:.3f} s (first C at {b[1]:.3f})   {100*(a[0]-b[0])/a[0]:+.1f} %")
