#!/bin/bash
# A/B of the round-6 schedule changes of the row-panel pipeline (start-up starvation and the
# write-back tail that a fast disk / the page cache expose): row slices of the whole-K panel launches
# ($BOF_PANEL_SLICES) and the anti-diagonal order of the ramp group ($BOF_PANEL_RAMP_ORDER), against the round-5
# schedule, on cfg2 from the page cache, cfg2 from O_DIRECT files and 65536^3 from O_DIRECT files.
# Usage: tools/r6/sched_ab.sh OUTDIR [ROUNDS]
out=${1:-gpurun_out/r6_sched}; rounds=${2:-2}
mkdir -p "$out"
run() {  # name round ENV... -- args
  name=$1; r=$2; shift 2
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 900 python tools/flash_e2e.py "$@" > "$out/$name.$r.json" 2> "$out/$name.$r.err"
  python3 - "$out/$name.$r.json" "$name.$r" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for m in ("odirect", "buffered"):
        if m in d and "seconds_all" in d[m]:
            print(f"{sys.argv[2]} {m}: {d[m]['seconds_all']} s  ok={d[m].get('whole_C_file_matches_closed_form')}", flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
}
old="BOF_PANEL_SLICES=1 BOF_PANEL_RAMP_ORDER=0"
for r in $(seq 1 "$rounds"); do
  run pc32k_old $r $old -- --n 32768 --direct 0 --path 2 --streams 1 --reps 4
  run pc32k_new $r X=1 -- --n 32768 --direct 0 --path 2 --streams 1 --reps 4
  run od32k_old $r $old -- --n 32768 --direct 1 --path 2 --streams 1 --reps 4
  run od32k_new $r X=1 -- --n 32768 --direct 1 --path 2 --streams 1 --reps 4
  run od64k_old $r $old -- --n 65536 --direct 1 --path 2 --streams 1 --reps 3
  run od64k_slices $r BOF_PANEL_RAMP_ORDER=0 -- --n 65536 --direct 1 --path 2 --streams 1 --reps 3
  run od64k_ramp $r BOF_PANEL_SLICES=1 -- --n 65536 --direct 1 --path 2 --streams 1 --reps 3
  run od64k_new $r X=1 -- --n 65536 --direct 1 --path 2 --streams 1 --reps 3
done
