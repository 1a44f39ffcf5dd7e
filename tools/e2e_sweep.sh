#!/bin/bash
# Knob sweeps of the level-3 GEMM pipeline on cfg2 files (tools/flash_e2e.py): how the defaults were
# picked and where the time went.  Results of the round-2 runs: profiles/r2/e2e_sweep_*.txt.
# Usage: tools/e2e_sweep.sh OUTDIR [knobs|writeback|engines ...]
out=${1:-gpurun_out/sweep}; shift
groups=${*:-knobs writeback engines}
mkdir -p "$out"
run() {  # name ENV=VAL... -- flash_e2e.py args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 600 python tools/flash_e2e.py --n 32768 "$@" > "$out/$name.json" 2> "$out/$name.err"
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    row = [sys.argv[2]]
    for m in ("odirect", "buffered"):
        if m in d and "seconds" in d[m]:
            row.append(f"{m}: {d[m]['seconds_all']} s best {d[m]['gflops']/1e3:.1f} TF ok={d[m]['whole_C_file_matches_closed_form']} req/unit={d[m]['requests_per_unit']}")
    print(" | ".join(row))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for g in $groups; do
case $g in
knobs)      # request / chunk size, thread counts, streams, NUMA binding; tile cache for comparison
  run tiles_default X=1 -- --path 1
  run panels_default X=1 -- --path 2
  run panels_trace BOF_TRACE=1 -- --path 2 --reps 1
  for r in 1024 2048 8192; do run panels_req${r}K BOF_IO_REQUEST_KIB=$r -- --path 2; done
  for c in 8 16 64; do run panels_chunk$c X=1 -- --path 2 --chunk-mib $c; done
  run panels_thr16 X=1 -- --path 2 --io-threads 16 --pinned 16
  run panels_thr4 X=1 -- --path 2 --io-threads 4
  run panels_group2 BOF_PANEL_GROUP=2 -- --path 2
  for s in 1 2 4; do run panels_streams$s BOF_PANEL_STREAMS=$s -- --path 2; done
  run panels_nonuma BOF_NUMA_BIND=0 -- --path 2 ;;
writeback)  # where does the O_DIRECT tail come from: verbose timeline, writer pool, ring depth
  run trace2_odirect BOF_TRACE=2 -- --path 2 --direct 1 --reps 1
  run trace2_buffered BOF_TRACE=2 -- --path 2 --direct 0 --reps 1
  for w in 2 8; do run writers$w BOF_PANEL_WRITERS=$w -- --path 2 --direct 1; done
  run writers8_pinned16 BOF_PANEL_WRITERS=8 -- --path 2 --direct 1 --pinned 16 ;;
r4)         # round 4: the write-back phase of the O_DIRECT call (reads end at ~0.45 s, writes trail to ~0.8 s)
  run base X=1 -- --path 2 --direct 1 --reps 3 --streams 1
  for w in 8 12 16; do run writers$w BOF_PANEL_WRITERS=$w -- --path 2 --direct 1 --reps 3 --streams 1; done
  run writers8_pinned16 BOF_PANEL_WRITERS=8 -- --path 2 --direct 1 --reps 3 --streams 1 --pinned 16
  run writers16_pinned16 BOF_PANEL_WRITERS=16 -- --path 2 --direct 1 --reps 3 --streams 1 --pinned 16
  run thr12_writers8 BOF_PANEL_WRITERS=8 -- --path 2 --direct 1 --reps 3 --streams 1 --io-threads 12 --pinned 12
  run writers8_req8M BOF_PANEL_WRITERS=8 BOF_IO_REQUEST_KIB=8192 -- --path 2 --direct 1 --reps 3 --streams 1
  run writers8_req2M BOF_PANEL_WRITERS=8 BOF_IO_REQUEST_KIB=2048 -- --path 2 --direct 1 --reps 3 --streams 1
  run writers8_uring BOF_PANEL_WRITERS=8 BOF_IO_ENGINE=uring -- --path 2 --direct 1 --reps 3 --streams 1
  run writers8_group3 BOF_PANEL_WRITERS=8 BOF_PANEL_GROUP=3 -- --path 2 --direct 1 --reps 3 --streams 1
  run writers8_trace2 BOF_PANEL_WRITERS=8 BOF_TRACE=2 -- --path 2 --direct 1 --reps 2 --streams 1
  run base_again X=1 -- --path 2 --direct 1 --reps 3 --streams 1 ;;
engines)    # kernel AIO vs io_uring (contexts / rings pooled), thread counts, request sizes
  for e in aio uring; do
    run ${e} BOF_IO_ENGINE=$e -- --path 2 --reps 3
    run ${e}_thr16 BOF_IO_ENGINE=$e -- --path 2 --direct 1 --reps 3 --io-threads 16 --pinned 16
    run ${e}_thr16_req2M BOF_IO_ENGINE=$e BOF_IO_REQUEST_KIB=2048 -- --path 2 --direct 1 --reps 3 --io-threads 16 --pinned 16
    run ${e}_tiles BOF_IO_ENGINE=$e -- --path 1 --direct 1 --reps 2
  done
  run aio_trace2 BOF_TRACE=2 BOF_IO_ENGINE=aio -- --path 2 --direct 1 --reps 2 ;;
esac
done
