#!/bin/bash
mkdir -p gpurun_out/pre_sweep
for pre in "" alloc resident "resident,release" dgemm "alloc,resident,dgemm,release"; do
  timeout 300 python tools/flash_e2e.py --n 32768 --path 2 --reps 2 --direct 0 --pre "$pre" > gpurun_out/pre_sweep/out.json 2> gpurun_out/pre_sweep/err.txt
  python - "$pre" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/pre_sweep/out.json").read().strip().splitlines()[-1]); print(repr(sys.argv[1]), d["buffered"]["seconds_all"], d["buffered"]["first_run_cold_cache_s"])
except Exception as e: print(repr(sys.argv[1]), "FAILED", e)
PY
done
