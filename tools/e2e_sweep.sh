#!/bin/bash
# Sweep of the level-3 GEMM pipeline knobs on cfg2 files (used to pick the defaults; results are
# copied to profiles/rNN/).  Usage: tools/e2e_sweep.sh OUTDIR
out=${1:-gpurun_out/sweep}
mkdir -p "$out"
run() {  # name, env..., -- args...
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
run tiles_default X=1 -- --path 1
run panels_default X=1 -- --path 2
run panels_trace BOF_TRACE=1 -- --path 2 --reps 1
run panels_req1M BOF_IO_REQUEST_KIB=1024 -- --path 2
run panels_req2M BOF_IO_REQUEST_KIB=2048 -- --path 2
run panels_req8M BOF_IO_REQUEST_KIB=8192 -- --path 2
run panels_chunk8 X=1 -- --path 2 --chunk-mib 8
run panels_chunk16 X=1 -- --path 2 --chunk-mib 16
run panels_chunk64 X=1 -- --path 2 --chunk-mib 64
run panels_thr16 X=1 -- --path 2 --io-threads 16 --pinned 16
run panels_thr16_req2M BOF_IO_REQUEST_KIB=2048 -- --path 2 --io-threads 16 --pinned 16
run panels_thr4 X=1 -- --path 2 --io-threads 4
run panels_group2 BOF_PANEL_GROUP=2 -- --path 2
run panels_streams1 X=1 -- --path 2 --streams 1
run panels_streams2 X=1 -- --path 2 --streams 2
run panels_nonuma BOF_NUMA_BIND=0 -- --path 2
