#!/bin/bash
# third sweep: AIO vs io_uring with pooled contexts/rings, thread counts, request sizes
out=${1:-gpurun_out/sweep3}
mkdir -p "$out"
run() {
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
            row.append(f"{m}: {d[m]['seconds_all']} s best {d[m]['gflops']/1e3:.1f} TF ok={d[m]['whole_C_file_matches_closed_form']}")
    print(" | ".join(row))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run aio_trace2 BOF_TRACE=2 BOF_IO_ENGINE=aio -- --path 2 --direct 1 --reps 3
run aio BOF_IO_ENGINE=aio -- --path 2 --reps 3
run uring BOF_IO_ENGINE=uring -- --path 2 --direct 1 --reps 3
run uring_trace2 BOF_TRACE=2 BOF_IO_ENGINE=uring -- --path 2 --direct 1 --reps 2
run aio_thr16 BOF_IO_ENGINE=aio -- --path 2 --direct 1 --reps 3 --io-threads 16 --pinned 16
run uring_thr16 BOF_IO_ENGINE=uring -- --path 2 --direct 1 --reps 3 --io-threads 16 --pinned 16
run aio_thr16_req2M BOF_IO_ENGINE=aio BOF_IO_REQUEST_KIB=2048 -- --path 2 --direct 1 --reps 3 --io-threads 16 --pinned 16
run uring_thr16_req2M BOF_IO_ENGINE=uring BOF_IO_REQUEST_KIB=2048 -- --path 2 --direct 1 --reps 3 --io-threads 16 --pinned 16
run uring_thr16_req8M BOF_IO_ENGINE=uring BOF_IO_REQUEST_KIB=8192 -- --path 2 --direct 1 --reps 3 --io-threads 16 --pinned 16
run uring_thr12_writers6 BOF_IO_ENGINE=uring BOF_PANEL_WRITERS=6 -- --path 2 --direct 1 --reps 3 --io-threads 12 --pinned 16
run uring_tiles BOF_IO_ENGINE=uring -- --path 1 --direct 1 --reps 2
run aio_tiles BOF_IO_ENGINE=aio -- --path 1 --direct 1 --reps 2
