#!/bin/bash
# second sweep: write-back diagnosis (verbose trace, writers, streams, io_uring)
out=${1:-gpurun_out/sweep2}
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
run trace2_odirect BOF_TRACE=2 -- --path 2 --direct 1 --reps 1
run trace2_buffered BOF_TRACE=2 -- --path 2 --direct 0 --reps 1
run default X=1 -- --path 2
run streams1 BOF_PANEL_STREAMS=1 -- --path 2
run streams4 BOF_PANEL_STREAMS=4 -- --path 2
run writers8 BOF_PANEL_WRITERS=8 -- --path 2 --direct 1
run writers2 BOF_PANEL_WRITERS=2 -- --path 2 --direct 1
run writers8_pinned16 BOF_PANEL_WRITERS=8 -- --path 2 --direct 1 --pinned 16
run thr16_req2M BOF_IO_REQUEST_KIB=2048 -- --path 2 --io-threads 16 --pinned 16
run thr16_req1M BOF_IO_REQUEST_KIB=1024 -- --path 2 --io-threads 16 --pinned 16 --direct 1
run uring BOF_IO_ENGINE=uring -- --path 2 --direct 1
run uring_thr16_req2M BOF_IO_ENGINE=uring BOF_IO_REQUEST_KIB=2048 -- --path 2 --direct 1 --io-threads 16 --pinned 16
run chunk16_thr16 X=1 -- --path 2 --direct 1 --chunk-mib 16 --io-threads 16 --pinned 16
env X=1 timeout 900 python tools/flash_e2e.py --n 65536 --path 2 --reps 1 > "$out/n65536.json" 2> "$out/n65536.err"; tail -c 1500 "$out/n65536.json"
