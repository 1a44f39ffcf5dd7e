#!/bin/bash
# Reader threads x row-block contexts of flash::csrmm on cfg3's 17.7 GB of files, interleaved rounds on one lease
# (VERDICT r5 item 9).  Usage: tools/r6/cfg3_sweep.sh OUT ROUNDS "thr ctx" ...
out=$1; rounds=$2; shift 2
variants=("$@")
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for v in "${variants[@]}"; do
    thr=${v% *}; ctx=${v#* }
    python tools/flash_e2e_csrmm.py --io-threads "$thr" --pinned "$ctx" --reps 3 2> /dev/null > "$out/t${thr}_c${ctx}.$r.json"
    python3 - "$out/t${thr}_c${ctx}.$r.json" "threads $thr contexts $ctx round $r" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {m: d[m].get("seconds_all") for m in d if isinstance(d[m], dict) and "seconds_all" in d[m]}, flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
  done
done
