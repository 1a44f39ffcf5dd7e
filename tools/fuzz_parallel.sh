#!/bin/bash
# N fuzz processes side by side on the one GPU (tests/test_gpu_fuzz.py, level 3 against the oracle, bit for bit),
# every gemm / kmeans call with BOF_VERIFY=1 hand-over checksums.  Usage: tools/fuzz_parallel.sh OUTDIR N SECONDS FIRST_SEED [extra args]
out=$1; n=$2; secs=$3; seed0=$4; shift 4
mkdir -p "$out"
export BOF_CRASH_TRACE=1      # a crash inside the library leaves its native stack and the event ring behind
pids=()
for i in $(seq 0 $((n - 1))); do
  s=$((seed0 + i))
  BOF_FUZZ_DUMP="${BOF_FUZZ_DUMP_DIR:-}" python3 tests/test_gpu_fuzz.py ${FUZZ_VERIFY---verify} --seconds "$secs" --seed "$s" "$@" > "$out/fuzz_seed$s.log" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
grep -h "^fuzz:" "$out"/fuzz_seed*.log | tee "$out/summary.txt"
for f in "$out"/fuzz_seed*.log; do
  if grep -q "FAIL\|BOF_VERIFY mismatch\|Segmentation" "$f"; then
    echo "== $f" | tee -a "$out/summary.txt"
    grep -h "^FAIL\|^\[bof\]" "$f" | sed -e "s/index=[0-9]*//" -e "s/[0-9a-f]\{16\}/H/g" | cut -c1-160 | sort | uniq -c | sort -rn | head -8 | tee -a "$out/summary.txt"; grep -m 3 -A 1 "^FAIL" "$f" | cut -c1-1200 | tee -a "$out/summary.txt"
    # keep the logs small enough to travel back
    head -c 400000 "$f" > "$f.head"; mv "$f.head" "$f"
  fi
done
exit $rc
