#!/bin/bash
# N fuzz processes side by side on the one GPU (tests/test_gpu_fuzz.py, level 3 against the oracle, bit for bit),
# every gemm / kmeans call with BOF_VERIFY=1 hand-over checksums.  Usage: tools/fuzz_parallel.sh OUTDIR N SECONDS FIRST_SEED [extra args]
out=$1; n=$2; secs=$3; seed0=$4; shift 4
mkdir -p "$out"
pids=()
for i in $(seq 0 $((n - 1))); do
  s=$((seed0 + i))
  BOF_FUZZ_DUMP="$out" python3 tests/test_gpu_fuzz.py --verify --seconds "$secs" --seed "$s" "$@" > "$out/fuzz_seed$s.log" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
grep -h "^fuzz:" "$out"/fuzz_seed*.log | tee "$out/summary.txt"
grep -l "FAIL\|BOF_VERIFY mismatch" "$out"/fuzz_seed*.log | tee -a "$out/summary.txt"
exit $rc
