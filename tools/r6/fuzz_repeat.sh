#!/bin/bash
# One drawn fuzz case repeated in PROCS processes side by side: tools/r6/fuzz_repeat.sh OUT SEED INDEX REPEAT PROCS [ENV=VAL ...]
out=$1; seed=$2; idx=$3; rep=$4; procs=$5; shift 5
mkdir -p "$out"
for p in $(seq 1 "$procs"); do
  env "$@" timeout 900 python3 tests/test_gpu_fuzz.py --seed "$seed" --only "$idx" --repeat "$rep" --verify 2>&1 | grep -v "^ok\|amdgpu.ids" > "$out/p$p.log" &
done
wait
echo "$* : $(cat "$out"/p*.log | grep -c '^FAIL') failures in $(grep -h '^fuzz:' "$out"/p*.log | awk '{c+=$2} END{print c}') cases"
grep -h "^FAIL" "$out"/p*.log | cut -c1-200 | sort | uniq -c | head -5
