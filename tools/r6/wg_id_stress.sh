#!/bin/bash
# N copies of tools/exp/wg_id_stress side by side (+ M processes of the CSR fuzz as background load): OUT SECONDS N M [SPIN]
out=$1; secs=$2; n=$3; m=$4; spin=${5:-2000}
mkdir -p "$out"
[ -x tools/exp/wg_id_stress ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/exp/wg_id_stress.hip -o tools/exp/wg_id_stress || exit 2
for i in $(seq 1 "$m"); do
  FUZZ_VERIFY= python3 tests/test_gpu_fuzz.py --seconds "$secs" --seed $((9500 + i)) --kind csr > "$out/load_fuzz$i.log" 2>&1 &
done
for i in $(seq 1 "$n"); do
  timeout $((secs + 120)) tools/exp/wg_id_stress "$secs" "$spin" > "$out/wg$i.json" 2> "$out/wg$i.err" &
done
wait
cat "$out"/wg*.json
grep -h "^fuzz:\|^FAIL" "$out"/load_fuzz*.log | cut -c1-200
