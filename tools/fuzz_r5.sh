#!/bin/bash
# The long instrumented fuzz of round 5: level 3 against the oracle bit for bit, BOF_VERIFY=1 inside
# every gemm / kmeans call -- producer-side hand-over sums, CONSUMER-side sums on the compute streams, a spot check of
# every launch, poisoned images -- with the round's defaults (compute streams per repetition of an ordinal; one chain
# over the whole K).  Few processes: on this pool the case rate FALLS with the process count (8: 141 cases/s, 16: 96,
# 24: 70 -- the driver serialises the contexts).  Three mixes side by side:
#   mix     the default draw of tests/test_gpu_fuzz.py (all kinds, all paths, device lists, peer_bcast, both arithmetics)
#   panels  gemm on row panels with devices [0,0,0] -- the configuration of the one unexplained mismatch of round 4
#   kmeans  kmeans through the tile cache with devices [0,0,0] -- the reproducer of rounds 3-4
# Usage: tools/fuzz_r5.sh OUTDIR SECONDS [N_MIX N_PANELS N_KMEANS]
out=$1; secs=$2; nm=${3:-5}; np=${4:-2}; nk=${5:-1}
mkdir -p "$out"
export BOF_FUZZ_DUMP_DIR="$out/dumps"; mkdir -p "$BOF_FUZZ_DUMP_DIR"
tools/fuzz_parallel.sh "$out/mix" "$nm" "$secs" 9001 > "$out/mix.txt" 2>&1 &
tools/fuzz_parallel.sh "$out/panels" "$np" "$secs" 9101 --kind gemm --set "devices=[0,0,0];gemm_path=2" > "$out/panels.txt" 2>&1 &
tools/fuzz_parallel.sh "$out/kmeans" "$nk" "$secs" 9201 --kind kmeans --set "devices=[0,0,0];gemm_path=1" > "$out/kmeans.txt" 2>&1 &
wait
for x in mix panels kmeans; do
  echo "== $x: $(grep -h '^fuzz:' "$out/$x.txt" | awk '{c+=$2; f+=$4; s+=$(NF-3)} END{print c" cases, "f" failures, "s" hand-over sums / spot checks compared"}')"
done | tee "$out/summary.txt"
grep -l "FAIL\|Segmentation\|BOF_VERIFY mismatch" "$out"/*/fuzz_seed*.log 2>/dev/null | head | tee -a "$out/summary.txt"
