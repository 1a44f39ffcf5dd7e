#!/bin/bash
# The long instrumented fuzz of round 4: level 3 against the oracle bit for bit with BOF_VERIFY=1
# inside every gemm / kmeans call, many processes side by side on the one GPU.  Usage: tools/fuzz_long.sh OUTDIR SECONDS
out=$1; secs=$2
mkdir -p "$out"
tools/fuzz_parallel.sh "$out/mix" 10 "$secs" 2001 > "$out/mix.txt" 2>&1 &
tools/fuzz_parallel.sh "$out/kmeans" 3 "$secs" 3001 --kind kmeans > "$out/kmeans.txt" 2>&1 &
tools/fuzz_parallel.sh "$out/gemm" 3 "$secs" 4001 --kind gemm > "$out/gemm.txt" 2>&1 &
# the window around the one unexplained mismatch of round 3 (seed 11, case 6130), again and again
python3 tests/test_gpu_fuzz.py --verify --seed 11 --range 6100 6161 --repeat 250 > "$out/seed11_window.txt" 2>&1 &
wait
tail -n 30 "$out"/mix.txt "$out"/kmeans.txt "$out"/gemm.txt
tail -n 3 "$out/seed11_window.txt"
