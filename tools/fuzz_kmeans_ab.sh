#!/bin/bash
# A/B of the kmeans norm-vector upload race: kmeans-only fuzz, tile cache, device list [0,0,0],
# N processes side by side, BOF_VERIFY on.  Phase A: the vectors go up with plain synchronous hipMemcpy calls from
# pageable memory and the first kernels are launched right behind them (rounds 1-3); phase B: the device is
# synchronised behind the upload (round 4).  Usage: tools/fuzz_kmeans_ab.sh OUTDIR N SECONDS
out=$1; n=$2; secs=$3
mkdir -p "$out"
export BOF_CRASH_TRACE=1
BOF_KMEANS_UPLOAD_SYNC=0 tools/fuzz_parallel.sh "$out/A_no_sync" "$n" "$secs" 5001 --kind kmeans --set "devices=[0,0,0];gemm_path=1" > "$out/A_no_sync.txt" 2>&1
BOF_KMEANS_UPLOAD_SYNC=1 tools/fuzz_parallel.sh "$out/B_sync" "$n" "$secs" 5001 --kind kmeans --set "devices=[0,0,0];gemm_path=1" > "$out/B_sync.txt" 2>&1
echo "== A: no synchronisation behind the upload"; grep "^fuzz:" "$out/A_no_sync.txt"
echo "== B: device synchronised behind the upload"; grep "^fuzz:" "$out/B_sync.txt"
