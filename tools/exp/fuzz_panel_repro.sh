#!/bin/bash
# Targeted reproduction of the one panel-path mismatch of the round-4 post-pooling fuzz (seed 3011 case 851: gemm, row
# panels, devices [0,0,0], forced k-major copies): that configuration only, BOF_VERIFY on, failures dumped (inputs,
# result, event ring).  Usage: tools/exp/fuzz_panel_repro.sh SECONDS [extra --set items]
secs=${1:-300}; extra=${2:-}
out=gpurun_out/fuzz_r4_panel
mkdir -p $out/dumps
# the case itself, many times
BOF_FUZZ_DUMP=$out/dumps python3 tests/test_gpu_fuzz.py --verify --seed 3011 --only 851 --repeat 400 > $out/replay_3011_851.txt 2>&1
grep -c "^ok" $out/replay_3011_851.txt | sed 's/^/replay of seed 3011 case 851: ok x /'
grep -c "^FAIL" $out/replay_3011_851.txt | sed 's/^/replay of seed 3011 case 851: FAIL x /'
BOF_FUZZ_DUMP_DIR=$out/dumps tools/fuzz_parallel.sh $out/run 16 "$secs" 8001 --kind gemm --set "devices=[0,0,0];panel_kmajor=2$extra" > $out/run.txt 2>&1
grep -h '^fuzz:' $out/run.txt | awk '{c+=$2; f+=$4} END{print "targeted: "c" cases, "f" failures"}'
tail -12 $out/run.txt | cut -c1-600
# keep what travels back small: the first four dumps
ls $out/dumps/*.npz 2>/dev/null | tail -n +5 | xargs -r rm -f
head -c 300000 $out/replay_3011_851.txt > $out/replay.head; mv $out/replay.head $out/replay_3011_851.txt
du -sh $out
