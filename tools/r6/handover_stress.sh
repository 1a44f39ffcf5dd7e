#!/bin/bash
# The variants of tools/exp/handover_stress.hip, SECS seconds each, one JSON line per variant.
# Usage: tools/r6/handover_stress.sh OUTDIR [SECS] [PROCS]   (PROCS > 1: that many copies of every variant at once)
out=${1:-gpurun_out/handover}; secs=${2:-45}; procs=${3:-1}
mkdir -p "$out"
bin=tools/exp/handover_stress
[ -x $bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -pthread tools/exp/handover_stress.hip -o $bin || exit 2
run() {  # name args...
  name=$1; shift
  for p in $(seq 1 "$procs"); do
    timeout $((secs + 60)) $bin --seconds "$secs" "$@" >> "$out/$name.jsonl" 2>> "$out/$name.err" &
  done
  wait
  tail -n "$procs" "$out/$name.jsonl"
}
run selftest_no_wait      --no-wait 1 --seconds 3
run base_fixed            --events fixed
run pooled                --events pooled
run fresh                 --events fresh
run launcher_thread       --events fixed --launcher 1
run launcher_pooled       --events pooled --launcher 1
run host_confirm          --events fixed --host-confirm 1
run copy2d                --events fixed --copy 2d
run copy2d_launcher       --events pooled --copy 2d --launcher 1
run chunks                --events fixed --copy chunks --readers 3
run wgs1                  --events fixed --wgs 1
run wgs64                 --events fixed --wgs 64
run cold_slots            --events fixed --prior-read 0
run cold_slots_pooled     --events pooled --prior-read 0 --launcher 1
run no_h2d_check          --events fixed --h2d-check 0
run big_slots             --events fixed --words 1048576
run small_slots           --events pooled --words 4096 --readers 3 --streams 3
# round-6 addition: several pipelines in ONE process (a device list that repeats its ordinal: more HIP streams than
# hardware queues) -- the ingredient of the one sighting that the single-pipeline variants above lack
run pipelines3_pooled_2d  --events pooled --copy 2d --launcher 1 --pipelines 3 --streams 3
run pipelines3_fixed      --events fixed --pipelines 3 --streams 4
run pipelines6_small      --events pooled --launcher 1 --pipelines 6 --streams 2 --words 16384
