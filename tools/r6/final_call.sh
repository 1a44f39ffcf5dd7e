#!/bin/bash
# The round's last full GPU call (second session): what the box's disks are, the driver's form of the bench, the GPU
# suite, smoke(), the rocprofv3 passes.  Usage: tools/r6/final_call.sh OUTDIR
out=${1:-gpurun_out/r6b}; mkdir -p "$out"
export TMPDIR=/tmp
{ echo "## lsblk"; lsblk -o NAME,SIZE,TYPE,MOUNTPOINT,MODEL 2>&1; echo "## df"; df -h /tmp /dev/shm . 2>&1; echo "## mounts"; mount 2>&1 | grep -v -E "proc|sysfs|cgroup|devpts|mqueue" | head -40; echo "## cpu"; nproc; free -g | head -2; } > "$out/box.txt" 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$out/bench_detail_steps20.json" > "$out/bench_line_steps20.json" 2> "$out/bench_steps20.err"
tail -c 600 "$out/bench_steps20.err"
python3 tools/r6/step_timeline.py "$out/bench_detail_steps20.json" > "$out/step_timeline.md" 2>&1
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15) > "$out/gpu_tests.txt"; tail -3 "$out/gpu_tests.txt"
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > "$out/smoke.txt"; cat "$out/smoke.txt"
timeout 900 bash tools/profile_bench.sh "$out/prof" 2>&1 | tail -12
