#!/bin/bash
# rocprofv3 passes behind profiles/rNN (run on the GPU box from the repo root): per-kernel stats
# of the default bench workload, the three PMC passes of MI355X_MICROARCH.md's HBM recipe, and
# (the bench headline IS the file-resident cfg2 call since round 4).  Usage: tools/profile_bench.sh OUTDIR
out=$(realpath -m "${1:-gpurun_out/prof}")
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- $B --steps 3 --warmup 1 > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o b -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o b -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/pmc_write.err"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d "$out/pmc_sq" -o b -- $B --steps 1 --warmup 0 > /dev/null 2> "$out/pmc_sq.err"
# the kernel alone, unperturbed by the profiler's D2H blit kernels: the HBM-resident tile DAG (no file I/O, no D2H)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/resident" -o res -- python3 $root/bench.py --resident-only --steps 3 > "$out/resident_under_rocprof.json" 2> "$out/resident.err"
cd "$root"
f=$(find "$out/resident" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/resident_kernel_stats.csv"
rm -rf "$out/resident"
python3 tools/pmc_summary.py sgemm_tile256_dma "$out/bench_gemm_pmc.json" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq"
f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_gemm_kernel_stats.csv" && python3 tools/kstats.py "$f" sgemm transpose
# keep only the summaries (the raw traces are tens of MB)
rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq"
