#!/bin/bash
# rocprofv3 passes for the CSR kernels at BASELINE sizes (cfg3 CSRMM, cfg5-size CSRGEMV), HBM-resident.
# Usage: tools/profile_csr.sh OUTDIR
out=$(realpath -m "${1:-gpurun_out/prof_csr}")
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
K="python3 $root/tools/kbench.py --what csr --rounds 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o k -- $K > "$out/kbench_csr_output.txt" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o k -- $K > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o k -- $K > /dev/null 2> "$out/pmc_write.err"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_tcc" -o k -- $K > /dev/null 2> "$out/pmc_tcc.err"
cd "$root"
python3 tools/pmc_csr_summary.py "$out/kbench_csr_pmc.json" 4 "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_tcc"
f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kbench_csr_kernel_stats.csv" && python3 tools/kstats.py "$f" csr radix gemv scan
rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_tcc"
