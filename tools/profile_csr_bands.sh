#!/bin/bash
# What could a column-band schedule of csrmm buy?  Upper bound, measured: the unchanged row-major
# kernel on matrices with the SAME 1e9 non-zeros (10M rows x 100) but all of them inside one column band of B --
# n = 1M (B = 512 MB, the real cfg3), 250k (128 MB: fits the 256 MB Infinity Cache), 62.5k (32 MB), 8192 (4 MB = one
# XCD's L2).  A band schedule would make every pass look like one of these AT BEST (before its own costs: a binary
# search per row and band, C re-read and re-written once per extra band).  Per shape: kernel time (rocprofv3 --stats),
# FETCH_SIZE, TCC hit rate.  Usage: tools/profile_csr_bands.sh OUTDIR
out=$(realpath -m "${1:-gpurun_out/prof_csr_bands}")
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for n in 1000000 250000 62500 8192; do
  K="python3 $root/tools/kbench.py --what csr --rounds 1 --csr-shape 10000000x${n}x100"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/s_$n" -o k -- $K > "$out/kbench_n$n.txt" 2> "$out/s_$n.err"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/f_$n" -o k -- $K > /dev/null 2> "$out/f_$n.err"
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/t_$n" -o k -- $K > /dev/null 2> "$out/t_$n.err"
  ( cd "$root" && python3 tools/pmc_csr_summary.py "$out/bands_n$n.json" 4 "$out/f_$n" "$out/t_$n" > /dev/null )
  f=$(find "$out/s_$n" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep "csrmm_rowmajor" "$f" | cut -d, -f1-4 > "$out/stats_n$n.csv"
  rm -rf "$out/s_$n" "$out/f_$n" "$out/t_$n" "$out"/*.err
done
cd "$root"
python3 - "$out" <<'PY'
import json, sys, os, re
out = sys.argv[1]
rows = []
for n in (1000000, 250000, 62500, 8192):
    d = json.load(open(os.path.join(out, f"bands_n{n}.json"))).get("csrmm_rowmajor", {})
    txt = open(os.path.join(out, f"kbench_n{n}.txt")).read()
    m = re.search(r"csrmm .*?: ([\d.]+) ms", txt)
    st = open(os.path.join(out, f"stats_n{n}.csv")).read().strip().split(",") if os.path.exists(os.path.join(out, f"stats_n{n}.csv")) else []
    fetch = d.get("FETCH_SIZE", {}).get("sum_over_launches_of_one_pass", 0) * 1024 * 2
    rows.append({"B_columns_n": n, "B_MB": round(n * 128 * 4 / 1e6, 1), "ms_per_pass_under_rocprof": float(m.group(1)) if m else None,
                 "kernel_avg_ns": float(st[3]) if len(st) > 3 else None, "kernel_calls": int(st[1]) if len(st) > 1 else None,
                 "fetch_GB_per_pass": round(fetch / 1e9, 1), "l2_hit_rate": round(d.get("l2_hit_rate", 0), 4)})
json.dump({"what": "csrmm_rowmajor_kernel, 1e9 nnz, 10M rows x 100, k = 128; all non-zeros inside a B of n rows",
           "rows": rows}, open(os.path.join(out, "csrmm_band_upper_bound.json"), "w"), indent=1)
for r in rows:
    print(r)
PY
