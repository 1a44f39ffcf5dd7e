#!/bin/bash
# Same-lease A/B of the driver-timed headline (VERDICT r5 "next round" item 1): the round-4 tree (_ab/r4, the
# tree BENCH_r04 was measured on, exported with `git archive ee48882` and built in place) against HEAD and HEAD with
# one round-5 change switched off at a time, interleaved, ROUNDS rounds.
# Usage: tools/r6/ab_headline.sh OUTDIR [ROUNDS] [STEPS] [variants...]
out=${1:-gpurun_out/ab}; rounds=${2:-3}; steps=${3:-10}; shift 3 2>/dev/null
variants=${*:-r4 head noprobe nohost chain1 streams0 writers8}
mkdir -p "$out"
root=$(pwd)
one() {  # name round
  local v=$1 r=$2 t0=$(date +%s.%N)
  local common="--no-cpu --no-extras --steps $steps --warmup 2"
  case $v in
    r4)       (cd _ab/r4 && timeout 900 python bench.py $common > "$root/$out/$v.$r.line" 2> "$root/$out/$v.$r.err"; cp -f bench_detail.json "$root/$out/$v.$r.detail.json" 2>/dev/null) ;;
    head)     timeout 900 python bench.py $common --detail-out "$out/$v.$r.detail.json" > "$out/$v.$r.line" 2> "$out/$v.$r.err" ;;
    noprobe)  timeout 900 python bench.py $common --no-probe --detail-out "$out/$v.$r.detail.json" > "$out/$v.$r.line" 2> "$out/$v.$r.err" ;;
    nohost)   BOF_HOST_HANDOVER=0 timeout 900 python bench.py $common --detail-out "$out/$v.$r.detail.json" > "$out/$v.$r.line" 2> "$out/$v.$r.err" ;;
    chain1)   timeout 900 python bench.py $common --opt gemm_chain=1 --detail-out "$out/$v.$r.detail.json" > "$out/$v.$r.line" 2> "$out/$v.$r.err" ;;
    streams0) timeout 900 python bench.py $common --streams 0 --detail-out "$out/$v.$r.detail.json" > "$out/$v.$r.line" 2> "$out/$v.$r.err" ;;
    writers8) BOF_PANEL_WRITERS=8 timeout 900 python bench.py $common --detail-out "$out/$v.$r.detail.json" > "$out/$v.$r.line" 2> "$out/$v.$r.err" ;;
    *)        # NAME:ENV=VAL,ENV=VAL:extra bench args   (free-form variant)
              IFS=: read -r nm envs extra <<< "$v"
              env $(echo "$envs" | tr ',' ' ') timeout 900 python bench.py $common $extra --detail-out "$out/$nm.$r.detail.json" > "$out/$nm.$r.line" 2> "$out/$nm.$r.err" ;;
  esac
  echo "$v round $r: $(python3 -c "import time;print(round(time.time()-$t0,1))") s wall" >> "$out/log.txt"
}
for r in $(seq 1 "$rounds"); do
  for v in $variants; do one "$v" "$r"; done
done
python3 tools/r6/ab_table.py "$out" > "$out/table.md" 2>&1
cat "$out/table.md"
