#!/bin/bash
# CSR-only fuzz, two arms side by side (4 processes each): the round-6 context rule (2 x readers) against the rule of
# rounds 1-5 (BOF_CSR_CONTEXTS=-1); failing cases leave their arrays in OUT/dumps.  tools/r6/fuzz_csr_ab.sh OUT SECONDS
out=$1; secs=$2
mkdir -p "$out/dumps" "$out/new" "$out/old"
export BOF_FUZZ_DUMP_DIR="$out/dumps"
FUZZ_VERIFY= tools/fuzz_parallel.sh "$out/new" 4 "$secs" 9301 --kind csr > "$out/new.txt" 2>&1 &
BOF_CSR_CONTEXTS=-1 FUZZ_VERIFY= tools/fuzz_parallel.sh "$out/old" 4 "$secs" 9401 --kind csr > "$out/old.txt" 2>&1 &
wait
for x in new old; do echo "== $x: $(grep -h '^fuzz:' "$out/$x.txt" | awk '{c+=$2; f+=$4} END{print c" cases, "f" failures"}')"; grep -h "^FAIL" "$out/$x"/fuzz_seed*.log | cut -c1-160 | head -5; done
ls -la "$out/dumps" | head
