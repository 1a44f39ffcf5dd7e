#!/bin/bash
# N fuzz processes under rocgdb (batch): on a fatal signal every thread's stack, the registers and the memory around
# the tile-cache slot table are printed.  Usage: tools/exp/fuzz_gdb.sh OUTDIR N SECONDS FIRST_SEED [fuzz args]
out=$1; n=$2; secs=$3; seed0=$4; shift 4
mkdir -p "$out"
cat > "$out/cmds.gdb" <<'G'
set pagination off
set confirm off
handle SIGSEGV stop print
handle SIGUSR1 nostop noprint pass
handle SIG34 nostop noprint pass
handle SIG35 nostop noprint pass
run
echo \n==== stopped ====\n
bt 12
info registers rdi rsi rbx r8 r12 r13 r14 r15 rip
echo \n---- frame of flush_wgroup ----\n
frame 2
info registers rbx r12 r14 r15 rip
x/24gx $r14+$r15
echo \n---- the whole slot table (first 6 slots) ----\n
x/138gx $r14
echo \n---- GemmRun: tiles / slots / wgroup vector headers ----\n
x/4gx $r12+0x1b0
x/4gx $r12+0x1f8
x/4gx $r12+0x4b8
thread apply all bt 8
kill
quit
G
pids=()
for i in $(seq 0 $((n - 1))); do
  s=$((seed0 + i))
  rocgdb -batch -x "$out/cmds.gdb" --args python3 tests/test_gpu_fuzz.py --verify --seconds "$secs" --seed "$s" "$@" > "$out/gdb_seed$s.log" 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
grep -l "==== stopped ====" "$out"/gdb_seed*.log
for f in "$out"/gdb_seed*.log; do tail -c 300000 "$f" > "$f.t"; mv "$f.t" "$f"; done
