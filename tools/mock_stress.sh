#!/bin/bash
# The host side of level 3 (readers, dispatchers, launchers, flushers, the pools) on the mock HIP runtime of
# tests/native/ -- four DISTINCT devices, asynchronous streams -- running drawn cases for SECONDS under ThreadSanitizer
# and under ASan + UBSan, with BOF_VERIFY=1 (producer- and consumer-side sums, spot checks, poison) inside every call.
# No GPU: this is the stream-ordering / data-race / lifetime hunt the GPU fuzz cannot do (tests/test_host_sanitizers.py
# runs the same binaries for a few seconds in the CPU suite).
# Usage: tools/mock_stress.sh OUTDIR SECONDS [N_TSAN N_ASAN SEED_BASE]
out=$1; secs=$2; nt=${3:-2}; na=${4:-2}; sb=${5:-0}
root=$(cd "$(dirname "$0")/.." && pwd); csrc=$root/blas-on-flash_amd/csrc; T=$root/tests/native
mkdir -p "$out"
build() {  # build NAME FLAGS...
  local name=$1; shift
  g++ -std=c++17 -g -O1 "$@" -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$root/include" -I"$csrc" "$csrc"/plan.cpp "$csrc"/fileio.cpp \
    "$csrc"/uring_io.cpp "$csrc"/flash_support.cpp "$csrc"/flash_runtime.cpp "$csrc"/flash_csr.cpp "$csrc"/flash_gemm_panels.cpp \
    -x c++ "$csrc"/c_api.hip -x none "$T"/mock_hip.cpp "$T"/host_pipeline.cpp -o "$out/host_pipeline_$name" -lpthread -ldl -lrt
}
build tsan -fsanitize=thread & build asan -fsanitize=address,undefined -fno-sanitize-recover=all & wait
export MOCK_HIP_ASYNC=1 MOCK_HIP_DEVICES=4 BOF_VERIFY=1 LSAN_OPTIONS="suppressions=$T/lsan.supp:print_suppressions=0"
unset BOF_DEVICES
pids=()
for i in $(seq 1 "$nt"); do
  d="$out/tsan_$i"; mkdir -p "$d"
  TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1" "$out/host_pipeline_tsan" "$d" stress $((sb + 100 + i)) "$secs" > "$out/tsan_$i.txt" 2>&1 &
  pids+=($!)
done
for i in $(seq 1 "$na"); do
  d="$out/asan_$i"; mkdir -p "$d"
  MOCK_HIP_JITTER_US=300 ASAN_OPTIONS="detect_leaks=1:handle_abort=1" UBSAN_OPTIONS="print_stacktrace=1" "$out/host_pipeline_asan" "$d" stress $((sb + 200 + i)) "$secs" > "$out/asan_$i.txt" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
grep -h "host_pipeline ok\|BOF_VERIFY:" "$out"/tsan_*.txt "$out"/asan_*.txt | tee "$out/summary.txt"
grep -l "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error\|CHECK failed\|want " "$out"/*.txt 2>/dev/null | tee -a "$out/summary.txt"
[ $rc -eq 0 ] && rm -rf "$out"/tsan_[0-9]*/ "$out"/asan_[0-9]*/ 2>/dev/null   # (a failed process leaves its case files behind)
exit $rc
