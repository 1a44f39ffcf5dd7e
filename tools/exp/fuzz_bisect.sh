# Which thread launches the tile kernels of a multi-slab tile-cache call?  (profiles/r4/fuzz_thread_bisect.md)
run() { name=$1; shift; env "$@" tools/fuzz_parallel.sh gpurun_out/fuzz_x/$name 12 ${SECS:-200} 7001 --kind ${KIND:-kmeans} "--set" "devices=[0,0,0];gemm_path=1" > gpurun_out/fuzz_x/$name.txt 2>&1; echo "== $name: $(grep -h '^fuzz:' gpurun_out/fuzz_x/$name.txt | awk '{c+=$2; f+=$4} END{print c" cases, "f" failures"}')"; }
mkdir -p gpurun_out/fuzz_x
for t in "$@"; do
case $t in
X0) run X0_fresh_thread BOF_DBG_SLAB_THREAD=1 ;;
X1) run X1_caller_thread BOF_DBG_SLAB_THREAD=0 ;;
X2) run X2_fresh_thread_sync_each BOF_DBG_SLAB_THREAD=1 BOF_DBG_KM_SYNC_EACH=1 ;;
X3) KIND=gemm run X3_gemm_fresh_thread BOF_DBG_SLAB_THREAD=1 ;;
# (X4 "fresh thread + warm-up launch" and X5 "persistent worker" of round 4 are gone: the library only knows
#  BOF_DBG_SLAB_THREAD=1 now -- the persistent launcher IS the default -- so those entries ran the default path under
#  their old labels; the numbers they produced are in profiles/r4/fuzz_thread_bisect.md)
# where the runtime puts kernel arguments (fresh launching threads, the reproducer): in device memory through the PCIe
# BAR (the default on this part) or in host memory; and the runtime's own HDP-flush workaround for the device placement
K0) run K0_fresh_thread_host_kernarg BOF_DBG_SLAB_THREAD=1 HIP_FORCE_DEV_KERNARG=0 ;;
K1) run K1_fresh_thread_hdp_flush_wa BOF_DBG_SLAB_THREAD=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 ;;
# the crash site of profiles/r4/fuzz_crash.md (hipEventRecord in flush_wgroup, 1 in ~300 000 under 16 processes): pooled
# events with timing disabled (default) against timing enabled; needs long runs (SECS=1800, 16 processes) to say anything
T0) run T0_events_timing_disabled BOF_EVENT_TIMING=0 ;;
T1) run T1_events_timing_enabled BOF_EVENT_TIMING=1 ;;
# compute streams per repetition of an ordinal instead of one set per ordinal fed by several dispatcher threads (section 6 of
# profiles/r4/fuzz_thread_bisect.md); KIND=gemm and --set 'devices=[0,0,0];gemm_path=2' is the configuration of the open mismatch
S1) run S1_streams_per_repetition BOF_STREAMS_PER_REP=1 ;;
S0) run S0_streams_shared_by_the_ordinal BOF_STREAMS_PER_REP=0 ;;     # (round 5: per repetition is the default; this is the old sharing)
esac
done
