# round-4 closing runs: the reproducer configuration (kmeans, tile cache, device list [0,0,0]) with the default build,
# then the default mix of all kinds -- every gemm / kmeans call with BOF_VERIFY=1
mkdir -p gpurun_out/fuzz_final
tools/fuzz_parallel.sh gpurun_out/fuzz_final/kmeans_000 12 240 8001 --kind kmeans --set "devices=[0,0,0];gemm_path=1" > gpurun_out/fuzz_final/kmeans_000.txt 2>&1
echo "== kmeans, tile cache, [0,0,0] (default build): $(grep -h '^fuzz:' gpurun_out/fuzz_final/kmeans_000.txt | awk '{c+=$2; f+=$4} END{print c" cases, "f" failures"}')"
tools/fuzz_parallel.sh gpurun_out/fuzz_final/gemm_00 12 200 8101 --kind gemm --set "devices=[0,0];gemm_path=2" > gpurun_out/fuzz_final/gemm_00.txt 2>&1
echo "== gemm, row panels, [0,0] (second dispatcher on a persistent launcher): $(grep -h '^fuzz:' gpurun_out/fuzz_final/gemm_00.txt | awk '{c+=$2; f+=$4} END{print c" cases, "f" failures"}')"
tools/fuzz_parallel.sh gpurun_out/fuzz_final/mix 12 420 8201 > gpurun_out/fuzz_final/mix.txt 2>&1
echo "== default mix: $(grep -h '^fuzz:' gpurun_out/fuzz_final/mix.txt | awk '{c+=$2; f+=$4} END{print c" cases, "f" failures"}')"
