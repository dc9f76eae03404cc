set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 280 python -u -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "million" --timeout 280 --timeout-method thread > gpurun_out/r3l_tests_million.txt 2>&1 || { tail -30 gpurun_out/r3l_tests_million.txt; exit 1; }
echo "million tests done"
timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3l_bench_cfg5_packets.json 2> gpurun_out/r3l_bench_cfg5_packets.err
echo "packets bench done"
RT_DEEP_NO_PACKETS=1 timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3l_bench_cfg5_nopackets.json 2> gpurun_out/r3l_bench_cfg5_nopackets.err
echo "no-packets bench done"
rm -rf gpurun_out/r3l_prof
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3l_prof -- python3 bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3l_prof.json 2> gpurun_out/r3l_prof.err
cp gpurun_out/r3l_prof/*/*kernel_stats.csv gpurun_out/r3l_cfg5_packets_kernel_stats.csv
echo "profile done"
