set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 280 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_render.py -x -q -m gpu -k "million or deep" --timeout 280 --timeout-method thread > gpurun_out/r3m_tests_million.txt 2>&1 || { tail -30 gpurun_out/r3m_tests_million.txt; exit 1; }
tail -3 gpurun_out/r3m_tests_million.txt
timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3m_bench_cfg5_wide.json 2> gpurun_out/r3m_bench_cfg5_wide.err
echo "wide bench done"
RT_DEEP_BINARY=1 timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3m_bench_cfg5_binary.json 2> gpurun_out/r3m_bench_cfg5_binary.err
echo "binary bench done"
