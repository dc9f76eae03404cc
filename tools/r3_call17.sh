set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3q_cfg5.json 2> gpurun_out/r3q_err.txt
echo "cfg5: $(head -c 60 gpurun_out/r3q_cfg5.json)"
timeout -k 10 280 python -u -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "million" --timeout 280 --timeout-method thread > gpurun_out/r3q_tests_million.txt 2>&1 || { tail -30 gpurun_out/r3q_tests_million.txt; exit 1; }
tail -2 gpurun_out/r3q_tests_million.txt
