set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for refill in 16 24 28 36 48; do
  timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --refill $refill > gpurun_out/r3n_cfg5_refill$refill.json 2> gpurun_out/r3n_err.txt
  echo "refill $refill: $(head -c 60 gpurun_out/r3n_cfg5_refill$refill.json)"
done
for nm in 8 16 24 40; do
  timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --node-min $nm > gpurun_out/r3n_cfg5_nodemin$nm.json 2> gpurun_out/r3n_err.txt
  echo "node-min $nm: $(head -c 60 gpurun_out/r3n_cfg5_nodemin$nm.json)"
done
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3n_tests_gpu.txt 2>&1 || { tail -30 gpurun_out/r3n_tests_gpu.txt; exit 1; }
tail -3 gpurun_out/r3n_tests_gpu.txt
