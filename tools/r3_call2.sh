set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 tools/ubench/node_fetch2 > gpurun_out/r3b_ubench_node_fetch2.txt 2>&1
echo "ubench done"
for v in 0 1 2 4; do
  RT_DEEP_VARIANT=$v timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3b_cfg5_variant$v.json 2> gpurun_out/r3b_cfg5_variant$v.err
  echo "variant $v done"
done
for v in 1 2; do for rf in 32 40 56; do
  RT_DEEP_VARIANT=$v timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --refill $rf > gpurun_out/r3b_cfg5_variant${v}_refill$rf.json 2>/dev/null
done; done
echo "refill sweep done"
RT_DEEP_VARIANT=2 timeout -k 10 200 python tools/phase_profile.py cfg5_sphere_cloud_4k 8 gpurun_out/executed_cfg5_r3b_v2.json > gpurun_out/phases_cfg5_r3b_v2.txt 2>&1
timeout -k 10 200 python tools/phase_profile.py cfg5_sphere_cloud_4k 8 gpurun_out/executed_cfg5_r3b_v0.json > gpurun_out/phases_cfg5_r3b_v0.txt 2>&1
echo "phases done"
timeout -k 10 600 python -u tools/parity_measure.py csg kitwhere > gpurun_out/r3b_parity_measure.txt 2>&1
echo "parity measure done"
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_edges.py -x -q --timeout 600 --timeout-method thread > gpurun_out/r3b_tests.txt 2>&1
echo "tests done"
