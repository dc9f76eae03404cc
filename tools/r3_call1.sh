# first GPU call of the round: parity measurements, the node-fetch microbenchmark, the new tests, executed-work records
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 tools/ubench/node_fetch2 > gpurun_out/r3_ubench_node_fetch2.txt 2>&1
echo "ubench done"
timeout -k 10 500 python -u tools/parity_measure.py known csg kit cfg5 > gpurun_out/r3_parity_measure.txt 2>&1
echo "parity measure done"
timeout -k 10 400 python -u -m pytest tests/test_zz_no_alloc.py tests/test_gpu_dist.py -x -v --timeout 300 --timeout-method thread > gpurun_out/r3_new_tests.txt 2>&1
echo "new tests done"
for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg2_random_spheres_1080p; do
  timeout -k 10 200 python tools/phase_profile.py $w 16 gpurun_out/executed_${w}_r3a.json > gpurun_out/phases_${w}_r3a.txt 2>&1
done
echo "phases done"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r3a_bench_cfg2.json 2> gpurun_out/r3a_bench_cfg2.err
echo "bench done"
timeout -k 10 300 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 > gpurun_out/r3a_bench_cfg5.json 2> gpurun_out/r3a_bench_cfg5.err
echo "bench cfg5 done"
