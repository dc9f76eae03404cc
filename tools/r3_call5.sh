set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_render.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3e_tests_render.txt 2>&1
echo "render tests done"
timeout -k 10 250 python bench.py --pipeline wavefront --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3e_bench_wavefront_cfg2.json 2> gpurun_out/r3e_bench_wavefront_cfg2.err
echo "wavefront bench done"
for w in cfg3_cornell_smoke_1080p f3_kit_world1_1080p cfg4_csg_textured_1080p; do
  for dm in 0 4 8 16; do
  RT_DEAR_MIN=$dm timeout -k 10 200 python bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3e_${w}_dear$dm.json 2>/dev/null
  done
done
echo "benches done"
timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3e_cfg5_default.json 2>/dev/null
RT_DEEP_PLAIN=1 timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3e_cfg5_plain.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3e_cfg2.json 2>/dev/null
echo "cfg5 done"
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3e_tests.txt 2>&1
echo "tests done"
