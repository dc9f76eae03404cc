set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 2; do
  RT_DEEP_VARIANT=$v timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3c_cfg5_variant$v.json 2> gpurun_out/r3c_cfg5_variant$v.err
  echo "variant $v done"
done
for v in 1 2; do for rf in 24 32 40; do
  RT_DEEP_VARIANT=$v timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --refill $rf > gpurun_out/r3c_cfg5_variant${v}_refill$rf.json 2>/dev/null
done; done
echo "refill sweep done"
for wps in 5 4; do
  RT_LIST1_WPS=$wps timeout -k 10 200 python bench.py --workload cfg3_cornell_smoke_1080p --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3c_cfg3_wps$wps.json 2>/dev/null
done
for w in f3_kit_world1_1080p f3_kit_world2_1080p cfg4_csg_textured_1080p; do
  timeout -k 10 200 python bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3c_$w.json 2>/dev/null
done
echo "list benches done"
timeout -k 10 200 python tools/phase_profile.py cfg3_cornell_smoke_1080p 16 gpurun_out/executed_cfg3_r3c.json > gpurun_out/phases_cfg3_r3c.txt 2>&1
timeout -k 10 900 python -u tools/parity_subsets.py > gpurun_out/r3c_parity_subsets.txt 2>&1
echo "subsets done"
timeout -k 10 900 python -u -m pytest tests/test_gpu_render.py tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_kit_textures.py -x -q --timeout 600 --timeout-method thread > gpurun_out/r3c_tests.txt 2>&1
echo "tests done"
