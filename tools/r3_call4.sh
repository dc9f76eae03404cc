set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 3 5 6; do
  RT_SHOW_OCCUPANCY=1 RT_DEEP_VARIANT=$v timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3d_cfg5_variant$v.json 2> gpurun_out/r3d_cfg5_variant$v.err
  echo "variant $v done"
done
for v in 1 3 5; do for rf in 28 36; do
  RT_DEEP_VARIANT=$v timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --refill $rf > gpurun_out/r3d_cfg5_variant${v}_refill$rf.json 2>/dev/null
done; done
timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --refill 32 > gpurun_out/r3d_cfg5_variant0_refill32.json 2>/dev/null
echo "refill sweep done"
for w in cfg3_cornell_smoke_1080p f3_kit_world1_1080p f3_kit_world2_1080p cfg4_csg_textured_1080p; do
  for hm in 8 1; do
  timeout -k 10 200 python bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --heavy-min $hm > gpurun_out/r3d_${w}_hm$hm.json 2>/dev/null
  done
done
echo "benches done"
timeout -k 10 200 python tools/phase_profile.py cfg4_csg_textured_1080p 16 gpurun_out/executed_cfg4_r3d.json > gpurun_out/phases_cfg4_r3d.txt 2>&1
timeout -k 10 900 python -u tools/parity_subsets.py > gpurun_out/r3d_parity_subsets.txt 2>&1
echo "subsets done"
timeout -k 10 600 python -u tools/parity_measure.py kit > gpurun_out/r3d_parity_kit.txt 2>&1
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3d_tests.txt 2>&1
echo "tests done"
