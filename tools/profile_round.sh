# The evidence set for the bench line: kernel trace + stats of the default bench, HBM traffic
# counters at the same launch size, shader-engine and memory-pipeline counters, per-phase
# lane statistics.  usage (GPU box): bash tools/profile_round.sh <tag>
set -e
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err
bash tools/measure_traffic.sh cfg2_random_spheres_1080p 1024 $TAG > /dev/null
bash tools/pmc_sq.sh cfg2_random_spheres_1080p 64 $TAG > /dev/null
bash tools/pmc_mem.sh cfg2_random_spheres_1080p 64 $TAG > /dev/null
for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do
  timeout -k 10 200 python tools/phase_profile.py $w 16 > gpurun_out/phases_${w}_$TAG.txt
done
for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do
  timeout -k 10 250 python bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${w}_$TAG.json 2>/dev/null
done
timeout -k 10 250 python bench.py --pipeline wavefront --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_wavefront_cfg2_$TAG.json 2>/dev/null
cat gpurun_out/prof_$TAG.json
