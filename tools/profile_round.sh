# The evidence set for the bench line: kernel trace + stats of the default bench, HBM traffic
# counters at the same launch size, shader-engine and memory-pipeline counters of every workload's
# kernel, per-phase lane statistics, executed-work records.  usage (GPU box): bash tools/profile_round.sh <tag>
set -e
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$TAG
# (the profiled command leaves the CPU baseline out: its parity leg renders the baseline's sample on the GPU with the same
# kernel, and those short launches would enter the kernel's average)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_nobase_$TAG.json 2> gpurun_out/prof_$TAG.err
cp gpurun_out/prof_$TAG/*/*kernel_stats.csv gpurun_out/${TAG}_fused_cfg2_1024spp_kernel_stats.csv
timeout -k 10 300 python bench.py > gpurun_out/prof_$TAG.json 2>> gpurun_out/prof_$TAG.err
echo "kernel trace done"
bash tools/measure_traffic.sh cfg2_random_spheres_1080p 1024 $TAG > /dev/null
bash tools/measure_traffic.sh cfg5_sphere_cloud_4k 16 $TAG > /dev/null
echo "traffic done"
for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do
  spp=64; [ $w = cfg5_sphere_cloud_4k ] && spp=16
  bash tools/pmc_sq.sh $w $spp $TAG > /dev/null
  bash tools/pmc_mem.sh $w $spp $TAG > /dev/null
  echo "pmc $w done"
done
for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do
  timeout -k 10 200 python tools/phase_profile.py $w 16 gpurun_out/executed_${w}_$TAG.json > gpurun_out/phases_${w}_$TAG.txt
done
echo "phases done"
bash tools/profile_workloads.sh $TAG
timeout -k 10 250 python bench.py --pipeline wavefront --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_wavefront_cfg2_$TAG.json 2>/dev/null
rm -rf gpurun_out/wfprof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wfprof_$TAG -- python3 bench.py --pipeline wavefront --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/wfprof_$TAG.json 2> gpurun_out/wfprof_$TAG.err
cp gpurun_out/wfprof_$TAG/*/*kernel_stats.csv gpurun_out/${TAG}_wavefront_cfg2_64spp_kernel_stats.csv
bash tools/wavefront_traffic.sh cfg2_random_spheres_1080p 64 $TAG > /dev/null
echo "wavefront done"
timeout -k 10 300 python bench.py --workload cfg5_sphere_cloud_4k --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg5_full_$TAG.json 2>/dev/null
timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 > gpurun_out/bench_cfg5_64spp_$TAG.json 2>/dev/null
RT_DEEP_BINARY=1 timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg5_binary_$TAG.json 2>/dev/null
RT_DEEP_PLAIN=1 timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg5_plain_$TAG.json 2>/dev/null
timeout -k 10 200 python tools/interactive_fps.py > gpurun_out/interactive_fps_$TAG.txt 2>&1
echo "cfg5 and interactive done"
for s in 1 16 64; do timeout -k 10 100 python bench.py --spp $s --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null; done > gpurun_out/bench_small_launches_$TAG.jsonl
cat gpurun_out/prof_$TAG.json
