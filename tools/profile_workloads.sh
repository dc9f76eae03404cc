# Kernel trace + stats of the bench on the other workloads (64 spp per launch; the million-sphere one at 16).
# usage (GPU box): bash tools/profile_workloads.sh <tag>
set -e
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k f2_bouncing_spheres_1080p f3_kit_world1_1080p f3_kit_world2_1080p; do
  rm -rf gpurun_out/wprof_${w}_$TAG
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wprof_${w}_$TAG -- python3 bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/wprof_${w}_$TAG.json 2> gpurun_out/wprof_${w}_$TAG.err
  cp gpurun_out/wprof_${w}_$TAG/*/*kernel_stats.csv gpurun_out/${TAG}_fused_${w}_64spp_kernel_stats.csv
  cat gpurun_out/wprof_${w}_$TAG.json >> gpurun_out/${TAG}_bench_all_workloads_64spp.jsonl
  echo "$w done"
done
