set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3t_cfg5_cluster.json 2> gpurun_out/r3t_err.txt
echo "sibling clusters: $(head -c 60 gpurun_out/r3t_cfg5_cluster.json)"
RT_TUNE_PREORDER=1 timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3t_cfg5_preorder.json 2> gpurun_out/r3t_err.txt
echo "pre-order: $(head -c 60 gpurun_out/r3t_cfg5_preorder.json)"
timeout -k 10 200 python tools/commit_time.py cfg5_sphere_cloud_4k 3 > gpurun_out/r3t_commit_time_cfg5.txt 2>&1
cat gpurun_out/r3t_commit_time_cfg5.txt
