set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 512 8192 65536 1000000; do
RT_TUNE_TOP4=$t timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3u_cfg5_top$t.json 2> gpurun_out/r3u_err.txt
echo "top $t: $(head -c 60 gpurun_out/r3u_cfg5_top$t.json)"
done
