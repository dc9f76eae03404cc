set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r3s
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3s -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_nobase_r3s.json 2> gpurun_out/prof_r3s.err
cp gpurun_out/prof_r3s/*/*kernel_stats.csv gpurun_out/r3s_fused_cfg2_1024spp_kernel_stats.csv
cat gpurun_out/prof_nobase_r3s.json | cut -c1-300
head -2 gpurun_out/r3s_fused_cfg2_1024spp_kernel_stats.csv | tail -1 | sed 's/.*)",//'
