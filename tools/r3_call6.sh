set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_render.py tests/test_progressive.py tests/test_motion.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3f_tests_render.txt 2>&1
echo "render tests done"
timeout -k 10 250 python bench.py --pipeline wavefront --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f_bench_wavefront_cfg2.json 2> gpurun_out/r3f_bench_wavefront_cfg2.err
echo "wavefront bench done"
rm -rf gpurun_out/r3f_wfprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3f_wfprof -- python3 bench.py --pipeline wavefront --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f_wfprof.json 2> gpurun_out/r3f_wfprof.err
cp gpurun_out/r3f_wfprof/*/*kernel_stats.csv gpurun_out/r3f_wavefront_cfg2_64spp_kernel_stats.csv
echo "wavefront profile done"
timeout -k 10 200 python tools/interactive_fps.py > gpurun_out/r3f_interactive_fps.txt 2>&1
echo "fps done"
for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p f3_kit_world1_1080p; do
  timeout -k 10 200 python bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f_$w.json 2>/dev/null
done
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3f_tests.txt 2>&1
echo "tests done"
