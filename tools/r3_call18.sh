set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3r_tests_gpu.txt 2>&1 || { tail -30 gpurun_out/r3r_tests_gpu.txt; exit 1; }
tail -2 gpurun_out/r3r_tests_gpu.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
bash tools/profile_round.sh r3r > gpurun_out/r3r_profile_round.log 2>&1 || { tail -30 gpurun_out/r3r_profile_round.log; exit 1; }
tail -3 gpurun_out/r3r_profile_round.log
