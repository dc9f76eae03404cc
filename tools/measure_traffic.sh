# HBM traffic of the fused kernel from the L2 fabric counters, one rocprofv3 pass per counter.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-cfg2_random_spheres_1080p}; SPP=${2:-64}
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$W -- python3 bench.py --workload $W --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch_$W.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$W -- python3 bench.py --workload $W --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write_$W.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/pmc_l2_$W -- python3 bench.py --workload $W --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_l2_$W.log 2>&1
echo done
