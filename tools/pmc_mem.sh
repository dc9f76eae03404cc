# Vector-memory pipeline counters (address unit, L1) of the dominant kernel.
# usage (GPU box): bash tools/pmc_mem.sh <workload> <spp> <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-cfg2_random_spheres_1080p}; SPP=${2:-64}; TAG=${3:-cur}
rm -rf gpurun_out/pmc_mem_${W}_$TAG
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY TD_TD_BUSY_sum --output-format csv -d gpurun_out/pmc_mem_${W}_$TAG -- python3 bench.py --workload $W --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_mem_${W}_$TAG.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_mem_${W}_$TAG > gpurun_out/pmc_mem_${W}_$TAG.json
cat gpurun_out/pmc_mem_${W}_$TAG.json
