# Instruction-cache counters of the dominant kernel.  usage (GPU box): bash tools/pmc_icache.sh <workload> <spp> <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-cfg4_csg_textured_1080p}; SPP=${2:-64}; TAG=${3:-cur}
rm -rf gpurun_out/pmc_ic_${W}_$TAG
timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d gpurun_out/pmc_ic_${W}_$TAG -- python3 bench.py --workload $W --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_ic_${W}_$TAG.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_ic_${W}_$TAG > gpurun_out/pmc_icache_${W}_$TAG.json
cat gpurun_out/pmc_icache_${W}_$TAG.json
