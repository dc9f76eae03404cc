# Shader-engine counters of the dominant kernel: one rocprofv3 pass per counter group.
# usage (on the GPU box): bash tools/pmc_sq.sh <workload> <spp> <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-cfg2_random_spheres_1080p}; SPP=${2:-64}; TAG=${3:-cur}
run() {
  rm -rf gpurun_out/pmc_$1_${W}_$TAG
  timeout -s KILL 150 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/pmc_$1_${W}_$TAG -- python3 bench.py --workload $W --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$1_${W}_$TAG.log 2>&1
}
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run b "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
python3 tools/pmc_summary.py gpurun_out/pmc_a_${W}_$TAG gpurun_out/pmc_b_${W}_$TAG > gpurun_out/pmc_sq_${W}_$TAG.tmp
python3 -c "import json,sys; d=json.load(open(sys.argv[1])); d['_spp']=int(sys.argv[2]); print(json.dumps(d, indent=1, sort_keys=True))" gpurun_out/pmc_sq_${W}_$TAG.tmp $SPP > gpurun_out/pmc_sq_${W}_$TAG.json
rm -f gpurun_out/pmc_sq_${W}_$TAG.tmp
cat gpurun_out/pmc_sq_${W}_$TAG.json
