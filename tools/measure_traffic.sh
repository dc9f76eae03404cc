# HBM traffic of the dominant kernel from the L2 fabric counters, one rocprofv3 pass per counter
# group, at the sample count the bench line is quoted on.
# usage (GPU box): bash tools/measure_traffic.sh <workload> <spp> <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-cfg2_random_spheres_1080p}; SPP=${2:-1024}; TAG=${3:-cur}
run() {
  rm -rf gpurun_out/pmc_$1_${W}_$TAG
  timeout -s KILL 150 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/pmc_$1_${W}_$TAG -- python3 bench.py --workload $W --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$1_${W}_$TAG.log 2>&1
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
python3 tools/pmc_summary.py gpurun_out/pmc_fetch_${W}_$TAG gpurun_out/pmc_write_${W}_$TAG gpurun_out/pmc_l2_${W}_$TAG > gpurun_out/traffic_${W}_$TAG.json
python3 tools/traffic_json.py $W $SPP gpurun_out/traffic_${W}_$TAG.json > gpurun_out/traffic_record_${W}_$TAG.json
cat gpurun_out/traffic_record_${W}_$TAG.json
