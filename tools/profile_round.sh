# The evidence set for the bench line: kernel trace + stats of the default bench, HBM traffic
# counters and the per-phase lane statistics.  usage (GPU box): bash tools/profile_round.sh <tag>
set -e
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err
bash tools/measure_traffic.sh cfg2_random_spheres_1080p 64
for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p; do
  timeout -k 10 200 python tools/phase_profile.py $w 16 > gpurun_out/phases_${w}_$TAG.txt
done
cat gpurun_out/prof_$TAG.json gpurun_out/phases_*_$TAG.txt
