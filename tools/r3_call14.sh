set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for combo in "8 40" "12 40" "16 40" "16 48" "16 56" "12 56" "20 48" "16 32"; do
  set -- $combo
  timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --refill $1 --node-min $2 > gpurun_out/r3o_cfg5_r$1_n$2.json 2> gpurun_out/r3o_err.txt
  echo "refill $1 node-min $2: $(head -c 60 gpurun_out/r3o_cfg5_r$1_n$2.json)"
done
bash tools/pmc_sq.sh cfg5_sphere_cloud_4k 16 r3o > /dev/null
bash tools/pmc_mem.sh cfg5_sphere_cloud_4k 16 r3o > /dev/null
echo pmc done
