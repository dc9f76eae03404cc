set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for combo in "16 24" "16 32" "20 32" "20 40" "24 32" "24 40" "28 40" "20 24"; do
  set -- $combo
  timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --refill $1 --node-min $2 > gpurun_out/r3p_cfg5_r$1_n$2.json 2> gpurun_out/r3p_err.txt
  echo "refill $1 node-min $2: $(head -c 60 gpurun_out/r3p_cfg5_r$1_n$2.json)"
done
for lm in 6 12 20 32; do
  RT_TUNE_LEAF_MAX=$lm timeout -k 10 100 python bench.py --workload cfg5_sphere_cloud_4k --spp 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3p_cfg5_lm$lm.json 2> gpurun_out/r3p_err.txt
  echo "leaf-max $lm: $(head -c 60 gpurun_out/r3p_cfg5_lm$lm.json)"
done
