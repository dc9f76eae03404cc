# Throughput against the scheduling thresholds of the fused kernel.  usage (GPU box): bash tools/sweep_knobs.sh <workload> <spp>
W=${1:-cfg2_random_spheres_1080p}; SPP=${2:-256}
for r in 36 40 44 48 52 56; do for nm in 4 8 12 16 24; do
  echo -n "$W refill $r node_min $nm: "
  timeout -k 10 120 python bench.py --workload $W --spp $SPP --steps 3 --warmup 1 --no-cpu-baseline --refill $r --node-min $nm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))" || exit 1
done; done
