# Throughput against the heavy-leaf threshold (RT_OPTION_HEAVY_MIN).  usage (GPU box): bash tools/sweep_heavy.sh
for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p f3_kit_world1_1080p f2_bouncing_spheres_1080p; do
  for hm in 1 4 8 16 32; do
    timeout -k 10 120 python bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --heavy-min $hm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w heavy_min $hm', round(d['value']))" || exit 1
  done
done
