for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p; do
  timeout -k 10 200 python bench.py --workload $w --spp 128 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w', round(d['value']), round(d['ms_per_step'],2))" || exit 1
done
