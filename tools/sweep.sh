for w in cfg2_random_spheres_1080p cfg4_csg_textured_1080p f3_kit_world1_1080p f3_kit_world2_1080p; do
  timeout -k 10 200 python bench.py --workload $w --spp 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w', round(d['value']), round(d['ms_per_step'],2))" || exit 1
done
