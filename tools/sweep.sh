for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p f2_bouncing_spheres_1080p f3_kit_world1_1080p; do
  timeout -k 10 200 python bench.py --workload $w --spp 100 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w spp 100', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done
timeout -k 10 300 python bench.py --no-cpu-baseline | cut -c1-300
timeout -k 10 100 python tools/phase_profile.py cfg2_random_spheres_1080p 64
