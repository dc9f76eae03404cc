for spp in 1 4 16 64 128 256; do
  timeout -k 10 200 python bench.py --spp $spp --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cfg2 spp $spp', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done
for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k f2_bouncing_spheres_1080p f3_kit_world1_1080p; do
  spp=256; [ $w = cfg5_sphere_cloud_4k ] && spp=64
  timeout -k 10 200 python bench.py --workload $w --spp $spp --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w spp $spp', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done
timeout -k 10 300 python bench.py | cut -c1-1200
