for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do for rf in 32 40 48; do
  spp=128; [ $w = cfg5_sphere_cloud_4k ] && spp=32
  timeout -k 10 200 python bench.py --workload $w --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --refill $rf 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w refill $rf', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done; done
timeout -k 10 300 python bench.py --no-cpu-baseline | cut -c1-200
