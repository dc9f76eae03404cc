for lm in 65 40 28 20; do for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do
  spp=128; [ $w = cfg5_sphere_cloud_4k ] && spp=32
  RT_LEAF_MAX=$lm timeout -k 10 200 python bench.py --workload $w --spp $spp --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('leaf_max $lm $w', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done; done
