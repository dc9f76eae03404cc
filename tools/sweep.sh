for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do for n in f16; do
  spp=32; [ $w = cfg5_sphere_cloud_4k ] && spp=16
  timeout -k 10 200 python bench.py --workload $w --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --nodes $n 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w $n', round(d['value']), round(d['ms_per_step'],2))" || exit 1
done; done
for nm in 16 24 32 40; do for rf in 40 48 56; do
  timeout -k 10 200 python bench.py --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --node-min $nm --refill $rf 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cfg2 node_min $nm refill $rf', round(d['value']), round(d['ms_per_step'],2))" || exit 1
done; done
