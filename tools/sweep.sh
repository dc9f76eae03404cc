for rf in 8 16 24 32 48; do for nm in 8 24; do
  timeout -k 10 250 python bench.py --workload cfg5_sphere_cloud_4k --spp 32 --steps 2 --warmup 1 --no-cpu-baseline --refill $rf --node-min $nm 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cfg5 refill $rf node_min $nm', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done; done
