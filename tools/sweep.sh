for rf in 48 52 56 60; do
  RT_LIB=$PWD/raytracer_amd/lib/var_empty.so timeout -k 10 200 python bench.py --spp 512 --steps 2 --warmup 1 --no-cpu-baseline --refill $rf 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('count-empty refill $rf', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done
for rf in 40 44; do
timeout -k 10 200 python bench.py --spp 512 --steps 2 --warmup 1 --no-cpu-baseline --refill $rf 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('baseline refill $rf', round(d['value']), round(d['ms_per_step'],3))"
done
