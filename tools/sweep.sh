for rm in 12 16 20 24 28 32; do for rf in 48 52; do
  RT_RAYGEN_MIN=$rm timeout -k 10 200 python bench.py --spp 512 --steps 2 --warmup 1 --no-cpu-baseline --refill $rf 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('raygen_min $rm refill $rf', round(d['value']), round(d['ms_per_step'],3))" || exit 1
done; done
