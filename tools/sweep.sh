for cfg in "--refill 40" "--refill 48" "--refill 52" "--refill 56" "--refill 60" "--refill 64"; do
  timeout -k 10 200 python bench.py --spp 64 --steps 3 --warmup 1 --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$cfg', round(d['value']), round(d['ms_per_step'],2))" || exit 1
done
