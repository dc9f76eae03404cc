set -e
timeout -k 10 600 python -u -m pytest tests -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -2
for cfg in "" "--refill 32" "--refill 40" "--node-min 24" "--node-min 40"; do
  timeout -k 10 200 python bench.py --spp 64 --steps 3 --warmup 1 --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$cfg', round(d['value']), round(d['ms_per_step'],2))"
done
