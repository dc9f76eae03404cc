set -e
timeout -k 10 600 python -u -m pytest tests -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -2
timeout -k 10 200 python bench.py --pipeline wavefront --spp 64 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('wavefront', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel_ms_avg'])"
