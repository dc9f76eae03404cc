for w in cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do
  timeout -k 10 400 python bench.py --workload $w --spp 32 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/err_$w.log | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w', round(d['value']), round(d['ms_per_step'],2), d.get('roofline',{}).get('frac'))" || tail -5 gpurun_out/err_$w.log
done
