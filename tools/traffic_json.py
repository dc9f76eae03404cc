"""Turns the counter summary of tools/measure_traffic.sh into the per-launch HBM traffic record that
bench.py quotes (profiles/traffic_<workload>.json).
usage: python tools/traffic_json.py <workload> <spp> <summary.json> > profiles/traffic_<workload>.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes  # noqa: E402

name, spp, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
c = json.load(open(path))
_, w, h, _, _, _ = scenes.WORKLOADS[name]
samples = w * h * spp
# rocprofv3's FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of the fetched bytes
hbm = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
out = {"workload": name, "spp_per_launch": spp, "samples_per_launch": samples,
       "FETCH_SIZE_KB_per_launch": c["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": c["WRITE_SIZE"],
       "correction": "gfx950 FETCH_SIZE counts half of the fetched bytes: doubled; WRITE_SIZE taken as is",
       "hbm_bytes_per_launch": hbm, "hbm_bytes_per_sample": hbm / samples,
       "TCC_HIT_sum": c.get("TCC_HIT_sum"), "TCC_MISS_sum": c.get("TCC_MISS_sum"),
       "TCP_TCC_READ_REQ_sum": c.get("TCP_TCC_READ_REQ_sum"), "TCP_TCC_WRITE_REQ_sum": c.get("TCP_TCC_WRITE_REQ_sum"),
       "launches_averaged": c.get("_launches"),
       "note": "three separate rocprofv3 --pmc passes (tools/measure_traffic.sh) of `bench.py --spp %d --steps 2 --warmup 1`" % spp}
print(json.dumps(out, indent=1))
