"""Writes tests/golden/counters_<workload>.json: operation counts per sample from an
instrumented run of the CPU oracle (float build) on each workload's scene and seed at
16 samples per pixel.  bench.py multiplies these counts by the per-operation constants
of raytracer_amd/roofline.py to get the algorithmic flops and bytes per sample.

Usage: python tests/golden/make_counters.py [workload ...]   (default: cfg1 cfg2 cfg3 cfg4 cfg5)
The 1080p and 4K workloads are counted at 480x270 (a decimation of the pixel grid: the
per-sample averages do not depend on the resolution) to keep the run to a minute each.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle_bridge as ob  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.scene import RenderConfig  # noqa: E402

names = sys.argv[1:] or ["cfg1_random_spheres_400x225", "cfg2_random_spheres_1080p", "cfg3_cornell_smoke_1080p",
                         "cfg4_csg_textured_1080p", "cfg5_sphere_cloud_4k"]
for name in names:
    scene, w, h, spp, depth, seed = scenes.make_workload(name)
    while w > 480:
        w, h = w // 2, h // 2
    orc = ob.oracle_scene_from(scene)
    t0 = time.time()
    # The float build of the oracle keeps the reference's sphere quadratic as written, which loses the
    # 0.05-0.25-unit spheres of the cloud 100-400 units from the ray origin in rounding (spurious hits,
    # 14 rays per sample instead of 2.5): that workload is counted in double precision.
    f64 = name.startswith("cfg5")
    _, threads, cnt = orc.render(scene.camera, RenderConfig(16, depth, 1, seed, 0), w, h,
                                 precision=ob.ORC_F64 if f64 else ob.ORC_F32, counters=True)
    out = {"workload": name, "counted_at": [w, h, 16, depth, seed], "precision": "f64" if f64 else "f32", "counters": cnt}
    with open(os.path.join(HERE, "counters_%s.json" % name), "w") as f:
        json.dump(out, f, indent=1)
    print(name, "%.1fs on %d threads" % (time.time() - t0, threads), cnt)
