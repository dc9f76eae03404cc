"""Writes tests/golden/counters_<workload>.json: operation counts per sample from an
instrumented run of the CPU oracle (float build) on each workload's scene and seed at
16 samples per pixel.  bench.py multiplies these counts by the per-operation constants
of raytracer_amd/roofline.py to get the algorithmic flops and bytes per sample.

Usage: python tests/golden/make_counters.py [workload ...]   (default: cfg1 cfg2 cfg3 cfg4)
The 1080p workloads are counted at 480x270 (a 4x4 decimation of the pixel grid: the
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
                         "cfg4_csg_textured_1080p"]
for name in names:
    scene, w, h, spp, depth, seed = scenes.make_workload(name)
    if w > 480:
        w, h = w // 4, h // 4
    orc = ob.oracle_scene_from(scene)
    t0 = time.time()
    _, threads, cnt = orc.render(scene.camera, RenderConfig(16, depth, 1, seed, 0), w, h, precision=ob.ORC_F32, counters=True)
    out = {"workload": name, "counted_at": [w, h, 16, depth, seed], "precision": "f32", "counters": cnt}
    with open(os.path.join(HERE, "counters_%s.json" % name), "w") as f:
        json.dump(out, f, indent=1)
    print(name, "%.1fs on %d threads" % (time.time() - t0, threads), cnt)
