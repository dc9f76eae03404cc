#!/usr/bin/env python3
"""Host time of rt_scene_commit (tree build + upload) for a named workload, per builder.

    python tools/commit_time.py [workload] [repeats]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from raytracer_amd import abi                                   # noqa: E402
from raytracer_amd.renderer import Context, NativeScene         # noqa: E402
from raytracer_amd.scenes import make_workload                  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_sphere_cloud_4k"
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    scene = make_workload(name)[0]
    ctx = Context(0)
    for label, builder in (("surface-area", abi.RT_BVH_SAH), ("median", abi.RT_BVH_MEDIAN_SPLIT)):
        times = []
        for _ in range(repeats):
            native = NativeScene(ctx)
            scene.encode(native)
            native.set_option(abi.RT_OPTION_BVH_BUILDER, builder)
            t0 = time.perf_counter()
            native.commit()
            times.append(time.perf_counter() - t0)
            st = native.stats()
            del native
        print("%s %s builder: commit %s s  objects %d nodes %d depth %d" % (
            name, label, " ".join("%.3f" % t for t in times), st["n_objects"], st["n_nodes"], st["bvh_depth"]), flush=True)


if __name__ == "__main__":
    main()
