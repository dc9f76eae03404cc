"""Parity evidence in one file: for every workload, first-hit agreement of the HIP path with the
double-precision CPU oracle on the workload's own camera rays, and PSNR / RMSE of the mean radiance
(same seeds on both sides, reduced image size so that the oracle finishes in seconds).
usage (GPU box): python tools/parity_report.py [workload ...] > gpurun_out/parity_report.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_bridge as ob  # noqa: E402
from raytracer_amd import abi, scenes  # noqa: E402
from raytracer_amd.renderer import Context, Renderer, camera_rays  # noqa: E402
from raytracer_amd.scene import RenderConfig  # noqa: E402

ctx = Context(0)
out = {}
names = sys.argv[1:] or ["cfg1_random_spheres_400x225", "cfg3_cornell_smoke_1080p", "cfg4_csg_textured_1080p",
                         "cfg5_sphere_cloud_4k", "f2_bouncing_spheres_1080p", "f3_kit_world1_1080p", "f3_kit_world2_1080p"]
for name in names:
    scene, w, h, _, depth, seed = scenes.make_workload(name)
    spp = 16
    while w > 480:
        w, h = w // 2, h // 2
    if w == 480:
        w, h = 240, 135
    orc = ob.oracle_scene_from(scene)
    entry = {"image": [w, h], "spp": spp, "depth": depth, "seed": seed}
    r = Renderer(scene, w, h, ctx=ctx)
    if not name.startswith("f2"):        # a moving scene's camera rays carry a time each; its first hits are tested per time
        rays = camera_rays(ctx, scene.camera, RenderConfig(1, 1, 1, seed, 0), w, h)
        g, o = r.native.first_hit(rays), orc.first_hit(rays)
        same = g["object"] == o["object"]
        hit = same & (o["object"] >= 0)
        dt = np.abs(g["t"][hit].astype(np.float64) - o["t"][hit]) / np.maximum(np.abs(o["t"][hit]), 1e-3)
        entry["first_hit"] = {"rays": int(len(rays)), "hits": int(hit.sum()), "object_mismatches": int((~same).sum()),
                              "material_mismatches": int((g["material_offset"][hit] != o["material_offset"][hit]).sum()),
                              "t_rel_err_median": float(np.median(dt)), "t_rel_err_p99": float(np.percentile(dt, 99)),
                              "normal_abs_err_max": float(np.abs(g["normal"][hit].astype(np.float64) - o["normal"][hit]).max())}
    ref, _, _ = orc.render(scene.camera, RenderConfig(spp, depth, 1, seed, 0), w, h, precision=ob.ORC_F64)
    ref = ref / spp
    for pname, pipeline in (("fused", abi.RT_PIPELINE_FUSED), ("wavefront", abi.RT_PIPELINE_WAVEFRONT)):
        rr = Renderer(scene, w, h, pipeline=pipeline, ctx=ctx)
        rr.draw(spp, depth, seed)
        torch.cuda.synchronize()
        img = rr.mean().cpu().numpy().astype(np.float64)
        a, b = np.minimum(img, 4.0), np.minimum(ref, 4.0)
        mse = float(np.mean((a - b) ** 2))
        entry[pname] = {"psnr_db": float(10 * np.log10(1.0 / max(mse, 1e-30))), "rmse": float(np.sqrt(mse)),
                        "pixels_equal_1e-3": float((np.abs(img - ref) <= 1e-3).mean())}
    out[name] = entry
    print(name, json.dumps(entry), file=sys.stderr)
print(json.dumps(out, indent=1))
