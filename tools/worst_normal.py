"""Prints the camera ray of a workload whose first-hit normal differs most from the double oracle's:
object, t on both sides, both normals, front_face, and normal . direction (a value near zero means a
grazing hit, where the side the normal is flipped to is decided by rounding).
usage (GPU box): python tools/worst_normal.py [workload] [width height]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_bridge as ob  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.renderer import Context, NativeScene, camera_rays  # noqa: E402
from raytracer_amd.scene import RenderConfig  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_random_spheres_400x225"
scene, w, h, _, depth, seed = scenes.make_workload(name)
if len(sys.argv) > 3:
    w, h = int(sys.argv[2]), int(sys.argv[3])
ctx = Context(0)
nat = NativeScene(ctx)
scene.encode(nat)
nat.commit()
orc = ob.oracle_scene_from(scene)
rays = camera_rays(ctx, scene.camera, RenderConfig(1, 1, 1, seed, 0), w, h)
g, o = nat.first_hit(rays), orc.first_hit(rays)
hit = (g["object"] == o["object"]) & (o["object"] >= 0)
dn = np.where(hit, np.abs(g["normal"].astype(np.float64) - o["normal"]).max(axis=1), 0)
for i in np.argsort(-dn)[:5]:
    d = rays[i, 3:].astype(np.float64)
    print("ray %d pixel (%d, %d): object %d, normal error %.4g" % (i, i % w, i // w, o["object"][i], dn[i]))
    print("   t gpu %.9g  oracle %.12g   front_face gpu %d oracle %d" % (g["t"][i], o["t"][i], g["front_face"][i], o["front_face"][i]))
    print("   normal gpu %s  oracle %s" % (g["normal"][i], o["normal"][i]))
    print("   normal . direction: gpu %.3g  oracle %.3g" % (float(g["normal"][i].astype(np.float64) @ d), float(o["normal"][i] @ d)))
