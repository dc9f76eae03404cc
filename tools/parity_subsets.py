"""Where does a converged image's mean differ between the HIP path and the oracle?  Renders sub-scenes of a workload
(the camera, the chosen objects) on both sides with the same sample streams and prints the relative difference of the
image means by maximum path length, for the double and the float build of the oracle.
usage (GPU box): python tools/parity_subsets.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_bridge as ob  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.renderer import Context, Renderer  # noqa: E402
from raytracer_amd.scene import RenderConfig, Scene  # noqa: E402

ctx = Context(0)
threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")))
full = scenes.kit_world1()
names = ["ground", "lens", "earth", "light", "mirror", "barrel0", "barrel1", "barrel2", "quad_v", "quad_h"]
subsets = {
    "ground+light": [0, 3],
    "ground+light+quads": [0, 3, 8, 9],
    "ground+light+quad_v": [0, 3, 8],
    "ground+light+quad_h": [0, 3, 9],
    "light+quads": [3, 8, 9],
    "ground+light+lens": [0, 3, 1],
    "ground+light+mirror": [0, 3, 4],
    "ground+light+earth": [0, 3, 2],
    "ground+light+barrel0": [0, 3, 5],
    "all": list(range(10)),
}
w, h, spp = 240, 135, 256
for label, idx in subsets.items():
    scene = Scene(full.camera, [full.objects[i] for i in idx], convention=full.convention)
    orc = ob.oracle_scene_from(scene)
    r = Renderer(scene, w, h, ctx=ctx)
    line = []
    for d in (3, 6, 50):
        r.reset()
        r.draw(spp, d, 3)
        torch.cuda.synchronize()
        got = r.mean().cpu().numpy().astype(np.float64)
        ref = orc.render(scene.camera, RenderConfig(spp, d, 1, 3, 0), w, h, precision=ob.ORC_F64, threads=threads)[0] / spp
        f32 = orc.render(scene.camera, RenderConfig(spp, d, 1, 3, 0), w, h, precision=ob.ORC_F32, threads=threads)[0] / spp
        bad = int((~np.isfinite(f32)).any(axis=2).sum())
        f32 = np.nan_to_num(f32, nan=0.0, posinf=0.0, neginf=0.0)
        line.append("depth %2d: f64 %.5f hip %+.5f f32 %+.5f (%d bad px)" % (d, ref.mean(), (got.mean() - ref.mean()) / max(ref.mean(), 1e-12),
                                                                          (f32.mean() - ref.mean()) / max(ref.mean(), 1e-12), bad))
    print("%-24s %s" % (label, " | ".join(line)), flush=True)
