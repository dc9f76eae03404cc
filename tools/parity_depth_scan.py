"""How the HIP image and the double-precision oracle's drift apart with path length: PSNR and the
share of pixels equal to 1e-3 at max_depth 1, 2, 3, 5, 50 (same sample streams on both sides).
usage (GPU box): python tools/parity_depth_scan.py <workload>"""
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
from raytracer_amd.scene import RenderConfig  # noqa: E402

ctx = Context(0)
scene, w, h, _, _, seed = scenes.make_workload(sys.argv[1])
w, h, spp = 240, 135, 16
orc = ob.oracle_scene_from(scene)
for depth in (1, 2, 3, 5, 50):
    ref, _, _ = orc.render(scene.camera, RenderConfig(spp, depth, 1, seed, 0), w, h, precision=ob.ORC_F64)
    ref = ref / spp
    r = Renderer(scene, w, h, ctx=ctx)
    r.draw(spp, depth, seed)
    torch.cuda.synchronize()
    img = r.mean().cpu().numpy().astype(np.float64)
    err = np.abs(img - ref).max(axis=-1)
    mse = float(np.mean((np.minimum(img, 4.0) - np.minimum(ref, 4.0)) ** 2))
    print("depth %2d  psnr %.1f dB  pixels equal to 1e-3: %.4f  to 1e-2: %.4f  mean %.4f vs %.4f" %
          (depth, 10 * np.log10(1.0 / max(mse, 1e-30)), (err <= 1e-3).mean(), (err <= 1e-2).mean(), img.mean(), ref.mean()))
