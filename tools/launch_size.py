"""Diagnostic: kernel time of the fused kernel against samples per launch and path depth (the fixed cost of a
launch is the ramp at its start and the longest paths at its end).  usage (GPU box): python tools/launch_size.py [workload [width height]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_random_spheres_1080p"
scene, w, h, _, _, seed = scenes.make_workload(name)
if len(sys.argv) > 3:
    w, h = int(sys.argv[2]), int(sys.argv[3])
print(name, w, h)
r = Renderer(scene, w, h, timing=True)
for depth in (1, 2, 5, 10, 50):
    row = []
    for spp in (1, 4, 16, 64, 256):
        r.draw(spp, depth, seed)
        torch.cuda.synchronize()
        r.kernel_time(reset=True)
        for _ in range(5):
            r.draw(spp, depth, seed)
        torch.cuda.synchronize()
        ms, n, _ = r.kernel_time(reset=True)
        row.append(ms / n)
    print("depth %2d  kernel ms at 1/4/16/64/256 spp: " % depth + "  ".join("%8.3f" % v for v in row)
          + "   per-spp slope 64->256: %.4f ms, fixed part at 64 spp: %.3f ms" % ((row[4] - row[3]) / 192, row[3] - 64 * (row[4] - row[3]) / 192))
