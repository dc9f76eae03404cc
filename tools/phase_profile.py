"""Diagnostic: per-phase wave executions, lane occupancy and cycle shares of k_fused
(run with RT_PROFILE_PHASES=1; the run time of this build is not meaningful)."""
import os, sys
os.environ["RT_PROFILE_PHASES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import scenes, abi
from raytracer_amd.renderer import Renderer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_random_spheres_1080p"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scene, w, h, _, depth, seed = scenes.make_workload(name)
r = Renderer(scene, w, h)
r.draw(spp, depth, seed)
torch.cuda.synchronize()
pc = r.workspace[256:256 + 96].cpu().numpy().view(np.uint64).reshape(4, 3)
tot = pc[:, 2].sum()
for nm, (ex, ln, cy) in zip(["shade", "raygen", "node", "leaf"], pc):
    print("%-7s execs %12d  lanes/exec %5.1f  cycles/exec %7.1f  cycle share %.3f" % (nm, ex, ln / max(ex, 1), cy / max(ex, 1), cy / tot))
print("samples", w * h * spp)
