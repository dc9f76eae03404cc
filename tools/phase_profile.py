"""Diagnostic: per-phase wave executions, lane occupancy and cycle shares of k_fused (the
diagnostic build of the kernel, RT_OPTION_PROFILE_PHASES; its run time is not meaningful).

usage (GPU box): python tools/phase_profile.py <workload> <spp> [executed.json]

For a spheres-only workload the counters also price the work the kernel EXECUTES per sample
(its own tree, nearest child first): node visits x two (or four) box tests + sphere tests + the shading,
ray generation and accumulation of the algorithmic count.  Written to the third argument as
the record bench.py quotes beside the algorithmic fraction (profiles/executed_<workload>.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import roofline, scenes  # noqa: E402
from raytracer_amd.renderer import Renderer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_random_spheres_1080p"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scene, w, h, _, depth, seed = scenes.make_workload(name)
import os as _os
from raytracer_amd import abi  # noqa: E402
_nodes = {"auto": abi.RT_NODES_AUTO, "f32": abi.RT_NODES_F32, "f16": abi.RT_NODES_F16}[_os.environ.get("RT_NODES", "auto")]
r = Renderer(scene, w, h, profile_phases=True, nodes=_nodes, refill=int(_os.environ["RT_REFILL"]) if "RT_REFILL" in _os.environ else None)
r.draw(spp, depth, seed)
torch.cuda.synchronize()
raw = r.workspace[1024:1024 + 112].cpu().numpy().view(np.uint64)
pc = raw[:12].reshape(4, 3)
object_flops, list_boxes = float(raw[12]), float(raw[13])
tot = pc[:, 2].sum()
for nm, (ex, ln, cy) in zip(["shade", "raygen", "node", "leaf"], pc):
    print("%-7s execs %12d  lanes/exec %5.1f  cycles/exec %7.1f  cycle share %.3f" % (nm, ex, ln / max(ex, 1), cy / max(ex, 1), cy / tot))
samples = w * h * spp
print("samples", samples)
print("node visits per sample %.2f  leaf tests per sample %.2f  flops of object tests per sample %.1f  list box tests per sample %.2f"
      % (pc[2, 1] / samples, pc[3, 1] / samples, object_flops / samples, list_boxes / samples))
st = r.native.stats()
fixture = os.path.join(ROOT, "tests", "golden", "counters_%s.json" % name)
if len(sys.argv) > 3 and os.path.exists(fixture):
    cnt = json.load(open(fixture))["counters"]
    n = float(cnt["samples"])
    shade = (sum(cnt["shades"][k] * f for k, f in roofline.F_SHADE.items()) + cnt["hits"] * roofline.F_HIT) / n
    nodes, leaves = pc[2, 1] / samples, pc[3, 1] / samples
    # the render kernel walks the four-children form when the scene has one, unless the environment asks otherwise
    boxes_per_visit = 4 if st["n_wide_nodes"] and not (_os.environ.get("RT_DEEP_BINARY") or _os.environ.get("RT_DEEP_PLAIN")) else 2
    executed = (roofline.F_RAYGEN + roofline.F_ACCUM + shade + nodes * boxes_per_visit * roofline.F_BOX + object_flops / samples
                + list_boxes / samples * roofline.F_BOX / 2)
    json.dump({"workload": name, "spp": spp, "node_visits_per_sample": nodes, "object_tests_per_sample": leaves,
               "object_test_flops_per_sample": object_flops / samples, "list_box_tests_per_sample": list_boxes / samples,
               "shade_flops_per_sample": shade, "executed_flops_per_sample": executed, "box_tests_per_node_visit": boxes_per_visit,
               "note": "phase counters of the diagnostic k_fused build (tools/phase_profile.py): lanes that took part in node steps "
                       "(a node visit is box_tests_per_node_visit box tests) and object tests, each object test priced by the kinds of primitive it "
                       "computes intervals of (roofline.py constants; a CSG shape: its leaves + two merge events per leaf); "
                       "shading from the oracle's event counts"}, open(sys.argv[3], "w"), indent=1)
    print("executed flops per sample %.1f" % executed)
