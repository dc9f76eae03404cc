"""What the parity bars of tests/ are set from: the measured differences between the HIP path and the
double-precision oracle, printed per case / class.
usage (GPU box): python tools/parity_measure.py [known] [csg] [kit] [cfg5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_bridge as ob  # noqa: E402
from common import check_event, load_metal_tests, random_rays, shape_from_json  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.renderer import Context, NativeScene, Renderer  # noqa: E402
from raytracer_amd.scene import RenderConfig, SceneEncoder  # noqa: E402

what = set(sys.argv[1:]) or {"known", "csg", "kit", "kitwhere", "cfg5"}
ctx = Context(0)
threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")))


class Ev:
    def __init__(self, e):
        self.t = float(e["t"]); self.point = [float(v) for v in e["point"]]
        self.normal = [float(v) for v in e["normal"]]; self.uv = [float(v) for v in e["uv"]]
        self.material_offset = int(e["material_offset"]); self.is_exit = int(e["is_exit"])


if "known" in what:
    fix = load_metal_tests()
    for case in fix["cases"]:
        sink = NativeScene(ctx)
        sink.add_dummy_materials(48)
        sid = shape_from_json(case["shape"]).as_impl(SceneEncoder(sink))
        sink.commit()
        ray = np.array([case["ray"][0] + case["ray"][1]], dtype=np.float32)
        events = [Ev(e) for e in sink.intervals(sid, ray, scripted=case["rng"])[0]]
        exp = case["hits"]
        pairs = list(zip(events, exp))[:1] if case.get("first_only") else list(zip(events, exp))
        passed = None
        for k in range(26, 10, -1):
            try:
                for i, (ev, ex) in enumerate(pairs):
                    check_event(ev, ex, 2.0 ** -k, case["name"])
                passed = k
                break
            except AssertionError:
                continue
        print("known-answer %-40s events %d/%d  passes at 2^-%s" % (case["name"], len(events), len(exp), passed))

if "csg" in what:
    for name, shape, center in (("composition", scenes._composition_object(), (0, 0, 0)), ("glass", scenes._glass(), (0, 6, 0))):
        nat = NativeScene(ctx)
        sid = shape.as_impl(SceneEncoder(nat))
        nat.commit()
        orc = ob.OracleScene()
        oid = shape.as_impl(SceneEncoder(orc))
        rays = random_rays(3000, 11, center=center, radius=9.0, aim_radius=2.5)
        got = nat.intervals(sid, rays)
        count_bad = 0
        rel, dn, graz_rel, graz_dn = [], [], [], []
        for i in range(len(rays)):
            ref = orc.enumerate(oid, rays[i, :3], rays[i, 3:], precision=ob.ORC_F64)
            ev = got[i]
            if len(ev) != len(ref):
                count_bad += 1
                continue
            d = rays[i, 3:].astype(np.float64)
            for a, b in zip(ev, ref):
                r = abs(float(a["t"]) - b.t) / max(abs(b.t), 1e-3)
                n = max(abs(float(x) - y) for x, y in zip(a["normal"], b.normal))
                nd = abs(float(np.dot(np.array(b.normal), d)))
                g = nd < 3.5e-3
                (graz_rel if g else rel).append(r)
                (graz_dn if g else dn).append(n)
                if r > 5e-6:
                    print("   event over 5e-6: ray %d t %.6f dt/|t| %.3g |n.d| %.4f material %d exit %d" % (i, b.t, r, nd, b.material_offset, b.is_exit))
        rel, dn = np.array(rel), np.array(dn)
        print("csg %-12s rays %d count mismatches %d; plain events %d: dt/|t| max %.3g p99.9 %.3g, over 1e-5: %d; normal max %.3g, over 5e-5: %d; "
              "grazing events %d: dt/|t| max %.3g normal max %.3g"
              % (name, len(rays), count_bad, len(rel), rel.max(), np.percentile(rel, 99.9), int((rel > 1e-5).sum()), dn.max(), int((dn > 5e-5).sum()),
                 len(graz_rel), max(graz_rel or [0]), max(graz_dn or [0])))


def image_stats(got, ref):
    a, b = np.minimum(got, 4.0), np.minimum(ref, 4.0)
    mse = float(np.mean((a - b) ** 2))
    lum = lambda x: 0.2126 * x[..., 0] + 0.7152 * x[..., 1] + 0.0722 * x[..., 2]
    return dict(psnr=10 * np.log10(1.0 / max(mse, 1e-30)),
                equal=float(np.mean(np.max(np.abs(got - ref), axis=2) <= 1e-3 * np.maximum(1.0, np.max(ref, axis=2)))),
                rel_mean=abs(got.mean() - ref.mean()) / max(ref.mean(), 1e-9),
                rel_mean_clipped4=abs(a.mean() - b.mean()) / max(b.mean(), 1e-9),
                rel_mean_clipped1=abs(np.minimum(got, 1).mean() - np.minimum(ref, 1).mean()) / max(np.minimum(ref, 1).mean(), 1e-9),
                rel_lum_clipped1=abs(np.minimum(lum(got), 1).mean() - np.minimum(lum(ref), 1).mean()) / np.minimum(lum(ref), 1).mean())


if "kit" in what:
    for name in ("f3_kit_world1_1080p",):
        scene, w, h, _, depth, seed = scenes.make_workload(name)
        orc = ob.oracle_scene_from(scene)
        for spp in (32, 128):
            r = Renderer(scene, w, h, ctx=ctx)
            r.draw(spp, depth, seed)
            torch.cuda.synchronize()
            got = r.mean().cpu().numpy().astype(np.float64)
            ref, _, _ = orc.render(scene.camera, RenderConfig(spp, depth, 1, seed, 0), w, h, precision=ob.ORC_F64, threads=threads)
            print(name, spp, "spp:", {k: round(float(v), 5) for k, v in image_stats(got, ref / spp).items()}, flush=True)

if "kitwhere" in what:
    # where in the image does the Kit world's mean differ?  per first-hit object, and per 4 x 4 block of the image
    from raytracer_amd.renderer import camera_rays
    name = "f3_kit_world1_1080p"
    scene, w, h, _, depth, seed = scenes.make_workload(name)
    w, h, spp = 480, 270, 256
    orc = ob.oracle_scene_from(scene)
    r = Renderer(scene, w, h, ctx=ctx)
    for d in (1, 2, 3, 5, depth):
        r.reset()
        r.draw(spp, d, seed)
        torch.cuda.synchronize()
        got = r.mean().cpu().numpy().astype(np.float64)
        ref, _, _ = orc.render(scene.camera, RenderConfig(spp, d, 1, seed, 0), w, h, precision=ob.ORC_F64, threads=threads)
        ref = ref / spp
        f32, _, _ = orc.render(scene.camera, RenderConfig(spp, d, 1, seed, 0), w, h, precision=ob.ORC_F32, threads=threads)
        f32 = f32 / spp
        print("kit world1 %dx%d x %d spp depth %2d: mean hip %.6f  oracle f64 %.6f  oracle f32 %.6f  (hip-f64)/f64 %+.5f  (f32-f64)/f64 %+.5f"
              % (w, h, spp, d, got.mean(), ref.mean(), f32.mean(), (got.mean() - ref.mean()) / ref.mean(), (f32.mean() - ref.mean()) / ref.mean()), flush=True)
    rays = camera_rays(ctx, scene.camera, RenderConfig(1, 1, 1, seed, 0), w, h)
    first = r.native.first_hit(rays)["object"].reshape(h, w)
    diff = (got - ref).sum(axis=2)
    for obj in np.unique(first):
        m = first == obj
        print("   first-hit object %3d: pixels %6d  sum hip %.3f  sum f64 %.3f  diff %+.4f" % (obj, m.sum(), got.sum(axis=2)[m].sum(), ref.sum(axis=2)[m].sum(), diff[m].sum()))

if "cfg5" in what:
    cloud, w, h, spp, depth, seed = scenes.make_workload("cfg5_sphere_cloud_4k")
    nat = NativeScene(ctx)
    cloud.encode(nat)
    nat.commit()
    orc = ob.OracleScene()
    cloud.encode(orc)
    orc.commit()
    x0, y0, ww, wh = 1800, 1000, 256, 128
    for d in (1, 2, 3):
        r = Renderer(cloud, w, h, native=nat)
        r.draw(4, d, seed)
        torch.cuda.synchronize()
        got = r.accum[y0:y0 + wh, x0:x0 + ww].cpu().numpy().astype(np.float64) / 4
        ref, _, _ = orc.render(cloud.camera, RenderConfig(4, d, 1, seed, 0), w, h, window=(x0, y0, ww, wh), threads=threads)
        print("cfg5 4K window 256x128 x 4 spp depth", d, {k: round(float(v), 5) for k, v in image_stats(got, ref / 4).items()}, flush=True)
