"""Converged-image parity: per-pixel mean radiance of the HIP render loop against the
double-precision CPU oracle on the same scene, seed and sample indices (GPU needed).

Both renderers draw sample s of pixel (x, y) from the same generator stream, so most
paths are the same path; the few that diverge (a float32 rounding flips a branch) are
independent samples of the same estimator.  Every material kind's scatter rule is in
these paths (a wrong Schlick exponent or reflection rule changes most pixels of its
scene), so the bars are the measured values less a margin, per scene: PSNR of the mean
image, the fraction of pixels equal to 1e-3, and the relative difference of the image-wide
mean.  The two pipelines must agree with each other far more tightly (they are the same
float32 code in a different schedule: PSNR >= 60 dB, differences only from the order of
float adds and from paths whose volume draws see a different generator state word).
"""
import numpy as np
import pytest
import torch

import oracle_bridge as ob
from raytracer_amd import abi, scenes
from raytracer_amd.scene import RenderConfig

pytestmark = pytest.mark.gpu


def psnr(a, b, peak=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    peak = peak if peak is not None else max(1.0, float(np.percentile(b, 99.9)))
    return 10 * np.log10(peak * peak / max(mse, 1e-30))


def gpu_mean(scene, w, h, spp, depth, seed, pipeline):
    from raytracer_amd.renderer import Renderer
    r = Renderer(scene, w, h, pipeline=pipeline)
    r.draw(spp, depth, seed)
    torch.cuda.synchronize()
    return r.mean().cpu().numpy()


CASES = [
    ("random_spheres", lambda: scenes.random_spheres(1), 160, 90, 16, 50, 1),
    ("cornell_smoke", scenes.cornell_smoke, 96, 96, 32, 50, 3),
    ("csg_textured", lambda: scenes.csg_textured(7), 128, 72, 16, 50, 7),
    ("quads", lambda: scenes.quads(9), 96, 64, 32, 10, 9),
    ("world1", lambda: scenes.world1(21), 120, 68, 32, 50, 21),
]
# (PSNR dB, fraction of pixels equal to 1e-3, relative difference of the image mean); measured on MI355X:
# random_spheres 59.7 / 0.987 / 1e-4, cornell_smoke 64.0 / 0.997 / 2e-4, csg_textured 59.0 / 0.995 / 1e-4,
# quads 69.7 (tree) and 84.6 (object list: the fog box stands on the floor quad, and a path that meets both at exactly the
# same distance takes the one its traversal meets first - with other trees of the same scene the tree walk gave 111.7) / 1.000 / 1.3e-4, world1 45.5 / 0.919 / 8e-3 (a defocused camera and a
# small light: few, bright paths)
BARS = {"random_spheres": (55.0, 0.97, 2e-3), "cornell_smoke": (58.0, 0.99, 2e-3), "csg_textured": (54.0, 0.98, 2e-3),
        "quads": (64.0, 0.999, 5e-4), "world1": (41.0, 0.89, 2e-2)}


@pytest.mark.parametrize("name,make,w,h,spp,depth,seed", CASES, ids=[c[0] for c in CASES])
def test_mean_radiance_matches_oracle(name, make, w, h, spp, depth, seed):
    scene = make()
    orc = ob.oracle_scene_from(scene)
    rc = RenderConfig(spp, depth, 1, seed, 0)
    ref_sum, _, _ = orc.render(scene.camera, rc, w, h, precision=ob.ORC_F64)
    ref = ref_sum / spp
    for pipeline in (abi.RT_PIPELINE_WAVEFRONT, abi.RT_PIPELINE_FUSED):
        got = gpu_mean(scene, w, h, spp, depth, seed, pipeline)
        p = psnr(np.minimum(got, 4.0), np.minimum(ref, 4.0))
        rel = abs(got.mean() - ref.mean()) / max(ref.mean(), 1e-9)
        same = np.mean(np.max(np.abs(got - ref), axis=2) <= 1e-3 * np.maximum(1.0, np.max(ref, axis=2)))
        print(name, "pipeline", pipeline, "PSNR %.1f dB" % p, "mean rel %.4f" % rel, "pixels equal to 1e-3: %.3f" % same)
        need_db, need_same, need_rel = BARS[name]
        assert p >= need_db, (name, pipeline, p)
        assert same >= need_same, (name, pipeline, same)
        assert rel <= need_rel, (name, pipeline, rel)


@pytest.mark.parametrize("name,make,w,h,spp,depth,seed", CASES[:3], ids=[c[0] for c in CASES[:3]])
def test_pipelines_agree(name, make, w, h, spp, depth, seed):
    scene = make()
    a = gpu_mean(scene, w, h, spp, depth, seed, abi.RT_PIPELINE_WAVEFRONT)
    b = gpu_mean(scene, w, h, spp, depth, seed, abi.RT_PIPELINE_FUSED)
    p = psnr(a, b)
    print(name, "wavefront vs fused PSNR %.1f dB" % p, "max abs", np.abs(a - b).max())
    assert p >= 60.0


def test_passes_accumulate_like_one_call():
    """Two passes of 8 samples continue the sample index, so they equal one pass of 16
    (round-off of the accumulation order aside) - and a second renderer rendering only the
    upper half of the sample range gives the difference: the sharding rule of multi-GPU."""
    from raytracer_amd.renderer import Renderer
    scene = scenes.random_spheres(1)
    r1 = Renderer(scene, 96, 54, pipeline=abi.RT_PIPELINE_FUSED)
    r1.draw(16, 20, 5)
    r2 = Renderer(scene, 96, 54, pipeline=abi.RT_PIPELINE_FUSED)
    r2.draw(8, 20, 5)
    r2.draw(8, 20, 5)
    torch.cuda.synchronize()
    a, b = r1.accum.cpu().numpy(), r2.accum.cpu().numpy()
    assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(a).max())
    r3 = Renderer(scene, 96, 54, pipeline=abi.RT_PIPELINE_WAVEFRONT)
    r3.draw(8, 20, 5, first_sample=8)
    r4 = Renderer(scene, 96, 54, pipeline=abi.RT_PIPELINE_WAVEFRONT)
    r4.draw(8, 20, 5, first_sample=0)
    torch.cuda.synchronize()
    c = r3.accum.cpu().numpy() + r4.accum.cpu().numpy()
    assert np.abs(a - c).max() <= 1e-4 * max(1.0, np.abs(a).max())
