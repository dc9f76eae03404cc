"""rt_render_static: the render entry point that allocates nothing (GPU needed).  Its output is
checked against rt_render's on the same scene, camera, seed and sample range - two calls in a row
on the library's per-GPU scratch array, the second on a smaller image - for a scene whose tree
needs the deep traversal-stack columns of the array and for one with CSG shapes, volumes and
textures.  Both entry points run the same kernel on the same random streams; the sums differ only
in the order of the float additions (tolerance 1e-5 of the image's largest sum)."""
import ctypes as C

import numpy as np
import pytest
import torch

from raytracer_amd import abi, scenes
from raytracer_amd.scene import RenderConfig

pytestmark = pytest.mark.gpu


def render_static(r, width, height, spp, depth, seed, first_sample=0):
    from raytracer_amd._native import check, lib
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device=r.ctx.device)
    rc = RenderConfig(spp, depth, 1, seed, first_sample).impl
    cc = r.camera.impl
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.rt_render_static(r.native.h, C.byref(cc), C.byref(rc), first_sample, width, height, accum.data_ptr(), stream))
    return accum


def deep_cloud():
    """200k nested spheres: a tree deeper than the 16 stack entries a lane keeps in LDS."""
    n = 200_000
    cloud = scenes.SphereCloud(n, 9)
    rng = np.random.default_rng(9)
    cloud.centers = (rng.standard_normal((n, 3)) * 0.01).astype(np.float32)
    cloud.radii = np.linspace(1.0, 60.0, n).astype(np.float32)
    return cloud


@pytest.mark.parametrize("make,sizes", [(deep_cloud, [(192, 108), (77, 41)]),
                                        (lambda: scenes.csg_textured(7), [(160, 90), (64, 24)]),
                                        (scenes.cornell_smoke, [(96, 96), (40, 56)])],
                         ids=["deep_cloud", "csg_textured", "cornell_smoke"])
def test_static_scratch_render_equals_rt_render(make, sizes):
    from raytracer_amd.renderer import Renderer
    scene = make()
    first = Renderer(scene, *sizes[0])
    if make is deep_cloud:
        assert first.native.stats()["bvh_depth"] > 16
    for k, (w, h) in enumerate(sizes):             # the second call is on a smaller image, on the same scratch
        ref = Renderer(scene, w, h, native=first.native)
        ref.draw(5, 12, 31 + k, first_sample=3)
        got = render_static(ref, w, h, 5, 12, 31 + k, first_sample=3)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all() and got.sum().item() > 0
        scale = max(1.0, ref.accum.abs().max().item())
        assert (got - ref.accum).abs().max().item() <= 1e-5 * scale, (w, h)
    # and once more at the first size: what the smaller call left in the scratch does not matter
    w, h = sizes[0]
    ref = Renderer(scene, w, h, native=first.native)
    ref.draw(2, 12, 77)
    got = render_static(ref, w, h, 2, 12, 77)
    torch.cuda.synchronize()
    assert (got - ref.accum).abs().max().item() <= 1e-5 * max(1.0, ref.accum.abs().max().item())


def test_static_scratch_render_refuses_oversized_images_and_bad_arguments():
    from raytracer_amd._native import lib
    from raytracer_amd.renderer import Renderer
    r = Renderer(scenes.simple_balls(), 16, 16)
    rc = RenderConfig(1, 2, 1, 1, 0).impl
    cc = r.camera.impl
    acc = torch.zeros((16, 16, 3), dtype=torch.float32, device=r.ctx.device)
    assert lib.rt_render_static(r.native.h, C.byref(cc), C.byref(rc), 0, 70000, 1, acc.data_ptr(), None) == -3      # RT_ERR_LIMIT
    assert b"image too large" in lib.rt_last_error()
    assert lib.rt_render_static(r.native.h, C.byref(cc), C.byref(rc), 0, 20000, 20000, acc.data_ptr(), None) == -3
    assert lib.rt_render_static(r.native.h, C.byref(cc), C.byref(rc), 0, 16, 16, None, None) == -1                  # RT_ERR_INVALID
    assert lib.rt_render_static(None, C.byref(cc), C.byref(rc), 0, 16, 16, acc.data_ptr(), None) == -1
    assert lib.rt_render_static(r.native.h, C.byref(cc), C.byref(rc), 0, 16, 16, acc.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert acc.sum().item() > 0


def test_static_scratch_render_allocates_nothing_and_runs_under_graph_capture():
    """The callers this entry point is for.  (1) Device memory: the free bytes the runtime reports
    (hipMemGetInfo) and torch's own allocator are where they were after a call.  (2) Stream capture:
    two calls (the second on a smaller image) recorded into a HIP graph on a side stream - a call that
    allocated, synchronised or queried the stream would fail the capture - then replayed twice; every
    replay adds the same sums rt_render gives."""
    from raytracer_amd._native import check, lib
    from raytracer_amd.renderer import Renderer
    scene = deep_cloud()
    sizes = [(192, 108), (77, 41)]
    first = Renderer(scene, *sizes[0])
    assert first.native.stats()["bvh_depth"] > 16
    refs, accs = [], []
    for k, (w, h) in enumerate(sizes):
        ref = Renderer(scene, w, h, native=first.native)
        ref.draw(3, 12, 91 + k, first_sample=2)
        refs.append(ref)
        accs.append(torch.zeros((h, w, 3), dtype=torch.float32, device=first.ctx.device))
    torch.cuda.synchronize()
    # (1) nothing allocated by a plain call
    warm = render_static(refs[0], *sizes[0], 1, 2, 5)       # torch allocates `warm` here, before the measurement
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    held0 = torch.cuda.memory_allocated()
    rc = RenderConfig(3, 12, 1, 91, 2).impl
    cc = refs[0].camera.impl
    check(lib.rt_render_static(first.native.h, C.byref(cc), C.byref(rc), 2, sizes[0][0], sizes[0][1], warm.data_ptr(),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0 and torch.cuda.memory_allocated() == held0
    # (2) capture and replay
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for k, (w, h) in enumerate(sizes):
            rc = RenderConfig(3, 12, 1, 91 + k, 2).impl
            cc = refs[k].camera.impl
            check(lib.rt_render_static(first.native.h, C.byref(cc), C.byref(rc), 2, w, h, accs[k].data_ptr(), stream))
    torch.cuda.synchronize()
    for replay in (1, 2):
        graph.replay()
        torch.cuda.synchronize()
        for k in range(2):
            want = refs[k].accum * replay         # the kernel adds to the accumulator it is given
            scale = max(1.0, want.abs().max().item())
            assert (accs[k] - want).abs().max().item() <= 1e-5 * scale, (replay, sizes[k])
