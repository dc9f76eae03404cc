"""Motion blur (RayTracingKit/Hittable.swift:528-602, Transform.swift:82-154): the
reference's own known answers (RayTracingKitTests/TransformTests.swift) against the oracle
and the host code, then the HIP path against the oracle."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_bridge as ob
from common import random_rays
from raytracer_amd import abi, scenes
from raytracer_amd.scene import (CameraConfig, ColoredLambertian, ColoredMetal, Cuboid, Cylinder, MotionBlur, RenderConfig,
                                 Scene, SceneEncoder, Sphere, Subtract, Transform)
from raytracer_amd.simd import Quat

T = Transform


def kit_quat(degrees, axis):
    """Quaternion.init(degrees:axis:normalized:), Transform.swift:193-206: (w, x, y, z), w >= 0."""
    r = math.radians(degrees)
    c = math.cos(r / 2)
    s = math.sin(r / 2) * (-1 if c < 0 else 1)
    a = np.asarray(axis, dtype=np.float64)
    return (C.c_double * 4)(abs(c), *(a * s))


def d3(*v):
    return (C.c_double * 3)(*v)


# ---------------------------------------------------------------- reference known answers
def test_reference_bounding_box_for_animation():
    """TransformTests.swift:32-37: a point at (10,0,0) turning 120 degrees about z sweeps a
    box of size (15, 10, 0)."""
    L = ob.lib()
    L.orc_animation_box_of_point.restype = None
    out = (C.c_double * 6)()
    L.orc_animation_box_of_point(kit_quat(120, (0, 0, 1)), d3(0, 0, 0), d3(10, 0, 0), out)
    size = np.array(out[3:]) - np.array(out[:3])
    assert np.sum((size - [15, 10, 0]) ** 2) < 1e-12            # eps = 1e-6 in the reference


def test_reference_animation_box_contains_the_path():
    """TransformTests.swift:38-51: with a translation as well, every tr.pow(i/100) image of
    the point lies in the box (threshold 1e-6)."""
    L = ob.lib()
    L.orc_animation_box_of_point.restype = None
    L.orc_transform_pow.restype = None
    q, tr, p = kit_quat(120, (0, 0, 1)), d3(-5, -5, 0), d3(10, 0, 0)
    box = (C.c_double * 6)()
    L.orc_animation_box_of_point(q, tr, p, box)
    lo, hi = np.array(box[:3]), np.array(box[3:])
    for i in range(100):
        out = d3(0, 0, 0)
        L.orc_transform_pow(q, tr, C.c_double(i * 0.01), p, out)
        pt = np.array(out[:])
        assert (pt >= lo - 1e-6).all() and (pt <= hi + 1e-6).all(), i


def test_reference_transform_inverse():
    """TransformTests.swift:14-31 on the host mirror of Transform."""
    tr = T.rotation_deg(72, "y")
    inv = tr.inverse
    ref = T.rotation_deg(-72, "y")
    assert np.allclose(inv.rotation.v, ref.rotation.v, atol=1e-7) and abs(inv.rotation.r - ref.rotation.r) < 1e-7
    assert np.allclose(T.translate(1, 2, 3).inverse.translation, [-1, -2, -3])
    both = T(Quat.from_angle_axis(math.radians(72), (0, 1, 0)), (1, 2, 3))
    p = np.array([10, 2, -1], dtype=np.float32)
    assert np.sum((both.inverse.act(both.act(p)) - p) ** 2) < 1e-10


def test_static_transformed_equals_moving_at_the_end_of_the_motion():
    """Transformed (Hittable.swift:563-602) folded into the records gives the hits MotionBlur
    gives at time 1, where transform.pow(1) is the transform itself."""
    from raytracer_amd.scene import Transformed
    m = [ColoredLambertian((0.1 * i, 0.5, 0.5)) for i in range(1, 7)]
    tr = T(Quat.from_angle_axis(0.9, (0.6, 0.0, 0.8)), (1.5, -0.5, 0.25))
    bases = [Sphere(radius=0.7, material=m[0], transform=T.translate(0.2, 0.1, 0) * T.rotation_deg(40, "x")),
             Cuboid(T.translate(-0.5, -0.4, -0.3), (1.0, 0.8, 0.6), m[1]),
             Cylinder(T.translate(1, -1, 0), 0.4, 2.0, bottom=m[2], top=m[3], side=m[4]),
             Subtract(Cuboid(T.translate(-1, -1, -1), (2, 2, 2), m[3]), Sphere((0, 0, 0), 1.25, m[5]))]
    rays = random_rays(150, 5, radius=7.0, aim_radius=2.5)
    for base in bases:
        orc = ob.OracleScene()
        enc = SceneEncoder(orc)
        moving = MotionBlur(tr, base).as_impl(enc)
        still = Transformed(tr, base).as_impl(enc)
        orc.set_time(1.0)
        for r in rays:
            a = orc.enumerate(moving, r[:3], r[3:], precision=ob.ORC_F64)
            b = orc.enumerate(still, r[:3], r[3:], precision=ob.ORC_F64)
            assert len(a) == len(b), type(base)
            for x, y in zip(a, b):
                assert abs(x.t - y.t) < 2e-5 and np.allclose(x.normal, y.normal, atol=2e-5)
                assert np.allclose(x.uv, y.uv, atol=2e-4) and x.material_offset == y.material_offset


# ---------------------------------------------------------------- oracle and host, no GPU
def moving_shapes():
    m = [ColoredLambertian((0.1 * i, 0.5, 0.5)) for i in range(1, 7)]
    spin = T(Quat.from_angle_axis(1.1, (0.6, 0.0, 0.8)), (1.5, -0.5, 0.25))
    return {
        "sphere_slide": MotionBlur(T.translate(2, 0, 0), Sphere((0, 0, 0), 0.7, m[0])),
        "box_spin": MotionBlur(spin, Cuboid(T.translate(-0.5, -0.4, -0.3), (1.0, 0.8, 0.6), m[1])),
        "cylinder_spin": MotionBlur(T(Quat.from_angle_axis(2.6, (0, 0, 1))), Cylinder(T.translate(1, -1, 0), 0.4, 2.0,
                                                                                bottom=m[2], top=m[3], side=m[4])),
        "csg_spin": MotionBlur(spin, Subtract(Cuboid(T.translate(-1, -1, -1), (2, 2, 2), m[3]), Sphere((0, 0, 0), 1.25, m[5]))),
    }


def test_oracle_motion_at_time_zero_and_half():
    m = ColoredLambertian((0.5, 0.5, 0.5))
    orc = ob.OracleScene()
    moving = MotionBlur(T.translate(2, 0, 0), Sphere((0, 0, 0), 0.7, m)).as_impl(SceneEncoder(orc))
    still0 = Sphere((0, 0, 0), 0.7, m).as_impl(SceneEncoder(orc))
    still1 = Sphere((1, 0, 0), 0.7, m).as_impl(SceneEncoder(orc))
    rays = random_rays(200, 3, radius=5.0, aim_radius=1.5)
    for time, still in ((0.0, still0), (0.5, still1)):
        orc.set_time(time)
        for r in rays:
            a = orc.enumerate(moving, r[:3], r[3:], precision=ob.ORC_F64)
            b = orc.enumerate(still, r[:3], r[3:], precision=ob.ORC_F64)
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert abs(x.t - y.t) < 1e-12 and np.allclose(x.normal, y.normal, atol=1e-12) and np.allclose(x.point, y.point, atol=1e-12)


def test_host_bounds_match_oracle_and_enclose_the_motion():
    from raytracer_amd._native import lib
    for name, shape in moving_shapes().items():
        h = C.c_void_p()
        assert lib.rt_scene_create(None, C.byref(h)) == 0

        class Sink:      # the native description-only scene as an encoder sink
            def __getattr__(self, fn):
                def call(*args):
                    if fn == "add_material":
                        return lib.rt_add_material(h, C.byref(args[0]), C.sizeof(args[0]))
                    if fn == "shape_csg":
                        arr = (C.c_int * len(args[1]))(*args[1])
                        return lib.rt_shape_csg(h, args[0], len(args[1]), arr)
                    if fn == "shape_motion":
                        return lib.rt_shape_motion(h, args[0], C.byref(args[1]))
                    return getattr(lib, "rt_" + fn)(h, C.byref(args[0]))
                return call
        sid = shape.as_impl(SceneEncoder(Sink()))
        assert sid >= 0, lib.rt_last_error()
        out = (C.c_float * 6)()
        assert lib.rt_shape_bounds(h, sid, out) == 0
        lo, hi = np.array(out[:3], dtype=np.float64), np.array(out[3:], dtype=np.float64)
        orc = ob.OracleScene()
        oid = shape.as_impl(SceneEncoder(orc))
        olo, ohi = orc.bounds(oid)
        assert np.allclose(lo, olo, atol=2e-5) and np.allclose(hi, ohi, atol=2e-5), name
        # the base box's corners stay inside at every time
        base = shape.base.bounding_box
        for k in range(101):
            tr = shape.pow(k / 100.0)
            for i in range(8):
                c = np.array([(base.max if i & 1 else base.min)[0], (base.max if i & 2 else base.min)[1],
                              (base.max if i & 4 else base.min)[2]], dtype=np.float64)
                p = tr.act(c).astype(np.float64)
                assert (p >= lo - 1e-4).all() and (p <= hi + 1e-4).all(), (name, k)
        # errors: a moving volume, a non-unit quaternion
        bad = abi.RtMotion((C.c_float * 4)(0.5, 0.5, 0, 0), (C.c_float * 3)(0, 0, 0))
        from raytracer_amd.scene import Transform as _Tr
        sph = abi.RtSphere(_Tr().as_impl, 1.0, 0)
        mat = abi.RtColoredLambertianMaterial(abi.material_kind_lambertian_colored, abi.F3(.5, .5, .5))
        lib.rt_add_material(h, C.byref(mat), C.sizeof(mat))
        s1 = lib.rt_shape_sphere(h, C.byref(sph))
        assert lib.rt_shape_motion(h, s1, C.byref(bad)) < 0 and b"unit quaternion" in lib.rt_last_error()
        fog = lib.rt_shape_cdv(h, s1, 0.5)
        good = abi.RtMotion((C.c_float * 4)(1, 0, 0, 0), (C.c_float * 3)(1, 0, 0))
        assert lib.rt_shape_motion(h, fog, C.byref(good)) < 0
        assert lib.rt_scene_set_time(h, 1.5) < 0 and lib.rt_scene_set_time(h, 0.25) == 0
        lib.rt_scene_destroy(h)


def test_oracle_render_draws_a_time_only_when_something_moves():
    """The still scene's streams must not shift when motion support exists."""
    scene = scenes.simple_balls()
    orc = ob.oracle_scene_from(scene)
    a, _, _ = orc.render(scene.camera, RenderConfig(2, 8, 1, 7, 0), 24, 14, threads=2)
    moving = Scene(scene.camera, list(scene.objects) + [MotionBlur(T.translate(0, 0.3, 0), Sphere((0.3, 0.2, 0.5), 0.1,
                                                                                           ColoredLambertian((0.9, 0.1, 0.1))))])
    b, _, _ = ob.oracle_scene_from(moving).render(scene.camera, RenderConfig(2, 8, 1, 7, 0), 24, 14, threads=2)
    assert np.isfinite(b).all() and not np.array_equal(a, b)
    c, _, _ = ob.oracle_scene_from(scene).render(scene.camera, RenderConfig(2, 8, 1, 7, 0), 24, 14, threads=1)
    assert np.array_equal(a, c)


# ---------------------------------------------------------------- HIP path against the oracle
@pytest.fixture(scope="module")
def ctx():
    from raytracer_amd.renderer import Context
    return Context(0)


@pytest.mark.gpu
@pytest.mark.parametrize("time", [0.0, 0.37, 0.99])
def test_moving_shape_intervals(ctx, time):
    from raytracer_amd.renderer import NativeScene
    from test_gpu_edges import _interval_mismatches
    for name, shape in moving_shapes().items():
        nat = NativeScene(ctx)
        sid = shape.as_impl(SceneEncoder(nat))
        nat.commit()
        nat.set_time(time)
        orc = ob.OracleScene()
        oid = shape.as_impl(SceneEncoder(orc))
        orc.set_time(time)
        rays = random_rays(1200, 31, radius=7.0, aim_radius=2.5)
        bad = _interval_mismatches(nat, sid, orc, oid, rays)
        print(name, time, "mismatches", bad)
        assert bad <= 3, name


@pytest.mark.gpu
@pytest.mark.parametrize("time", [0.0, 0.5, 0.93])
def test_first_hits_of_a_moving_scene(ctx, time):
    from raytracer_amd.renderer import NativeScene, camera_rays
    from test_gpu_parity import compare_first_hits
    scene = scenes.bouncing_spheres(1)
    nat = NativeScene(ctx)
    scene.encode(nat)
    nat.commit()
    nat.set_time(time)
    orc = ob.oracle_scene_from(scene)
    orc.set_time(time)
    rays = np.concatenate([camera_rays(ctx, scene.camera, RenderConfig(1, 1, 1, 5, 0), 160, 90),
                           random_rays(6000, 8, center=(0, 0.5, 0), radius=9.0, aim_radius=6.0)]).astype(np.float32)
    # a moving sphere's normal goes through the frame of transform.pow(t): float sinf / cosf of the
    # half angle add to the 0.2-unit spheres' own floor (measured: at most 9.9e-5)
    compare_first_hits(nat.first_hit(rays), orc.first_hit(rays), rays, scene, what="bouncing t=%.2f" % time, normal_tol=2e-4)
    assert nat.stats()["n_objects"] == len(scene.objects)


@pytest.mark.gpu
def test_moving_scene_render_matches_oracle(ctx):
    import torch
    from raytracer_amd.renderer import Renderer
    from test_gpu_render import psnr
    m = ColoredLambertian((0.8, 0.3, 0.3))
    spin = T(Quat.from_angle_axis(1.3, (0.2, 1.0, 0.1) / np.linalg.norm((0.2, 1.0, 0.1))), (0.4, 0.0, 0.0))
    scene = Scene(CameraConfig(vertical_fov=50, look_from=(0, 1.5, 5), look_at=(0, 0.3, 0)), [
        Sphere((0, -100.5, 0), 100.0, ColoredLambertian((0.5, 0.5, 0.5))),
        MotionBlur(T.translate(0, 0.8, 0), Sphere((-1.2, 0, 0), 0.5, m)),
        MotionBlur(spin, Cuboid(T.translate(0.6, -0.5, -0.4), (0.9, 0.9, 0.8), ColoredMetal((0.8, 0.8, 0.9), 0.1))),
        Sphere((0, 0, -1.5), 0.5, ColoredLambertian((0.2, 0.3, 0.8))),
    ])
    w, h, spp, depth, seed = 128, 72, 16, 12, 9
    orc = ob.oracle_scene_from(scene)
    ref, _, _ = orc.render(scene.camera, RenderConfig(spp, depth, 1, seed, 0), w, h, precision=ob.ORC_F64)
    ref = ref / spp
    imgs = []
    for pipeline in (abi.RT_PIPELINE_FUSED, abi.RT_PIPELINE_WAVEFRONT):
        r = Renderer(scene, w, h, pipeline=pipeline, ctx=ctx)
        r.draw(spp, depth, seed)
        torch.cuda.synchronize()
        img = r.mean().cpu().numpy().astype(np.float64)
        imgs.append(img)
        p = psnr(img, ref)
        same = (np.abs(img - ref) <= 1e-3).mean()
        print("pipeline", pipeline, "PSNR %.1f dB, pixels equal to 1e-3: %.3f" % (p, same))
        assert p >= 30.0 and same > 0.8
    assert psnr(imgs[0], imgs[1]) >= 60.0
    # the blur is really there: the still scene differs
    still = Scene(scene.camera, [o.base if isinstance(o, MotionBlur) else o for o in scene.objects])
    r = Renderer(still, w, h, ctx=ctx)
    r.draw(spp, depth, seed)
    torch.cuda.synchronize()
    assert psnr(r.mean().cpu().numpy().astype(np.float64), ref) < 30.0


@pytest.mark.gpu
def test_moving_spheres_only_render_matches_oracle(ctx):
    """Scenes that hold nothing but spheres and moving spheres run the spheres kernel with the frame
    change in front of its sphere test: translations (no rotation is computed), a sphere turning about an
    axis away from its centre, a moving glass ball, more than 16 objects (the tree is walked, not the list)."""
    import torch
    from raytracer_amd.renderer import Renderer
    from raytracer_amd.scene import Dielectric
    from test_gpu_render import psnr
    rs = np.random.RandomState(4)
    axis = np.array((0.3, 1.0, -0.2)) / np.linalg.norm((0.3, 1.0, -0.2))
    objs = [Sphere((0, -100.5, 0), 100.0, ColoredLambertian((0.5, 0.5, 0.5))),
            MotionBlur(T(Quat.from_angle_axis(1.1, axis), (0.0, 0.3, 0.0)), Sphere((1.4, 0.1, 0.2), 0.45, ColoredMetal((0.9, 0.7, 0.4), 0.05))),
            MotionBlur(T.translate(0.5, 0.0, 0.0), Sphere((-0.3, 0.0, 0.8), 0.4, Dielectric(1.5))),
            Sphere((0, 0.3, -1.5), 0.8, ColoredMetal((0.8, 0.8, 0.9), 0.0))]
    for k in range(24):
        c = (rs.uniform(-3, 3), -0.3, rs.uniform(-3, 2))
        ball = Sphere(c, 0.2, ColoredLambertian(tuple(rs.uniform(0.1, 0.9, 3))))
        objs.append(MotionBlur(T.translate(0, rs.uniform(0, 0.5), 0), ball) if k % 3 else ball)
    scene = Scene(CameraConfig(vertical_fov=50, look_from=(0, 1.5, 5), look_at=(0, 0.2, 0)), objs)
    w, h, spp, depth, seed = 160, 90, 16, 12, 3
    orc = ob.oracle_scene_from(scene)
    ref, _, _ = orc.render(scene.camera, RenderConfig(spp, depth, 1, seed, 0), w, h, precision=ob.ORC_F64)
    ref = ref / spp
    imgs = []
    for pipeline in (abi.RT_PIPELINE_FUSED, abi.RT_PIPELINE_WAVEFRONT):
        r = Renderer(scene, w, h, pipeline=pipeline, ctx=ctx)
        assert r.native.stats()["n_objects"] == 28
        r.draw(spp, depth, seed)
        torch.cuda.synchronize()
        img = r.mean().cpu().numpy().astype(np.float64)
        imgs.append(img)
        p = psnr(img, ref)
        same = (np.abs(img - ref) <= 1e-3).mean()
        print("pipeline", pipeline, "PSNR %.1f dB, pixels equal to 1e-3: %.3f" % (p, same))
        assert p >= 80.0 and same > 0.995     # measured on MI355X: 93.7 dB, every pixel equal
    assert psnr(imgs[0], imgs[1]) >= 60.0


@pytest.mark.gpu
@pytest.mark.parametrize("textured", [False, True], ids=["plain", "textured"])
def test_small_moving_sphere_scenes(ctx, textured):
    """Four spheres, one of them moving (the object list, not the tree), without and with an image
    texture on the moving one: the kernels chosen for these feature mixes against the oracle."""
    import torch
    from raytracer_amd.renderer import Renderer
    from raytracer_amd.scene import ImageTexture, TexturedLambertian
    from test_gpu_render import psnr
    axis = np.array((0.1, 1.0, 0.3)) / np.linalg.norm((0.1, 1.0, 0.3))
    skin = TexturedLambertian(ImageTexture.procedural("stripes", 64, 32, 5)) if textured else ColoredLambertian((0.7, 0.4, 0.2))
    scene = Scene(CameraConfig(vertical_fov=40, look_from=(0, 1.2, 5), look_at=(0, 0.3, 0)), [
        Sphere((0, -100.5, 0), 100.0, ColoredLambertian((0.5, 0.5, 0.5))),
        MotionBlur(T(Quat.from_angle_axis(0.9, axis), (0.3, 0.2, 0.0)), Sphere((-0.6, 0.2, 0), 0.6, skin)),
        Sphere((1.0, 0.0, -0.4), 0.5, ColoredMetal((0.8, 0.8, 0.9), 0.1)),
        Sphere((0.2, 2.6, -1.0), 0.5, ColoredLambertian((0.9, 0.9, 0.9))),
    ])
    w, h, spp, depth, seed = 128, 72, 16, 10, 21
    orc = ob.oracle_scene_from(scene)
    ref, _, _ = orc.render(scene.camera, RenderConfig(spp, depth, 1, seed, 0), w, h, precision=ob.ORC_F64)
    ref = ref / spp
    for traversal in (abi.RT_TRAVERSAL_LIST, abi.RT_TRAVERSAL_TREE):
        r = Renderer(scene, w, h, ctx=ctx, traversal=traversal)
        r.draw(spp, depth, seed)
        torch.cuda.synchronize()
        img = r.mean().cpu().numpy().astype(np.float64)
        p = psnr(img, ref)
        same = (np.abs(img - ref) <= 1e-3).mean()
        print("textured", textured, "traversal", traversal, "PSNR %.1f dB, pixels equal to 1e-3: %.3f" % (p, same))
        assert p >= 62.0 and same > 0.995     # measured on MI355X: 70.8 and 69.9 dB, every pixel equal
