"""CPU-only checks: the C-ABI library loads and exports every symbol of
include/raytracer_amd.h, record layouts match the header, and the host scene API
flattens every reference demo scene (into the oracle here: no GPU calls)."""
import ctypes as C
import os
import re

import numpy as np

import oracle_bridge as ob
from raytracer_amd import abi, scenes
from raytracer_amd.scene import Cylinder, Quad, Transform
from raytracer_amd.simd import Quat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "raytracer_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header)))
    from raytracer_amd import _native
    assert sorted(_native.EXPORTS) == declared
    for name in declared:
        assert hasattr(_native.lib, name), name
    assert _native.lib.rt_abi_version() == 3


def test_record_sizes_match_the_reference_layout():
    # sizes of the reference's records with Apple simd types (16-byte vector_float3):
    # MetalRayTracer/Renderable.h:13-63, Config.h:18-33, Materials.h:38-93
    assert C.sizeof(abi.RtTransform) == 64
    assert C.sizeof(abi.RtSphere) == 80
    assert C.sizeof(abi.RtCylinder) == 96
    assert C.sizeof(abi.RtCuboid) == 128
    assert C.sizeof(abi.RtQuad) == 96
    assert C.sizeof(abi.RtPerlinNoiseTexture) == 4912
    assert C.sizeof(abi.RtPerlinNoiseLambertianMaterial) == 4928        # the reference's largest material record
    assert C.sizeof(abi.RtColoredLambertianMaterial) == 32
    assert C.sizeof(abi.RtCameraConfig) == 80
    assert C.sizeof(abi.RtRenderConfig) == 24
    assert C.sizeof(abi.RtHit) == 48 and C.sizeof(abi.RtEvent) == 48


def reference_sphere_bytes(center, radius, material_offset, columns=((1, 0, 0), (0, 1, 0), (0, 0, 1))):
    """The 80 bytes of the reference's `Sphere` (Renderable.h:13-25) as a Swift host lays them out:
    three 16-byte rotation columns, a 16-byte translation, radius at 64, material_offset at 72."""
    import struct
    b = b"".join(struct.pack("<4f", *col, 0.0) for col in columns)
    b += struct.pack("<4f", *center, 0.0)
    b += struct.pack("<f4xQ", radius, material_offset)
    assert len(b) == 80
    return b


def test_raw_reference_layout_bytes_are_accepted():
    """A byte image of the reference's records goes through the C ABI as it is (no ctypes record
    involved), at an odd address too: bounds come back as centre +- radius."""
    lib, h = _host_scene()
    mat = bytes([1, 0, 0, 0]) + bytes(12) + np.array([.5, .5, .5, 0], dtype=np.float32).tobytes()   # ColoredLambertianMaterial, 32 bytes
    assert lib.rt_add_material(h, mat, len(mat)) == 0
    raw = reference_sphere_bytes((1.0, 2.0, 3.0), 0.5, 0)
    buf = C.create_string_buffer(b"\0" * 4 + raw)            # 4 bytes off any 16-byte boundary
    sid = lib.rt_shape_sphere(h, C.cast(C.addressof(buf) + 4, C.POINTER(abi.RtSphere)))
    assert sid == 0, lib.rt_last_error()
    out = (C.c_float * 6)()
    assert lib.rt_shape_bounds(h, sid, out) == 0
    assert np.allclose(list(out), [0.5, 1.5, 2.5, 1.5, 2.5, 3.5])
    # the ctypes record is the same bytes
    rec = abi.RtSphere(abi.RtTransform(abi.RtMatrix3x3(), abi.F3(1, 2, 3)), 0.5, 0)
    for c in range(3):
        rec.transform.rotation.columns[c] = abi.F3(*[1.0 if r == c else 0.0 for r in range(3)])
    assert bytes(rec) == raw
    lib.rt_scene_destroy(h)


def test_moving_shapes_are_refused_as_operands():
    """CSG leaves and volume bases are tested as plain primitives on the device, so a moving operand
    is an error at description time, in the library and in the host API."""
    import pytest
    from raytracer_amd.scene import (ColoredLambertian, ConstantDensityVolume, MotionBlur, Sphere, Subtract, Union)
    lib, h = _host_scene()
    mat = abi.RtColoredLambertianMaterial(abi.material_kind_lambertian_colored, abi.F3(.5, .5, .5))
    assert lib.rt_add_material(h, C.byref(mat), C.sizeof(mat)) == 0
    sph = abi.RtSphere(Transform().as_impl, 1.0, 0)
    a, b, c = (lib.rt_shape_sphere(h, C.byref(sph)) for _ in range(3))
    motion = abi.RtMotion((C.c_float * 4)(1, 0, 0, 0), (C.c_float * 3)(1, 0, 0))
    moving = lib.rt_shape_motion(h, a, C.byref(motion))
    assert moving >= 0
    two = (C.c_int * 2)(moving, b)
    assert lib.rt_shape_csg(h, abi.RT_CSG_UNION, 2, two) == -1 and b"moving" in lib.rt_last_error()
    assert lib.rt_shape_cdv(h, moving, 0.5) == -1 and b"moving" in lib.rt_last_error()
    lib.rt_scene_destroy(h)
    m = ColoredLambertian((.5, .5, .5))
    mb = MotionBlur(Transform.translate(1, 0, 0), Sphere((0, 0, 0), 1, m))
    with pytest.raises(TypeError):
        Union(mb, Sphere((0, 0, 0), 1, m))
    with pytest.raises(TypeError):
        Subtract(Sphere((0, 0, 0), 1, m), mb)
    with pytest.raises(TypeError):
        ConstantDensityVolume(mb, 0.5)


def test_error_paths_without_a_gpu():
    from raytracer_amd._native import lib
    # a null scene is rejected with a message, not a crash
    assert lib.rt_scene_commit(None) < 0
    assert b"null" in lib.rt_last_error()
    # header + the deep-stack columns of the largest grid: independent of the image, modest
    assert 4096 <= lib.rt_render_workspace_size(abi.RT_PIPELINE_FUSED, 1920, 1080) <= 64 << 20
    assert lib.rt_render_workspace_size(abi.RT_PIPELINE_FUSED, 64, 64) == lib.rt_render_workspace_size(abi.RT_PIPELINE_FUSED, 1920, 1080)
    ws = lib.rt_render_workspace_size(abi.RT_PIPELINE_WAVEFRONT, 1920, 1080)
    assert 1920 * 1080 * 8 * 160 < ws < 1920 * 1080 * 8 * 161          # 160 bytes per resident path: two queues of four 16-byte streams, hit, sample


def test_quaternion_matrix_matches_reference_expectations():
    # CuboidTests.swift:104-135 pins simd_matrix3x3(q2 * q1) through the entry normal
    q1 = Quat.from_angle_axis(np.float32(np.pi / 3), (0, 1, 0))
    q2 = Quat.from_angle_axis(np.float32(np.pi / 4), (0, 0, 1))
    cols = (q2 * q1).matrix_columns()
    # entry face of that test is `back` (normal = -column 2)
    assert np.allclose(-cols[2], [-0.61237234, -0.61237246, -0.5], atol=1e-6)
    assert np.allclose(cols[0], [0.35355335, 0.3535534, -0.8660254], atol=1e-6)


def test_cylinder_between_points_frame():
    c = Cylinder.between((1, 2, 3), (1, 2, 8), 0.5, material=None)
    assert abs(c.height - 5) < 1e-6
    assert np.allclose(c.transform.rotation.act((0, 1, 0)), [0, 0, 1], atol=1e-6)
    assert np.allclose(c.transform.translation, [1, 2, 3])


def test_demo_scenes_flatten_and_trace_on_the_oracle():
    for name, make in scenes.DEMO_SCENES.items():
        scene = make()
        orc = ob.oracle_scene_from(scene)
        rays = np.array([list(scene.camera.look_from) + list(
            (scene.camera.look_at - scene.camera.look_from) / np.linalg.norm(scene.camera.look_at - scene.camera.look_from))])
        hit = orc.first_hit(rays)
        assert hit["object"][0] >= -1, name


def test_synthetic_workloads_are_seeded_and_sized():
    a = scenes.random_spheres(1)
    b = scenes.random_spheres(1)
    assert len(a.objects) == len(b.objects)
    assert 470 <= len(a.objects) <= 490
    assert np.allclose(a.objects[17].transform.translation, b.objects[17].transform.translation)
    assert len(scenes.cornell_smoke().objects) == 8
    assert len(scenes.csg_textured().objects) == 2 + 1 + 64 + 2
    cloud = scenes.SphereCloud(1000, 42)
    assert cloud.centers.shape == (1000, 3) and cloud.radii.min() >= 0.05 and cloud.radii.max() <= 0.25


def test_oracle_render_is_deterministic_and_thread_independent():
    scene = scenes.simple_balls()
    orc = ob.oracle_scene_from(scene)
    from raytracer_amd.scene import RenderConfig
    rc = RenderConfig(2, 8, 1, 99, 0)
    a, _, _ = orc.render(scene.camera, rc, 32, 18, threads=1)
    b, _, cnt = orc.render(scene.camera, rc, 32, 18, threads=4, counters=True)
    assert np.array_equal(a, b)
    assert cnt["samples"] == 32 * 18 * 2 and cnt["rays"] >= cnt["samples"]


def test_quad_record_follows_reference_formulas():
    class Sink:
        def add_material(self, rec): return 0
        def shape_quad(self, rec): self.rec = rec; return 0
    from raytracer_amd.scene import ColoredLambertian, SceneEncoder
    s = Sink()
    Quad((1, 2, 3), (2, 0, 0), (0, 3, 0), ColoredLambertian((1, 1, 1))).as_impl(SceneEncoder(s))
    assert np.allclose(list(s.rec.normal), [0, 0, 1]) and abs(s.rec.d - 3) < 1e-6
    assert np.allclose(list(s.rec.w), [0, 0, 1 / 6.0])


def _host_scene():
    from raytracer_amd._native import lib
    h = C.c_void_p()
    assert lib.rt_scene_create(None, C.byref(h)) == 0
    return lib, h


def test_description_limits_and_errors_without_a_gpu():
    """What the reference answers with a crash at pipeline creation
    (ContentView.swift:136-138) comes back as an error code here."""
    lib, h = _host_scene()
    mat = abi.RtColoredLambertianMaterial(abi.material_kind_lambertian_colored, abi.F3(.5, .5, .5))
    assert lib.rt_add_material(h, C.byref(mat), C.sizeof(mat)) == 0
    assert lib.rt_add_material(h, C.byref(mat), 7) < 0                     # size does not match the kind
    bad = abi.RtColoredLambertianMaterial(77, abi.F3(0, 0, 0))
    assert lib.rt_add_material(h, C.byref(bad), C.sizeof(bad)) < 0
    tex = abi.RtTexturedLambertianMaterial(abi.material_kind_lambertian_textured, abi.RtImageTexture(3))
    assert lib.rt_add_material(h, C.byref(tex), C.sizeof(tex)) < 0         # no such texture
    sph = abi.RtSphere(Transform().as_impl, 1.0, 5)
    assert lib.rt_shape_sphere(h, C.byref(sph)) < 0 and b"material" in lib.rt_last_error()
    sph.material_offset = 0
    ids = [lib.rt_shape_sphere(h, C.byref(sph)) for _ in range(10)]
    assert ids == list(range(10))
    many = [lib.rt_shape_sphere(h, C.byref(sph)) for _ in range(65)]
    arr65 = (C.c_int * 65)(*many)
    assert lib.rt_shape_csg(h, abi.RT_CSG_UNION, 65, arr65) == -3          # RT_ERR_LIMIT: more than 64 leaves
    arr9 = (C.c_int * 9)(*many[:9])
    assert lib.rt_shape_csg(h, abi.RT_CSG_UNION, 9, arr9) == 75            # more than 8 leaves: a wide shape
    arr8 = (C.c_int * 8)(*ids[:8])
    u = lib.rt_shape_csg(h, abi.RT_CSG_UNION, 8, arr8)
    assert u == 76
    assert lib.rt_shape_csg(h, abi.RT_CSG_UNION, 8, arr8) < 0              # children already in use
    assert lib.rt_add_object(h, ids[0]) < 0                                # a CSG child is not an object
    fog = lib.rt_shape_cdv(h, ids[8], 0.5)
    assert fog == 77
    assert lib.rt_shape_cdv(h, fog, 0.5) < 0                               # volume of a volume
    two = (C.c_int * 2)(ids[9], fog)
    assert lib.rt_shape_csg(h, abi.RT_CSG_SUBTRACT, 2, two) < 0            # volume as a CSG operand
    assert lib.rt_shape_cdv(h, ids[9], 0.0) < 0                            # density must be positive
    assert lib.rt_add_object(h, u) == 0 and lib.rt_add_object(h, fog) == 1
    assert lib.rt_add_object(h, u) < 0                                     # twice
    assert lib.rt_scene_set_option(h, abi.RT_OPTION_BVH_BUILDER, 7) < 0
    assert lib.rt_scene_set_option(h, abi.RT_OPTION_BVH_BUILDER, abi.RT_BVH_MEDIAN_SPLIT) == 0
    assert lib.rt_scene_commit(h) == -2 and b"context" in lib.rt_last_error()   # RT_ERR_STATE: no device
    st = abi.RtSceneStats()
    assert lib.rt_scene_stats(h, C.byref(st)) == -2
    assert lib.rt_first_hit(h, None, 0, 0.0, 1.0, 0, None, None) == -2
    lib.rt_scene_destroy(h)
