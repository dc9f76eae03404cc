"""ctypes binding of the CPU oracle (oracle/liboracle.so) for the tests.

The oracle is test infrastructure: this file lives under tests/ and nothing in the
raytracer_amd package imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

ORC_F32, ORC_F64 = 0, 1


class OrcMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", C.c_float * 3), ("fuzz", C.c_float),
                ("refraction_index", C.c_float), ("texture", C.c_uint32)]


# The oracle keeps its own packed input records (oracle/oracle_api.h); the records of
# include/raytracer_amd.h are converted field by field.
_P3 = C.c_float * 3


class OrcTransform(C.Structure):
    _fields_ = [("rotation", C.c_float * 9), ("translation", _P3)]


class OrcSphere(C.Structure):
    _fields_ = [("transform", OrcTransform), ("radius", C.c_float), ("material_offset", C.c_size_t)]


class OrcCylinder(C.Structure):
    _fields_ = [("transform", OrcTransform), ("radius", C.c_float), ("height", C.c_float),
                ("bottom_material_offset", C.c_size_t), ("top_material_offset", C.c_size_t),
                ("side_material_offset", C.c_size_t)]


class OrcCuboid(C.Structure):
    _fields_ = [("transform", OrcTransform), ("size", _P3), ("material_offset", C.c_size_t * 6)]


class OrcQuad(C.Structure):
    _fields_ = [("origin", _P3), ("u", _P3), ("v", _P3), ("w", _P3), ("normal", _P3),
                ("d", C.c_float), ("material_offset", C.c_size_t)]


class OrcPerlin(C.Structure):
    _fields_ = [("colors", _P3 * 2), ("frequency", C.c_float), ("turbulence", C.c_uint32),
                ("vectors", _P3 * 256), ("permutations", (C.c_uint8 * 256) * 3)]


class OrcCameraConfig(C.Structure):
    _fields_ = [("vertical_FOV", C.c_float), ("look_from", _P3), ("look_at", _P3), ("up", _P3),
                ("defocus_angle", C.c_float), ("focus_distance", C.c_float), ("background", C.c_uint32)]


def _p3(v):
    return _P3(v.x, v.y, v.z)


def _orc_transform(t):
    o = OrcTransform()
    for c in range(3):
        col = t.rotation.columns[c]
        o.rotation[3 * c], o.rotation[3 * c + 1], o.rotation[3 * c + 2] = col.x, col.y, col.z
    o.translation = _p3(t.translation)
    return o


def _orc_camera(cc):
    return OrcCameraConfig(cc.vertical_FOV, _p3(cc.look_from), _p3(cc.look_at), _p3(cc.up), cc.defocus_angle,
                           cc.focus_distance, cc.background)


class OrcEvent(C.Structure):
    _fields_ = [("t", C.c_double), ("point", C.c_double * 3), ("normal", C.c_double * 3), ("uv", C.c_double * 2),
                ("material_offset", C.c_uint64), ("is_exit", C.c_int32), ("pad", C.c_int32)]


class OrcHit(C.Structure):
    _fields_ = [("t", C.c_double), ("point", C.c_double * 3), ("normal", C.c_double * 3), ("uv", C.c_double * 2),
                ("material_offset", C.c_uint64), ("object", C.c_int32), ("front_face", C.c_int32)]


class OrcCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("hits", C.c_uint64), ("box_tests", C.c_uint64),
                ("prim_tests", C.c_uint64 * 4), ("csg_events", C.c_uint64), ("cdv_intervals", C.c_uint64),
                ("shades", C.c_uint64 * 10)]

    def as_dict(self):
        return {"samples": self.samples, "rays": self.rays, "hits": self.hits, "box_tests": self.box_tests,
                "prim_tests": list(self.prim_tests), "csg_events": self.csg_events,
                "cdv_intervals": self.cdv_intervals, "shades": list(self.shades)}


class OrcRenderConfig(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32), ("first_sample", C.c_uint32),
                ("rng_seed", C.c_uint64)]


HIT_DTYPE = np.dtype([("t", "f8"), ("point", "f8", 3), ("normal", "f8", 3), ("uv", "f8", 2),
                      ("material_offset", "u8"), ("object", "i4"), ("front_face", "i4")])
assert HIT_DTYPE.itemsize == C.sizeof(OrcHit)

SCATTER_QUERY = np.dtype([("direction", "f8", 3), ("normal", "f8", 3), ("point", "f8", 3), ("uv", "f8", 2), ("front_face", "i4"),
                          ("material_offset", "u4"), ("sample", "u4"), ("x", "u4"), ("y", "u4"), ("pad", "u4")])
SCATTER_RESULT = np.dtype([("scattered", "i4"), ("pad", "i4"), ("direction", "f8", 3), ("attenuation", "f8", 3), ("emitted", "f8", 3)])

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_last_error.restype = C.c_char_p
        for name in ("orc_scene_destroy", "orc_add_material", "orc_add_texture_rgb8", "orc_add_perlin",
                     "orc_shape_sphere", "orc_shape_cylinder", "orc_shape_cuboid", "orc_shape_quad",
                     "orc_add_kit_perlin", "orc_add_kit_texture", "orc_kit_texture_color", "orc_shape_csg", "orc_shape_cdv", "orc_shape_motion", "orc_shape_bounds", "orc_add_object", "orc_commit", "orc_enumerate",
                     "orc_first_hit", "orc_render"):
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
    return rc


class OracleScene:
    """Sink for raytracer_amd.scene.SceneEncoder that feeds the oracle.  The geometry
    records are passed as they are (the oracle's input structs have the same layout as
    include/raytracer_amd.h); material records are flattened to the oracle's one struct."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_create())
        self.committed = False

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_destroy(self.h)
            self.h = None

    # ---- sink protocol
    def add_material(self, rec):
        m = OrcMaterial()
        m.kind = rec.kind
        k = rec.kind
        if k in (abi.material_kind_lambertian_colored, abi.material_kind_metal_colored,
                 abi.material_kind_emissive_colored, abi.material_kind_isotropic_colored):
            m.albedo = _p3(rec.albedo)
        if k in (abi.material_kind_lambertian_textured, abi.material_kind_metal_textured):
            m.texture = rec.albedo.texture_ptr
        if k in (abi.material_kind_lambertian_perlin_noise, abi.material_kind_metal_perlin_noise):
            src = rec.albedo
            p = OrcPerlin()
            p.colors[0], p.colors[1] = _p3(src.colors[0]), _p3(src.colors[1])
            p.frequency, p.turbulence = src.frequency, src.turbulence
            vec = np.frombuffer(bytes(src.vectors), dtype=np.float32).reshape(256, 4)[:, :3]
            C.memmove(p.vectors, np.ascontiguousarray(vec).tobytes(), 256 * 12)
            C.memmove(p.permutations, bytes(src.permutations), 768)
            m.texture = _check(self.L.orc_add_perlin(self.h, C.byref(p)))
        if k in (abi.material_kind_metal_colored, abi.material_kind_metal_textured, abi.material_kind_metal_perlin_noise):
            m.fuzz = rec.fuzz
        if k == abi.material_kind_dielectric:
            m.refraction_index = rec.refraction_index
        if k in (abi.material_kind_lambertian_kit_texture, abi.material_kind_emissive_kit_texture):
            m.texture = rec.texture
        return _check(self.L.orc_add_material(self.h, C.byref(m)))

    def add_dummy_materials(self, n):
        for _ in range(n):
            self.add_material(abi.RtColoredLambertianMaterial(abi.material_kind_lambertian_colored, abi.F3(.5, .5, .5)))

    def add_texture_rgb8(self, w, h, pixels):
        buf = np.ascontiguousarray(pixels, dtype=np.uint8)
        return _check(self.L.orc_add_texture_rgb8(self.h, C.c_int(w), C.c_int(h), buf.ctypes.data_as(C.c_void_p)))

    def add_kit_perlin(self, rec):
        return _check(self.L.orc_add_kit_perlin(self.h, C.byref(rec)))      # same layout as OrcKitPerlin

    def add_kit_texture(self, rec):
        return _check(self.L.orc_add_kit_texture(self.h, C.byref(rec)))     # same layout as OrcKitTexture

    def kit_texture_color(self, texture, u, v, point, precision=None):
        out = (C.c_double * 3)()
        _check(self.L.orc_kit_texture_color(self.h, C.c_int(texture), C.c_int(ORC_F64 if precision is None else precision),
                                            C.c_double(u), C.c_double(v), (C.c_double * 3)(*[float(x) for x in point]), out))
        return np.array(out[:])

    def shape_sphere(self, rec):
        o = OrcSphere(_orc_transform(rec.transform), rec.radius, rec.material_offset)
        return _check(self.L.orc_shape_sphere(self.h, C.byref(o)))

    def shape_cylinder(self, rec):
        o = OrcCylinder(_orc_transform(rec.transform), rec.radius, rec.height, rec.bottom_material_offset,
                        rec.top_material_offset, rec.side_material_offset)
        return _check(self.L.orc_shape_cylinder(self.h, C.byref(o)))

    def shape_cuboid(self, rec):
        o = OrcCuboid(_orc_transform(rec.transform), _p3(rec.size), rec.material_offset)
        return _check(self.L.orc_shape_cuboid(self.h, C.byref(o)))

    def shape_quad(self, rec):
        o = OrcQuad(_p3(rec.origin), _p3(rec.u), _p3(rec.v), _p3(rec.w), _p3(rec.normal), rec.d, rec.material_offset)
        return _check(self.L.orc_shape_quad(self.h, C.byref(o)))

    def shape_csg(self, op, ids):
        arr = (C.c_int * len(ids))(*ids)
        return _check(self.L.orc_shape_csg(self.h, C.c_int(op), C.c_int(len(ids)), arr))

    def shape_cdv(self, base, density):
        return _check(self.L.orc_shape_cdv(self.h, C.c_int(base), C.c_float(density)))

    def shape_motion(self, base, rec):
        q = (C.c_double * 4)(*[float(x) for x in rec.rotation])
        t = (C.c_double * 3)(*[float(x) for x in rec.translation])
        return _check(self.L.orc_shape_motion(self.h, C.c_int(base), q, t))

    def bounds(self, shape):
        out = (C.c_double * 6)()
        _check(self.L.orc_shape_bounds(self.h, C.c_int(shape), out))
        return np.array(out[:3]), np.array(out[3:])

    def set_convention(self, convention):
        self.L.orc_set_convention.restype = None
        self.L.orc_set_convention(self.h, C.c_int(1 if convention == abi.RT_CONVENTION_KIT else 0))

    def set_time(self, time):
        self.L.orc_set_query_time.restype = None
        self.L.orc_set_query_time(self.h, C.c_double(time))

    def add_object(self, shape):
        return _check(self.L.orc_add_object(self.h, C.c_int(shape)))

    def add_spheres_bulk(self, centers, radii, mat_ids):
        cr = np.ascontiguousarray(np.concatenate([centers, radii[:, None]], axis=1), dtype=np.float32)
        m = np.ascontiguousarray(mat_ids, dtype=np.uint32)
        return _check(self.L.orc_add_sphere_objects(self.h, C.c_int(len(m)), cr.ctypes.data_as(C.c_void_p),
                                                    m.ctypes.data_as(C.c_void_p)))

    def commit(self):
        _check(self.L.orc_commit(self.h))
        self.committed = True

    # ---- queries
    def enumerate(self, shape, origin, direction, precision=ORC_F32, scripted=None, max_out=32):
        o = (C.c_double * 3)(*[float(v) for v in origin])
        d = (C.c_double * 3)(*[float(v) for v in direction])
        ev = (OrcEvent * max_out)()
        if scripted:
            s = (C.c_float * len(scripted))(*scripted)
            n = self.L.orc_enumerate(self.h, C.c_int(shape), C.c_int(precision), o, d, s, C.c_int(len(scripted)), ev,
                                     C.c_int(max_out))
        else:
            n = self.L.orc_enumerate(self.h, C.c_int(shape), C.c_int(precision), o, d, None, C.c_int(0), ev,
                                     C.c_int(max_out))
        _check(n)
        return [ev[i] for i in range(n)]

    def first_hit(self, rays, tmin=1e-3, tmax=float("inf"), precision=ORC_F64, volume_key=0):
        if not self.committed:
            self.commit()
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        _check(self.L.orc_first_hit(self.h, C.c_int(precision), C.c_int(len(rays)), rays.ctypes.data_as(C.c_void_p),
                                    C.c_double(tmin), C.c_double(tmax), C.c_uint64(volume_key),
                                    out.ctypes.data_as(C.c_void_p)))
        return out

    def scatter(self, queries, rng_seed, precision=ORC_F64):
        """queries: array with the fields of raytracer_amd.renderer.SCATTER_QUERY_DTYPE."""
        n = len(queries)
        q = np.zeros(n, dtype=SCATTER_QUERY)
        for name in ("direction", "normal", "point", "uv", "front_face", "material_offset"):
            q[name] = queries[name]
        q["sample"], q["x"], q["y"] = queries["rng_sample"], queries["rng_x"], queries["rng_y"]
        out = np.zeros(n, dtype=SCATTER_RESULT)
        _check(self.L.orc_scatter(self.h, C.c_int(precision), C.c_int(n), q.ctypes.data_as(C.c_void_p), C.c_uint64(rng_seed),
                                  out.ctypes.data_as(C.c_void_p)))
        return out

    def render(self, camera, config, width, height, window=None, precision=ORC_F64, threads=0, counters=False):
        """Returns (sum image [h, w, 3] float64, threads used, counters dict or None)."""
        if not self.committed:
            self.commit()
        x0, y0, w, h = window if window else (0, 0, width, height)
        cc = _orc_camera(camera.impl)
        rc = OrcRenderConfig(config.samples_per_pixel, config.max_depth, config.first_sample, config.rng_seed)
        out = np.zeros((h, w, 3), dtype=np.float64)
        cnt = OrcCounters()
        used = _check(self.L.orc_render(self.h, C.byref(cc), C.byref(rc), C.c_int(precision), C.c_int(width),
                                        C.c_int(height), C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h),
                                        C.c_int(threads), out.ctypes.data_as(C.c_void_p),
                                        C.byref(cnt) if counters else None))
        return out, used, (cnt.as_dict() if counters else None)


def camera_rays(camera, config, width, height, window=None, precision=ORC_F32):
    x0, y0, w, h = window if window else (0, 0, width, height)
    cc = _orc_camera(camera.impl)
    rc = OrcRenderConfig(config.samples_per_pixel, config.max_depth, config.first_sample, config.rng_seed)
    out = np.zeros((w * h * config.samples_per_pixel, 6), dtype=np.float64)
    lib().orc_camera_rays(C.byref(cc), C.c_int(precision), C.c_int(width), C.c_int(height), C.byref(rc),
                          C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), out.ctypes.data_as(C.c_void_p))
    return out


def pcg32(initstate, initseq, n):
    out = (C.c_uint32 * n)()
    lib().orc_pcg32(C.c_uint64(initstate), C.c_uint64(initseq), C.c_int(n), out)
    return list(out)


def wyrand(seed, n):
    out = (C.c_uint64 * n)()
    lib().orc_wyrand(C.c_uint64(seed), C.c_int(n), out)
    return list(out)


def oracle_scene_from(scene, dedup=True):
    sink = OracleScene()
    scene.encode(sink, dedup=dedup)
    sink.commit()
    return sink


def progressive_present(pass_sum, samples_per_pixel, pass_counter, acc):
    """Shaders.metal:151-158 on the CPU.  pass_sum: [n, 3] float32 sums of one pass; acc:
    [n] uint32 rgb10a2 words (updated copy returned).  Returns (acc, bgra [n, 4] uint8)."""
    L = lib()
    ps = np.ascontiguousarray(pass_sum, dtype=np.float32).reshape(-1, 3)
    a = np.ascontiguousarray(acc, dtype=np.uint32).copy()
    out = np.zeros((len(ps), 4), dtype=np.uint8)
    L.orc_progressive_present.restype = None
    L.orc_progressive_present(ps.ctypes.data_as(C.c_void_p), C.c_uint32(samples_per_pixel), C.c_uint32(pass_counter),
                              a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(len(ps)))
    return a, out


def resolve_rgb8(sums, n_samples):
    """Camera.swift:92 + Vector3D.swift:185-211 in float on the CPU: [n, 3] float32 sums -> [n, 3] uint8."""
    L = lib()
    a = np.ascontiguousarray(sums, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((len(a), 4), dtype=np.uint8)
    L.orc_resolve_rgb8.restype = None
    L.orc_resolve_rgb8(a.ctypes.data_as(C.c_void_p), C.c_uint32(n_samples), C.c_int(len(a)), out.ctypes.data_as(C.c_void_p))
    return out[:, :3]


def sample_stream(rng_seed, sample, x, y):
    """(state, inc) of the PCG32 stream of one (pixel, sample)."""
    out = (C.c_uint64 * 2)()
    lib().orc_sample_stream(C.c_uint64(rng_seed), C.c_uint32(sample), C.c_uint32(x), C.c_uint32(y), out)
    return int(out[0]), int(out[1])


def sample_stream_words(rng_seed, sample, x, y, n):
    out = (C.c_uint32 * n)()
    lib().orc_sample_stream_words(C.c_uint64(rng_seed), C.c_uint32(sample), C.c_uint32(x), C.c_uint32(y), C.c_int(n), out)
    return list(out)
