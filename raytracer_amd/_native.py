"""ctypes binding of raytracer_amd/lib/libraytracer_amd.so (the C ABI of
include/raytracer_amd.h).  There is no CPU fallback: if the library is missing or does
not load, importing this module raises."""
import ctypes as C
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(HERE, "lib", "libraytracer_amd.so")   # RT_LIB: A/B builds

# every symbol include/raytracer_amd.h declares
EXPORTS = [
    "rt_context_create", "rt_context_destroy", "rt_last_error", "rt_abi_version",
    "rt_scene_create", "rt_scene_destroy", "rt_add_material", "rt_add_texture_rgb8", "rt_add_kit_perlin", "rt_add_kit_texture",
    "rt_shape_sphere", "rt_shape_cylinder", "rt_shape_cuboid", "rt_shape_quad", "rt_shape_csg", "rt_shape_cdv",
    "rt_shape_motion", "rt_shape_bounds", "rt_scene_set_time",
    "rt_add_object", "rt_add_sphere_objects", "rt_scene_commit", "rt_scene_stats", "rt_scene_set_option",
    "rt_first_hit", "rt_intervals", "rt_scatter", "rt_camera_rays", "rt_rng_words",
    "rt_render_workspace_size", "rt_render", "rt_render_static", "rt_band_rows", "rt_render_bands", "rt_resolve_rgb8", "rt_progressive_present",
    "rt_kernel_time",
]


class NativeError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C raytracer_amd/csrc).  There is no fallback path." % LIB_PATH)
    # torch carries its own copy of the HIP runtime; it has to be in the process first so that this
    # library binds to the same one (two runtimes in one process cannot both open the device)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    sig = {
        "rt_context_create": (i32, [i32, C.POINTER(vp)]),
        "rt_context_destroy": (None, [vp]),
        "rt_last_error": (C.c_char_p, []),
        "rt_abi_version": (i32, []),
        "rt_scene_create": (i32, [vp, C.POINTER(vp)]),
        "rt_scene_destroy": (None, [vp]),
        "rt_add_material": (i32, [vp, vp, C.c_size_t]),
        "rt_add_texture_rgb8": (i32, [vp, i32, i32, vp]),
        "rt_add_kit_perlin": (i32, [vp, C.POINTER(abi.RtKitPerlin)]),
        "rt_add_kit_texture": (i32, [vp, C.POINTER(abi.RtKitTexture)]),
        "rt_shape_sphere": (i32, [vp, C.POINTER(abi.RtSphere)]),
        "rt_shape_cylinder": (i32, [vp, C.POINTER(abi.RtCylinder)]),
        "rt_shape_cuboid": (i32, [vp, C.POINTER(abi.RtCuboid)]),
        "rt_shape_quad": (i32, [vp, C.POINTER(abi.RtQuad)]),
        "rt_shape_csg": (i32, [vp, i32, i32, C.POINTER(i32)]),
        "rt_shape_cdv": (i32, [vp, i32, f32]),
        "rt_shape_motion": (i32, [vp, i32, C.POINTER(abi.RtMotion)]),
        "rt_shape_bounds": (i32, [vp, i32, C.POINTER(f32)]),
        "rt_scene_set_time": (i32, [vp, f32]),
        "rt_add_object": (i32, [vp, i32]),
        "rt_add_sphere_objects": (i32, [vp, i32, vp, vp]),
        "rt_scene_commit": (i32, [vp]),
        "rt_scene_stats": (i32, [vp, C.POINTER(abi.RtSceneStats)]),
        "rt_scene_set_option": (i32, [vp, i32, i32]),
        "rt_first_hit": (i32, [vp, vp, i32, f32, f32, u64, vp, vp]),
        "rt_intervals": (i32, [vp, i32, vp, i32, vp, i32, u64, vp, vp, vp]),
        "rt_camera_rays": (i32, [vp, C.POINTER(abi.RtCameraConfig), C.POINTER(abi.RtRenderConfig), u32, i32, i32, i32, i32,
                                 i32, i32, vp, vp]),
        "rt_rng_words": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "rt_scatter": (i32, [vp, vp, i32, u64, vp, vp]),
        "rt_render_workspace_size": (C.c_size_t, [i32, i32, i32]),
        "rt_render": (i32, [vp, C.POINTER(abi.RtCameraConfig), C.POINTER(abi.RtRenderConfig), u32, i32, i32, i32, vp, vp,
                            C.c_size_t, vp]),
        "rt_render_static": (i32, [vp, C.POINTER(abi.RtCameraConfig), C.POINTER(abi.RtRenderConfig), u32, i32, i32, vp, vp]),
        "rt_band_rows": (i32, [i32, i32, i32]),
        "rt_render_bands": (i32, [vp, C.POINTER(abi.RtCameraConfig), C.POINTER(abi.RtRenderConfig), u32, i32, i32, i32, i32,
                                  vp, vp, C.c_size_t, vp]),
        "rt_resolve_rgb8": (i32, [vp, vp, u32, i32, i32, vp, vp]),
        "rt_progressive_present": (i32, [vp, vp, u32, u32, vp, vp, i32, i32, vp]),
        "rt_kernel_time": (i32, [vp, i32, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(C.c_char_p)]),
    }
    for name in EXPORTS:
        fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = sig[name]
    if L.rt_abi_version() != 3:
        raise ImportError("libraytracer_amd.so has ABI version %d, expected 3" % L.rt_abi_version())
    return L


lib = _load()


def check(rc):
    if rc < 0:
        raise NativeError("raytracer_amd error %d: %s" % (rc, lib.rt_last_error().decode()))
    return rc
