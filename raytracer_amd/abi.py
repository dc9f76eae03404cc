"""ctypes mirror of include/raytracer_amd.h (records only; no compute).

Field names follow the reference's device ABI: MetalRayTracer/Renderable.h:13-63,
Materials.h:16-93, Config.h:13-33.
"""
import ctypes as C

RT_CSG_UNION, RT_CSG_INTERSECTION, RT_CSG_SUBTRACT = 0, 1, 2
RT_PIPELINE_WAVEFRONT, RT_PIPELINE_FUSED = 0, 1
RT_OPTION_BVH_BUILDER, RT_OPTION_REFILL_THRESHOLD, RT_OPTION_NODE_MIN, RT_OPTION_NODE_FORMAT = 1, 2, 3, 4
RT_NODES_AUTO, RT_NODES_F32, RT_NODES_F16 = 0, 1, 2
RT_BVH_MEDIAN_SPLIT, RT_BVH_SAH = 0, 1
RT_CSG_MAX_LEAVES = 64
RT_MAX_EVENTS = 16
RT_PERLIN_TABLE_SIZE = 256

# Materials.h:16-26
material_kind_lambertian_colored = 1
material_kind_lambertian_textured = 2
material_kind_lambertian_perlin_noise = 3
material_kind_metal_colored = 4
material_kind_metal_textured = 5
material_kind_metal_perlin_noise = 6
material_kind_dielectric = 7
material_kind_emissive_colored = 8
material_kind_isotropic_colored = 9

# Config.h:13-16
background_lighting_none = 0
background_lighting_sky = 1

RT_OPTION_KERNEL_TIMING, RT_OPTION_PROFILE_PHASES, RT_OPTION_CONVENTION, RT_OPTION_HEAVY_MIN = 5, 6, 7, 8
RT_OPTION_TRAVERSAL = 9
RT_TRAVERSAL_AUTO, RT_TRAVERSAL_TREE, RT_TRAVERSAL_LIST = 0, 1, 2
RT_CONVENTION_METAL, RT_CONVENTION_KIT = 0, 1


class _Record(C.Structure):
    """Structure whose positional constructor skips the padding fields (named _pad*), so a
    record is built from its meaningful fields in header order."""

    def __init__(self, *args, **kw):
        super().__init__()
        names = [f[0] for f in self._fields_ if not f[0].startswith("_pad")]
        if len(args) > len(names):
            raise TypeError("too many initialisers for %s" % type(self).__name__)
        for name, value in zip(names, args):
            setattr(self, name, value)
        for name, value in kw.items():
            setattr(self, name, value)


class RtFloat3(C.Structure):
    """vector_float3: three floats in 16 bytes."""
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("unused", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z, 0.0)

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]


F3 = RtFloat3


class RtMatrix3x3(C.Structure):
    _fields_ = [("columns", RtFloat3 * 3)]


class RtTransform(_Record):
    _fields_ = [("rotation", RtMatrix3x3), ("translation", F3)]


class RtSphere(_Record):
    _fields_ = [("transform", RtTransform), ("radius", C.c_float), ("_pad0", C.c_uint32), ("material_offset", C.c_size_t)]


class RtCylinder(_Record):
    _fields_ = [("transform", RtTransform), ("radius", C.c_float), ("height", C.c_float),
                ("bottom_material_offset", C.c_size_t), ("top_material_offset", C.c_size_t),
                ("side_material_offset", C.c_size_t)]


class RtCuboid(_Record):
    _fields_ = [("transform", RtTransform), ("size", F3), ("material_offset", C.c_size_t * 6)]


class RtQuad(_Record):
    _fields_ = [("origin", F3), ("u", F3), ("v", F3), ("w", F3), ("normal", F3),
                ("d", C.c_float), ("_pad0", C.c_uint32), ("material_offset", C.c_size_t)]


class RtImageTexture(_Record):
    _fields_ = [("texture_ptr", C.c_uint64)]


class RtPerlinNoiseTexture(_Record):
    _fields_ = [("colors", F3 * 2), ("frequency", C.c_float), ("turbulence", C.c_uint32), ("_pad0", C.c_uint32 * 2),
                ("vectors", F3 * RT_PERLIN_TABLE_SIZE),
                ("permutations", (C.c_uint8 * RT_PERLIN_TABLE_SIZE) * 3)]


class RtColoredLambertianMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("albedo", F3)]


class RtTexturedLambertianMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32), ("albedo", RtImageTexture)]


class RtPerlinNoiseLambertianMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("albedo", RtPerlinNoiseTexture)]


class RtColoredMetalMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("albedo", F3), ("fuzz", C.c_float), ("_pad1", C.c_uint32 * 3)]


class RtTexturedMetalMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32), ("albedo", RtImageTexture), ("fuzz", C.c_float), ("_pad1", C.c_uint32)]


class RtPerlinNoiseMetalMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("albedo", RtPerlinNoiseTexture), ("fuzz", C.c_float),
                ("_pad1", C.c_uint32 * 3)]


class RtDielectricMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("refraction_index", C.c_float)]


class RtColoredEmissiveMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("albedo", F3)]


class RtColoredIsotropicMaterial(_Record):
    _fields_ = [("kind", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("albedo", F3)]


class RtCameraConfig(_Record):
    _fields_ = [("vertical_FOV", C.c_float), ("_pad0", C.c_uint32 * 3), ("look_from", F3), ("look_at", F3), ("up", F3),
                ("defocus_angle", C.c_float), ("focus_distance", C.c_float), ("background", C.c_uint32), ("_pad1", C.c_uint32)]


class RtRenderConfig(_Record):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("pass_counter", C.c_uint32), ("_pad0", C.c_uint32), ("rng_seed", C.c_uint64)]


# sizes and offsets of the reference's records (MetalRayTracer/Renderable.h, Config.h, Materials.h with simd types)
assert [C.sizeof(t) for t in (RtTransform, RtSphere, RtCylinder, RtCuboid, RtQuad, RtCameraConfig, RtRenderConfig)] == \
    [64, 80, 96, 128, 96, 80, 24]
assert [C.sizeof(t) for t in (RtColoredLambertianMaterial, RtTexturedLambertianMaterial, RtPerlinNoiseLambertianMaterial,
                              RtColoredMetalMaterial, RtTexturedMetalMaterial, RtPerlinNoiseMetalMaterial,
                              RtDielectricMaterial, RtPerlinNoiseTexture)] == [32, 16, 4928, 48, 24, 4944, 8, 4912]
assert RtSphere.material_offset.offset == 72 and RtQuad.material_offset.offset == 88 and RtCuboid.material_offset.offset == 80
assert RtCameraConfig.look_from.offset == 16 and RtCameraConfig.background.offset == 72 and RtRenderConfig.rng_seed.offset == 16
assert RtPerlinNoiseTexture.vectors.offset == 48 and RtPerlinNoiseTexture.permutations.offset == 4144


P3 = C.c_float * 3      # packed float triple of the result records


class RtHit(C.Structure):
    _fields_ = [("t", C.c_float), ("point", P3), ("normal", P3), ("uv", C.c_float * 2),
                ("object", C.c_int32), ("material_offset", C.c_uint32), ("front_face", C.c_int32)]


class RtEvent(C.Structure):
    _fields_ = [("t", C.c_float), ("point", P3), ("normal", P3), ("uv", C.c_float * 2),
                ("material_offset", C.c_uint32), ("is_exit", C.c_int32), ("pad", C.c_int32)]


class RtKitPerlin(C.Structure):
    _fields_ = [("vectors", C.c_float * 3 * 256), ("perm_x", C.c_uint8 * 256), ("perm_y", C.c_uint8 * 256),
                ("perm_z", C.c_uint8 * 256)]


class RtKitTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("scale", C.c_float), ("even", C.c_int32), ("odd", C.c_int32),
                ("image", C.c_int32), ("origin_x", C.c_int32), ("origin_y", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("wrap_x", C.c_uint32), ("wrap_y", C.c_uint32), ("perlin", C.c_int32)]


class RtKitTexturedMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32)]


RT_KIT_TEXTURE_SOLID, RT_KIT_TEXTURE_CHECKER, RT_KIT_TEXTURE_IMAGE, RT_KIT_TEXTURE_NOISE, RT_KIT_TEXTURE_MARBLE = range(5)
material_kind_lambertian_kit_texture, material_kind_emissive_kit_texture = 10, 11


class RtMotion(C.Structure):
    _fields_ = [("rotation", C.c_float * 4), ("translation", C.c_float * 3)]


class RtSceneStats(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("n_nodes", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("n_spheres", C.c_uint32), ("n_cylinders", C.c_uint32), ("n_cuboids", C.c_uint32),
                ("n_quads", C.c_uint32), ("n_csg", C.c_uint32), ("n_cdv", C.c_uint32),
                ("n_materials", C.c_uint32), ("n_textures", C.c_uint32), ("n_perlin", C.c_uint32),
                ("device_bytes", C.c_uint64), ("n_wide_nodes", C.c_uint32), ("reserved", C.c_uint32)]


assert C.sizeof(RtHit) == 48 and C.sizeof(RtEvent) == 48
