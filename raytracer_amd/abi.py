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
RT_CSG_MAX_LEAVES = 8
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

F3 = C.c_float * 3


class RtTransform(C.Structure):
    _fields_ = [("rotation", C.c_float * 9), ("translation", F3)]


class RtSphere(C.Structure):
    _fields_ = [("transform", RtTransform), ("radius", C.c_float), ("material_offset", C.c_size_t)]


class RtCylinder(C.Structure):
    _fields_ = [("transform", RtTransform), ("radius", C.c_float), ("height", C.c_float),
                ("bottom_material_offset", C.c_size_t), ("top_material_offset", C.c_size_t),
                ("side_material_offset", C.c_size_t)]


class RtCuboid(C.Structure):
    _fields_ = [("transform", RtTransform), ("size", F3), ("material_offset", C.c_size_t * 6)]


class RtQuad(C.Structure):
    _fields_ = [("origin", F3), ("u", F3), ("v", F3), ("w", F3), ("normal", F3),
                ("d", C.c_float), ("material_offset", C.c_size_t)]


class RtImageTexture(C.Structure):
    _fields_ = [("texture_id", C.c_uint32)]


class RtPerlinNoiseTexture(C.Structure):
    _fields_ = [("colors", F3 * 2), ("frequency", C.c_float), ("turbulence", C.c_uint32),
                ("vectors", F3 * RT_PERLIN_TABLE_SIZE),
                ("permutations", (C.c_uint8 * RT_PERLIN_TABLE_SIZE) * 3)]


class RtColoredLambertianMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", F3)]


class RtTexturedLambertianMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", RtImageTexture)]


class RtPerlinNoiseLambertianMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", RtPerlinNoiseTexture)]


class RtColoredMetalMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", F3), ("fuzz", C.c_float)]


class RtTexturedMetalMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", RtImageTexture), ("fuzz", C.c_float)]


class RtPerlinNoiseMetalMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", RtPerlinNoiseTexture), ("fuzz", C.c_float)]


class RtDielectricMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("refraction_index", C.c_float)]


class RtColoredEmissiveMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", F3)]


class RtColoredIsotropicMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", F3)]


class RtCameraConfig(C.Structure):
    _fields_ = [("vertical_FOV", C.c_float), ("look_from", F3), ("look_at", F3), ("up", F3),
                ("defocus_angle", C.c_float), ("focus_distance", C.c_float), ("background", C.c_uint32)]


class RtRenderConfig(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("pass_counter", C.c_uint32), ("first_sample", C.c_uint32), ("rng_seed", C.c_uint64)]


class RtHit(C.Structure):
    _fields_ = [("t", C.c_float), ("point", F3), ("normal", F3), ("uv", C.c_float * 2),
                ("object", C.c_int32), ("material_offset", C.c_uint32), ("front_face", C.c_int32)]


class RtEvent(C.Structure):
    _fields_ = [("t", C.c_float), ("point", F3), ("normal", F3), ("uv", C.c_float * 2),
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
                ("device_bytes", C.c_uint64)]


assert C.sizeof(RtHit) == 48 and C.sizeof(RtEvent) == 48
