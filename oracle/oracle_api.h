/*
 * oracle_api.h - C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the
 * reference path tracer's algorithms (nickolas-pohilets/RayTracer).  It is the
 * checker for the HIP path: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Nothing under raytracer_amd/
 * links, imports or calls it.
 *
 * Parity status: the hit enumerators (sphere, cylinder, cuboid, quad,
 * composition, constant-density volume) and PCG32 are pinned by the
 * reference's own known-answer tests (the .swift files of MetalRayTracerTests/) held as
 * tests/golden/metal_tests.json.  Camera, materials, textures, BVH and the
 * integrator have no reference test: for those the parity is UNPINNED and
 * rests on this restatement alone.
 *
 * The input records mirror the reference's device ABI
 * (MetalRayTracer/Renderable.h:13-63, Materials.h:16-93, Config.h:13-33) with
 * simd vector types spelled as plain float arrays.
 */
#ifndef ORACLE_API_H
#define ORACLE_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Renderable.h:13-16.  rotation is column-major: rotation[3*c + r] = columns[c][r]. */
typedef struct { float rotation[9]; float translation[3]; } OrcTransform;
/* Renderable.h:18-25 */
typedef struct { OrcTransform transform; float radius; size_t material_offset; } OrcSphere;
/* Renderable.h:27-37 */
typedef struct {
    OrcTransform transform; float radius; float height;
    size_t bottom_material_offset, top_material_offset, side_material_offset;
} OrcCylinder;
/* Renderable.h:39-52; Face order left,right,bottom,top,back,front */
typedef struct { OrcTransform transform; float size[3]; size_t material_offset[6]; } OrcCuboid;
/* Renderable.h:54-63 */
typedef struct {
    float origin[3], u[3], v[3], w[3], normal[3]; float d; size_t material_offset;
} OrcQuad;

/* Materials.h:16-26 */
enum {
    orc_material_kind_lambertian_colored = 1,
    orc_material_kind_lambertian_textured,
    orc_material_kind_lambertian_perlin_noise,
    orc_material_kind_metal_colored,
    orc_material_kind_metal_textured,
    orc_material_kind_metal_perlin_noise,
    orc_material_kind_dielectric,
    orc_material_kind_emissive_colored,
    orc_material_kind_isotropic_colored,
    orc_material_kind_lambertian_kit_texture,   /* RayTracingKit/Material.swift:17-41 over any Texture */
    orc_material_kind_emissive_kit_texture,     /* Material.swift:75-93 */
};

/* One flat record for every material kind (the reference packs nine structs,
 * Materials.h:47-93; unused fields are ignored per kind). */
typedef struct {
    uint32_t kind;
    float albedo[3];
    float fuzz;              /* metal kinds */
    float refraction_index;  /* dielectric */
    uint32_t texture;        /* image id (textured kinds) or perlin id (perlin kinds) */
} OrcMaterial;

/* Materials.h:38-45 */
typedef struct {
    float colors[2][3];
    float frequency;
    uint32_t turbulence;
    float vectors[256][3];
    uint8_t permutations[3][256];
} OrcPerlin;

/* Config.h:13-26 */
typedef struct {
    float vertical_FOV;
    float look_from[3], look_at[3], up[3];
    float defocus_angle;
    float focus_distance;
    uint32_t background;   /* 0 none, 1 sky */
} OrcCameraConfig;

/* Config.h:28-33 (pass_counter has no meaning for the float accumulators) */
typedef struct {
    uint32_t samples_per_pixel;
    uint32_t max_depth;
    uint32_t first_sample;   /* global index of the first sample: shards use disjoint ranges */
    uint64_t rng_seed;
} OrcRenderConfig;

enum { ORC_OP_UNION = 0, ORC_OP_INTERSECTION = 1, ORC_OP_SUBTRACT = 2 };
enum { ORC_F32 = 0, ORC_F64 = 1 };

/* One enumerated hit (RenderableImpl.h enumerator protocol). */
typedef struct {
    double t;
    double point[3];
    double normal[3];      /* as enumerated: NOT flipped against the ray */
    double uv[2];
    uint64_t material_offset;
    int32_t is_exit;
    int32_t pad;
} OrcEvent;

/* Closest accepted hit of a ray against the whole scene
 * (Shaders.metal:171-203 wrapper + HitTesting.h:24-32 flip). */
typedef struct {
    double t;
    double point[3];
    double normal[3];      /* flipped to oppose the ray */
    double uv[2];
    uint64_t material_offset;
    int32_t object;        /* index into the scene's object list, -1 = miss */
    int32_t front_face;    /* 1 front, 0 back */
} OrcHit;

typedef struct {
    uint64_t samples, rays, hits;
    uint64_t box_tests;
    uint64_t prim_tests[4];   /* sphere, cylinder, cuboid, quad */
    uint64_t csg_events;
    uint64_t cdv_intervals;
    uint64_t shades[10];      /* by material kind */
} OrcCounters;

typedef struct OrcScene OrcScene;

OrcScene *orc_scene_create(void);
void orc_scene_destroy(OrcScene *);

int orc_add_material(OrcScene *, const OrcMaterial *);                 /* -> material_offset handle */
int orc_add_texture_rgb8(OrcScene *, int w, int h, const uint8_t *rgb); /* rows top to bottom */
/* RayTracingKit textures (Texture.swift); same record layouts as include/raytracer_amd.h. */
typedef struct OrcKitPerlin { float vectors[256][3]; uint8_t perm_x[256], perm_y[256], perm_z[256]; } OrcKitPerlin;
typedef struct OrcKitTexture {
    uint32_t kind;                  /* 0 solid, 1 checker, 2 image, 3 noise, 4 marble */
    float color[3];
    float scale;
    int32_t even, odd;
    int32_t image;
    int32_t origin_x, origin_y, width, height;
    uint32_t wrap_x, wrap_y;
    int32_t perlin;
} OrcKitTexture;
int orc_add_kit_perlin(OrcScene *, const OrcKitPerlin *);
int orc_add_kit_texture(OrcScene *, const OrcKitTexture *);
/* One texture lookup: Texture.subscript(coordinates, point:). */
int orc_kit_texture_color(const OrcScene *, int texture, int precision, double u, double v, const double point[3], double out_rgb[3]);
int orc_add_perlin(OrcScene *, const OrcPerlin *);

int orc_shape_sphere(OrcScene *, const OrcSphere *);     /* -> shape id */
/* n plain spheres added as objects in one call (center_radius: n x 4 floats); -> first object index. */
int orc_add_sphere_objects(OrcScene *, int n, const float *center_radius, const uint32_t *material_offsets);
int orc_shape_cylinder(OrcScene *, const OrcCylinder *);
int orc_shape_cuboid(OrcScene *, const OrcCuboid *);
int orc_shape_quad(OrcScene *, const OrcQuad *);
int orc_shape_csg(OrcScene *, int op, int n, const int *children);
int orc_shape_cdv(OrcScene *, int base, float density);
/* MotionBlur (RayTracingKit/Hittable.swift:528-555): `base` seen through transform.pow(time),
 * transform = unit quaternion (w, x, y, z; Transform.swift:169-259) + translation. */
int orc_shape_motion(OrcScene *, int base, const double quat_wxyz[4], const double translation[3]);
/* Time used by orc_enumerate / orc_first_hit (orc_render draws one per sample). */
void orc_set_query_time(OrcScene *, double time);
/* 0 (default): texture coordinates and metal rule of MetalRayTracer; 1: RayTracingKit's
 * (Hittable.swift:188,245-248,281-282; Material.swift:53-57). */
void orc_set_convention(OrcScene *, int kit);
/* Bounding box of a shape as the BVH uses it (Hittable.boundingBox): min xyz, max xyz. */
int orc_shape_bounds(OrcScene *, int shape, double out_min_max[6]);
/* Transform3D.boundingBoxForAnimation of a point and Transform3D.pow(t).transform(p)
 * (Transform.swift:82-140), as RayTracingKitTests/TransformTests.swift exercises them. */
void orc_animation_box_of_point(const double quat_wxyz[4], const double translation[3], const double p[3], double out_min_max[6]);
void orc_transform_pow(const double quat_wxyz[4], const double translation[3], double t, const double p[3], double out[3]);
int orc_add_object(OrcScene *, int shape);               /* -> object index */
int orc_commit(OrcScene *);                               /* builds the BVH */

/* Enumerate the ordered hits of one shape along one ray.  scripted_rng (may be
 * NULL) is the reference's test RNG: values cycled in order (RNG.h:15-27). */
int orc_enumerate(const OrcScene *, int shape, int precision,
                  const double origin[3], const double direction[3],
                  const float *scripted_rng, int n_scripted,
                  OrcEvent *out, int max_out);

/* Closest hit for n rays (origin xyz, direction xyz interleaved, 6 doubles per
 * ray).  cdv_state: PCG32 state word used to key volume free-flight draws. */
int orc_first_hit(const OrcScene *, int precision, int n, const double *rays,
                  double tmin, double tmax, uint64_t cdv_state, OrcHit *out);

/* One scattering event of a material at a given hit (MaterialsImpl.h:147-215 / Material.swift),
 * drawing from the generator stream of (rng_seed, sample, x, y). */
typedef struct {
    double direction[3], normal[3], point[3], uv[2];
    int32_t front_face;
    uint32_t material_offset;
    uint32_t sample, x, y;
    uint32_t pad;
} OrcScatterQuery;
typedef struct {
    int32_t scattered, pad;
    double direction[3], attenuation[3], emitted[3];
} OrcScatterResult;
int orc_scatter(const OrcScene *, int precision, int n, const OrcScatterQuery *queries, uint64_t rng_seed, OrcScatterResult *out);

/* Camera rays as the renderer generates them: out = n*6 doubles. */
int orc_camera_rays(const OrcCameraConfig *, int precision, int width, int height,
                    const OrcRenderConfig *, int x0, int y0, int w, int h, double *out);

/* Render the window [x0,x0+w) x [y0,y0+h) of a width x height image:
 * out_sum = per-pixel RGB radiance SUMS over samples_per_pixel samples
 * (w*h*3 doubles, row-major).  threads <= 0 means all hardware threads. */
int orc_render(const OrcScene *, const OrcCameraConfig *, const OrcRenderConfig *,
               int precision, int width, int height, int x0, int y0, int w, int h,
               int threads, double *out_sum, OrcCounters *counters);

/* PCG32 (RNG.h:31-52) and WyRand (PRNG.swift:10-22) raw streams. */
void orc_pcg32(uint64_t initstate, uint64_t initseq, int n, uint32_t *out);
void orc_wyrand(uint64_t seed, int n, uint64_t *out);
/* First n PCG32 words of the stream of one (pixel, sample). */
void orc_sample_stream_words(uint64_t rng_seed, uint32_t sample, uint32_t x, uint32_t y, int n, uint32_t *out);
/* Seeding of the per-(pixel, sample) stream; returns state and inc. */
void orc_sample_stream(uint64_t rng_seed, uint32_t sample, uint32_t x, uint32_t y, uint64_t out_state_inc[2]);

/* Progressive display step of the app's kernel (Shaders.metal:151-158): the pass mean is
 * clamped to 1, blended into the 10-bit running mean held in `acc` (rgb10a2Uint words: r in
 * bits 0-9, g 10-19, b 20-29) with weight 1/pass_counter, and sqrt(mean) is written as the
 * drawable's bgra8Unorm pixel (round to nearest even).  pass_sum: n_pixels*3 radiance sums. */
void orc_progressive_present(const float *pass_sum, uint32_t samples_per_pixel, uint32_t pass_counter,
                             uint32_t *acc, uint8_t *bgra, int n_pixels);

/* Output stage of the CLI renderer (Camera.swift:92, Vector3D.swift:185-191,208-211) in float:
 * mean, square root, clamp, floor(nextDown(256) * x) -> RGBX8.  sums: n_pixels*3 radiance sums. */
void orc_resolve_rgb8(const float *sums, uint32_t n_samples, int n_pixels, uint8_t *rgbx);

const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
