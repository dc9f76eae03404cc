/*
 * raytracer_amd.h - C ABI of the MI355X path-tracing hot path.
 *
 * The reference (nickolas-pohilets/RayTracer) has no FFI of its own: its seam is
 * "Swift scene API -> C structs shared with the kernel -> kernel argument table"
 * (MetalRayTracer/Renderable.h, Materials.h, Config.h; SceneBuffers.swift;
 * ContentView.swift Renderer).  This header is that seam as a C ABI: the records
 * keep the reference's field names, meaning, enum values AND byte layout (simd
 * vectors are 16-byte RtFloat3, the Metal resource id an 8-byte word holding a
 * texture id), so a host that already builds the reference's records passes them
 * as they are.
 *
 * Conventions
 *  - every function returns >= 0 on success (an id where documented) and a
 *    negative RT_ERR_* code on failure; rt_last_error() then describes it.
 *    Nothing throws or aborts across the boundary (the reference crashes on
 *    `!`/precondition, e.g. SceneBuffers.swift:21,50).
 *  - records passed to rt_add_* / rt_shape_* are copied; the caller may free them
 *    at once (the reference copies into device buffers too,
 *    SceneBuffers.swift:92-101).
 *  - pointers named *_dev are DEVICE pointers owned by the caller; `stream` is a
 *    hipStream_t passed as void* (NULL = default stream).  Calls that take a
 *    stream are asynchronous with respect to the host.
 *  - one host thread per scene; calls on one scene are serialised by the caller.
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3

enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument / id / record */
    RT_ERR_STATE = -2,        /* scene not committed, or already committed */
    RT_ERR_LIMIT = -3,        /* a documented limit was exceeded */
    RT_ERR_DEVICE = -4,       /* HIP runtime error */
};

/* ---- simd types ---------------------------------------------------------------
 * The reference's records are built from Apple simd types: vector_float3 is 16 bytes
 * with 16-byte alignment (three floats + one unused), matrix_float3x3 is three such
 * columns.  The records below have exactly those sizes and field offsets (checked by
 * static_asserts in rt_host.cpp), so the bytes of a reference `Sphere`, `Cuboid`,
 * `CameraConfig`, ... can be handed over as they are. */
#define RT_ALIGN16 __attribute__((aligned(16)))
typedef struct RT_ALIGN16 { float x, y, z, unused; } RtFloat3;        /* vector_float3 / simd_float3 */
typedef struct { RtFloat3 columns[3]; } RtMatrix3x3;                   /* matrix_float3x3: column-major */

/* ---- geometry records: MetalRayTracer/Renderable.h:13-63 ------------------- */

/* Renderable.h:13-16 (64 bytes). */
typedef struct { RtMatrix3x3 rotation; RtFloat3 translation; } RtTransform;

/* Renderable.h:18-25 (80 bytes: radius at 64, material_offset at 72). */
typedef struct { RtTransform transform; float radius; size_t material_offset; } RtSphere;

/* Renderable.h:27-37 (96 bytes): local frame has the axis on +y and the base at the origin. */
typedef struct {
    RtTransform transform;
    float radius, height;
    size_t bottom_material_offset, top_material_offset, side_material_offset;
} RtCylinder;

/* Renderable.h:39-52 (128 bytes): box [0,size] in the local frame. */
enum RtFace { rt_face_left, rt_face_right, rt_face_bottom, rt_face_top, rt_face_back, rt_face_front };
typedef struct { RtTransform transform; RtFloat3 size; size_t material_offset[6]; } RtCuboid;

/* Renderable.h:54-63 (96 bytes): w = n/|n|^2, normal = n/|n|, d = dot(origin, normal),
 * n = cross(u, v) (Primitives.swift:266-274). */
typedef struct { RtFloat3 origin, u, v, w, normal; float d; size_t material_offset; } RtQuad;

/* ---- material records: MetalRayTracer/Materials.h:16-93 -------------------- */

enum RtMaterialKind {
    rt_material_kind_lambertian_colored = 1,
    rt_material_kind_lambertian_textured,
    rt_material_kind_lambertian_perlin_noise,
    rt_material_kind_metal_colored,
    rt_material_kind_metal_textured,
    rt_material_kind_metal_perlin_noise,
    rt_material_kind_dielectric,
    rt_material_kind_emissive_colored,
    rt_material_kind_isotropic_colored,
    /* RayTracingKit materials over an arbitrary Texture (Material.swift:17-41,75-93): no Metal counterpart */
    rt_material_kind_lambertian_kit_texture,
    rt_material_kind_emissive_kit_texture,
};

typedef RtFloat3 RtSolidColor;                                         /* Materials.h:28 */

/* Materials.h:30-36: the 8-byte MTLResourceID holds the id rt_add_texture_rgb8 returned. */
typedef struct { uint64_t texture_ptr; } RtImageTexture;

/* Materials.h:38-45 (4912 bytes: vectors at 48, permutations at 4144). */
#define RT_PERLIN_TABLE_SIZE 256
typedef struct {
    RtSolidColor colors[2];
    float frequency;
    uint32_t turbulence;
    RtFloat3 vectors[RT_PERLIN_TABLE_SIZE];
    uint8_t permutations[3][RT_PERLIN_TABLE_SIZE];
} RtPerlinNoiseTexture;

/* Materials.h:47-93.  The first 4 bytes of every record are its kind; sizes 32, 16, 4928,
 * 48, 24, 4944, 8, 32, 32 bytes. */
typedef struct { uint32_t kind; RtSolidColor albedo; } RtColoredLambertianMaterial;
typedef struct { uint32_t kind; RtImageTexture albedo; } RtTexturedLambertianMaterial;
typedef struct { uint32_t kind; RtPerlinNoiseTexture albedo; } RtPerlinNoiseLambertianMaterial;
typedef struct { uint32_t kind; RtSolidColor albedo; float fuzz; } RtColoredMetalMaterial;
typedef struct { uint32_t kind; RtImageTexture albedo; float fuzz; } RtTexturedMetalMaterial;
typedef struct { uint32_t kind; RtPerlinNoiseTexture albedo; float fuzz; } RtPerlinNoiseMetalMaterial;
typedef struct { uint32_t kind; float refraction_index; } RtDielectricMaterial;
typedef struct { uint32_t kind; RtSolidColor albedo; } RtColoredEmissiveMaterial;
typedef struct { uint32_t kind; RtSolidColor albedo; } RtColoredIsotropicMaterial;

/* ---- configuration: MetalRayTracer/Config.h:13-33 -------------------------- */

enum RtBackgroundLighting { rt_background_lighting_none, rt_background_lighting_sky };

/* Config.h:18-26 (80 bytes: look_from at 16, look_at 32, up 48, defocus_angle 64,
 * focus_distance 68, background 72). */
typedef struct {
    float vertical_FOV;
    RtFloat3 look_from, look_at, up;
    float defocus_angle;
    float focus_distance;
    uint32_t background;
} RtCameraConfig;

/* Config.h:28-33 (24 bytes: rng_seed at 16).  samples_per_pixel more samples are ADDED to
 * the accumulator by one rt_render call; pass_counter keeps the reference's meaning (number
 * of the pass, 1 based).  Which samples a call renders is the `first_sample` argument of
 * rt_render, not a field: this record is the reference's, byte for byte. */
typedef struct {
    uint32_t samples_per_pixel;
    uint32_t max_depth;
    uint32_t pass_counter;
    uint64_t rng_seed;
} RtRenderConfig;

/* ---- results ----------------------------------------------------------------- */

/* Closest accepted hit (Shaders.metal:171-203 + HitTesting.h:17-33). */
typedef struct {
    float t;
    float point[3];
    float normal[3];          /* flipped to oppose the ray */
    float uv[2];
    int32_t object;           /* index returned by rt_add_object, -1 = miss */
    uint32_t material_offset; /* handle returned by rt_add_material */
    int32_t front_face;       /* 1 = front, 0 = back */
} RtHit;

/* One boundary crossing of a shape along a ray (the hit enumerators of
 * Impl/RenderableImpl.h: hasNext/isExit/t/point/normal/material_offset/
 * texture_coordinates). */
typedef struct {
    float t;
    float point[3];
    float normal[3];          /* as enumerated: not flipped against the ray */
    float uv[2];
    uint32_t material_offset;
    int32_t is_exit;
    int32_t pad;
} RtEvent;

typedef struct {
    uint32_t n_objects, n_nodes, bvh_depth;
    uint32_t n_spheres, n_cylinders, n_cuboids, n_quads, n_csg, n_cdv;
    uint32_t n_materials, n_textures, n_perlin;
    uint64_t device_bytes;
    uint32_t n_wide_nodes;    /* nodes of the four-children form a deep tree of spheres is rendered from (0: the scene has none) */
    uint32_t reserved;
} RtSceneStats;

enum RtCsgOp { RT_CSG_UNION = 0, RT_CSG_INTERSECTION = 1, RT_CSG_SUBTRACT = 2 };

/* Which form of the render loop runs (both give the same estimator, sample for
 * sample; they differ in where path state lives). */
enum RtPipeline {
    RT_PIPELINE_WAVEFRONT = 0,   /* ray-gen / extend / shade kernels over path queues in HBM */
    RT_PIPELINE_FUSED = 1,       /* one persistent kernel, the stages as wave-wide phases, paths regenerated in place */
};

#define RT_CSG_MAX_LEAVES 64     /* leaf primitives under one CSG object (up to 8 are walked from registers, more by recomputation) */
#define RT_MAX_EVENTS 16         /* boundary crossings rt_intervals can report per ray */

typedef struct RtContext RtContext;
typedef struct RtScene RtScene;

/* ---- context ----------------------------------------------------------------- */

/* Replaces MTLCreateSystemDefaultDevice + makeCommandQueue (ContentView.swift:122-128). */
int rt_context_create(int device_ordinal, RtContext **out);
void rt_context_destroy(RtContext *);
const char *rt_last_error(void);
int rt_abi_version(void);

/* ---- scene description: replaces SceneBuffers.init (SceneBuffers.swift:15-72) -- */

/* ctx may be NULL for a description-only scene (validated, but rt_scene_commit fails). */
int rt_scene_create(RtContext *, RtScene **out);
void rt_scene_destroy(RtScene *);

/* MaterialEncoder.encode (Material.swift:46-56): appends one record (any of the nine
 * structs above; size must match its kind) and returns its handle, to be stored in the
 * *_material_offset fields.  Handles count 0, 1, 2 ... in call order. */
int rt_add_material(RtScene *, const void *record, size_t size);
/* TextureLoader.load (Texture.swift:20-41): 8-bit RGB, rows top to bottom; texture
 * coordinate v = 0 is the bottom row as with the reference's bottom-left origin. */
int rt_add_texture_rgb8(RtScene *, int width, int height, const uint8_t *rgb);

/* ---- RayTracingKit textures (Texture.swift): solid, checker over two textures, bilinear
 * image window, noise, marble.  A material of kind rt_material_kind_lambertian_kit_texture /
 * rt_material_kind_emissive_kit_texture names one by the handle rt_add_kit_texture returned. */
typedef struct RtKitPerlin {             /* PerlinNoise, Texture.swift:166-182 */
    float vectors[256][3];               /* components in [-1, 1], not normalised */
    uint8_t perm_x[256], perm_y[256], perm_z[256];
} RtKitPerlin;
enum RtKitTextureKind {
    RT_KIT_TEXTURE_SOLID = 0,            /* SolidColor, :13-25 */
    RT_KIT_TEXTURE_CHECKER = 1,          /* CheckerTexture, :27-51: even / odd are texture handles */
    RT_KIT_TEXTURE_IMAGE = 2,            /* ImageTexture, :53-120: bilinear, window, mirrored, wrapped or clamped */
    RT_KIT_TEXTURE_NOISE = 3,            /* NoiseTexture, :136-149 */
    RT_KIT_TEXTURE_MARBLE = 4            /* Marble, :151-164 */
};
typedef struct RtKitTexture {
    uint32_t kind;
    float color[3];                      /* solid */
    float scale;                         /* checker, noise, marble */
    int32_t even, odd;                   /* checker */
    int32_t image;                       /* image: handle of rt_add_texture_rgb8 (rows top to bottom) */
    int32_t origin_x, origin_y;          /* image: first texel column / row (:55-56) */
    int32_t width, height;               /* image: window size; negative = runs towards smaller indices (:72-75) */
    uint32_t wrap_x, wrap_y;             /* image: wrap (1) or clamp (0) outside the window (:103-120) */
    int32_t perlin;                      /* noise, marble: handle of rt_add_kit_perlin */
} RtKitTexture;
typedef struct { uint32_t kind; uint32_t texture; } RtKitTexturedMaterial;
int rt_add_kit_perlin(RtScene *, const RtKitPerlin *);     /* -> perlin handle */
int rt_add_kit_texture(RtScene *, const RtKitTexture *);   /* -> texture handle */

/* Renderable.asImpl (Primitives.swift:39-42,133-145,186-193,266-274): returns a shape id. */
int rt_shape_sphere(RtScene *, const RtSphere *);
int rt_shape_cylinder(RtScene *, const RtCylinder *);
int rt_shape_cuboid(RtScene *, const RtCuboid *);
int rt_shape_quad(RtScene *, const RtQuad *);
/* Subtract / Union / Intersection (CSG.swift:8-119; Impl/RenderableImpl.h:506-523):
 * children are shape ids (primitives or other CSG shapes); for RT_CSG_SUBTRACT child 0
 * is the left-hand side.  At most RT_CSG_MAX_LEAVES primitives in total. */
int rt_shape_csg(RtScene *, int op, int n_children, const int *children);
/* ConstantDensityVolume (ConstantDensityVolume.swift:8-33; RenderableImpl.h:717-725). */
int rt_shape_cdv(RtScene *, int base_shape, float density);
/* RenderableGrouper.add (SceneBuffers.swift:116-119): makes a shape a scene object. */
/* MotionBlur (RayTracingKit/Hittable.swift:528-555): `base` (a primitive or CSG shape) seen
 * through transform.pow(time), time in [0, 1) drawn once per sample after the camera ray
 * (Camera.swift:89; only drawn when the scene holds a moving object).  The BVH box is
 * Transform3D.boundingBoxForAnimation (Transform.swift:93-154). */
typedef struct RtMotion {
    float rotation[4];      /* unit quaternion w, x, y, z with w >= 0 (Transform.swift:169-206) */
    float translation[3];
} RtMotion;
int rt_shape_motion(RtScene *, int base_shape, const RtMotion *);
/* Hittable.boundingBox (Hittable.swift:98) of any shape before padding: min xyz, max xyz. */
int rt_shape_bounds(const RtScene *, int shape, float out_min_max[6]);
/* Time of the rays given to rt_first_hit / rt_intervals (rt_render draws its own). */
int rt_scene_set_time(RtScene *, float time);

int rt_add_object(RtScene *, int shape);
/* Bulk form of rt_shape_sphere + rt_add_object for n plain spheres (identity rotation):
 * center_radius holds n x 4 floats (cx, cy, cz, r).  No counterpart in the reference,
 * whose scenes hold a handful of objects; it exists so that a million-sphere scene does
 * not cost a million calls.  Must come after every rt_add_object; returns the object
 * index of the first sphere. */
int rt_add_sphere_objects(RtScene *, int n, const float *center_radius, const uint32_t *material_offsets);
/* Acceleration-structure build + upload (SceneBuffers.swift:42-71;
 * BoundingVolumeNode.init, RayTracingKit/Hittable.swift:609-631).  Scenes of 32768 objects
 * or more are built with up to 15 helper threads that end before the call returns; the tree
 * does not depend on how many of them could be started. */
int rt_scene_commit(RtScene *);
int rt_scene_stats(const RtScene *, RtSceneStats *out);

/* Options; no counterpart in the reference (Metal's acceleration structure is opaque).
 * RT_OPTION_BVH_BUILDER (before commit): RT_BVH_MEDIAN_SPLIT is the reference's rule
 * (RayTracingKit/Hittable.swift:609-631), RT_BVH_SAH (default) a binned surface-area
 * heuristic; hit results do not depend on the choice.  RT_OPTION_REFILL_THRESHOLD: how many
 * lanes of a wave must be waiting before the fused kernel leaves its traversal loop (1..64; left
 * unset, each kernel uses its own: 48, and 24 for the kernel of trees deeper than 16 levels).
 * RT_OPTION_NODE_MIN: below how many lanes on inner nodes the kernel turns to waiting leaf tests
 * (0..64; left unset: 8, and 40 for the kernel of trees deeper than 16 levels).
 * RT_OPTION_NODE_FORMAT (before commit): how rt_render's fused kernel reads BVH nodes.
 * RT_NODES_F16 stores the child boxes as halves rounded outward (32-byte nodes, half the
 * fetch work; boxes only cull, so hits are the same), RT_NODES_F32 keeps the 64-byte nodes,
 * RT_NODES_AUTO (default) takes halves unless a box plane lies beyond +-60000.
 * RT_OPTION_HEAVY_MIN: how many lanes of a wave must wait at a CSG shape, volume or moving object
 * before the fused kernel runs those tests (they wait unless nothing else in the wave can move).
 * RT_OPTION_TRAVERSAL: how rt_render's fused kernel finds the closest hit.  RT_TRAVERSAL_LIST tests every
 * object in turn (Array<any Hittable>.hit, RayTracingKit/Hittable.swift:662-671), all lanes of a wave
 * the same object at once: for scenes of at most 16 objects, where a tree only adds divergence;
 * RT_TRAVERSAL_TREE walks the BVH; RT_TRAVERSAL_AUTO (default) takes the list when the scene has one.
 * Hits are the same (ties between two objects at exactly the same distance may resolve differently).
 * RT_OPTION_KERNEL_TIMING: 1 records HIP events around every launch of the render loop's
 * dominant kernel for rt_kernel_time (off by default: a viewer that never asks pays nothing).
 * RT_OPTION_PROFILE_PHASES: 1 runs the diagnostic build of the fused kernel that counts
 * per-phase executions, lanes and cycles into the workspace header.
 * RT_OPTION_CONVENTION (before commit): where the reference's two renderers disagree,
 * RT_CONVENTION_METAL (default) follows MetalRayTracer (sphere / tube u of
 * Impl/RenderableImpl.h:83,205, cap uv of :113-121, metal scatter of MaterialsImpl.h:163-171),
 * RT_CONVENTION_KIT follows RayTracingKit (sphere u of Hittable.swift:188, cap and tube uv of
 * :245-248,281-282, metal scatter of Material.swift:53-57: absorb on <= 0). */
enum RtOption {
    RT_OPTION_BVH_BUILDER = 1, RT_OPTION_REFILL_THRESHOLD = 2, RT_OPTION_NODE_MIN = 3, RT_OPTION_NODE_FORMAT = 4,
    RT_OPTION_KERNEL_TIMING = 5, RT_OPTION_PROFILE_PHASES = 6, RT_OPTION_CONVENTION = 7, RT_OPTION_HEAVY_MIN = 8,
    RT_OPTION_TRAVERSAL = 9
};
enum RtTraversal { RT_TRAVERSAL_AUTO = 0, RT_TRAVERSAL_TREE = 1, RT_TRAVERSAL_LIST = 2 };
enum RtBvhBuilder { RT_BVH_MEDIAN_SPLIT = 0, RT_BVH_SAH = 1 };
enum RtNodeFormat { RT_NODES_AUTO = 0, RT_NODES_F32 = 1, RT_NODES_F16 = 2 };
enum RtConvention { RT_CONVENTION_METAL = 0, RT_CONVENTION_KIT = 1 };
int rt_scene_set_option(RtScene *, int option, int value);

/* ---- queries (parity surface) ------------------------------------------------ */

/* Closest hit of n rays (rays_dev: n x 6 floats, origin then direction) with
 * tmin <= t <= tmax.  volume_key is the 64-bit word that keys the free-flight draws of
 * constant-density volumes (the renderer passes the path generator's state). */
int rt_first_hit(RtScene *, const float *rays_dev, int n, float tmin, float tmax,
                 uint64_t volume_key, RtHit *out_dev, void *stream);

/* Ordered boundary crossings of one shape along n rays: events_dev holds
 * n x RT_MAX_EVENTS records, counts_dev n counts.  scripted_u_dev (may be NULL) replaces
 * the volume's random numbers by the given values, cycled in order, which is the
 * reference's test generator (Impl/RNG.h:15-27). */
int rt_intervals(RtScene *, int shape, const float *rays_dev, int n,
                 const float *scripted_u_dev, int n_scripted, uint64_t volume_key,
                 RtEvent *events_dev, int32_t *counts_dev, void *stream);

/* One scattering event of a material at a given hit (MaterialsImpl.h:147-215; under
 * RT_CONVENTION_KIT the metal rule of RayTracingKit/Material.swift:53-57), with the random
 * numbers of the generator stream of (rng_seed, rng_sample, rng_x, rng_y): what the shade stage
 * does to one path, exposed so that every material kind can be compared with a host
 * implementation draw for draw.  queries_dev / out_dev: n records. */
typedef struct {
    float direction[3];       /* incoming ray direction (unit) */
    float normal[3];          /* hit normal, already flipped to oppose the ray */
    float point[3];
    float uv[2];
    int32_t front_face;
    uint32_t material_offset;
    uint32_t rng_sample, rng_x, rng_y;
} RtScatterQuery;
typedef struct {
    int32_t scattered;        /* 0: the path ends here (emissive, or absorbed) */
    float direction[3];       /* outgoing direction when scattered */
    float attenuation[3];
    float emitted[3];
    uint32_t pad[2];
} RtScatterResult;
int rt_scatter(RtScene *, const RtScatterQuery *queries_dev, int n, uint64_t rng_seed,
               RtScatterResult *out_dev, void *stream);

/* The ray-generation stage alone (Shaders.metal:30-67): rays of samples
 * first_sample .. first_sample+samples_per_pixel-1 of the pixels of the window
 * [x0,x0+w) x [y0,y0+h), pixel-major then sample; out n x 6 floats. */
int rt_camera_rays(RtContext *, const RtCameraConfig *, const RtRenderConfig *, uint32_t first_sample,
                   int width, int height, int x0, int y0, int w, int h,
                   float *rays_out_dev, void *stream);

/* Raw words of the device generators (Impl/RNG.h:31-52 PCG32; RayTracingKit/PRNG.swift:17-21
 * WyRand) for bit-exact comparison with a host implementation.  streams_dev holds n records
 * of 8 uint32 words: rng_seed low word, rng_seed high word, sample, x, y, 0, 0, 0.  For
 * record i the first words_per_stream PCG32 outputs of the stream of that (pixel, sample) go
 * to out_dev[i * words_per_stream ...], and the 64-bit WyRand output number `sample` of
 * rng_seed (the per-sample seed) to wyrand_out_dev[2 i] (low word), [2 i + 1] (high word). */
int rt_rng_words(RtContext *, const uint32_t *streams_dev, int n, int words_per_stream,
                 uint32_t *out_dev, uint32_t *wyrand_out_dev, void *stream);

/* ---- render: replaces Renderer.draw (ContentView.swift:182-224) --------------- */

/* Device scratch the render loop needs for a width x height image. */
size_t rt_render_workspace_size(int pipeline, int width, int height);

/* Adds config->samples_per_pixel samples per pixel to accum_dev: width*height*3 floats,
 * the per-pixel RGB radiance SUMS (row-major, interleaved RGB).  The caller keeps the
 * sample count (the reference keeps passCounter on the host too, ContentView.swift:187).
 * first_sample is the global index of the call's first sample: sample s of pixel (x, y)
 * always uses the same random stream, whichever call, pipeline or GPU renders it (the
 * reference draws a fresh rng_seed per pass instead; a caller that does so passes 0).
 * workspace_dev must hold rt_render_workspace_size bytes. */
int rt_render(RtScene *, const RtCameraConfig *, const RtRenderConfig *, uint32_t first_sample, int pipeline,
              int width, int height, float *accum_dev,
              void *workspace_dev, size_t workspace_bytes, void *stream);

/* rt_render on the fused pipeline without a workspace argument and without any allocation.
 * No counterpart in the reference.  It is for callers that cannot allocate device memory or
 * call anything that might: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  The kernel's scratch (work counter, camera, deep traversal-stack columns:
 * 40 MiB + 4 KiB) is an array inside the library, one copy per GPU, resolved when
 * rt_context_create runs on that GPU and held until the process ends.  The call itself makes no
 * hipMalloc / hipFree, creates no events (rt_kernel_time does not see it), never synchronises and
 * asks the runtime for nothing but the current device (hipGetDevice, and hipSetDevice there and back
 * when the caller's current device is not the scene's; the occupancy query was made by
 * rt_scene_commit): two kernel launches on `stream` are all it does, so it may be recorded into a
 * HIP graph, whose replays render the camera and configuration given at capture time.  The caller
 * still passes the accumulator.
 * Limits: those of rt_render (an image beyond 65535 pixels a side or 2^28 pixels is refused with
 * RT_ERR_LIMIT); on a GPU with more than 256 compute units the grid is sized for 256.
 * Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 * scenes) race; calls in stream order do not (each launch re-initialises what it reads). */
int rt_render_static(RtScene *, const RtCameraConfig *, const RtRenderConfig *, uint32_t first_sample,
                     int width, int height, float *accum_dev, void *stream);

/* Image-space sharding for several GPUs; no counterpart in the reference (one device).
 * The image is cut into bands of 8 pixel rows; band b belongs to shard b % band_count.
 * This call renders the bands of shard band_index only (fused pipeline) and adds them to
 * accum_bands_dev, which holds just those bands stacked in order:
 * rt_band_rows(height, band_index, band_count) rows of width*3 floats.  Random streams are
 * those of the full image, so the shards of all ranks put together are the image rt_render
 * gives.  Equal-sized shard buffers (rt_band_rows(height, 0, band_count) rows) gathered
 * from all ranks hold every pixel exactly once: no arithmetic in the exchange. */
int rt_band_rows(int height, int band_index, int band_count);
int rt_render_bands(RtScene *, const RtCameraConfig *, const RtRenderConfig *, uint32_t first_sample,
                    int band_index, int band_count, int width, int height, float *accum_bands_dev,
                    void *workspace_dev, size_t workspace_bytes, void *stream);

/* Camera.swift:92 / Vector3D.swift:181-211: mean -> min(sqrt(x), 1) ->
 * floor(255.999.. * x), written as RGBX8 (width*height*4 bytes); n_samples is the number
 * of samples per pixel the sums hold. */
int rt_resolve_rgb8(RtContext *, const float *accum_dev, uint32_t n_samples, int width, int height,
                    uint8_t *rgbx_out_dev, void *stream);

/* The app's accumulate-and-present step, Shaders.metal:151-158 / ContentView.swift:182-235:
 * pass_sum_dev holds ONE pass (rt_render into a zeroed W*H*3 buffer with samples_per_pixel
 * samples); its mean, clamped to 1, is blended with weight 1/pass_counter into the 10-bit
 * running mean `acc_rgb10a2_dev` (the app's rgb10a2Uint accumulator texture, one word per
 * pixel: r bits 0-9, g 10-19, b 20-29) and sqrt(mean) is written to `bgra8_out_dev` as the
 * drawable's bgra8Unorm pixel.  pass_counter = 1 starts a new image (the old accumulator
 * content has weight 0). */
int rt_progressive_present(RtContext *, const float *pass_sum_dev, uint32_t samples_per_pixel, uint32_t pass_counter,
                           uint32_t *acc_rgb10a2_dev, uint8_t *bgra8_out_dev, int width, int height, void *stream);

/* Timing of the render loop's dominant kernel on its own stream, when RT_OPTION_KERNEL_TIMING
 * is on (HIP events around every launch since the last reset): total milliseconds and
 * launch count. */
int rt_kernel_time(RtScene *, int reset, double *total_ms, uint64_t *launches, const char **kernel_name);

#ifdef __cplusplus
}
#endif
#endif
