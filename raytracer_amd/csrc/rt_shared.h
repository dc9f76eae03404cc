// rt_shared.h - plain-data layouts shared by the host flattener and the HIP kernels.
//
// Device layout (all buffers live in HBM, written once at rt_scene_commit):
//   nodes      4 x float4 per BVH node: both child boxes and both child references, so one
//              64-byte fetch serves two box tests.
//   nodes16    the same nodes in 32 bytes: box planes as IEEE halves rounded outward (a
//              superset of every box, so culling with them loses no hit), refs as they are.
//              A wave-wide node fetch occupies the CU's address unit for as many cycles as
//              each lane requests bytes, which makes the 64-byte form the render kernel's
//              tightest resource; this form halves it.
//   spheres    float4 (cx, cy, cz, r) per sphere: the whole geometry of the common case
//              in one 16-byte load; material ids and UV rotations are separate streams
//              that only the shade stage touches.
//   xprims     4 x float4 per cylinder / cuboid (rotation, translation, extents) / quad (origin, v x w, w x u, -, normal, d).
//   prim_mat   per-face material ids of the xprims (6 per record), sphere_mat for spheres.
//   csg / cdv  small records naming leaf primitives; a CSG tree is flattened to a truth
//              table over its leaves.
//   materials  2 x float4 per material; texels RGBA8; perlin tables as in the C ABI.
#pragma once

#include <stdint.h>

#define RT_REF_NONE 0xFFFFFFFFu
#define RT_REF_LEAF 0x80000000u
#define RT_KIND_SHIFT 28
#define RT_INDEX_MASK 0x0FFFFFFFu

enum RtPrimKind : uint32_t {
    RT_KIND_SPHERE = 0,
    RT_KIND_CYLINDER = 1,
    RT_KIND_CUBOID = 2,
    RT_KIND_QUAD = 3,
    RT_KIND_CSG = 4,
    RT_KIND_CDV = 5,
    RT_KIND_MOTION = 6,
};

// Traversal stack entries per lane (24 KB of LDS per 256-thread workgroup).  The kernels' LDS
// stacks, the builder's depth limit, the commit-time depth check and the deep-stack columns of
// the fused workspace all follow this one definition.
#define RT_STACK_DEPTH 24
// The four-children form of a tree of RT_STACK_DEPTH levels has half the levels and leaves at most three entries on the
// stack per level: the deep-stack columns of the fused workspace are sized for this many entries per lane.
#define RT_WIDE_STACK_DEPTH (3 * ((RT_STACK_DEPTH + 1) / 2))
#define RT_BLOCK 256             // threads per workgroup of every traversal kernel
#define RT_TOP_NODES_MAX 1024    // nodes the builder of a deep tree moves to the head of the node array, largest boxes first
#define RT_LINEAR_MAX 16         // scenes up to this many objects are rendered without a tree
#define RT_CSG_LEAVES 8           // leaves whose intervals the walk over a CSG shape keeps in registers
#define RT_CSG_WIDE_LEAVES 64     // leaves of a shape that is walked by recomputing intervals (RT_CSG_MAX_LEAVES of the C ABI)
// Postfix program of a wide shape: a token below 64 pushes "inside leaf <token>", the others combine the two on top.
#define RT_CSG_TOKEN_UNION 64u
#define RT_CSG_TOKEN_INTERSECTION 65u
#define RT_CSG_TOKEN_SUBTRACT 66u

static inline uint32_t rt_make_ref(uint32_t kind, uint32_t index) { return (kind << RT_KIND_SHIFT) | index; }

// Up to RT_CSG_LEAVES leaves: the record holds the leaves and the truth table.  More (a "wide" shape): leaf[0] is
// the shape's offset in RtDevScene::csg_pool, where flip mask (2 words), token count, the leaf references and the
// postfix program of the expression follow one another; flip_mask and truth are unused.
struct RtDevCsg {
    uint32_t n_leaves;
    uint32_t flip_mask;                 // bit i: leaf i sits under an odd number of subtract right-hand sides
    uint32_t leaf[RT_CSG_LEAVES];       // primitive references (kind | index), depth-first order
    uint32_t truth[8];                  // inside(root) for every leaf-membership mask (256 bits)
};
#define RT_CSG_POOL_HEADER 3u

struct RtDevCdv {
    uint32_t base;                      // primitive or CSG reference
    float neg_inv_density;
};

// MotionBlur (RayTracingKit/Hittable.swift:528-555): `base` seen through transform.pow(time).
struct RtDevMotion {
    uint32_t base;                      // primitive or CSG reference
    float qx, qy, qz;                   // vector part of the unit quaternion (w = cos(half_angle) >= 0)
    float half_angle, inv_sin;          // acos(w) and 1 / sin(acos(w)); inv_sin = 0 for the identity rotation
    float tx, ty, tz;
    uint32_t pad[3];
};

struct RtDevTexture {
    uint32_t offset;                    // first texel in the RGBA8 pool
    uint32_t width, height, pad;
};

// RayTracingKit textures (Texture.swift).  One record per texture; checker records name two others.
struct RtDevKitTexture {
    uint32_t kind;                      // RtKitTextureKind
    float color[3];
    float scale;                        // checker: 1 / scale
    int32_t even, odd;
    uint32_t image;                     // index into `textures`
    int32_t origin_x, origin_y, width, height;   // width, height > 0
    int32_t step_x, step_y;             // +1 / -1
    uint32_t wrap_x, wrap_y;
    uint32_t perlin;
};
struct RtDevKitPerlin {
    float vectors[256][3];
    uint8_t perm[3][256];
};

struct RtDevPerlin {
    float colors[2][3];
    float frequency;
    uint32_t turbulence;
    float vectors[256][3];
    uint8_t permutations[3][256];
};

// Everything a kernel needs to trace and shade; passed by value.
struct RtDevScene {
    const float *nodes;            // n_nodes x 16 floats
    const uint32_t *nodes16;       // n_nodes x 8 words: the same boxes as 12 halves rounded outward (per axis the two
                                   // children's min planes, then their max planes), then the two refs
    const float *spheres;          // n x 4
    const float *sphere_rot;       // n x 12 (three padded columns) or nullptr when every rotation is identity
    const uint32_t *sphere_mat;
    const float *sphere_mrec;      // n x 8: the material record of each sphere, by value (shade needs no second lookup)
    const uint32_t *sphere_obj;
    const float *xprims;           // n x 16
    const uint32_t *xprim_mat;     // n x 6
    const uint32_t *xprim_obj;
    const RtDevCsg *csg;
    const uint32_t *csg_obj;
    const uint32_t *csg_pool;      // leaves and postfix programs of the wide CSG shapes
    const RtDevCdv *cdv;
    const uint32_t *cdv_obj;
    const RtDevMotion *motion;
    const uint32_t *motion_obj;
    const float *materials;        // n x 8
    const RtDevTexture *textures;
    const uint32_t *texels;
    const RtDevPerlin *perlin;
    const RtDevKitTexture *kit_textures;
    const RtDevKitPerlin *kit_perlin;
    // Scenes of at most RT_LINEAR_MAX objects also carry their objects as a plain list (reference and
    // padded box, 8 words per object: ref, 6 planes min/max per axis, pad): the render kernel then
    // tests every object in turn, the same object in all lanes at once, instead of walking a tree.
    const uint32_t *object_list;
    uint32_t n_listed;             // 0 when there is no list
    uint32_t n_wide_csg;           // CSG shapes with more than RT_CSG_LEAVES leaves (0: the kernels skip the check)
    uint32_t root;                 // child reference of the root (RT_REF_NONE for an empty scene)
    uint32_t n_nodes;
    uint32_t kit_convention;       // 1: RayTracingKit's texture coordinates and metal rule (RT_CONVENTION_KIT)
    uint32_t root4;                // the root's reference in nodes4 (byte offset or leaf)
    // A deep tree of spheres also has its nodes in a four-children form, two levels of the binary tree per node (same
    // boxes, same leaves): 64 bytes = per axis the four children's min planes and max planes as halves rounded outward
    // (children 0,1 | 2,3 | 0,1 | 2,3), then four refs (an inner child as its byte offset in this array; an absent
    // child is RT_REF_NONE behind the empty box).  nullptr when the scene has none.
    const uint32_t *nodes4;
    uint32_t n_nodes4;
    uint32_t pad;
};

// Camera as the kernels use it (Shaders.metal:20-48 computed once on the host instead of
// once per thread).
struct RtDevCamera {
    float center[3];
    float viewport_u[3], viewport_v[3], viewport_center[3];
    float defocus_u[3], defocus_v[3];
    float inv_w, inv_h;
    uint32_t defocus;        // 1 when defocus_angle > 0
    uint32_t background;
};

struct RtDevRender {
    uint32_t width, height;
    uint32_t spp, max_depth;
    uint32_t first_sample;
    uint32_t seed_lo, seed_hi;
    uint32_t animated;       // 1 when the scene holds moving objects: every sample draws a time after its ray
    // Image-space sharding: the accumulator holds only the 8-row bands b with b % band_count ==
    // band_index, stacked; band_count = 1 is the whole image.
    uint32_t band_index, band_count;
};
