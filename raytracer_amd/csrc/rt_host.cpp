// rt_host.cpp - host side of the C ABI (include/raytracer_amd.h): scene description,
// flattening into device buffers, BVH build, and the entry points that launch kernels.
//
// Flattening replaces MetalRayTracer/SceneBuffers.swift:15-72: instead of one AoS buffer per
// concrete Swift type it writes structure-of-arrays streams split by what each stage reads
// (see rt_shared.h).  The BVH follows RayTracingKit/Hittable.swift:609-631: longest axis
// of the node's box (ties as AABB.swift:59-66), objects ordered by box centre on that
// axis, split at (n+1)/2 (or, by default, a binned surface-area split) - linearised into
// 64-byte nodes that carry both child boxes, plus a 32-byte half-precision copy of them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "rt_launch.h"
#include "rt_shared.h"

#define RT_DEFAULT_HEAVY_MIN 8u      // measured: +10 % on the CSG / textured workload, +12 % on the Kit's world, none lost elsewhere

// The records are the reference's, byte for byte (MetalRayTracer/Renderable.h:13-63,
// Config.h:18-33, Materials.h:38-93 with 16-byte simd vectors).
static_assert(sizeof(RtFloat3) == 16 && alignof(RtFloat3) == 16, "vector_float3");
static_assert(sizeof(RtMatrix3x3) == 48, "matrix_float3x3");
static_assert(sizeof(RtTransform) == 64 && offsetof(RtTransform, translation) == 48, "Transform");
static_assert(sizeof(RtSphere) == 80 && offsetof(RtSphere, radius) == 64 && offsetof(RtSphere, material_offset) == 72, "Sphere");
static_assert(sizeof(RtCylinder) == 96 && offsetof(RtCylinder, height) == 68 && offsetof(RtCylinder, bottom_material_offset) == 72, "Cylinder");
static_assert(sizeof(RtCuboid) == 128 && offsetof(RtCuboid, size) == 64 && offsetof(RtCuboid, material_offset) == 80, "Cuboid");
static_assert(sizeof(RtQuad) == 96 && offsetof(RtQuad, normal) == 64 && offsetof(RtQuad, d) == 80 && offsetof(RtQuad, material_offset) == 88, "Quad");
static_assert(sizeof(RtCameraConfig) == 80 && offsetof(RtCameraConfig, look_from) == 16 && offsetof(RtCameraConfig, up) == 48 &&
              offsetof(RtCameraConfig, defocus_angle) == 64 && offsetof(RtCameraConfig, background) == 72, "CameraConfig");
static_assert(sizeof(RtRenderConfig) == 24 && offsetof(RtRenderConfig, rng_seed) == 16, "RenderConfig");
static_assert(sizeof(RtPerlinNoiseTexture) == 4912 && offsetof(RtPerlinNoiseTexture, vectors) == 48 &&
              offsetof(RtPerlinNoiseTexture, permutations) == 4144, "PerlinNoiseTexture");
static_assert(sizeof(RtColoredLambertianMaterial) == 32 && offsetof(RtColoredLambertianMaterial, albedo) == 16, "ColoredLambertianMaterial");
static_assert(sizeof(RtTexturedLambertianMaterial) == 16 && offsetof(RtTexturedLambertianMaterial, albedo) == 8, "TexturedLambertianMaterial");
static_assert(sizeof(RtPerlinNoiseLambertianMaterial) == 4928, "PerlinNoiseLambertianMaterial");
static_assert(sizeof(RtColoredMetalMaterial) == 48 && offsetof(RtColoredMetalMaterial, fuzz) == 32, "ColoredMetalMaterial");
static_assert(sizeof(RtTexturedMetalMaterial) == 24 && offsetof(RtTexturedMetalMaterial, fuzz) == 16, "TexturedMetalMaterial");
static_assert(sizeof(RtPerlinNoiseMetalMaterial) == 4944 && offsetof(RtPerlinNoiseMetalMaterial, fuzz) == 4928, "PerlinNoiseMetalMaterial");
static_assert(sizeof(RtDielectricMaterial) == 8, "DielectricMaterial");
static_assert(sizeof(RtScatterQuery) == 64 && sizeof(RtScatterResult) == 48, "scatter records");

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
int fail_hip(hipError_t e, const char *what) {
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return RT_ERR_DEVICE;
}

struct Box3 {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    void add(const float p[3]) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    void add(const Box3 &b) { add(b.mn); add(b.mx); }
};

struct HostShape {
    uint32_t kind = 0;
    float rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, trans[3] = {0, 0, 0};
    float radius = 0, height = 0, size[3] = {0, 0, 0};
    float quad[16] = {0};
    uint32_t mat[6] = {0, 0, 0, 0, 0, 0};
    int op = 0;
    std::vector<int> children;
    int base = -1;
    float density = 0;
    float m_quat[4] = {1, 0, 0, 0}, m_trans[3] = {0, 0, 0};      // motion blur: unit quaternion (w, x, y, z), translation
    int parent = -1;        // CSG / volume that uses this shape, -1 if none
    int object = -1;        // scene object made of this shape, -1 if none
    uint32_t ref = RT_REF_NONE;
    Box3 box;
};

struct HostTexture { int w, h; std::vector<uint32_t> rgba; };

}  // namespace

struct RtContext {
    int device = 0;
    int cus = 256;
};

struct RtScene {
    RtContext *ctx = nullptr;
    std::vector<HostShape> shapes;
    std::vector<int> objects;
    std::vector<float> materials;          // 8 floats each
    std::vector<uint32_t> material_kinds;
    std::vector<HostTexture> textures;
    std::vector<RtDevPerlin> perlin;
    std::vector<RtDevKitTexture> kit_textures;
    std::vector<RtDevKitPerlin> kit_perlin;
    // bulk spheres (kept as arrays: a million HostShape records would be wasteful)
    std::vector<float> bulk_spheres;       // cx, cy, cz, r
    std::vector<uint32_t> bulk_mats;
    int bvh_builder = RT_BVH_SAH;
    uint32_t refill = 0;               // 0: the kernel's own default (48; 28 for the deep-tree kernel that keeps the top of the tree in LDS)
    int node_min = -1;      // -1: chosen at render time from what the scene holds
    int heavy_min = -1;     // -1: default
    int traversal = RT_TRAVERSAL_AUTO;
    int node_format = RT_NODES_AUTO;
    int convention = RT_CONVENTION_METAL;
    bool profile_phases = false;
    float query_time = 0.f;     // time of rt_first_hit / rt_intervals queries
    bool use_nodes16 = false;
    bool committed = false;
    std::vector<void *> device_allocs;
    RtDevScene dev{};
    uint32_t feat = 0;
    RtSceneStats stats{};
    RtKernelTimer timer;
    RtWfProbe wf_probe;            // live-count read-back of the wavefront loop (created by the first wavefront render)

    ~RtScene() { release(); }
    void release() {
        for (void *p : device_allocs) (void)hipFree(p);
        device_allocs.clear();
    }
};

namespace {

template <class T> int upload(RtScene *sc, const std::vector<T> &v, const T **out) {
    *out = nullptr;
    if (v.empty()) return RT_OK;
    void *p = nullptr;
    size_t bytes = v.size() * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc");
    sc->device_allocs.push_back(p);
    e = hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpy");
    sc->stats.device_bytes += bytes;
    *out = (const T *)p;
    return RT_OK;
}

void xform(const HostShape &s, const float p[3], float out[3]) {
    for (int r = 0; r < 3; r++) out[r] = s.rot[r] * p[0] + s.rot[3 + r] * p[1] + s.rot[6 + r] * p[2] + s.trans[r];
}

// ---- where a moving object can be during [0, 1): Transform3D.boundingBoxForAnimation,
// RayTracingKit/Transform.swift:93-154, in double precision.
struct Quatd { double w, x, y, z; };
Quatd quat_mul(Quatd a, Quatd b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
void quat_rotate(Quatd q, const double v[3], double out[3]) {
    Quatd r = quat_mul(quat_mul(q, Quatd{0, v[0], v[1], v[2]}), Quatd{q.w, -q.x, -q.y, -q.z});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void sweep_point(Quatd q, const double tr[3], const double p[3], double mn[3], double mx[3]) {
    auto add = [&](const double v[3]) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], v[a]); mx[a] = std::max(mx[a], v[a]); } };
    double p1[3], p2[3];
    quat_rotate(q, p, p1);
    for (int a = 0; a < 3; a++) p2[a] = p1[a] + tr[a];
    add(p);
    add(p2);
    if (q.w == 1.0) return;
    // each coordinate of p(t) = (sin((1-t)th) p + sin(t th) p1) / sin(th) + t tr has at most two
    // stationary points in [0, 1]: A sin(t th) + B cos(t th) = C  (:99-136)
    const double tau = 2 * 3.14159265358979323846264338327950288;
    double th = 2 * std::acos(q.w), sin_th = std::sin(th), cos_th = std::cos(th);
    auto norm_angle = [&](double x) { return x < 0 ? x + tau : (x >= tau ? x - tau : x); };
    for (int a = 0; a < 3; a++) {
        double A = th * sin_th * p[a], B = th * (cos_th * p[a] - p1[a]), C = tr[a] * sin_th;
        double Z = C / std::sqrt(A * A + B * B);
        if (!std::isfinite(Z) || std::fabs(Z) > 1) continue;
        double phi = std::atan2(A, B), acos_z = std::acos(Z);
        double ts[2] = {norm_angle(phi + acos_z) / th, norm_angle(phi - acos_z) / th};
        for (double t : ts) {
            if (!(std::isfinite(t) && t >= 0 && t <= 1)) continue;
            double e[3];
            for (int c = 0; c < 3; c++) e[c] = (std::sin((1 - t) * th) / sin_th) * p[c] + (std::sin(t * th) / sin_th) * p1[c] + t * tr[c];
            add(e);
        }
    }
}
Box3 sweep_box(const float quat[4], const float trans[3], const Box3 &box) {
    Quatd q{quat[0], quat[1], quat[2], quat[3]};
    double tr[3] = {trans[0], trans[1], trans[2]};
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (q.w == 1.0) {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min((double)box.mn[a], box.mn[a] + tr[a]);
            mx[a] = std::max((double)box.mx[a], box.mx[a] + tr[a]);
        }
    } else {
        for (int i = 0; i < 8; i++) {
            double p[3] = {(i & 1) ? box.mx[0] : box.mn[0], (i & 2) ? box.mx[1] : box.mn[1], (i & 4) ? box.mx[2] : box.mn[2]};
            sweep_point(q, tr, p, mn, mx);
        }
    }
    Box3 out;
    for (int a = 0; a < 3; a++) {     // to float, outward
        out.mn[a] = std::nextafter((float)mn[a], -INFINITY);
        out.mx[a] = std::nextafter((float)mx[a], INFINITY);
    }
    return out;
}

// Bounding boxes: Primitives.swift:29-33 (sphere), :112-125 (cylinder, disk formula),
// :163-180 (cuboid corners), :253-260 (quad corners); CSG.swift:21-23 (subtract = lhs),
// :60-66 (union), :102-108 (intersection); ConstantDensityVolume.swift:21-23.
Box3 shape_box(const RtScene &sc, int id) {
    const HostShape &s = sc.shapes[id];
    Box3 b;
    switch (s.kind) {
    case RT_KIND_SPHERE:
        for (int a = 0; a < 3; a++) { b.mn[a] = s.trans[a] - s.radius; b.mx[a] = s.trans[a] + s.radius; }
        break;
    case RT_KIND_CYLINDER: {
        const float *n = s.rot + 3;
        for (int a = 0; a < 3; a++) {
            float rp = std::sqrt(std::max(0.f, 1.f - n[a] * n[a])) * s.radius;
            float bc = s.trans[a], tc = s.trans[a] + s.height * n[a];
            b.mn[a] = std::min(bc, tc) - rp;
            b.mx[a] = std::max(bc, tc) + rp;
        }
        break;
    }
    case RT_KIND_CUBOID:
        for (int i = 0; i < 8; i++) {
            float p[3] = {(i & 1) ? s.size[0] : 0.f, (i & 2) ? s.size[1] : 0.f, (i & 4) ? s.size[2] : 0.f}, w[3];
            xform(s, p, w);
            b.add(w);
        }
        break;
    case RT_KIND_QUAD:
        for (int i = 0; i < 4; i++) {
            float p[3];
            for (int a = 0; a < 3; a++) p[a] = s.quad[a] + ((i & 1) ? s.quad[3 + a] : 0.f) + ((i & 2) ? s.quad[6 + a] : 0.f);
            b.add(p);
        }
        break;
    case RT_KIND_CSG:
        if (s.op == RT_CSG_SUBTRACT) {
            b = shape_box(sc, s.children[0]);
        } else if (s.op == RT_CSG_UNION) {
            for (int c : s.children) b.add(shape_box(sc, c));
        } else {
            for (int a = 0; a < 3; a++) { b.mn[a] = -INFINITY; b.mx[a] = INFINITY; }
            for (int c : s.children) {
                Box3 cb = shape_box(sc, c);
                for (int a = 0; a < 3; a++) { b.mn[a] = std::max(b.mn[a], cb.mn[a]); b.mx[a] = std::min(b.mx[a], cb.mx[a]); }
            }
        }
        break;
    case RT_KIND_MOTION:
        b = sweep_box(s.m_quat, s.m_trans, shape_box(sc, s.base));
        break;
    default:
        b = shape_box(sc, s.base);
        break;
    }
    return b;
}

// Boxes are widened a little: a flat box (axis-aligned quad) gets thickness, and every box
// a few float ulps, so that rounding in the slab test can never cull a hit the primitive
// test would accept.
void pad_box(Box3 &b) {
    for (int a = 0; a < 3; a++) {
        if (!(b.mn[a] <= b.mx[a])) continue;    // empty (e.g. disjoint intersection): leave it
        float m = std::max(std::fabs(b.mn[a]), std::fabs(b.mx[a]));
        float pad = std::isfinite(m) ? 4e-6f * std::max(m, 1e-3f) : 0.f;
        if (b.mx[a] - b.mn[a] < 1e-4f) pad = std::max(1e-4f, std::isfinite(m) ? m * (1.f / 65536.f) : 0.f);
        b.mn[a] -= pad;
        b.mx[a] += pad;
    }
}

void collect_leaves(const RtScene &sc, int id, bool flipped, std::vector<int> &leaves, std::vector<bool> &flips) {
    const HostShape &s = sc.shapes[id];
    if (s.kind == RT_KIND_CSG) {
        for (size_t i = 0; i < s.children.size(); i++)
            collect_leaves(sc, s.children[i], flipped != (s.op == RT_CSG_SUBTRACT && i > 0), leaves, flips);
    } else {
        leaves.push_back(id);
        flips.push_back(flipped);
    }
}

// The expression of a CSG shape as a postfix program over its leaves (numbered depth first, like
// collect_leaves): a node of k operands is k - 1 binary operations folded from the left, which for a
// subtraction is "the first operand and none of the others".
void csg_tokens(const RtScene &sc, int id, int &next_leaf, std::vector<uint32_t> &out) {
    const HostShape &s = sc.shapes[id];
    if (s.kind != RT_KIND_CSG) { out.push_back((uint32_t)next_leaf++); return; }
    const uint32_t op = s.op == RT_CSG_UNION ? RT_CSG_TOKEN_UNION : (s.op == RT_CSG_INTERSECTION ? RT_CSG_TOKEN_INTERSECTION : RT_CSG_TOKEN_SUBTRACT);
    for (size_t i = 0; i < s.children.size(); i++) {
        csg_tokens(sc, s.children[i], next_leaf, out);
        if (i > 0) out.push_back(op);
    }
}

// inside(shape) for a given leaf-membership mask: the depth counting of
// Impl/RenderableImpl.h:678-691 in closed form (union: any child; intersection: all
// children; subtract: first child and none of the others).
bool eval_csg(const RtScene &sc, int id, uint32_t mask, int &next_leaf) {
    const HostShape &s = sc.shapes[id];
    if (s.kind != RT_KIND_CSG) return (mask >> next_leaf++) & 1u;
    int count = 0, depth = 0;
    for (size_t i = 0; i < s.children.size(); i++) {
        bool in = eval_csg(sc, s.children[i], mask, next_leaf);
        count += in ? 1 : 0;
        if (s.op == RT_CSG_SUBTRACT) depth += (i == 0) ? (in ? 1 : 0) : (in ? -1 : 0);
    }
    if (s.op == RT_CSG_UNION) return count >= 1;
    if (s.op == RT_CSG_INTERSECTION) return count == (int)s.children.size();
    return depth >= 1;
}

struct BuildItem { uint32_t ref; Box3 box; float c[3]; };

// IEEE half nearest to x on the side asked for (up: >= x, down: <= x), so that a box
// written in halves encloses the float box.  Out-of-range planes land on +-65504 or +-inf.
uint16_t half_toward(float x, bool up) {
    _Float16 h = (_Float16)x;                 // round to nearest
    float back = (float)h;
    uint16_t bits;
    memcpy(&bits, &h, 2);
    if (up ? back < x : back > x) {
        const bool neg = (bits & 0x8000u) != 0;
        if (up) bits = neg ? (bits == 0x8000u ? (uint16_t)0x0001u : (uint16_t)(bits - 1)) : (uint16_t)(bits + 1);
        else bits = neg ? (uint16_t)(bits + 1) : (bits == 0x0000u ? (uint16_t)0x8001u : (uint16_t)(bits - 1));
    }
    return bits;
}

// The 32-byte node form from the 64-byte one.
std::vector<uint32_t> pack_nodes16(const std::vector<float> &nodes) {
    std::vector<uint32_t> out(nodes.size() / 2);
    for (size_t i = 0; i < nodes.size() / 16; i++) {
        const float *n = nodes.data() + 16 * i;
        uint32_t *o = out.data() + 8 * i;
        for (int a = 0; a < 3; a++) {        // per axis: (left min, right min), (left max, right max)
            o[2 * a] = (uint32_t)half_toward(n[2 * a], false) | ((uint32_t)half_toward(n[6 + 2 * a], false) << 16);
            o[2 * a + 1] = (uint32_t)half_toward(n[2 * a + 1], true) | ((uint32_t)half_toward(n[6 + 2 * a + 1], true) << 16);
        }
        memcpy(o + 6, n + 12, 8);
        // a reference to an inner node is kept as that node's byte offset in this array: the kernel adds it to the
        // array's address without a shift
        for (int c = 6; c < 8; c++)
            if (!(o[c] & RT_REF_LEAF)) o[c] *= 32u;
    }
    return out;
}

// The four-children node form (RtDevScene::nodes4) of a binary tree: every node takes the children of its two children
// in their place (a leaf child stays as it is), so the tree has half the levels, the same boxes and the same leaves.
// The `top_n` nodes with the largest boxes come first in the array (as Builder::hoist_top does for the binary form),
// the others with the inner children of a node next to each other, such groups in depth-first order (two siblings share
// a 128-byte line; measured on a million spheres against plain pre-order, -1.4 %, and against ordering 8 192 / 65 536 /
// all nodes by box area, -0.3 / -1.1 / -2.1 %).  root4: the root's reference; depth4: levels of inner nodes.
std::vector<uint32_t> pack_nodes4(const std::vector<float> &nodes, uint32_t root, size_t top_n, uint32_t &root4, uint32_t &depth4) {
    struct Wide { float box[4][6]; uint32_t ref[4]; };
    depth4 = 0;
    if (root & RT_REF_LEAF) { root4 = root; return {}; }
    std::vector<Wide> wide;
    auto child_ref = [&](uint32_t node, int k) { uint32_t r; memcpy(&r, nodes.data() + 16 * (size_t)node + 12 + k, 4); return r; };
    // pre-order, without recursion: (binary node, index of its wide node, level)
    struct Todo { uint32_t node, index, level; };
    std::vector<Todo> todo;
    wide.emplace_back();
    todo.push_back({root, 0u, 1u});
    while (!todo.empty()) {
        const Todo cur = todo.back();
        todo.pop_back();
        depth4 = std::max(depth4, cur.level);
        const float *picked_box[4];
        uint32_t picked_ref[4];
        int n = 0;
        for (int k = 0; k < 2; k++) {
            const uint32_t r = child_ref(cur.node, k);
            if (r == RT_REF_NONE) continue;
            if (r & RT_REF_LEAF) { picked_box[n] = nodes.data() + 16 * (size_t)cur.node + 6 * k; picked_ref[n++] = r; continue; }
            for (int j = 0; j < 2; j++) {
                const uint32_t rr = child_ref(r, j);
                if (rr == RT_REF_NONE) continue;
                picked_box[n] = nodes.data() + 16 * (size_t)r + 6 * j;
                picked_ref[n++] = rr;
            }
        }
        uint32_t inner_index[4];
        for (int c = 0; c < n; c++)
            if (!(picked_ref[c] & RT_REF_LEAF)) { inner_index[c] = (uint32_t)wide.size(); wide.emplace_back(); }
        Wide &w = wide[cur.index];
        for (int c = 0; c < 4; c++) {
            if (c < n) {
                memcpy(w.box[c], picked_box[c], 24);
                w.ref[c] = (picked_ref[c] & RT_REF_LEAF) ? picked_ref[c] : inner_index[c];
            } else {
                for (int a = 0; a < 3; a++) { w.box[c][2 * a] = INFINITY; w.box[c][2 * a + 1] = -INFINITY; }
                w.ref[c] = RT_REF_NONE;
            }
        }
        for (int c = n - 1; c >= 0; c--)
            if (!(picked_ref[c] & RT_REF_LEAF)) todo.push_back({picked_ref[c], inner_index[c], cur.level + 1u});
    }
    // the nodes with the largest boxes first
    const size_t count = wide.size();
    std::vector<uint32_t> new_index(count, UINT32_MAX);
    {
        auto area = [](const float *b) {
            const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
            if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
            const float a = dx * dy + dy * dz + dz * dx;
            return std::isfinite(a) ? a : 3.0e38f;
        };
        std::vector<std::pair<float, uint32_t>> heap;
        auto less = [](const std::pair<float, uint32_t> &a, const std::pair<float, uint32_t> &b) {
            return a.first < b.first || (a.first == b.first && a.second > b.second);
        };
        heap.push_back({3.4e38f, 0u});
        uint32_t next = 0;
        while (!heap.empty() && next < top_n) {
            std::pop_heap(heap.begin(), heap.end(), less);
            const uint32_t node = heap.back().second;
            heap.pop_back();
            new_index[node] = next++;
            for (int c = 0; c < 4; c++)
                if (!(wide[node].ref[c] & RT_REF_LEAF)) { heap.push_back({area(wide[node].box[c]), wide[node].ref[c]}); std::push_heap(heap.begin(), heap.end(), less); }
        }
        for (size_t i = 0; i < count; i++)
            if (new_index[i] == UINT32_MAX) new_index[i] = next++;
    }
    std::vector<uint32_t> out(16 * count);
    for (size_t i = 0; i < count; i++) {
        const Wide &w = wide[i];
        uint32_t *o = out.data() + 16 * (size_t)new_index[i];
        for (int a = 0; a < 3; a++)
            for (int pair = 0; pair < 2; pair++) {
                o[4 * a + pair] = (uint32_t)half_toward(w.box[2 * pair][2 * a], false) | ((uint32_t)half_toward(w.box[2 * pair + 1][2 * a], false) << 16);
                o[4 * a + 2 + pair] = (uint32_t)half_toward(w.box[2 * pair][2 * a + 1], true) | ((uint32_t)half_toward(w.box[2 * pair + 1][2 * a + 1], true) << 16);
            }
        for (int c = 0; c < 4; c++) o[12 + c] = (w.ref[c] & RT_REF_LEAF) ? w.ref[c] : new_index[w.ref[c]] * 64u;
    }
    root4 = new_index[0] * 64u;
    return out;
}

struct Builder {
    std::vector<BuildItem> items;
    std::vector<float> nodes;      // 16 floats per node
    std::atomic<uint32_t> max_depth{0};
    std::atomic<bool> failed{false};   // a helper thread of the build ended with an exception
    size_t sweep_below = 0;        // ranges of at most this many items are split by an exact sweep instead of bins
    bool sah = false;
    uint32_t depth_limit = RT_STACK_DEPTH;   // small scenes are kept within 16 levels: the 16-entry-stack kernel holds more waves

    static void put_box(float *dst, const Box3 &b) { for (int a = 0; a < 3; a++) { dst[2 * a] = b.mn[a]; dst[2 * a + 1] = b.mx[a]; } }

    static float half_area(const Box3 &b) {
        float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;      // empty
        float a = dx * dy + dy * dz + dz * dx;
        return std::isfinite(a) ? a : 3.0e38f;
    }

    // Reference rule, RayTracingKit/Hittable.swift:615-620: longest axis of the node's box
    // (ties as AABB.swift:59-66), order by box centre, split at (n+1)/2.  Only the split
    // matters (each half is re-ordered on its own axis one level down), so a selection
    // stands in for the reference's full sort.
    size_t split_median(size_t lo, size_t hi, const Box3 &all) {
        float sx = all.mx[0] - all.mn[0], sy = all.mx[1] - all.mn[1], sz = all.mx[2] - all.mn[2];
        int axis = sx > sy ? (sx > sz ? 0 : 2) : (sy > sz ? 1 : 2);
        size_t mid = lo + (hi - lo + 1) / 2;
        std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                         [axis](const BuildItem &a, const BuildItem &b) { return a.c[axis] < b.c[axis]; });
        return mid;
    }

    // Surface-area split over every position of the centroid order on each axis (small ranges: an exact sweep
    // costs nothing there and the bottom levels of the tree are where most node visits go).
    size_t split_sah_sweep(size_t lo, size_t hi) {
        const size_t n = hi - lo;
        float best_cost = INFINITY;
        int best_axis = -1;
        size_t best_pos = 0;
        std::vector<uint32_t> order(n);
        std::vector<float> right_area(n);
        for (int axis = 0; axis < 3; axis++) {
            for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return items[lo + a].c[axis] < items[lo + b].c[axis]; });
            Box3 acc;
            for (size_t i = n; i-- > 1;) { acc.add(items[lo + order[i]].box); right_area[i] = half_area(acc); }
            Box3 left;
            for (size_t i = 0; i + 1 < n; i++) {
                left.add(items[lo + order[i]].box);
                float cost = half_area(left) * (float)(i + 1) + right_area[i + 1] * (float)(n - i - 1);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_pos = i + 1; }
            }
        }
        if (best_axis < 0 || !std::isfinite(best_cost)) return lo;
        const int axis = best_axis;
        std::stable_sort(items.begin() + lo, items.begin() + hi, [=](const BuildItem &a, const BuildItem &b) { return a.c[axis] < b.c[axis]; });
        return lo + best_pos;
    }

    // Binned surface-area heuristic (16 bins per axis on the box centres): minimises
    // area(left)*count(left) + area(right)*count(right).  Returns lo when no split helps.
    size_t split_sah(size_t lo, size_t hi) {
        if (hi - lo <= sweep_below) return split_sah_sweep(lo, hi);
        constexpr int BINS = 16;
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = lo; i < hi; i++)
            for (int a = 0; a < 3; a++) { cmin[a] = std::min(cmin[a], items[i].c[a]); cmax[a] = std::max(cmax[a], items[i].c[a]); }
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = 0;
        float best_scale = 0.f;
        for (int axis = 0; axis < 3; axis++) {
            float ext = cmax[axis] - cmin[axis];
            if (!(ext > 0.f) || !std::isfinite(ext)) continue;
            float scale = (float)BINS / ext;
            Box3 bb[BINS];
            uint32_t bc[BINS] = {0};
            for (size_t i = lo; i < hi; i++) {
                int b = std::min(BINS - 1, (int)((items[i].c[axis] - cmin[axis]) * scale));
                bb[b].add(items[i].box);
                bc[b]++;
            }
            float right_area[BINS];
            uint32_t right_cnt[BINS];
            Box3 acc;
            uint32_t cnt = 0;
            for (int b = BINS - 1; b > 0; b--) {
                if (bc[b]) acc.add(bb[b]);
                cnt += bc[b];
                right_area[b] = half_area(acc);
                right_cnt[b] = cnt;
            }
            Box3 left;
            uint32_t lcnt = 0;
            for (int b = 0; b < BINS - 1; b++) {
                if (bc[b]) left.add(bb[b]);
                lcnt += bc[b];
                if (lcnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = half_area(left) * (float)lcnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; best_scale = scale; }
            }
        }
        if (best_axis < 0) return lo;
        const int axis = best_axis, bin = best_bin;
        const float c0 = cmin[axis], scale = best_scale;
        auto mid = std::partition(items.begin() + lo, items.begin() + hi, [=](const BuildItem &it) {
            return std::min(BINS - 1, (int)((it.c[axis] - c0) * scale)) <= bin;
        });
        return (size_t)(mid - items.begin());
    }

    // A subtree over n items has exactly n - 1 inner nodes, so the place of every node in pre-order (node, left
    // subtree, right subtree) is known before its subtree is built: the node of a range is `idx`, its left subtree
    // starts at idx + 1 and its right subtree at idx + (mid - lo).  Both halves of a large range can therefore be
    // built by different threads into the same array, and the result is the one a single thread writes.
    static constexpr size_t PAR_MIN_ITEMS = 32768;
    static constexpr uint32_t PAR_LEVELS = 4;      // at most 2^4 ranges in flight

    void note_depth(uint32_t d) {
        uint32_t seen = max_depth.load(std::memory_order_relaxed);
        while (d > seen && !max_depth.compare_exchange_weak(seen, d, std::memory_order_relaxed)) {}
    }

    // Moves the `top_n` nodes with the largest boxes to the head of the array (the root stays node 0), the others keeping
    // their pre-order among themselves.  A child's box lies inside its parent's, so the nodes picked form the top of the
    // tree; a ray visits a node about as often as its box is large, so these are the nodes most often read - the render
    // kernel of a deep tree keeps them in LDS.  Only the numbering changes: every reference is rewritten, the tree and
    // its boxes are the same.
    void hoist_top(size_t top_n) {
        const size_t n = nodes.size() / 16;
        if (n < 2 || top_n < 2) return;
        auto child = [&](size_t node, int k) { uint32_t r; memcpy(&r, nodes.data() + 16 * node + 12 + k, 4); return r; };
        auto child_area = [&](size_t node, int k) {
            const float *b = nodes.data() + 16 * node + 6 * k;
            Box3 box;
            for (int a = 0; a < 3; a++) { box.mn[a] = b[2 * a]; box.mx[a] = b[2 * a + 1]; }
            return half_area(box);
        };
        std::vector<uint32_t> new_index(n, UINT32_MAX);
        std::vector<std::pair<float, uint32_t>> heap;          // (area, old index): largest area first, ties by lower index
        auto less = [](const std::pair<float, uint32_t> &a, const std::pair<float, uint32_t> &b) {
            return a.first < b.first || (a.first == b.first && a.second > b.second);
        };
        heap.push_back({3.4e38f, 0u});
        uint32_t next = 0;
        while (!heap.empty() && next < top_n) {
            std::pop_heap(heap.begin(), heap.end(), less);
            const uint32_t node = heap.back().second;
            heap.pop_back();
            new_index[node] = next++;
            for (int k = 0; k < 2; k++) {
                const uint32_t r = child(node, k);
                if (!(r & RT_REF_LEAF)) { heap.push_back({child_area(node, k), r}); std::push_heap(heap.begin(), heap.end(), less); }
            }
        }
        for (size_t i = 0; i < n; i++)
            if (new_index[i] == UINT32_MAX) new_index[i] = next++;
        std::vector<float> moved(nodes.size());
        for (size_t i = 0; i < n; i++) {
            float *dst = moved.data() + 16 * (size_t)new_index[i];
            memcpy(dst, nodes.data() + 16 * i, 64);
            for (int k = 0; k < 2; k++) {
                uint32_t r;
                memcpy(&r, dst + 12 + k, 4);
                if (!(r & RT_REF_LEAF)) { r = new_index[r]; memcpy(dst + 12 + k, &r, 4); }
            }
        }
        nodes.swap(moved);
    }

    // Builds the whole tree; returns the root's reference.
    uint32_t build_all(Box3 &box) {
        nodes.assign(16 * (items.size() - 1), 0.f);
        const uint32_t root = build(0, items.size(), box, 0, 0);
        if (failed.load()) throw std::runtime_error("a tree-building thread failed");
        return root;
    }

    // Builds the subtree over items[lo, hi) with its root at node `idx`; returns its reference and box.
    uint32_t build(size_t lo, size_t hi, Box3 &box, uint32_t depth, uint32_t idx) {
        if (hi - lo == 1) {
            box = items[lo].box;
            return RT_REF_LEAF | items[lo].ref;
        }
        note_depth(depth + 1);
        Box3 all;
        for (size_t i = lo; i < hi; i++) all.add(items[i].box);
        size_t mid = lo;
        if (sah) {
            // an unbalanced split is only taken while a balanced subtree below it still
            // fits the traversal stack
            uint32_t need = 1;
            while ((1ull << need) < (hi - lo)) need++;
            if (depth + 2 + need < depth_limit) mid = split_sah(lo, hi);
        }
        if (mid == lo || mid == hi) mid = split_median(lo, hi, all);
        Box3 lb, rb;
        uint32_t l, r;
        const uint32_t left_idx = idx + 1, right_idx = idx + (uint32_t)(mid - lo);
        // The helper thread is joined on every way out of this scope (an exception on this thread included), and an
        // exception inside it (std::bad_alloc from a split's vectors) is recorded in `failed` instead of ending the
        // process: build_all() turns the flag into an exception on the calling thread once every thread has ended.
        struct Joiner {
            std::thread t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } other;
        if (depth < PAR_LEVELS && hi - lo >= PAR_MIN_ITEMS) {
            // a process that may start no more threads builds this range on the one it has
            try {
                other.t = std::thread([&] {
                    try { l = build(lo, mid, lb, depth + 1, left_idx); } catch (...) { failed.store(true, std::memory_order_relaxed); }
                });
            } catch (const std::system_error &) {}
        }
        if (!other.t.joinable()) l = build(lo, mid, lb, depth + 1, left_idx);
        r = build(mid, hi, rb, depth + 1, right_idx);
        if (other.t.joinable()) other.t.join();
        float *n = nodes.data() + 16 * (size_t)idx;
        put_box(n, lb);
        put_box(n + 6, rb);
        memcpy(n + 12, &l, 4);
        memcpy(n + 13, &r, 4);
        box = all;
        return idx;
    }
};

int check_material(const RtScene *sc, size_t m) {
    return m < sc->material_kinds.size() ? RT_OK : fail(RT_ERR_INVALID, "material_offset does not name a material added with rt_add_material");
}

int push_shape(RtScene *sc, HostShape &&s) {
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    sc->shapes.push_back(std::move(s));
    return (int)sc->shapes.size() - 1;
}

void set_xf(HostShape &s, const RtTransform &t) {
    for (int c = 0; c < 3; c++) memcpy(s.rot + 3 * c, &t.rotation.columns[c].x, 12);
    memcpy(s.trans, &t.translation.x, 12);
}

// A caller's record may sit at any address (a byte buffer from another language); the
// 16-byte alignment of the simd types is only promised for copies made here.
template <class T> T copy_record(const T *p) {
    T r;
    memcpy((void *)&r, (const void *)p, sizeof(T));
    return r;
}

// Launches go to the scene's / context's device whatever the caller's current device is.
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
        else prev = -1;        // nothing to restore
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

hipStream_t as_stream(void *p) { return (hipStream_t)p; }

// Shaders.metal:30-48, once per call instead of once per thread.
void make_camera(const RtCameraConfig &c, int w, int h, RtDevCamera &out) {
    auto sub = [](const float *a, const float *b, float *o) { for (int i = 0; i < 3; i++) o[i] = a[i] - b[i]; };
    auto norm = [](float *v) { float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (int i = 0; i < 3; i++) v[i] /= l; };
    auto cross = [](const float *a, const float *b, float *o) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    const float pi = 3.14159265358979323846f;
    float half = c.vertical_FOV * pi / 360.f;
    float vh = 2.f * std::tan(half) * c.focus_distance;
    float vw = vh * (float)w / (float)h;
    const float *look_from = &c.look_from.x, *look_at = &c.look_at.x, *up = &c.up.x;
    float wv[3], uv[3], vv[3];
    sub(look_from, look_at, wv);
    norm(wv);
    cross(up, wv, uv);
    norm(uv);
    cross(wv, uv, vv);
    float dr = c.focus_distance * std::tan(c.defocus_angle * pi / 360.f);
    for (int i = 0; i < 3; i++) {
        out.center[i] = look_from[i];
        out.viewport_u[i] = vw * uv[i];
        out.viewport_v[i] = -vh * vv[i];
        out.viewport_center[i] = look_from[i] - c.focus_distance * wv[i];
        out.defocus_u[i] = uv[i] * dr;
        out.defocus_v[i] = vv[i] * dr;
    }
    out.inv_w = 1.f / (float)w;
    out.inv_h = 1.f / (float)h;
    out.defocus = c.defocus_angle > 0.f ? 1u : 0u;
    out.background = c.background;
}

RtDevRender make_render(const RtRenderConfig &rc, uint32_t first_sample, int w, int h) {
    RtDevRender r;
    r.width = (uint32_t)w; r.height = (uint32_t)h;
    r.spp = rc.samples_per_pixel; r.max_depth = rc.max_depth;
    r.first_sample = first_sample;
    r.seed_lo = (uint32_t)rc.rng_seed; r.seed_hi = (uint32_t)(rc.rng_seed >> 32);
    r.animated = 0;
    r.band_index = 0; r.band_count = 1;
    return r;
}

}  // namespace

// ============================================================================ C ABI

const char *rt_last_error(void) { return g_error.c_str(); }
int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_context_create(int device_ordinal, RtContext **out) {
    if (!out) return fail(RT_ERR_INVALID, "out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail_hip(e, "hipGetDeviceCount");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(RT_ERR_INVALID, "no such device");
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    // the address of this device's copy of the library's scratch array (rt_render_static)
    e = rt_static_scratch_init(device_ordinal);
    if (e != hipSuccess) return fail_hip(e, "hipGetSymbolAddress of the static scratch");
    RtContext *c = new RtContext();
    c->device = device_ordinal;
    (void)hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device_ordinal);
    *out = c;
    return RT_OK;
}

void rt_context_destroy(RtContext *c) { delete c; }

int rt_scene_create(RtContext *ctx, RtScene **out) {
    if (!out) return fail(RT_ERR_INVALID, "null argument");
    // A null context gives a description-only scene: everything up to rt_scene_commit works
    // (and is validated) without a device.
    RtScene *s = new RtScene();
    s->ctx = ctx;
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(RtScene *s) { delete s; }

int rt_add_material(RtScene *sc, const void *record, size_t size) {
    if (!sc || !record || size < 4) return fail(RT_ERR_INVALID, "bad material record");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    uint32_t kind;
    memcpy(&kind, record, 4);
    float rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t tex = 0;
    auto want = [&](size_t n) { return size == n ? RT_OK : fail(RT_ERR_INVALID, "material record size does not match its kind"); };
    // Fields are read at their offsets with memcpy: the record may sit at any address.
    auto f32_at = [&](size_t off) { float v; memcpy(&v, (const char *)record + off, 4); return v; };
    auto albedo_at = [&](size_t off) { rec[4] = f32_at(off); rec[5] = f32_at(off + 4); rec[6] = f32_at(off + 8); };
    auto image_at = [&](size_t off) {
        uint64_t id;
        memcpy(&id, (const char *)record + off, 8);
        tex = (uint32_t)id;
        return id < sc->textures.size() ? RT_OK : fail(RT_ERR_INVALID, "texture_ptr does not name a texture added with rt_add_texture_rgb8");
    };
    auto add_perlin = [&](size_t off) {
        const char *p = (const char *)record + off;
        RtDevPerlin d;
        for (int i = 0; i < 2; i++) memcpy(d.colors[i], p + offsetof(RtPerlinNoiseTexture, colors) + 16 * i, 12);
        memcpy(&d.frequency, p + offsetof(RtPerlinNoiseTexture, frequency), 4);
        memcpy(&d.turbulence, p + offsetof(RtPerlinNoiseTexture, turbulence), 4);
        for (int i = 0; i < RT_PERLIN_TABLE_SIZE; i++) memcpy(d.vectors[i], p + offsetof(RtPerlinNoiseTexture, vectors) + 16 * i, 12);
        memcpy(d.permutations, p + offsetof(RtPerlinNoiseTexture, permutations), sizeof(d.permutations));
        sc->perlin.push_back(d);
        return (uint32_t)sc->perlin.size() - 1;
    };
    switch (kind) {
    case rt_material_kind_lambertian_colored:
    case rt_material_kind_emissive_colored:
    case rt_material_kind_isotropic_colored:
        if (want(sizeof(RtColoredLambertianMaterial)) < 0) return RT_ERR_INVALID;
        albedo_at(offsetof(RtColoredLambertianMaterial, albedo));
        break;
    case rt_material_kind_lambertian_textured:
        if (want(sizeof(RtTexturedLambertianMaterial)) < 0) return RT_ERR_INVALID;
        if (image_at(offsetof(RtTexturedLambertianMaterial, albedo)) < 0) return RT_ERR_INVALID;
        break;
    case rt_material_kind_lambertian_perlin_noise:
        if (want(sizeof(RtPerlinNoiseLambertianMaterial)) < 0) return RT_ERR_INVALID;
        tex = add_perlin(offsetof(RtPerlinNoiseLambertianMaterial, albedo));
        break;
    case rt_material_kind_metal_colored:
        if (want(sizeof(RtColoredMetalMaterial)) < 0) return RT_ERR_INVALID;
        albedo_at(offsetof(RtColoredMetalMaterial, albedo));
        rec[2] = f32_at(offsetof(RtColoredMetalMaterial, fuzz));
        break;
    case rt_material_kind_metal_textured:
        if (want(sizeof(RtTexturedMetalMaterial)) < 0) return RT_ERR_INVALID;
        if (image_at(offsetof(RtTexturedMetalMaterial, albedo)) < 0) return RT_ERR_INVALID;
        rec[2] = f32_at(offsetof(RtTexturedMetalMaterial, fuzz));
        break;
    case rt_material_kind_metal_perlin_noise:
        if (want(sizeof(RtPerlinNoiseMetalMaterial)) < 0) return RT_ERR_INVALID;
        tex = add_perlin(offsetof(RtPerlinNoiseMetalMaterial, albedo));
        rec[2] = f32_at(offsetof(RtPerlinNoiseMetalMaterial, fuzz));
        break;
    case rt_material_kind_lambertian_kit_texture:
    case rt_material_kind_emissive_kit_texture:
        if (want(sizeof(RtKitTexturedMaterial)) < 0) return RT_ERR_INVALID;
        memcpy(&tex, (const char *)record + offsetof(RtKitTexturedMaterial, texture), 4);
        if (tex >= sc->kit_textures.size()) return fail(RT_ERR_INVALID, "texture does not name a texture added with rt_add_kit_texture");
        break;
    case rt_material_kind_dielectric:
        if (want(sizeof(RtDielectricMaterial)) < 0) return RT_ERR_INVALID;
        rec[3] = f32_at(offsetof(RtDielectricMaterial, refraction_index));
        break;
    default:
        return fail(RT_ERR_INVALID, "unknown material kind");
    }
    memcpy(&rec[0], &kind, 4);
    memcpy(&rec[1], &tex, 4);
    sc->materials.insert(sc->materials.end(), rec, rec + 8);
    sc->material_kinds.push_back(kind);
    return (int)sc->material_kinds.size() - 1;
}

int rt_add_kit_perlin(RtScene *sc, const RtKitPerlin *p) {
    if (!sc || !p) return fail(RT_ERR_INVALID, "null argument");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    RtDevKitPerlin d;
    memcpy(d.vectors, p->vectors, sizeof(d.vectors));
    memcpy(d.perm[0], p->perm_x, 256);
    memcpy(d.perm[1], p->perm_y, 256);
    memcpy(d.perm[2], p->perm_z, 256);
    sc->kit_perlin.push_back(d);
    return (int)sc->kit_perlin.size() - 1;
}

int rt_add_kit_texture(RtScene *sc, const RtKitTexture *t) {
    if (!sc || !t) return fail(RT_ERR_INVALID, "null argument");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    RtDevKitTexture d;
    memset(&d, 0, sizeof(d));
    d.kind = t->kind;
    const int n = (int)sc->kit_textures.size();
    switch (t->kind) {
    case RT_KIT_TEXTURE_SOLID:
        memcpy(d.color, t->color, 12);
        break;
    case RT_KIT_TEXTURE_CHECKER:
        // children come first, so chains of checkers end
        if (t->even < 0 || t->even >= n || t->odd < 0 || t->odd >= n) return fail(RT_ERR_INVALID, "checker children must be textures added before it");
        if (!(t->scale != 0.f)) return fail(RT_ERR_INVALID, "checker scale must not be zero");
        d.scale = 1.f / t->scale;        // Texture.swift:33
        d.even = t->even; d.odd = t->odd;
        break;
    case RT_KIT_TEXTURE_IMAGE: {
        if (t->image < 0 || t->image >= (int)sc->textures.size()) return fail(RT_ERR_INVALID, "image does not name a texture added with rt_add_texture_rgb8");
        if (t->width == 0 || t->height == 0) return fail(RT_ERR_INVALID, "image window must not be empty");
        const HostTexture &img = sc->textures[(size_t)t->image];
        d.image = (uint32_t)t->image;
        d.origin_x = t->origin_x; d.origin_y = t->origin_y;
        d.width = std::abs(t->width); d.height = std::abs(t->height);      // Texture.swift:72-75
        d.step_x = t->width < 0 ? -1 : 1; d.step_y = t->height < 0 ? -1 : 1;
        d.wrap_x = t->wrap_x ? 1u : 0u; d.wrap_y = t->wrap_y ? 1u : 0u;
        long x_end = (long)d.origin_x + (long)(d.width - 1) * d.step_x, y_end = (long)d.origin_y + (long)(d.height - 1) * d.step_y;
        if (d.origin_x < 0 || d.origin_x >= img.w || x_end < 0 || x_end >= img.w || d.origin_y < 0 || d.origin_y >= img.h ||
            y_end < 0 || y_end >= img.h)
            return fail(RT_ERR_INVALID, "image window reaches outside the image");   // the reference asserts, Image.swift:65-68
        break;
    }
    case RT_KIT_TEXTURE_NOISE:
    case RT_KIT_TEXTURE_MARBLE:
        if (t->perlin < 0 || t->perlin >= (int)sc->kit_perlin.size()) return fail(RT_ERR_INVALID, "perlin does not name tables added with rt_add_kit_perlin");
        d.perlin = (uint32_t)t->perlin;
        d.scale = t->scale;
        break;
    default:
        return fail(RT_ERR_INVALID, "unknown texture kind");
    }
    sc->kit_textures.push_back(d);
    return n;
}

int rt_add_texture_rgb8(RtScene *sc, int width, int height, const uint8_t *rgb) {
    if (!sc || !rgb || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad texture");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    HostTexture t;
    t.w = width; t.h = height;
    t.rgba.resize((size_t)width * height);
    for (size_t i = 0; i < t.rgba.size(); i++)
        t.rgba[i] = (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16) | 0xFF000000u;
    sc->textures.push_back(std::move(t));
    return (int)sc->textures.size() - 1;
}

int rt_shape_sphere(RtScene *sc, const RtSphere *rec) {
    if (!sc || !rec) return fail(RT_ERR_INVALID, "null argument");
    const RtSphere p = copy_record(rec);
    if (!(p.radius > 0.f)) return fail(RT_ERR_INVALID, "sphere radius must be positive");
    if (check_material(sc, p.material_offset) < 0) return RT_ERR_INVALID;
    HostShape s;
    s.kind = RT_KIND_SPHERE;
    set_xf(s, p.transform);
    s.radius = p.radius;
    s.mat[0] = (uint32_t)p.material_offset;
    return push_shape(sc, std::move(s));
}

int rt_shape_cylinder(RtScene *sc, const RtCylinder *rec) {
    if (!sc || !rec) return fail(RT_ERR_INVALID, "null argument");
    const RtCylinder p = copy_record(rec);
    if (!(p.radius >= 0.f) || !(p.height >= 0.f)) return fail(RT_ERR_INVALID, "cylinder radius/height must not be negative");
    if (check_material(sc, p.bottom_material_offset) < 0 || check_material(sc, p.top_material_offset) < 0 ||
        check_material(sc, p.side_material_offset) < 0)
        return RT_ERR_INVALID;
    HostShape s;
    s.kind = RT_KIND_CYLINDER;
    set_xf(s, p.transform);
    s.radius = p.radius; s.height = p.height;
    s.mat[0] = (uint32_t)p.bottom_material_offset;
    s.mat[1] = (uint32_t)p.top_material_offset;
    s.mat[2] = (uint32_t)p.side_material_offset;
    return push_shape(sc, std::move(s));
}

int rt_shape_cuboid(RtScene *sc, const RtCuboid *rec) {
    if (!sc || !rec) return fail(RT_ERR_INVALID, "null argument");
    const RtCuboid p = copy_record(rec);
    HostShape s;
    s.kind = RT_KIND_CUBOID;
    set_xf(s, p.transform);
    memcpy(s.size, &p.size.x, 12);
    for (int i = 0; i < 6; i++) {
        if (check_material(sc, p.material_offset[i]) < 0) return RT_ERR_INVALID;
        s.mat[i] = (uint32_t)p.material_offset[i];
    }
    return push_shape(sc, std::move(s));
}

int rt_shape_quad(RtScene *sc, const RtQuad *rec) {
    if (!sc || !rec) return fail(RT_ERR_INVALID, "null argument");
    const RtQuad p = copy_record(rec);
    if (check_material(sc, p.material_offset) < 0) return RT_ERR_INVALID;
    HostShape s;
    s.kind = RT_KIND_QUAD;
    memcpy(s.quad + 0, &p.origin.x, 12);
    memcpy(s.quad + 3, &p.u.x, 12);
    memcpy(s.quad + 6, &p.v.x, 12);
    memcpy(s.quad + 9, &p.w.x, 12);
    memcpy(s.quad + 12, &p.normal.x, 12);
    s.quad[15] = p.d;
    s.mat[0] = (uint32_t)p.material_offset;
    return push_shape(sc, std::move(s));
}

int rt_shape_csg(RtScene *sc, int op, int n_children, const int *children) {
    if (!sc || !children || n_children < 1) return fail(RT_ERR_INVALID, "a CSG shape needs children");
    if (op < RT_CSG_UNION || op > RT_CSG_SUBTRACT) return fail(RT_ERR_INVALID, "unknown CSG operation");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    HostShape s;
    s.kind = RT_KIND_CSG;
    s.op = op;
    for (int i = 0; i < n_children; i++) {
        int c = children[i];
        if (c < 0 || c >= (int)sc->shapes.size()) return fail(RT_ERR_INVALID, "bad child shape id");
        if (sc->shapes[c].kind == RT_KIND_CDV)
            return fail(RT_ERR_INVALID, "a constant-density volume cannot be a CSG operand (the reference's Composition "
                                        "builds its children without a generator, RenderableImpl.h:645-654)");
        if (sc->shapes[c].kind == RT_KIND_MOTION)
            return fail(RT_ERR_INVALID, "a moving shape cannot be a CSG operand: wrap the whole CSG shape in the motion instead");
        if (sc->shapes[c].parent >= 0 || sc->shapes[c].object >= 0) return fail(RT_ERR_INVALID, "shape is already in use");
        s.children.push_back(c);
    }
    int id = (int)sc->shapes.size();
    std::vector<int> leaves;
    std::vector<bool> flips;
    sc->shapes.push_back(s);
    collect_leaves(*sc, id, false, leaves, flips);
    if (leaves.size() > RT_CSG_MAX_LEAVES) {
        sc->shapes.pop_back();
        return fail(RT_ERR_LIMIT, "more than RT_CSG_MAX_LEAVES primitives under one CSG shape");
    }
    for (int c : s.children) sc->shapes[c].parent = id;
    return id;
}

int rt_shape_cdv(RtScene *sc, int base, float density) {
    if (!sc || base < 0 || base >= (int)sc->shapes.size()) return fail(RT_ERR_INVALID, "bad base shape id");
    if (!(density > 0.f)) return fail(RT_ERR_INVALID, "density must be positive");
    if (sc->shapes[base].kind == RT_KIND_CDV) return fail(RT_ERR_INVALID, "a volume cannot wrap a volume");
    if (sc->shapes[base].kind == RT_KIND_MOTION) return fail(RT_ERR_INVALID, "a volume cannot wrap a moving shape");
    if (sc->shapes[base].parent >= 0 || sc->shapes[base].object >= 0) return fail(RT_ERR_INVALID, "shape is already in use");
    HostShape s;
    s.kind = RT_KIND_CDV;
    s.base = base;
    s.density = density;
    int id = push_shape(sc, std::move(s));
    if (id >= 0) sc->shapes[base].parent = id;
    return id;
}

int rt_shape_motion(RtScene *sc, int base, const RtMotion *m) {
    if (!sc || !m || base < 0 || base >= (int)sc->shapes.size()) return fail(RT_ERR_INVALID, "bad base shape id");
    const uint32_t bk = sc->shapes[base].kind;
    if (bk == RT_KIND_CDV || bk == RT_KIND_MOTION)
        return fail(RT_ERR_INVALID, "motion blur takes a primitive or a CSG shape");
    if (sc->shapes[base].parent >= 0 || sc->shapes[base].object >= 0) return fail(RT_ERR_INVALID, "shape is already in use");
    double len = 0;
    for (int i = 0; i < 4; i++) len += (double)m->rotation[i] * m->rotation[i];
    if (!(std::fabs(len - 1.0) < 1e-4) || !(m->rotation[0] >= 0.f))
        return fail(RT_ERR_INVALID, "rotation must be a unit quaternion (w, x, y, z) with w >= 0");
    HostShape s;
    s.kind = RT_KIND_MOTION;
    s.base = base;
    memcpy(s.m_quat, m->rotation, 16);
    memcpy(s.m_trans, m->translation, 12);
    int id = push_shape(sc, std::move(s));
    if (id >= 0) sc->shapes[base].parent = id;
    return id;
}

int rt_shape_bounds(const RtScene *sc, int shape, float out_min_max[6]) {
    if (!sc || !out_min_max || shape < 0 || shape >= (int)sc->shapes.size()) return fail(RT_ERR_INVALID, "bad shape id");
    Box3 b = shape_box(*sc, shape);
    for (int a = 0; a < 3; a++) { out_min_max[a] = b.mn[a]; out_min_max[3 + a] = b.mx[a]; }
    return RT_OK;
}

int rt_scene_set_time(RtScene *sc, float time) {
    if (!sc) return fail(RT_ERR_INVALID, "null scene");
    if (!(time >= 0.f && time <= 1.f)) return fail(RT_ERR_INVALID, "time must lie in [0, 1]");
    sc->query_time = time;
    return RT_OK;
}

int rt_add_object(RtScene *sc, int shape) {
    if (!sc || shape < 0 || shape >= (int)sc->shapes.size()) return fail(RT_ERR_INVALID, "bad shape id");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    if (sc->shapes[shape].parent >= 0 || sc->shapes[shape].object >= 0) return fail(RT_ERR_INVALID, "shape is already in use");
    if (!sc->bulk_spheres.empty()) return fail(RT_ERR_STATE, "rt_add_sphere_objects must be the last objects added");
    sc->objects.push_back(shape);
    sc->shapes[shape].object = (int)sc->objects.size() - 1;
    return sc->shapes[shape].object;
}

int rt_add_sphere_objects(RtScene *sc, int n, const float *center_radius, const uint32_t *material_offsets) {
    if (!sc || n < 0 || (n > 0 && (!center_radius || !material_offsets))) return fail(RT_ERR_INVALID, "bad argument");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    for (int i = 0; i < n; i++) {
        if (check_material(sc, material_offsets[i]) < 0) return RT_ERR_INVALID;
        if (!(center_radius[4 * (size_t)i + 3] > 0.f)) return fail(RT_ERR_INVALID, "sphere radius must be positive");
    }
    int first = (int)sc->objects.size() + (int)sc->bulk_mats.size();
    sc->bulk_spheres.insert(sc->bulk_spheres.end(), center_radius, center_radius + 4 * (size_t)n);
    sc->bulk_mats.insert(sc->bulk_mats.end(), material_offsets, material_offsets + n);
    return first;
}

static int scene_commit(RtScene *sc);

// No exception crosses the C boundary: running out of host memory while flattening or building the tree (on this
// thread or on one of the build's helper threads) comes back as an error code like everything else.
int rt_scene_commit(RtScene *sc) {
    try {
        return scene_commit(sc);
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_LIMIT, "out of host memory while committing the scene");
    } catch (const std::exception &e) {
        return fail(RT_ERR_STATE, (std::string("scene commit failed: ") + e.what()).c_str());
    }
}

static int scene_commit(RtScene *sc) {
    if (!sc) return fail(RT_ERR_INVALID, "null scene");
    if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
    if (!sc->ctx) return fail(RT_ERR_STATE, "scene was created without a device context");
    DeviceScope on_device(sc->ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    sc->stats = RtSceneStats{};

    // ---- device slots for every shape (objects or not: rt_intervals may ask for any)
    std::vector<float> spheres, sphere_rot, xprims;
    std::vector<uint32_t> sphere_mat, sphere_obj, xprim_mat, xprim_obj, csg_obj, cdv_obj;
    std::vector<RtDevCsg> csg;
    std::vector<uint32_t> csg_pool;
    uint32_t n_wide_csg = 0;
    std::vector<RtDevCdv> cdv;
    bool any_rot = false;
    auto owner_object = [&](int id) {
        int cur = id;
        while (sc->shapes[cur].parent >= 0) cur = sc->shapes[cur].parent;
        return sc->shapes[cur].object >= 0 ? (uint32_t)sc->shapes[cur].object : 0u;
    };
    for (size_t id = 0; id < sc->shapes.size(); id++) {
        HostShape &s = sc->shapes[id];
        uint32_t obj = owner_object((int)id);
        if (s.kind == RT_KIND_SPHERE) {
            s.ref = rt_make_ref(RT_KIND_SPHERE, (uint32_t)(spheres.size() / 4));
            spheres.insert(spheres.end(), {s.trans[0], s.trans[1], s.trans[2], s.radius});
            const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (memcmp(s.rot, ident, sizeof(ident)) != 0) any_rot = true;
            // rows of transpose(R) = columns of R, padded to float4
            for (int c = 0; c < 3; c++) sphere_rot.insert(sphere_rot.end(), {s.rot[3 * c], s.rot[3 * c + 1], s.rot[3 * c + 2], 0.f});
            sphere_mat.push_back(s.mat[0]);
            sphere_obj.push_back(obj);
        } else if (s.kind <= RT_KIND_QUAD) {
            s.ref = rt_make_ref(s.kind, (uint32_t)(xprims.size() / 16));
            float rec[16] = {0};
            if (s.kind == RT_KIND_QUAD) {
                // The planar coordinates of RenderableImpl.h:418-426, alpha = w . (q x v) and beta = w . (u x q)
                // for q = hit point - origin, are the triple products q . (v x w) and q . (w x u): the two
                // cross products are taken here once (in double), the kernels evaluate two dot products.
                memcpy(rec, s.quad, sizeof(rec));
                const float *v = s.quad + 6, *u = s.quad + 3, *w = s.quad + 9;
                const double A[3] = {(double)v[1] * w[2] - (double)v[2] * w[1], (double)v[2] * w[0] - (double)v[0] * w[2], (double)v[0] * w[1] - (double)v[1] * w[0]};
                const double B[3] = {(double)w[1] * u[2] - (double)w[2] * u[1], (double)w[2] * u[0] - (double)w[0] * u[2], (double)w[0] * u[1] - (double)w[1] * u[0]};
                for (int a = 0; a < 3; a++) { rec[3 + a] = (float)A[a]; rec[6 + a] = (float)B[a]; rec[9 + a] = 0.f; }
            } else {
                memcpy(rec, s.rot, 36);
                memcpy(rec + 9, s.trans, 12);
                if (s.kind == RT_KIND_CYLINDER) { rec[12] = s.radius; rec[13] = s.height; }
                else { rec[12] = s.size[0]; rec[13] = s.size[1]; rec[14] = s.size[2]; }
            }
            xprims.insert(xprims.end(), rec, rec + 16);
            xprim_mat.insert(xprim_mat.end(), s.mat, s.mat + 6);
            xprim_obj.push_back(obj);
        }
    }
    for (size_t id = 0; id < sc->shapes.size(); id++) {
        HostShape &s = sc->shapes[id];
        if (s.kind != RT_KIND_CSG) continue;
        std::vector<int> leaves;
        std::vector<bool> flips;
        collect_leaves(*sc, (int)id, false, leaves, flips);
        RtDevCsg rec;
        memset(&rec, 0, sizeof(rec));
        rec.n_leaves = (uint32_t)leaves.size();
        if (leaves.size() > RT_CSG_LEAVES) {
            // a wide shape: leaves and the expression as a postfix program go to the pool
            uint64_t flip_mask = 0;
            for (size_t i = 0; i < leaves.size(); i++)
                if (flips[i]) flip_mask |= 1ull << i;
            std::vector<uint32_t> tokens;
            int next = 0;
            csg_tokens(*sc, (int)id, next, tokens);
            rec.leaf[0] = (uint32_t)csg_pool.size();
            csg_pool.push_back((uint32_t)flip_mask);
            csg_pool.push_back((uint32_t)(flip_mask >> 32));
            csg_pool.push_back((uint32_t)tokens.size());
            for (int leaf : leaves) csg_pool.push_back(sc->shapes[leaf].ref);
            csg_pool.insert(csg_pool.end(), tokens.begin(), tokens.end());
            n_wide_csg++;
        } else {
            for (size_t i = 0; i < leaves.size(); i++) {
                rec.leaf[i] = sc->shapes[leaves[i]].ref;
                if (flips[i]) rec.flip_mask |= 1u << i;
            }
            for (uint32_t mask = 0; mask < (1u << leaves.size()); mask++) {
                int next = 0;
                if (eval_csg(*sc, (int)id, mask, next)) rec.truth[mask >> 5] |= 1u << (mask & 31u);
            }
        }
        s.ref = rt_make_ref(RT_KIND_CSG, (uint32_t)csg.size());
        csg.push_back(rec);
        csg_obj.push_back(owner_object((int)id));
    }
    for (size_t id = 0; id < sc->shapes.size(); id++) {
        HostShape &s = sc->shapes[id];
        if (s.kind != RT_KIND_CDV) continue;
        RtDevCdv rec;
        rec.base = sc->shapes[s.base].ref;
        rec.neg_inv_density = -1.f / s.density;
        s.ref = rt_make_ref(RT_KIND_CDV, (uint32_t)cdv.size());
        cdv.push_back(rec);
        cdv_obj.push_back(owner_object((int)id));
    }
    std::vector<RtDevMotion> motion;
    std::vector<uint32_t> motion_obj;
    for (size_t id = 0; id < sc->shapes.size(); id++) {
        HostShape &s = sc->shapes[id];
        if (s.kind != RT_KIND_MOTION) continue;
        RtDevMotion rec;
        memset(&rec, 0, sizeof(rec));
        rec.base = sc->shapes[s.base].ref;
        // Quaternion.pow, Transform.swift:243-250: (cos(a t), v sin(a t) / sin(a)), a = acos(w)
        double w = std::min(1.0, (double)s.m_quat[0]);
        double a = std::acos(w), sin_a = std::sqrt(1 - w * w);
        rec.qx = s.m_quat[1]; rec.qy = s.m_quat[2]; rec.qz = s.m_quat[3];
        rec.half_angle = (float)a;
        rec.inv_sin = (w == 1.0 || !std::isfinite(1.0 / sin_a)) ? 0.f : (float)(1.0 / sin_a);
        if (rec.inv_sin == 0.f) rec.half_angle = 0.f;
        rec.tx = s.m_trans[0]; rec.ty = s.m_trans[1]; rec.tz = s.m_trans[2];
        s.ref = rt_make_ref(RT_KIND_MOTION, (uint32_t)motion.size());
        motion.push_back(rec);
        motion_obj.push_back(owner_object((int)id));
    }
    // bulk spheres come after the described ones
    uint32_t first_bulk_object = (uint32_t)sc->objects.size();
    uint32_t first_bulk_sphere = (uint32_t)(spheres.size() / 4);
    size_t n_bulk = sc->bulk_mats.size();
    if (n_bulk) {
        spheres.insert(spheres.end(), sc->bulk_spheres.begin(), sc->bulk_spheres.end());
        sphere_mat.insert(sphere_mat.end(), sc->bulk_mats.begin(), sc->bulk_mats.end());
        for (size_t i = 0; i < n_bulk; i++) sphere_obj.push_back(first_bulk_object + (uint32_t)i);
        if (any_rot) sphere_rot.resize(spheres.size() * 3, 0.f);
        if (any_rot)
            for (size_t i = first_bulk_sphere; i < spheres.size() / 4; i++) {
                float *r = sphere_rot.data() + 12 * i;
                r[0] = 1; r[5] = 1; r[10] = 1;
            }
    }
    if (spheres.size() / 4 > RT_INDEX_MASK || xprims.size() / 16 > RT_INDEX_MASK)
        return fail(RT_ERR_LIMIT, "too many primitives");

    // ---- BVH over the objects
    Builder bld;
    bld.sah = sc->bvh_builder == RT_BVH_SAH;
    if (sc->objects.size() + sc->bulk_mats.size() <= 4096) bld.depth_limit = 16;
    bld.items.reserve(sc->objects.size() + n_bulk);
    // scenes of up to 65536 objects: ranges of up to 4096 items are split by the exact sweep (+2.7 % on the random-spheres
    // workload over 16 bins at the same number of node visits; on a million spheres it doubles the build time for +0.3 %)
    bld.sweep_below = sc->objects.size() + n_bulk <= 65536 ? 4096 : 0;
    uint32_t feat = 0;
    for (size_t o = 0; o < sc->objects.size(); o++) {
        const HostShape &s = sc->shapes[sc->objects[o]];
        BuildItem it;
        it.ref = s.ref;
        it.box = shape_box(*sc, sc->objects[o]);
        pad_box(it.box);
        if (s.kind == RT_KIND_MOTION) {
            feat |= RT_FEAT_MOTION;
            if (sc->shapes[s.base].kind != RT_KIND_SPHERE) feat |= RT_FEAT_XPRIM;
        } else if (s.kind != RT_KIND_SPHERE) {
            feat |= RT_FEAT_XPRIM;
        }
        bld.items.push_back(it);
    }
    for (size_t i = 0; i < n_bulk; i++) {
        const float *g = sc->bulk_spheres.data() + 4 * i;
        BuildItem it;
        it.ref = rt_make_ref(RT_KIND_SPHERE, first_bulk_sphere + (uint32_t)i);
        for (int a = 0; a < 3; a++) { it.box.mn[a] = g[a] - g[3]; it.box.mx[a] = g[a] + g[3]; }
        pad_box(it.box);
        bld.items.push_back(it);
    }
    for (auto &it : bld.items)
        for (int a = 0; a < 3; a++) {
            float c = 0.5f * (it.box.mn[a] + it.box.mx[a]);
            it.c[a] = std::isfinite(c) ? c : 0.f;
        }
    for (uint32_t k : sc->material_kinds)
        if (k == 2 || k == 3 || k == 5 || k == 6 || k >= 10) feat |= RT_FEAT_TEXTURE;

    uint32_t root = RT_REF_NONE;
    if (bld.items.size() == 1) {
        // a single object: one node whose right child is absent
        bld.nodes.assign(16, 0.f);
        Builder::put_box(bld.nodes.data(), bld.items[0].box);
        Box3 none;
        Builder::put_box(bld.nodes.data() + 6, none);
        uint32_t l = RT_REF_LEAF | bld.items[0].ref, r = RT_REF_NONE;
        memcpy(bld.nodes.data() + 12, &l, 4);
        memcpy(bld.nodes.data() + 13, &r, 4);
        root = 0;
        bld.max_depth = 1;
    } else if (bld.items.size() > 1) {
        Box3 all;
        root = bld.build_all(all);
        // a tree deeper than the 16 stack entries the spheres kernel keeps per lane is walked by the kernel that holds the
        // head of the node array in LDS
        if (bld.max_depth > 16) bld.hoist_top(RT_TOP_NODES_MAX);
    }
    if (bld.max_depth > RT_STACK_DEPTH) return fail(RT_ERR_LIMIT, "BVH deeper than the traversal stack");

    // ---- textures
    std::vector<RtDevTexture> tex_desc;
    std::vector<uint32_t> texels;
    for (const HostTexture &t : sc->textures) {
        RtDevTexture d;
        d.offset = (uint32_t)texels.size();
        d.width = (uint32_t)t.w; d.height = (uint32_t)t.h; d.pad = 0;
        tex_desc.push_back(d);
        texels.insert(texels.end(), t.rgba.begin(), t.rgba.end());
    }

    // ---- upload
    RtDevScene &d = sc->dev;
    memset(&d, 0, sizeof(d));
    if (!any_rot) sphere_rot.clear();
    int rc = RT_OK;
    if ((rc = upload(sc, bld.nodes, &d.nodes)) < 0) return rc;
    {
        // halves hold box planes up to 65504; beyond that (or on request) the render kernel
        // reads the float nodes
        float reach = 0.f;
        for (size_t i = 0; i < bld.nodes.size(); i += 16)
            for (int k = 0; k < 12; k++)
                if (std::isfinite(bld.nodes[i + k])) reach = std::max(reach, std::fabs(bld.nodes[i + k]));
        sc->use_nodes16 = sc->node_format == RT_NODES_F16 || (sc->node_format == RT_NODES_AUTO && reach <= 60000.f);
        if (bld.nodes.size() / 16 >= (1u << 26)) sc->use_nodes16 = false;      // byte offsets of that form must stay below the leaf bit
        if (sc->use_nodes16) {
            std::vector<uint32_t> n16 = pack_nodes16(bld.nodes);
            if ((rc = upload(sc, n16, &d.nodes16)) < 0) return rc;
            // a deep tree of plain spheres is walked in its four-children form; its stack holds at most three entries
            // per level
            if ((feat & 7u) == 0 && bld.max_depth > 16 && !(root & RT_REF_LEAF)) {
                uint32_t root4 = RT_REF_NONE, depth4 = 0;
                std::vector<uint32_t> n4 = pack_nodes4(bld.nodes, root, RT_TOP_NODES_MAX / 2, root4, depth4);
                if (getenv("RT_SHOW_OCCUPANCY")) fprintf(stderr, "four-children form: %zu nodes, %u levels (binary: %zu nodes, %u levels)\n", n4.size() / 16, depth4, bld.nodes.size() / 16, (unsigned)bld.max_depth);
                if (3u * depth4 <= RT_WIDE_STACK_DEPTH && n4.size() / 16 < (1u << 25)) {
                    if ((rc = upload(sc, n4, &d.nodes4)) < 0) return rc;
                    d.root4 = root4;
                    d.n_nodes4 = (uint32_t)(n4.size() / 16);
                }
            }
        }
    }
    if ((rc = upload(sc, spheres, &d.spheres)) < 0) return rc;
    if ((rc = upload(sc, sphere_rot, &d.sphere_rot)) < 0) return rc;
    if ((rc = upload(sc, sphere_mat, &d.sphere_mat)) < 0) return rc;
    {
        std::vector<float> mrec(sphere_mat.size() * 8);
        for (size_t i = 0; i < sphere_mat.size(); i++) memcpy(mrec.data() + 8 * i, sc->materials.data() + 8 * (size_t)sphere_mat[i], 32);
        if ((rc = upload(sc, mrec, &d.sphere_mrec)) < 0) return rc;
    }
    if ((rc = upload(sc, sphere_obj, &d.sphere_obj)) < 0) return rc;
    if ((rc = upload(sc, xprims, &d.xprims)) < 0) return rc;
    if ((rc = upload(sc, xprim_mat, &d.xprim_mat)) < 0) return rc;
    if ((rc = upload(sc, xprim_obj, &d.xprim_obj)) < 0) return rc;
    if ((rc = upload(sc, csg, &d.csg)) < 0) return rc;
    if ((rc = upload(sc, csg_obj, &d.csg_obj)) < 0) return rc;
    if ((rc = upload(sc, csg_pool, &d.csg_pool)) < 0) return rc;
    d.n_wide_csg = n_wide_csg;
    if (n_wide_csg) feat |= RT_FEAT_WIDE_CSG;
    if ((rc = upload(sc, cdv, &d.cdv)) < 0) return rc;
    if ((rc = upload(sc, cdv_obj, &d.cdv_obj)) < 0) return rc;
    if ((rc = upload(sc, motion, &d.motion)) < 0) return rc;
    if ((rc = upload(sc, motion_obj, &d.motion_obj)) < 0) return rc;
    if ((rc = upload(sc, sc->materials, &d.materials)) < 0) return rc;
    if ((rc = upload(sc, tex_desc, &d.textures)) < 0) return rc;
    if ((rc = upload(sc, texels, &d.texels)) < 0) return rc;
    if ((rc = upload(sc, sc->perlin, &d.perlin)) < 0) return rc;
    if ((rc = upload(sc, sc->kit_textures, &d.kit_textures)) < 0) return rc;
    if ((rc = upload(sc, sc->kit_perlin, &d.kit_perlin)) < 0) return rc;
    if (!bld.items.empty() && bld.items.size() <= RT_LINEAR_MAX) {
        // Array<any Hittable>.hit of RayTracingKit/Hittable.swift:662-671: the plain list of objects
        std::vector<uint32_t> list(8 * bld.items.size(), 0u);
        for (size_t i = 0; i < bld.items.size(); i++) {
            const BuildItem &it = bld.items[i];
            list[8 * i] = it.ref;
            for (int a = 0; a < 3; a++) {
                memcpy(&list[8 * i + 1 + 2 * a], &it.box.mn[a], 4);
                memcpy(&list[8 * i + 2 + 2 * a], &it.box.mx[a], 4);
            }
        }
        if ((rc = upload(sc, list, &d.object_list)) < 0) return rc;
        d.n_listed = (uint32_t)bld.items.size();
    }
    d.root = root;
    d.n_nodes = (uint32_t)(bld.nodes.size() / 16);
    d.kit_convention = sc->convention == RT_CONVENTION_KIT ? 1u : 0u;
    sc->feat = feat;

    RtSceneStats &st = sc->stats;
    st.n_objects = (uint32_t)bld.items.size();
    st.n_nodes = d.n_nodes;
    st.n_wide_nodes = d.n_nodes4;
    st.bvh_depth = bld.max_depth;
    st.n_spheres = (uint32_t)(spheres.size() / 4);
    for (const HostShape &s : sc->shapes) {
        if (s.kind == RT_KIND_CYLINDER) st.n_cylinders++;
        if (s.kind == RT_KIND_CUBOID) st.n_cuboids++;
        if (s.kind == RT_KIND_QUAD) st.n_quads++;
    }
    st.n_csg = (uint32_t)csg.size();
    st.n_cdv = (uint32_t)cdv.size();
    st.n_materials = (uint32_t)sc->material_kinds.size();
    st.n_textures = (uint32_t)sc->textures.size();
    st.n_perlin = (uint32_t)sc->perlin.size();
    sc->committed = true;
    {
        // ask now what a render call of this scene would ask the runtime, so that rendering queries nothing
        RtDevCamera cam{};
        RtDevRender r{};
        r.band_count = 1;
        (void)rt_render_fused(sc->dev, sc->feat, st.bvh_depth, cam, r, sc->use_nodes16 ? 0x20000u : 0u, nullptr, nullptr, 0, nullptr,
                              nullptr, sc->ctx->cus, true, 0u);
        (void)rt_render_fused(sc->dev, sc->feat, st.bvh_depth, cam, r, sc->use_nodes16 ? 0x20000u : 0u, nullptr, nullptr, 0, nullptr,
                              nullptr, sc->ctx->cus, true, 0x100u);
    }
    return RT_OK;
}

int rt_scene_set_option(RtScene *sc, int option, int value) {
    if (!sc) return fail(RT_ERR_INVALID, "null scene");
    switch (option) {
    case RT_OPTION_BVH_BUILDER:
        if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
        if (value != RT_BVH_MEDIAN_SPLIT && value != RT_BVH_SAH) return fail(RT_ERR_INVALID, "unknown BVH builder");
        sc->bvh_builder = value;
        return RT_OK;
    case RT_OPTION_REFILL_THRESHOLD:
        if (value < 1 || value > 64) return fail(RT_ERR_INVALID, "refill threshold must be 1..64");
        sc->refill = (uint32_t)value;
        return RT_OK;
    case RT_OPTION_NODE_FORMAT:
        if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
        if (value != RT_NODES_AUTO && value != RT_NODES_F32 && value != RT_NODES_F16) return fail(RT_ERR_INVALID, "unknown node format");
        sc->node_format = value;
        return RT_OK;
    case RT_OPTION_NODE_MIN:
        if (value < 0 || value > 64) return fail(RT_ERR_INVALID, "node minimum must be 0..64");
        sc->node_min = value;
        return RT_OK;
    case RT_OPTION_HEAVY_MIN:
        if (value < 0 || value > 64) return fail(RT_ERR_INVALID, "heavy minimum must be 0..64");
        sc->heavy_min = value;
        return RT_OK;
    case RT_OPTION_TRAVERSAL:
        if (value != RT_TRAVERSAL_AUTO && value != RT_TRAVERSAL_TREE && value != RT_TRAVERSAL_LIST) return fail(RT_ERR_INVALID, "unknown traversal");
        sc->traversal = value;
        return RT_OK;
    case RT_OPTION_KERNEL_TIMING:
        if (!value) sc->timer.collect();
        sc->timer.enabled = value != 0;
        return RT_OK;
    case RT_OPTION_PROFILE_PHASES:
        sc->profile_phases = value != 0;
        return RT_OK;
    case RT_OPTION_CONVENTION:
        if (sc->committed) return fail(RT_ERR_STATE, "scene already committed");
        if (value != RT_CONVENTION_METAL && value != RT_CONVENTION_KIT) return fail(RT_ERR_INVALID, "unknown convention");
        sc->convention = value;
        return RT_OK;
    default:
        return fail(RT_ERR_INVALID, "unknown option");
    }
}

int rt_scene_stats(const RtScene *sc, RtSceneStats *out) {
    if (!sc || !out) return fail(RT_ERR_INVALID, "null argument");
    if (!sc->committed) return fail(RT_ERR_STATE, "scene not committed");
    *out = sc->stats;
    return RT_OK;
}

int rt_first_hit(RtScene *sc, const float *rays_dev, int n, float tmin, float tmax, uint64_t volume_key, RtHit *out_dev,
                 void *stream) {
    if (!sc || n < 0 || (n > 0 && (!rays_dev || !out_dev))) return fail(RT_ERR_INVALID, "bad argument");
    if (!sc->committed) return fail(RT_ERR_STATE, "scene not committed");
    DeviceScope on_device(sc->ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    hipError_t e = rt_launch_first_hit(sc->dev, sc->feat, rays_dev, n, tmin, tmax, volume_key, sc->query_time, out_dev, as_stream(stream));
    return e == hipSuccess ? RT_OK : fail_hip(e, "first-hit launch");
}

int rt_intervals(RtScene *sc, int shape, const float *rays_dev, int n, const float *scripted_u_dev, int n_scripted,
                 uint64_t volume_key, RtEvent *events_dev, int32_t *counts_dev, void *stream) {
    if (!sc || n < 0 || (n > 0 && (!rays_dev || !events_dev || !counts_dev))) return fail(RT_ERR_INVALID, "bad argument");
    if (!sc->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (shape < 0 || shape >= (int)sc->shapes.size()) return fail(RT_ERR_INVALID, "bad shape id");
    if (scripted_u_dev && n_scripted <= 0) return fail(RT_ERR_INVALID, "scripted values given but n_scripted <= 0");
    DeviceScope on_device(sc->ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    const HostShape &s = sc->shapes[shape];
    uint32_t object = s.object >= 0 ? (uint32_t)s.object : 0u;
    hipError_t e = rt_launch_intervals(sc->dev, s.ref, rays_dev, n, scripted_u_dev, n_scripted, volume_key, object,
                                       sc->query_time, events_dev, counts_dev, as_stream(stream));
    return e == hipSuccess ? RT_OK : fail_hip(e, "intervals launch");
}

int rt_camera_rays(RtContext *ctx, const RtCameraConfig *cc, const RtRenderConfig *rc, uint32_t first_sample, int width,
                   int height, int x0, int y0, int w, int h, float *rays_out_dev, void *stream) {
    if (!ctx || !cc || !rc || !rays_out_dev || width <= 0 || height <= 0 || w < 0 || h < 0)
        return fail(RT_ERR_INVALID, "bad argument");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    RtDevCamera cam;
    make_camera(copy_record(cc), width, height, cam);
    RtDevRender r = make_render(copy_record(rc), first_sample, width, height);
    hipError_t e = rt_launch_camera_rays(cam, r, x0, y0, w, h, rays_out_dev, as_stream(stream));
    return e == hipSuccess ? RT_OK : fail_hip(e, "camera-rays launch");
}

int rt_scatter(RtScene *sc, const RtScatterQuery *queries_dev, int n, uint64_t rng_seed, RtScatterResult *out_dev, void *stream) {
    if (!sc || n < 0 || (n > 0 && (!queries_dev || !out_dev))) return fail(RT_ERR_INVALID, "bad argument");
    if (!sc->committed) return fail(RT_ERR_STATE, "scene not committed");
    if (sc->material_kinds.empty()) return fail(RT_ERR_STATE, "scene has no materials");
    DeviceScope on_device(sc->ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    hipError_t e = rt_launch_scatter(sc->dev, queries_dev, n, rng_seed, out_dev, as_stream(stream));
    return e == hipSuccess ? RT_OK : fail_hip(e, "scatter launch");
}

int rt_rng_words(RtContext *ctx, const uint32_t *streams_dev, int n, int words_per_stream, uint32_t *out_dev,
                 uint32_t *wyrand_out_dev, void *stream) {
    if (!ctx || n < 0 || words_per_stream < 0 || (n > 0 && (!streams_dev || !out_dev || !wyrand_out_dev)))
        return fail(RT_ERR_INVALID, "bad argument");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    hipError_t e = rt_launch_rng_words(streams_dev, n, words_per_stream, out_dev, wyrand_out_dev, as_stream(stream));
    return e == hipSuccess ? RT_OK : fail_hip(e, "rng-words launch");
}

// The fused workspace is sized for the current device (the call has no context argument); a
// caller that drives several devices asks with the scene's device current.
size_t rt_render_workspace_size(int pipeline, int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    if (pipeline == RT_PIPELINE_FUSED) return rt_fused_workspace_bytes(rt_current_device_cus());
    return rt_wavefront_bytes((uint32_t)width * (uint32_t)height * RT_WAVEFRONT_BATCH_SPP);
}

int rt_band_rows(int height, int band_index, int band_count) {
    if (height <= 0 || band_count <= 0 || band_index < 0 || band_index >= band_count) return fail(RT_ERR_INVALID, "bad band arguments");
    int bands = (height + 7) / 8;
    int mine = band_index < bands ? (bands - band_index + band_count - 1) / band_count : 0;
    return 8 * mine;
}

namespace {

// Knobs of the fused kernel: node steps go on while 8 lanes want one.  Sphere tests cost about
// one node step, so in sphere scenes 12 lanes waiting at a leaf get their turn even then (swept 8..64; 20 when the
// spheres move: the change of frame makes their test dearer); tests
// of cylinders, boxes, CSG shapes and volumes cost several node steps and wait for fuller waves.
uint32_t fused_knobs(const RtScene *sc, const RtDevCamera &cam) {
    // refill = 0 and node_min = 255 stand for "not set": the launcher puts the chosen kernel's defaults in
    const uint32_t node_min = sc->node_min >= 0 ? (uint32_t)std::min(sc->node_min, 254) : 255u;
    return sc->refill | (node_min << 8) | (sc->profile_phases ? 0x10000u : 0u) | (sc->use_nodes16 ? 0x20000u : 0u) |
           ((cam.defocus ? 16u : 0u) << 18) | (((sc->feat & RT_FEAT_XPRIM) ? 65u : ((sc->feat & RT_FEAT_MOTION) ? 20u : 12u)) << 25);
}

uint32_t fused_knobs2(const RtScene *sc) {
    const uint32_t heavy = sc->heavy_min >= 0 ? (uint32_t)sc->heavy_min : RT_DEFAULT_HEAVY_MIN;
    return heavy | (sc->traversal == RT_TRAVERSAL_TREE ? 0x100u : 0u);
}

int render_checks(const RtScene *sc, const RtCameraConfig *cc, const RtRenderConfig *rc, const float *accum_dev, int width, int height) {
    if (!sc || !cc || !rc || !accum_dev || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad argument");
    if (!sc->committed) return fail(RT_ERR_STATE, "scene not committed");
    if ((uint64_t)width * (uint64_t)height > (1u << 28) || width > 65535 || height > 65535)     // a lane keeps its pixel as x | y << 16
        return fail(RT_ERR_LIMIT, "image too large");
    return RT_OK;
}

}  // namespace

int rt_render(RtScene *sc, const RtCameraConfig *cc, const RtRenderConfig *rc, uint32_t first_sample, int pipeline, int width,
              int height, float *accum_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    int chk = render_checks(sc, cc, rc, accum_dev, width, height);
    if (chk < 0) return chk;
    if (!workspace_dev) return fail(RT_ERR_INVALID, "bad argument");
    DeviceScope on_device(sc->ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    RtDevCamera cam;
    make_camera(copy_record(cc), width, height, cam);
    RtDevRender r = make_render(copy_record(rc), first_sample, width, height);
    r.animated = (sc->feat & RT_FEAT_MOTION) ? 1u : 0u;
    hipError_t e;
    if (pipeline == RT_PIPELINE_FUSED) {
        if (workspace_bytes < RT_FUSED_HEADER) return fail(RT_ERR_INVALID, "workspace too small");
        e = rt_render_fused(sc->dev, sc->feat, sc->stats.bvh_depth, cam, r, fused_knobs(sc, cam), accum_dev, workspace_dev,
                            workspace_bytes, as_stream(stream), &sc->timer, sc->ctx->cus, false, fused_knobs2(sc));
    } else if (pipeline == RT_PIPELINE_WAVEFRONT) {
        if (workspace_bytes < rt_wavefront_bytes((uint32_t)width * (uint32_t)height))
            return fail(RT_ERR_INVALID, "workspace too small");
        e = rt_render_wavefront(sc->dev, sc->feat, cam, r, accum_dev, workspace_dev, workspace_bytes, as_stream(stream), &sc->timer,
                                sc->ctx->cus, &sc->wf_probe, sc->stats.bvh_depth);
    } else {
        return fail(RT_ERR_INVALID, "unknown pipeline");
    }
    return e == hipSuccess ? RT_OK : fail_hip(e, "render");
}

int rt_render_static(RtScene *sc, const RtCameraConfig *cc, const RtRenderConfig *rc, uint32_t first_sample, int width, int height,
                     float *accum_dev, void *stream) {
    int chk = render_checks(sc, cc, rc, accum_dev, width, height);
    if (chk < 0) return chk;
    DeviceScope on_device(sc->ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    RtDevCamera cam;
    make_camera(copy_record(cc), width, height, cam);
    RtDevRender r = make_render(copy_record(rc), first_sample, width, height);
    r.animated = (sc->feat & RT_FEAT_MOTION) ? 1u : 0u;
    // the diagnostic build reports through a workspace the caller can read: not on this path
    const uint32_t knobs = fused_knobs(sc, cam) & ~0x10000u;
    hipError_t e = rt_render_fused_static(sc->dev, sc->feat, sc->stats.bvh_depth, cam, r, knobs, accum_dev, as_stream(stream),
                                          sc->ctx->cus, sc->ctx->device, fused_knobs2(sc));
    return e == hipSuccess ? RT_OK : fail_hip(e, "render");
}

int rt_render_bands(RtScene *sc, const RtCameraConfig *cc, const RtRenderConfig *rc, uint32_t first_sample, int band_index,
                    int band_count, int width, int height, float *accum_bands_dev, void *workspace_dev, size_t workspace_bytes,
                    void *stream) {
    int chk = render_checks(sc, cc, rc, accum_bands_dev, width, height);
    if (chk < 0) return chk;
    if (!workspace_dev || workspace_bytes < RT_FUSED_HEADER) return fail(RT_ERR_INVALID, "workspace too small");
    if (band_count <= 0 || band_index < 0 || band_index >= band_count) return fail(RT_ERR_INVALID, "bad band arguments");
    DeviceScope on_device(sc->ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    RtDevCamera cam;
    make_camera(copy_record(cc), width, height, cam);
    RtDevRender r = make_render(copy_record(rc), first_sample, width, height);
    r.animated = (sc->feat & RT_FEAT_MOTION) ? 1u : 0u;
    r.band_index = (uint32_t)band_index;
    r.band_count = (uint32_t)band_count;
    hipError_t e = rt_render_fused(sc->dev, sc->feat, sc->stats.bvh_depth, cam, r, fused_knobs(sc, cam), accum_bands_dev, workspace_dev,
                                   workspace_bytes, as_stream(stream), &sc->timer, sc->ctx->cus, false, fused_knobs2(sc));
    return e == hipSuccess ? RT_OK : fail_hip(e, "render");
}

int rt_resolve_rgb8(RtContext *ctx, const float *accum_dev, uint32_t n_samples, int width, int height,
                    uint8_t *rgbx_out_dev, void *stream) {
    if (!ctx || !accum_dev || !rgbx_out_dev || width <= 0 || height <= 0 || n_samples == 0)
        return fail(RT_ERR_INVALID, "bad argument");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    hipError_t e = rt_launch_resolve(accum_dev, (float)n_samples, width, height, rgbx_out_dev, as_stream(stream));
    return e == hipSuccess ? RT_OK : fail_hip(e, "resolve launch");
}

int rt_progressive_present(RtContext *ctx, const float *pass_sum_dev, uint32_t samples_per_pixel, uint32_t pass_counter,
                           uint32_t *acc_rgb10a2_dev, uint8_t *bgra8_out_dev, int width, int height, void *stream) {
    if (!ctx || !pass_sum_dev || !acc_rgb10a2_dev || !bgra8_out_dev || width <= 0 || height <= 0)
        return fail(RT_ERR_INVALID, "bad argument");
    if (samples_per_pixel == 0 || pass_counter == 0) return fail(RT_ERR_INVALID, "samples_per_pixel and pass_counter start at 1");
    DeviceScope on_device(ctx->device);
    if (!on_device.ok) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    hipError_t e = rt_launch_progressive_present(pass_sum_dev, samples_per_pixel, pass_counter, width, height, acc_rgb10a2_dev,
                                                 bgra8_out_dev, as_stream(stream));
    return e == hipSuccess ? RT_OK : fail_hip(e, "present launch");
}

int rt_kernel_time(RtScene *sc, int reset, double *total_ms, uint64_t *launches, const char **kernel_name) {
    if (!sc) return fail(RT_ERR_INVALID, "null scene");
    DeviceScope on_device(sc->ctx ? sc->ctx->device : 0);
    sc->timer.collect();
    if (total_ms) *total_ms = sc->timer.total_ms;
    if (launches) *launches = sc->timer.launches;
    if (kernel_name) *kernel_name = sc->timer.name;
    if (reset) sc->timer.reset();
    return RT_OK;
}
