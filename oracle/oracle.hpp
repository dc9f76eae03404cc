// oracle.hpp - CPU restatement of the reference path tracer (test infrastructure).
//
// TEST INFRASTRUCTURE ONLY: see oracle_api.h.  Every routine cites the reference
// file:line it restates (paths relative to the reference checkout).  The code is
// templated on the arithmetic type: R = float reproduces the Metal-side headers'
// precision, R = double the RayTracingKit precision (Vector3D.swift:20-23).
//
// Where the two reference renderers disagree, the choice made here is listed in
// the block comment in front of the routine.
#pragma once

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "oracle_api.h"

namespace orc {

// ---------------------------------------------------------------- vectors
template <class R> struct V3 {
    R x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(R a, R b, R c) : x(a), y(b), z(c) {}
    template <class S> explicit V3(const V3<S> &o) : x(R(o.x)), y(R(o.y)), z(R(o.z)) {}
    R operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class R> V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> V3<R> operator*(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class R> V3<R> operator*(V3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> V3<R> operator*(R s, V3<R> a) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> V3<R> operator/(V3<R> a, R s) { return {a.x / s, a.y / s, a.z / s}; }
template <class R> V3<R> operator/(V3<R> a, V3<R> b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
template <class R> R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> V3<R> cross(V3<R> a, V3<R> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class R> R length_squared(V3<R> a) { return dot(a, a); }
template <class R> V3<R> normalize(V3<R> a) { return a / std::sqrt(dot(a, a)); }

template <class R> constexpr R kPi() { return R(3.14159265358979323846264338327950288L); }
template <class R> constexpr R kInf() { return std::numeric_limits<R>::infinity(); }

// 3x3 rotation (column-major, Renderable.h:13-16) + translation.
template <class R> struct Xf {
    V3<R> c0, c1, c2, t;
    V3<R> rot(V3<R> v) const { return c0 * v.x + c1 * v.y + c2 * v.z; }            // R * v
    V3<R> rot_t(V3<R> v) const { return {dot(c0, v), dot(c1, v), dot(c2, v)}; }      // transpose(R) * v
};

template <class R> struct Ray {
    V3<R> origin, direction;
    V3<R> at(R t) const { return origin + t * direction; }   // HitTesting.h:42-44
};

// RenderableImpl.h:16-19
template <class R> Ray<R> inverse_transform(Ray<R> r, const Xf<R> &t) {
    return {t.rot_t(r.origin - t.t), t.rot_t(r.direction)};
}
// RenderableImpl.h:25-28
template <class R> V3<R> inverse_transform_point(V3<R> p, const Xf<R> &t) { return t.rot_t(p - t.t); }

// ---------------------------------------------------------------- RNG
// WyRand, PRNG.swift:10-22.  The state after k steps is seed + k*C, so output k is a
// pure function of (seed, k): that is what makes the per-sample seeds counter based.
inline uint64_t wyrand_at(uint64_t seed, uint64_t k) {
    uint64_t s = seed + (k + 1) * 0xa0761d6478bd642fULL;
    unsigned __int128 m = (unsigned __int128)s * (unsigned __int128)(s ^ 0xe7037ed1a0b428dbULL);
    return (uint64_t)(m >> 64) ^ (uint64_t)m;
}

// PCG32, RNG.h:31-52.
struct Pcg32 {
    uint64_t state = 0, inc = 1;
    Pcg32() {}
    Pcg32(uint64_t initstate, uint64_t initseq) {
        state = 0u;
        inc = (initseq << 1u) | 1u;
        next_u32();
        state += initstate;
        next_u32();
    }
    uint32_t next_u32() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
    }
};

// Stream of one (pixel, sample): the per-pass seed is WyRand output number `sample` of
// rng_seed (the reference app draws a fresh 64-bit seed per one-sample pass,
// ContentView.swift:201-202), and the pixel offsets are Shaders.metal:141-143
// (32-bit wrap-around products widened to 64 bits).
inline Pcg32 sample_stream(uint64_t rng_seed, uint32_t sample, uint32_t x, uint32_t y) {
    uint64_t pass = wyrand_at(rng_seed, sample);
    uint32_t lo = (uint32_t)pass, hi = (uint32_t)(pass >> 32);
    return Pcg32((uint64_t)(uint32_t)(x * 5569u + lo), (uint64_t)(uint32_t)(y * 2707u + hi));
}

// Random source: PCG32, or the scripted test RNG of RNG.h:15-27.
struct Rng {
    Pcg32 pcg;
    const float *script = nullptr;
    int n_script = 0, i_script = 0;

    // RNG.h:54-56 returns u32 * 2^-32, which rounds to 1.0f for the top 128 values and
    // then yields log(0) in the volume sampler.  Here: (u32 >> 8) * 2^-24, in [0, 1) and
    // exact in float, so both precisions see the same number.
    template <class R> R random_f() {
        if (script) {
            float v = script[i_script];
            i_script = (i_script + 1) % n_script;
            return R(v);
        }
        return R(pcg.next_u32() >> 8) * R(1.0 / 16777216.0);
    }
};

constexpr double kMinVectorLengthSquared = 1e-24;   // RNG.h:13

// RNG.h:59-68 / Vector3D.swift:129-143: rejection from the square, normalised ONTO the circle.
template <class R> void random_unit_vector_2d(Rng &rng, R &ox, R &oy) {
    while (true) {
        R x = rng.random_f<R>() * 2 - 1;
        R y = rng.random_f<R>() * 2 - 1;
        R len_sq = x * x + y * y;
        if (R(kMinVectorLengthSquared) < len_sq && len_sq <= 1) {
            R l = std::sqrt(len_sq);
            ox = x / l;
            oy = y / l;
            return;
        }
    }
}

// Vector3D.swift:113-127.  (RNG.h:70-80 omits z*z from the length: a defect that yields
// non-unit vectors; the RayTracingKit form is the one restated.)
template <class R> V3<R> random_unit_vector_3d(Rng &rng) {
    while (true) {
        R x = rng.random_f<R>() * 2 - 1;
        R y = rng.random_f<R>() * 2 - 1;
        R z = rng.random_f<R>() * 2 - 1;
        R len_sq = x * x + y * y + z * z;
        if (R(kMinVectorLengthSquared) < len_sq && len_sq <= 1) {
            return V3<R>(x, y, z) / std::sqrt(len_sq);
        }
    }
}

// ---------------------------------------------------------------- scene storage
enum ShapeKind { SPHERE = 0, CYLINDER = 1, CUBOID = 2, QUAD = 3, COMPOSITION = 4, CDV = 5, MOTION = 6 };

struct Box {
    double mn[3] = {INFINITY, INFINITY, INFINITY};
    double mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    void add(const double p[3]) {
        for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); }
    }
    void add(const Box &b) { add(b.mn); add(b.mx); }
    double center(int a) const { return (mn[a] + mx[a]) / 2; }
};

struct Shape {
    ShapeKind kind;
    // primitives (values exactly as given, converted to R at use)
    double rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, trans[3] = {0, 0, 0};
    double radius = 0, height = 0, size[3] = {0, 0, 0};
    double q_origin[3], q_u[3], q_v[3], q_w[3], q_normal[3], q_d = 0;
    uint64_t mat[6] = {0, 0, 0, 0, 0, 0};
    // composition (RenderableImpl.h:506-523)
    int min_count = 0;
    bool subtract = false;
    std::vector<int> children;
    // constant-density volume (RenderableImpl.h:717-725)
    int base = -1;
    double density = 0;
    // motion blur (RayTracingKit/Hittable.swift:528-555): `base` moved by transform.pow(time)
    double m_quat[4] = {1, 0, 0, 0}, m_trans[3] = {0, 0, 0};     // w, x, y, z
    // texture coordinates of RayTracingKit (Hittable.swift:188,245-248,281-282) instead of the
    // Metal side's (RenderableImpl.h:83,113-121,205); copied from the scene when the shape is added
    bool kit_uv = false;
    Box box;
};

struct Image { int w = 0, h = 0; std::vector<uint8_t> rgb; };

struct BvhNode {
    Box box;
    int left = -1, right = -1;        // >= 0: node index; < 0: ~object index
};

struct Counters {
    uint64_t samples = 0, rays = 0, hits = 0, box_tests = 0, prim_tests[4] = {0, 0, 0, 0};
    uint64_t csg_events = 0, cdv_intervals = 0, shades[10] = {0};
    void add(const Counters &o) {
        samples += o.samples; rays += o.rays; hits += o.hits; box_tests += o.box_tests;
        for (int i = 0; i < 4; i++) prim_tests[i] += o.prim_tests[i];
        csg_events += o.csg_events; cdv_intervals += o.cdv_intervals;
        for (int i = 0; i < 10; i++) shades[i] += o.shades[i];
    }
};

struct Scene {
    bool animated = false;      // some shape is a MOTION: camera samples draw a time
    bool kit_convention = false;   // RayTracingKit's texture coordinates and metal rule (Material.swift:53-57)
    std::vector<Shape> shapes;
    std::vector<int> objects;
    std::vector<OrcMaterial> materials;
    std::vector<Image> images;
    std::vector<OrcPerlin> perlins;
    std::vector<OrcKitTexture> kit_textures;
    std::vector<OrcKitPerlin> kit_perlins;
    std::vector<BvhNode> nodes;
    int root = -1;          // node index, or ~object when there is a single object
    bool committed = false;
};

template <class R> V3<R> v3(const double p[3]) { return V3<R>(R(p[0]), R(p[1]), R(p[2])); }
template <class R> Xf<R> xf_of(const Shape &s) {
    Xf<R> x;
    x.c0 = V3<R>(R(s.rot[0]), R(s.rot[1]), R(s.rot[2]));
    x.c1 = V3<R>(R(s.rot[3]), R(s.rot[4]), R(s.rot[5]));
    x.c2 = V3<R>(R(s.rot[6]), R(s.rot[7]), R(s.rot[8]));
    x.t = v3<R>(s.trans);
    return x;
}

// ---------------------------------------------------------------- hit enumerators
// The enumerator protocol of RenderableImpl.h: hasNext/move/isExit/t/point/normal/
// material_offset/texture_coordinates, hits ordered by t.
template <class R> struct Enum {
    virtual ~Enum() {}
    virtual bool hasNext() const = 0;
    virtual void move() = 0;
    virtual bool isExit() const = 0;
    virtual R t() const = 0;
    virtual V3<R> point() const = 0;
    virtual V3<R> normal() const = 0;
    virtual uint64_t material_offset() const = 0;
    virtual void texture_coordinates(R &u, R &v) const = 0;
};

// RenderableImpl.h:30-87
template <class R> struct SphereEnum final : Enum<R> {
    Xf<R> xf; R radius; uint64_t mat; Ray<R> ray; R ts[2]; int index; bool kit_uv;
    SphereEnum(const Shape &s, Ray<R> r) : xf(xf_of<R>(s)), radius(R(s.radius)), mat(s.mat[0]), ray(r), kit_uv(s.kit_uv) {
        V3<R> oc = ray.origin - xf.t;
        R a = dot(ray.direction, ray.direction);
        R b_2 = dot(ray.direction, oc);
        R c = dot(oc, oc) - radius * radius;
        R D_4 = b_2 * b_2 - a * c;
        if (D_4 < 0) {
            index = 2;
            ts[0] = ts[1] = std::numeric_limits<R>::quiet_NaN();
        } else {
            ts[0] = (-b_2 - std::sqrt(D_4)) / a;
            ts[1] = (-b_2 + std::sqrt(D_4)) / a;
            index = 0;
        }
    }
    bool hasNext() const override { return index < 2; }
    void move() override { index++; }
    bool isExit() const override { return index == 1; }
    R t() const override { return ts[index]; }
    V3<R> point() const override { return ray.at(t()); }
    V3<R> normal() const override { return (point() - xf.t) / radius; }
    uint64_t material_offset() const override { return mat; }
    void texture_coordinates(R &u, R &v) const override {
        V3<R> n = xf.rot_t(normal());
        if (kit_uv) u = (std::atan2(-n.z, n.x) + kPi<R>()) / (2 * kPi<R>());      // Hittable.swift:188
        else u = (std::atan2(n.z, -n.x) + kPi<R>()) / (2 * kPi<R>());             // RenderableImpl.h:83
        v = std::acos(-n.y) / kPi<R>();
    }
};

template <class R> struct PlainHit { R t = 0; V3<R> normal; uint64_t mat = 0; R u = 0, v = 0; int face = 0; };

// RenderableImpl.h:89-255.  One departure: the tube quadratic's leading coefficient is
// d.d - d.y^2 (Hittable.swift:264) rather than 1 - d.y^2 (RenderableImpl.h:156), which is
// the same number for the unit directions the renderer produces and stays correct for
// the unnormalised rays a first-hit query may pass.
template <class R> struct CylinderEnum final : Enum<R> {
    Ray<R> ray; PlainHit<R> hit[2]; int index;

    bool kit_uv;
    void plane_uv(V3<R> lp, R radius, R flipX, R &u, R &v) const {
        if (kit_uv) {                                                         // Hittable.swift:245,248
            u = R(0.5) + lp.x / (2 * radius);
            v = R(0.5) + lp.z / (2 * radius);
        } else {                                                              // RenderableImpl.h:141-146
            u = R(0.5) + lp.z * flipX / (2 * radius);
            v = R(0.5) + lp.x / (2 * radius);
        }
    }
    void tube_uv(V3<R> lp, R height, R &u, R &v) const {
        if (kit_uv) u = std::atan2(lp.z, lp.x) / (2 * kPi<R>());              // Hittable.swift:281
        else u = (std::atan2(lp.z, -lp.x) + kPi<R>()) / (2 * kPi<R>());       // RenderableImpl.h:203-208
        v = lp.y / height;
    }
    CylinderEnum(const Shape &s, Ray<R> r) : ray(r), kit_uv(s.kit_uv) {
        Xf<R> xf = xf_of<R>(s);
        R radius = R(s.radius), height = R(s.height);
        Ray<R> lr = inverse_transform(ray, xf);
        PlainHit<R> planeIn, planeOut, tubeIn, tubeOut;
        {   // hit_plane :102-139
            R denom = lr.direction.y;
            R tb = -lr.origin.y / denom;
            R tt = (height - lr.origin.y) / denom;
            if (std::isfinite(tb) && std::isfinite(tt)) {
                planeIn.t = tb;
                planeIn.normal = -xf.c1;
                planeIn.mat = s.mat[0];
                plane_uv(lr.at(tb), radius, R(-1), planeIn.u, planeIn.v);
                planeOut.t = tt;
                planeOut.normal = xf.c1;
                planeOut.mat = s.mat[1];
                plane_uv(lr.at(tt), radius, R(+1), planeOut.u, planeOut.v);
                if (planeIn.t > planeOut.t) std::swap(planeIn, planeOut);
            } else if (lr.origin.y >= 0 && lr.origin.y < height) {
                planeIn.t = -kInf<R>(); planeOut.t = kInf<R>();
            } else {
                planeIn.t = kInf<R>(); planeOut.t = -kInf<R>();
            }
        }
        {   // hit_tube :148-201
            V3<R> d = lr.direction, o = lr.origin;
            R a = std::max(R(0), -d.y * d.y + dot(d, d));
            R b_2 = -d.y * o.y + dot(d, o);
            R c = -o.y * o.y + dot(o, o) - radius * radius;
            R D_4 = b_2 * b_2 - a * c;
            if (D_4 < 0) {
                tubeIn.t = kInf<R>(); tubeOut.t = -kInf<R>();
            } else {
                R root = std::sqrt(D_4);
                R t1 = (-b_2 - root) / a;
                R t2 = (-b_2 + root) / a;
                if (std::isfinite(t1) && std::isfinite(t2)) {
                    V3<R> lp1 = lr.at(t1), lp2 = lr.at(t2);
                    tubeIn.t = t1;
                    tubeIn.normal = xf.rot(V3<R>(lp1.x, 0, lp1.z) / radius);
                    tubeIn.mat = s.mat[2];
                    tube_uv(lp1, height, tubeIn.u, tubeIn.v);
                    tubeOut.t = t2;
                    tubeOut.normal = xf.rot(V3<R>(lp2.x, 0, lp2.z) / radius);
                    tubeOut.mat = s.mat[2];
                    tube_uv(lp2, height, tubeOut.u, tubeOut.v);
                } else if (o.x * o.x + o.z * o.z > radius * radius) {
                    tubeIn.t = kInf<R>(); tubeOut.t = -kInf<R>();
                } else {
                    tubeIn.t = -kInf<R>(); tubeOut.t = kInf<R>();
                }
            }
        }
        hit[0] = planeIn.t > tubeIn.t ? planeIn : tubeIn;       // :222-224
        hit[1] = planeOut.t < tubeOut.t ? planeOut : tubeOut;
        index = hit[0].t <= hit[1].t ? 0 : 2;
    }
    bool hasNext() const override { return index < 2; }
    void move() override { index++; }
    bool isExit() const override { return index == 1; }
    R t() const override { return hit[index].t; }
    V3<R> point() const override { return ray.at(t()); }
    V3<R> normal() const override { return hit[index].normal; }
    uint64_t material_offset() const override { return hit[index].mat; }
    void texture_coordinates(R &u, R &v) const override { u = hit[index].u; v = hit[index].v; }
};

// RenderableImpl.h:257-393.  Face order left,right,bottom,top,back,front (Renderable.h:40-44).
template <class R> struct CuboidEnum final : Enum<R> {
    Xf<R> xf; V3<R> size; uint64_t mat[6]; Ray<R> ray; PlainHit<R> hit[2]; int index;

    static void hit_plane(R size, R origin, R direction, V3<R> top_normal, int axis,
                          PlainHit<R> &in, PlainHit<R> &out) {                // :269-308
        R tb = -origin / direction;
        R tt = (size - origin) / direction;
        if (std::isfinite(tb) && std::isfinite(tt)) {
            in.t = tb; in.normal = -top_normal; in.face = 2 * axis;
            out.t = tt; out.normal = top_normal; out.face = 2 * axis + 1;
            if (in.t > out.t) std::swap(in, out);
        } else if (origin >= 0 && origin < size) {
            in.t = -kInf<R>(); out.t = kInf<R>();
        } else {
            in.t = kInf<R>(); out.t = -kInf<R>();
        }
    }
    CuboidEnum(const Shape &s, Ray<R> r) : xf(xf_of<R>(s)), size(v3<R>(s.size)), ray(r) {
        for (int i = 0; i < 6; i++) mat[i] = s.mat[i];
        Ray<R> lr = inverse_transform(ray, xf);
        PlainHit<R> xIn, xOut, yIn, yOut, zIn, zOut;
        hit_plane(size.x, lr.origin.x, lr.direction.x, xf.c0, 0, xIn, xOut);
        hit_plane(size.y, lr.origin.y, lr.direction.y, xf.c1, 1, yIn, yOut);
        hit_plane(size.z, lr.origin.z, lr.direction.z, xf.c2, 2, zIn, zOut);
        hit[0] = xIn.t > yIn.t ? (xIn.t > zIn.t ? xIn : zIn) : (yIn.t > zIn.t ? yIn : zIn);          // :331-333
        hit[1] = xOut.t < yOut.t ? (xOut.t < zOut.t ? xOut : zOut) : (yOut.t < zOut.t ? yOut : zOut);
        index = hit[0].t <= hit[1].t ? 0 : 2;
    }
    bool hasNext() const override { return index < 2; }
    void move() override { index++; }
    bool isExit() const override { return index == 1; }
    R t() const override { return hit[index].t; }
    V3<R> point() const override { return ray.at(t()); }
    V3<R> normal() const override { return hit[index].normal; }
    uint64_t material_offset() const override { return mat[hit[index].face]; }
    void texture_coordinates(R &u, R &v) const override {                    // :360-392
        V3<R> lp = inverse_transform_point(point(), xf) / size;
        switch (hit[index].face) {
        case 0: u = 1 - lp.z; v = lp.y; break;       // left
        case 5: u = lp.x;     v = lp.y; break;       // front
        case 1: u = lp.z;     v = lp.y; break;       // right
        case 4: u = 1 - lp.x; v = lp.y; break;       // back
        case 3: u = lp.x;     v = lp.z; break;       // top
        default: u = lp.x;    v = 1 - lp.z; break;   // bottom
        }
    }
};

// RenderableImpl.h:395-467: the quad as a half-space interval [t, +inf) or (-inf, t].
template <class R> struct QuadEnum final : Enum<R> {
    int index; R hit[2]; V3<R> pt, nrm; R uu = 0, vv = 0; uint64_t mat;
    QuadEnum(const Shape &s, Ray<R> ray) : mat(s.mat[0]) {
        V3<R> normal = v3<R>(s.q_normal);
        R denom = dot(ray.direction, normal);
        R t = (R(s.q_d) - dot(ray.origin, normal)) / denom;
        hit[0] = hit[1] = 0;
        if (std::isfinite(t)) {
            hit[0] = denom < 0 ? t : -kInf<R>();
            hit[1] = denom < 0 ? kInf<R>() : t;
            pt = ray.at(t);
            nrm = normal;
            V3<R> p = pt - v3<R>(s.q_origin);
            uu = dot(v3<R>(s.q_w), cross(p, v3<R>(s.q_v)));
            vv = dot(v3<R>(s.q_w), cross(v3<R>(s.q_u), p));
            index = (uu < 0 || uu > 1 || vv < 0 || vv > 1) ? 2 : 0;
        } else {
            index = 2;
        }
    }
    bool hasNext() const override { return index < 2; }
    void move() override { index++; }
    bool isExit() const override { return index == 1; }
    R t() const override { return hit[index]; }
    V3<R> point() const override { return pt; }
    V3<R> normal() const override { return nrm; }
    uint64_t material_offset() const override { return mat; }
    void texture_coordinates(R &u, R &v) const override { u = uu; v = vv; }
};

template <class R> struct TraceCtx {
    const Scene *scene;
    Counters *counters;
    Rng *scripted;           // test RNG for volume draws, or nullptr
    uint64_t cdv_state;      // PCG32 state word that keys volume draws when not scripted
    double time = 0;         // the sample's time in [0, 1), Camera.swift:89
};

template <class R> std::unique_ptr<Enum<R>> make_enum(const TraceCtx<R> &ctx, int shape, Ray<R> ray, int object);

// RenderableImpl.h:658-715 with the helpers of :525-656.
template <class R> struct CompositionEnum final : Enum<R> {
    std::vector<std::unique_ptr<Enum<R>>> children;
    int min_count; bool subtract; int depth = 0; int cur = -1; bool dead = false;
    Counters *counters;

    bool anyHasNext() const {                                                // :550-558
        for (auto &c : children) if (c->hasNext()) return true;
        return false;
    }
    void chooseChild() {                                                     // :532-548: ties keep the HIGHEST index
        cur = -1;
        R best = kInf<R>();
        for (int i = (int)children.size() - 1; i >= 0; i--) {
            if (children[i]->hasNext()) {
                R t = children[i]->t();
                if (t < best) { cur = i; best = t; }
            }
        }
    }
    bool shouldSwap() const { return subtract && cur > 0; }                  // :693-695
    void scanDepth() {                                                       // :678-691
        bool wasInside = depth >= min_count;
        while (anyHasNext()) {
            chooseChild();
            if (cur < 0) { dead = true; return; }   // only hits at +inf are left: the reference indexes child -1 here
            if (counters) counters->csg_events++;
            if (children[cur]->isExit() != (subtract && cur > 0)) depth--; else depth++;
            bool isInside = depth >= min_count;
            if (isInside != wasInside) break;
            children[cur]->move();
        }
    }
    CompositionEnum(const TraceCtx<R> &ctx, const Shape &s, Ray<R> ray, int object)
        : min_count(s.min_count), subtract(s.subtract), counters(ctx.counters) {
        for (int c : s.children) children.push_back(make_enum<R>(ctx, c, ray, object));
        scanDepth();
    }
    bool hasNext() const override { return !dead && anyHasNext(); }
    void move() override { children[cur]->move(); scanDepth(); }             // :704-707
    bool isExit() const override { return children[cur]->isExit() != shouldSwap(); }
    R t() const override { return children[cur]->t(); }
    V3<R> point() const override { return children[cur]->point(); }
    V3<R> normal() const override { return children[cur]->normal() * R(shouldSwap() ? -1 : +1); }
    uint64_t material_offset() const override { return children[cur]->material_offset(); }
    void texture_coordinates(R &u, R &v) const override { children[cur]->texture_coordinates(u, v); }
};

// RayTracingKit/Hittable.swift:394-526: the Kit's Composition.  Where the Metal side merges the events of all children
// at once and counts depth (CompositionEnum above), the Kit keeps a list of ranges per operand and folds the operands
// from the left, two lists at a time: enumerateHits (:432-460) walks both lists in order of t - the right-hand list
// first on a tie - and makeUnion / makeIntersection / makeDifference (:462-526) rebuild ranges from the depth count;
// a hit of a subtracted operand is `inverted` (:64-68: its side flips).  Used for scenes under the Kit's conventions,
// so that images of the Kit's worlds are checked against the Kit's own interval algebra.  The events come out in the
// enumerator protocol of the Metal side (normals as enumerated; the Kit stores a normal already turned against the
// ray, which is the same normal after HitInfo::set_normal), so the normal of an inverted hit is negated here.
template <class R> struct KitCompositionEnum final : Enum<R> {
    struct Ev { R t; bool exit; V3<R> p, n; uint64_t mat; R u, v; };
    std::vector<Ev> events;          // entry, exit, entry, exit, ... of the result
    size_t idx = 0;

    static std::vector<Ev> collect(Enum<R> &e) {
        std::vector<Ev> out;
        for (; e.hasNext(); e.move()) {
            Ev v;
            v.t = e.t(); v.exit = e.isExit();
            const bool fin = std::isfinite(v.t);          // the open end of a quad's half space has no surface
            v.p = fin ? e.point() : V3<R>{}; v.n = fin ? e.normal() : V3<R>{};
            v.mat = fin ? e.material_offset() : 0;
            v.u = v.v = 0;
            if (fin) e.texture_coordinates(v.u, v.v);
            out.push_back(v);
        }
        return out;
    }
    static Ev inverted(Ev v) { v.exit = !v.exit; v.n = v.n * R(-1); return v; }

    // enumerateHits + the three make... functions; op: 0 union, 1 intersection, 2 difference
    static std::vector<Ev> fold(const std::vector<Ev> &lhs, const std::vector<Ev> &rhs, int op, Counters *counters) {
        std::vector<Ev> result;
        bool has_entry = false;
        Ev entry{};
        int depth = 0;
        size_t i = 0, j = 0;
        while (i < lhs.size() || j < rhs.size()) {
            bool is_right;
            if (i < lhs.size() && j < rhs.size()) is_right = !(lhs[i].t < rhs[j].t);      // :439-445
            else is_right = i >= lhs.size();
            const Ev &hit = is_right ? rhs[j++] : lhs[i++];
            if (counters) counters->csg_events++;
            const bool front = !hit.exit;
            if (op == 0) {                                                            // :462-481
                if (front) { if (!has_entry) { entry = hit; has_entry = true; } depth++; }
                else { depth--; if (depth == 0) { result.push_back(entry); result.push_back(hit); has_entry = false; } }
            } else if (op == 1) {                                                     // :483-502
                if (front) { depth++; if (depth == 2) { entry = hit; has_entry = true; } }
                else { if (depth == 2) { result.push_back(entry); result.push_back(hit); has_entry = false; } depth--; }
            } else {                                                                  // :504-526
                if (front != is_right) depth++; else depth--;
                if (depth == 1) { entry = is_right ? inverted(hit) : hit; has_entry = true; }
                else if (has_entry) { result.push_back(entry); result.push_back(is_right ? inverted(hit) : hit); has_entry = false; }
            }
        }
        return result;
    }

    KitCompositionEnum(const TraceCtx<R> &ctx, const Shape &s, Ray<R> ray, int object) {
        const int op = s.subtract ? 2 : (s.min_count > 1 ? 1 : 0);
        for (size_t k = 0; k < s.children.size(); k++) {
            auto child = make_enum<R>(ctx, s.children[k], ray, object);
            std::vector<Ev> next = collect(*child);
            events = k == 0 ? next : fold(events, next, op, ctx.counters);           // :416-430
        }
    }
    bool hasNext() const override { return idx < events.size(); }
    void move() override { idx++; }
    bool isExit() const override { return events[idx].exit; }
    R t() const override { return events[idx].t; }
    V3<R> point() const override { return events[idx].p; }
    V3<R> normal() const override { return events[idx].n; }
    uint64_t material_offset() const override { return events[idx].mat; }
    void texture_coordinates(R &u, R &v) const override { u = events[idx].u; v = events[idx].v; }
};

// RenderableImpl.h:727-802.  The reference draws from the path's generator inside
// traversal and keeps the advance even for candidates that lose, which makes the result
// depend on the traversal order of an opaque BVH.  Here (and in the HIP path) the draws
// of one volume come from a private PCG32 stream seeded with (state word of the path's
// generator at the start of the trace, object index + 1): order independent.
template <class R> struct CdvEnum final : Enum<R> {
    std::unique_ptr<Enum<R>> impl; Ray<R> ray; Rng own; Rng *rng; R neg_inv_density;
    bool exit_ = true; R t_ = 0; V3<R> hp, hn; uint64_t hmat = 0; R hu = 0, hv = 0;
    Counters *counters;

    void scan() {                                                            // :737-777
        while (true) {
            if (!impl->hasNext()) { exit_ = true; break; }
            R t1 = std::max(R(0), impl->t());
            uint64_t material = impl->material_offset();
            R tu, tv;
            impl->texture_coordinates(tu, tv);
            impl->move();
            R t2 = impl->hasNext() ? impl->t() : kInf<R>();
            if (counters) counters->cdv_intervals++;
            if (t1 < t2) {
                R t = t1 + std::log(1 - rng->random_f<R>()) * neg_inv_density;
                if (t <= t2) {
                    exit_ = false;
                    t_ = t;
                    hp = ray.at(t);
                    hn = -ray.direction;
                    hmat = material;
                    hu = tu; hv = tv;
                    break;
                }
            }
            if (impl->hasNext()) impl->move(); else { exit_ = true; break; }
        }
    }
    CdvEnum(const TraceCtx<R> &ctx, const Shape &s, Ray<R> r, int object)
        : ray(r), neg_inv_density(R(-1) / R(s.density)), counters(ctx.counters) {
        impl = make_enum<R>(ctx, s.base, r, object);
        if (ctx.scripted) {
            rng = ctx.scripted;
        } else {
            own.pcg = Pcg32(ctx.cdv_state, (uint64_t)(object + 1));
            rng = &own;
        }
        scan();
    }
    bool hasNext() const override { return !exit_ || impl->hasNext(); }     // :784
    void move() override { if (exit_) { impl->move(); scan(); } else { exit_ = true; } }
    bool isExit() const override { return exit_; }
    R t() const override { return exit_ ? impl->t() : t_; }
    V3<R> point() const override { return exit_ ? impl->point() : hp; }
    V3<R> normal() const override { return exit_ ? impl->normal() : hn; }
    uint64_t material_offset() const override { return exit_ ? impl->material_offset() : hmat; }
    void texture_coordinates(R &u, R &v) const override {
        if (exit_) impl->texture_coordinates(u, v); else { u = hu; v = hv; }
    }
};

// ---------------------------------------------------------------- motion blur
// Unit quaternions as RayTracingKit/Transform.swift:169-259 (w, x, y, z; w >= 0).
struct Quat { double w, x, y, z; };
inline Quat quat_mul(Quat a, Quat b) {                                       // :234-241
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline void quat_rotate(Quat q, const double v[3], double out[3]) {          // :228-232
    Quat r = quat_mul(quat_mul(q, Quat{0, v[0], v[1], v[2]}), Quat{q.w, -q.x, -q.y, -q.z});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
inline Quat quat_pow(Quat q, double t) {                                     // :243-250
    if (q.w == 1.0) return q;
    double th = std::acos(q.w);
    double sin_th = std::sqrt(1 - q.w * q.w);
    double w2 = std::cos(th * t);
    double scale = std::sin(th * t) / sin_th;
    if (!std::isfinite(scale)) return {w2, 0, 0, 0};
    return {w2, q.x * scale, q.y * scale, q.z * scale};
}

// Transform3D.boundingBoxForAnimation of a point, Transform.swift:93-140.
inline Box animation_box_of_point(Quat q, const double tr[3], const double p[3]) {
    double p1[3], p2[3];
    quat_rotate(q, p, p1);
    for (int a = 0; a < 3; a++) p2[a] = p1[a] + tr[a];
    Box result;
    result.add(p);
    result.add(p2);
    if (q.w != 1.0) {
        const double kTau = 2 * 3.14159265358979323846264338327950288;
        double th = 2 * std::acos(q.w);
        double sin_th = std::sin(th), cos_th = std::cos(th);
        auto norm_angle = [&](double x) { return x < 0 ? x + kTau : (x >= kTau ? x - kTau : x); };   // :157-166
        auto get_p = [&](double t, double out[3]) {
            for (int a = 0; a < 3; a++)
                out[a] = (std::sin((1 - t) * th) / sin_th) * p[a] + (std::sin(t * th) / sin_th) * p1[a] + t * tr[a];
        };
        for (int a = 0; a < 3; a++) {
            double A = th * sin_th * p[a];
            double B = th * (cos_th * p[a] - p1[a]);
            double C = tr[a] * sin_th;
            double Z = C / std::sqrt(A * A + B * B);
            if (!std::isfinite(Z) || std::fabs(Z) > 1) continue;
            double phi = std::atan2(A, B);
            double acos_z = std::acos(Z);
            double t1 = norm_angle(phi + acos_z) / th, t2 = norm_angle(phi - acos_z) / th;
            double e[3];
            if (std::isfinite(t1) && t1 >= 0 && t1 <= 1) { get_p(t1, e); result.add(e); }
            if (std::isfinite(t2) && t2 >= 0 && t2 <= 1) { get_p(t2, e); result.add(e); }
        }
    }
    return result;
}

// Transform3D.boundingBoxForAnimation of a box, Transform.swift:142-154.
inline Box animation_box(Quat q, const double tr[3], const Box &box) {
    Box result;
    if (q.w == 1.0) {
        result.add(box);
        double a[3], b[3];
        for (int i = 0; i < 3; i++) { a[i] = box.mn[i] + tr[i]; b[i] = box.mx[i] + tr[i]; }
        result.add(a);
        result.add(b);
    } else {
        for (int i = 0; i < 8; i++) {
            double p[3] = {(i & 1) ? box.mx[0] : box.mn[0], (i & 2) ? box.mx[1] : box.mn[1], (i & 4) ? box.mx[2] : box.mn[2]};
            result.add(animation_box_of_point(q, tr, p));
        }
    }
    return result;
}

// MotionBlur / Transformed, Hittable.swift:543-555,578-601: the ray goes into the frame of
// transform.pow(time), the base is enumerated there, points and normals come back.
template <class R> struct MotionEnum final : Enum<R> {
    std::unique_ptr<Enum<R>> impl;
    Xf<R> xf;
    MotionEnum(const TraceCtx<R> &ctx, const Shape &s, Ray<R> ray, int object) {
        Quat q = quat_pow(Quat{s.m_quat[0], s.m_quat[1], s.m_quat[2], s.m_quat[3]}, ctx.time);
        double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1}, c0[3], c1[3], c2[3];
        quat_rotate(q, ex, c0); quat_rotate(q, ey, c1); quat_rotate(q, ez, c2);
        xf.c0 = v3<R>(c0); xf.c1 = v3<R>(c1); xf.c2 = v3<R>(c2);
        xf.t = V3<R>(R(ctx.time * s.m_trans[0]), R(ctx.time * s.m_trans[1]), R(ctx.time * s.m_trans[2]));
        impl = make_enum<R>(ctx, s.base, inverse_transform(ray, xf), object);
    }
    bool hasNext() const override { return impl->hasNext(); }
    void move() override { impl->move(); }
    bool isExit() const override { return impl->isExit(); }
    R t() const override { return impl->t(); }
    V3<R> point() const override { return xf.rot(impl->point()) + xf.t; }
    V3<R> normal() const override { return xf.rot(impl->normal()); }
    uint64_t material_offset() const override { return impl->material_offset(); }
    void texture_coordinates(R &u, R &v) const override { impl->texture_coordinates(u, v); }
};

template <class R> std::unique_ptr<Enum<R>> make_enum(const TraceCtx<R> &ctx, int shape, Ray<R> ray, int object) {
    const Shape &s = ctx.scene->shapes[shape];
    if (ctx.counters && s.kind <= QUAD) ctx.counters->prim_tests[s.kind]++;
    switch (s.kind) {
    case SPHERE: return std::make_unique<SphereEnum<R>>(s, ray);
    case CYLINDER: return std::make_unique<CylinderEnum<R>>(s, ray);
    case CUBOID: return std::make_unique<CuboidEnum<R>>(s, ray);
    case QUAD: return std::make_unique<QuadEnum<R>>(s, ray);
    case COMPOSITION:
        if (ctx.scene->kit_convention) return std::make_unique<KitCompositionEnum<R>>(ctx, s, ray, object);
        return std::make_unique<CompositionEnum<R>>(ctx, s, ray, object);
    case MOTION: return std::make_unique<MotionEnum<R>>(ctx, s, ray, object);
    default: return std::make_unique<CdvEnum<R>>(ctx, s, ray, object);
    }
}

// ---------------------------------------------------------------- closest hit
// HitTesting.h:17-33
template <class R> struct HitInfo {
    R t = 0; V3<R> point, normal; bool front = true; uint64_t material_offset = 0; R u = 0, v = 0; int object = -1;
    void set_normal(V3<R> front_normal, V3<R> dir) {
        if (dot(front_normal, dir) > 0) { normal = -front_normal; front = false; }
        else { normal = front_normal; front = true; }
    }
};

// Shaders.metal:171-203: the first enumerated hit with lo <= t and t inside the upper
// bound wins for this object.  The upper bound is inclusive for the caller's own tmax
// (Shaders.metal:188) and exclusive once a closer hit of another object has shrunk it
// (half-open Range of Hittable.swift:643).
// A hit is a finite distance.  The two reference renderers part ways here: RayTracingKit's range is
// 0.001 ..< infinity (Camera.swift:117 with the half-open Range of Hittable.swift:111-122), which no event at
// +infinity can lie in; Shaders.metal:187-190 compares `distance <= maxDistance` against an infinite maximum and so
// accepts the open end of a quad's half space (RenderableImpl.h:416-417) for a ray that leaves the quad's back -
// shading the plane point once more.  The Kit is the renderer the images are judged against, so its rule is kept.
template <class R, class E> bool first_in_range(E &e, Ray<R> ray, R lo, R hi, bool hi_open, int object, HitInfo<R> &out) {
    for (; e.hasNext(); e.move()) {
        R d = e.t();
        bool ok = d >= lo && (hi_open ? d < hi : d <= hi) && std::isfinite(d);
        if (ok) {
            out.t = d;
            out.point = e.point();
            out.set_normal(e.normal(), ray.direction);
            out.material_offset = e.material_offset();
            e.texture_coordinates(out.u, out.v);
            out.object = object;
            return true;
        }
    }
    return false;
}

template <class R> bool hit_object(const TraceCtx<R> &ctx, int object, Ray<R> ray, R lo, R hi, bool hi_open, HitInfo<R> &out) {
    int shape = ctx.scene->objects[object];
    const Shape &s = ctx.scene->shapes[shape];
    if (ctx.counters && s.kind <= QUAD) ctx.counters->prim_tests[s.kind]++;
    switch (s.kind) {   // plain primitives: enumerator on the stack
    case SPHERE: { SphereEnum<R> e(s, ray); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
    case CYLINDER: { CylinderEnum<R> e(s, ray); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
    case CUBOID: { CuboidEnum<R> e(s, ray); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
    case QUAD: { QuadEnum<R> e(s, ray); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
    case COMPOSITION:
        if (ctx.scene->kit_convention) { KitCompositionEnum<R> e(ctx, s, ray, object); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
        else { CompositionEnum<R> e(ctx, s, ray, object); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
    case MOTION: { MotionEnum<R> e(ctx, s, ray, object); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
    default: { CdvEnum<R> e(ctx, s, ray, object); return first_in_range(e, ray, lo, hi, hi_open, object, out); }
    }
}

// AABB.swift:78-93: axes with non-finite slab distances are skipped; a miss is
// tEntry >= tExit; no clipping against the ray's own range.
template <class R> bool box_hit(const Box &b, Ray<R> ray) {
    R tEntry = -kInf<R>(), tExit = kInf<R>();
    for (int a = 0; a < 3; a++) {
        R t1 = (R(b.mn[a]) - ray.origin[a]) / ray.direction[a];
        R t2 = (R(b.mx[a]) - ray.origin[a]) / ray.direction[a];
        if (std::isfinite(t1) && std::isfinite(t2)) {
            tEntry = std::max(tEntry, std::min(t1, t2));
            tExit = std::min(tExit, std::max(t1, t2));
        }
    }
    return !(tEntry >= tExit);
}

// Hittable.swift:637-645: left child first, right child with the range shrunk to the left
// hit, `right ?? left`.
template <class R> bool hit_node(const TraceCtx<R> &ctx, int ref, Ray<R> ray, R lo, R hi, bool hi_open, HitInfo<R> &out) {
    if (ref < 0) return hit_object(ctx, ~ref, ray, lo, hi, hi_open, out);
    const BvhNode &n = ctx.scene->nodes[ref];
    if (ctx.counters) ctx.counters->box_tests++;
    if (!box_hit(n.box, ray)) return false;
    HitInfo<R> l, r;
    bool hl = hit_node(ctx, n.left, ray, lo, hi, hi_open, l);
    bool hr = hl ? hit_node(ctx, n.right, ray, lo, l.t, true, r) : hit_node(ctx, n.right, ray, lo, hi, hi_open, r);
    if (hr) { out = r; return true; }
    if (hl) { out = l; return true; }
    return false;
}

template <class R> bool scene_hit(const TraceCtx<R> &ctx, Ray<R> ray, R lo, R hi, HitInfo<R> &out) {
    if (ctx.counters) ctx.counters->rays++;
    const Scene &sc = *ctx.scene;
    if (sc.objects.empty()) return false;
    bool h = hit_node(ctx, sc.root, ray, lo, hi, false, out);
    if (h && ctx.counters) ctx.counters->hits++;
    return h;
}

// ---------------------------------------------------------------- bounding boxes + BVH build
inline void xform_point(const Shape &s, const double p[3], double out[3]) {
    for (int r = 0; r < 3; r++) out[r] = s.rot[0 + r] * p[0] + s.rot[3 + r] * p[1] + s.rot[6 + r] * p[2] + s.trans[r];
}

// MetalRayTracer/Primitives.swift:29-33,112-125,163-180,253-260; CSG.swift:21-23,60-66,102-108;
// ConstantDensityVolume.swift:21-23.
inline Box shape_box(const Scene &sc, int id) {
    const Shape &s = sc.shapes[id];
    Box b;
    switch (s.kind) {
    case SPHERE:
        for (int a = 0; a < 3; a++) { b.mn[a] = s.trans[a] - s.radius; b.mx[a] = s.trans[a] + s.radius; }
        break;
    case CYLINDER: {
        double n[3] = {s.rot[3], s.rot[4], s.rot[5]};    // R * (0,1,0)
        for (int a = 0; a < 3; a++) {
            double rp = std::sqrt(std::max(0.0, 1.0 - n[a] * n[a])) * s.radius;
            double bc = s.trans[a], tc = s.trans[a] + s.height * n[a];
            b.mn[a] = std::min(bc, tc) - rp;
            b.mx[a] = std::max(bc, tc) + rp;
        }
        break;
    }
    case CUBOID:
        for (int i = 0; i < 8; i++) {
            double p[3] = {(i & 1) ? s.size[0] : 0, (i & 2) ? s.size[1] : 0, (i & 4) ? s.size[2] : 0}, w[3];
            xform_point(s, p, w);
            b.add(w);
        }
        break;
    case QUAD:
        for (int i = 0; i < 4; i++) {
            double p[3];
            for (int a = 0; a < 3; a++) p[a] = s.q_origin[a] + ((i & 1) ? s.q_u[a] : 0) + ((i & 2) ? s.q_v[a] : 0);
            b.add(p);
        }
        break;
    case COMPOSITION:
        if (s.subtract) {
            b = shape_box(sc, s.children[0]);
        } else if (s.min_count == 1) {
            for (int c : s.children) b.add(shape_box(sc, c));
        } else {
            for (int a = 0; a < 3; a++) { b.mn[a] = -INFINITY; b.mx[a] = INFINITY; }
            for (int c : s.children) {
                Box cb = shape_box(sc, c);
                for (int a = 0; a < 3; a++) { b.mn[a] = std::max(b.mn[a], cb.mn[a]); b.mx[a] = std::min(b.mx[a], cb.mx[a]); }
            }
        }
        break;
    case CDV:
        b = shape_box(sc, s.base);
        break;
    case MOTION:
        b = animation_box(Quat{s.m_quat[0], s.m_quat[1], s.m_quat[2], s.m_quat[3]}, s.m_trans, shape_box(sc, s.base));
        break;
    }
    return b;
}

// Hittable.swift:609-631 with AABB.swift:59-66 for the axis.  Swift's sort is not
// guaranteed stable, so the order of equal centroids is unspecified there; a stable sort
// is one of the admissible orders.
inline int build_bvh(Scene &sc, std::vector<int> items) {
    if (items.size() == 1) return ~items[0];
    BvhNode node;
    for (int o : items) node.box.add(sc.shapes[sc.objects[o]].box);
    double sx = node.box.mx[0] - node.box.mn[0], sy = node.box.mx[1] - node.box.mn[1], sz = node.box.mx[2] - node.box.mn[2];
    int axis = sx > sy ? (sx > sz ? 0 : 2) : (sy > sz ? 1 : 2);
    std::stable_sort(items.begin(), items.end(), [&](int a, int b) {
        return sc.shapes[sc.objects[a]].box.center(axis) < sc.shapes[sc.objects[b]].box.center(axis);
    });
    size_t mid = (items.size() + 1) / 2;
    int idx = (int)sc.nodes.size();
    sc.nodes.push_back(node);
    int l = build_bvh(sc, std::vector<int>(items.begin(), items.begin() + mid));
    int r = build_bvh(sc, std::vector<int>(items.begin() + mid, items.end()));
    sc.nodes[idx].left = l;
    sc.nodes[idx].right = r;
    return idx;
}

inline void commit(Scene &sc) {
    for (size_t i = 0; i < sc.shapes.size(); i++) sc.shapes[i].box = shape_box(sc, (int)i);
    // AABB.swift:91 treats tEntry >= tExit as a miss, so a box with zero extent on an axis
    // (an axis-aligned quad) can never be hit.  Such boxes are widened so that the
    // reference's own test keeps them; this changes no hit result.
    for (auto &s : sc.shapes)
        for (int a = 0; a < 3; a++)
            if (s.box.mx[a] - s.box.mn[a] < 1e-4) { s.box.mn[a] -= 1e-4; s.box.mx[a] += 1e-4; }
    sc.nodes.clear();
    std::vector<int> all(sc.objects.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
    sc.root = all.empty() ? -1 : build_bvh(sc, all);
    sc.committed = true;
}

// ---------------------------------------------------------------- materials and textures
// MaterialsImpl.h:16-24 / Vector3D.swift:213-221
template <class R> R reflectance(R cos_theta, R eta_ratio) {
    R sqR0 = (1 - eta_ratio) / (1 + eta_ratio);
    R R0 = sqR0 * sqR0;
    R x = 1 + cos_theta;
    R x2 = x * x;
    R x4 = x2 * x2;
    return R0 + (1 - R0) * (x4 * x);
}
template <class R> V3<R> reflect(V3<R> v, V3<R> n) { return v - (2 * dot(v, n)) * n; }   // Vector3D.swift:152-154

// MaterialsImpl.h:26-47 / Vector3D.swift:156-173
template <class R> V3<R> refract_or_reflect(V3<R> v, V3<R> normal, R eta_ratio, R reflectance_random) {
    R cos_theta = std::max(dot(v, normal), R(-1));
    V3<R> r_perp = eta_ratio * (v - cos_theta * normal);
    R r_perp_len_sq = length_squared(r_perp);
    bool refracts = true;
    if (r_perp_len_sq > 1) refracts = false;
    else if (reflectance_random <= 0) refracts = false;
    else if (reflectance_random < 1 && reflectance(cos_theta, eta_ratio) > reflectance_random) refracts = false;
    if (!refracts) return reflect(v, normal);
    V3<R> r_parallel = normal * -(std::sqrt(1 - r_perp_len_sq));
    return r_perp + r_parallel;
}

// MaterialsImpl.h:53-58: normalised coordinates, nearest filter, clamp-to-edge (the
// sampler default); MTKTextureLoader's bottom-left origin puts v = 0 on the bottom row
// (MetalRayTracer/Texture.swift:35-37).  Rows are stored top to bottom; bytes are taken
// as linear values / 255 (the loader's sRGB handling is outside the reference sources).
template <class R> V3<R> sample_image(const Image &im, R u, R v) {
    int x = (int)std::floor(u * R(im.w));
    int y = (int)std::floor(v * R(im.h));
    x = std::min(std::max(x, 0), im.w - 1);
    y = std::min(std::max(y, 0), im.h - 1);
    const uint8_t *p = &im.rgb[3 * ((size_t)(im.h - 1 - y) * im.w + x)];
    return V3<R>(R(p[0]) / 255, R(p[1]) / 255, R(p[2]) / 255);
}

template <class R> R mix(R a, R b, R t) { return a + (b - a) * t; }

// MaterialsImpl.h:60-121
template <class R> R perlin_noise(const OrcPerlin &tex, V3<R> p) {
    auto hash = [&](int x, int y, int z) -> int {
        return tex.permutations[2][(tex.permutations[1][(tex.permutations[0][x] + y) & 255] + z) & 255];
    };
    auto fade = [](R t) { return t * t * t * (t * (R(6) * t - R(15)) + R(10)); };
    R fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    int xi0 = ((int)fx) & 255, yi0 = ((int)fy) & 255, zi0 = ((int)fz) & 255;
    int xi1 = (xi0 + 1) & 255, yi1 = (yi0 + 1) & 255, zi1 = (zi0 + 1) & 255;
    R tx = p.x - fx, ty = p.y - fy, tz = p.z - fz;
    R u = fade(tx), v = fade(ty), w = fade(tz);
    auto grad = [&](int xi, int yi, int zi, R x, R y, R z) {
        const float *g = tex.vectors[hash(xi, yi, zi)];
        return R(g[0]) * x + R(g[1]) * y + R(g[2]) * z;
    };
    R x0 = tx, x1 = tx - 1, y0 = ty, y1 = ty - 1, z0 = tz, z1 = tz - 1;
    R a = mix(grad(xi0, yi0, zi0, x0, y0, z0), grad(xi1, yi0, zi0, x1, y0, z0), u);
    R b = mix(grad(xi0, yi1, zi0, x0, y1, z0), grad(xi1, yi1, zi0, x1, y1, z0), u);
    R c = mix(grad(xi0, yi0, zi1, x0, y0, z1), grad(xi1, yi0, zi1, x1, y0, z1), u);
    R d = mix(grad(xi0, yi1, zi1, x0, y1, z1), grad(xi1, yi1, zi1, x1, y1, z1), u);
    return mix(mix(a, b, v), mix(c, d, v), w);
}

// MaterialsImpl.h:123-139
template <class R> V3<R> perlin_color(const OrcPerlin &tex, V3<R> point) {
    R t = 0;
    if (tex.turbulence == 0) {
        t = 1 + perlin_noise(tex, point * R(tex.frequency));
    } else {
        R f = R(tex.frequency), weight = 1;
        for (unsigned i = 0; i < tex.turbulence; i++) {
            t += perlin_noise(tex, point * f) * weight;
            weight *= R(0.5);
            f *= 2;
        }
        t = std::fabs(t);
    }
    V3<R> c0(R(tex.colors[0][0]), R(tex.colors[0][1]), R(tex.colors[0][2]));
    V3<R> c1(R(tex.colors[1][0]), R(tex.colors[1][1]), R(tex.colors[1][2]));
    return c0 + (c1 - c0) * t;
}

// ---- RayTracingKit textures, Texture.swift
// PerlinNoise, :184-223: Hermite weights, cell from Int() (truncation), fraction from floor().
template <class R> R kit_noise(const OrcKitPerlin &n, V3<R> p) {
    R u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
    auto hermitian = [](R x) { return x * x * (3 - 2 * x); };
    R uu = hermitian(u), vv = hermitian(v), ww = hermitian(w);
    long i = (long)p.x, j = (long)p.y, k = (long)p.z;
    R result = 0;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                int index = n.perm_x[(i + di) & 255] ^ n.perm_y[(j + dj) & 255] ^ n.perm_z[(k + dk) & 255];
                R weight = (di == 0 ? 1 - uu : uu) * (dj == 0 ? 1 - vv : vv) * (dk == 0 ? 1 - ww : ww);
                V3<R> g(R(n.vectors[index][0]), R(n.vectors[index][1]), R(n.vectors[index][2]));
                result += weight * dot(g, V3<R>(u - R(di), v - R(dj), w - R(dk)));
            }
    return result;
}
template <class R> R kit_noise_turbulence(const OrcKitPerlin &n, V3<R> point, int turbulence) {
    R accum = 0, weight = 1;
    V3<R> p = point;
    for (int t = 0; t < turbulence; t++) {
        accum += weight * kit_noise<R>(n, p);
        weight *= R(0.5);
        p = p * R(2);
    }
    return accum;
}
inline long kit_clip(long index, long size, bool wrap) {                    // :103-120
    if (index < 0) return wrap ? size + (index % size) : 0;
    if (index >= size) return wrap ? index % size : size - 1;
    return index;
}
template <class R> V3<R> kit_texture_color(const Scene &sc, int id, R u, R v, V3<R> point) {
    const OrcKitTexture &t = sc.kit_textures[(size_t)id];
    switch (t.kind) {
    case 1: {                                                                // CheckerTexture, :44-50
        R inv_scale = R(1) / R(t.scale);
        long xi = (long)std::floor(inv_scale * point.x), yi = (long)std::floor(inv_scale * point.y), zi = (long)std::floor(inv_scale * point.z);
        bool is_even = (xi + yi + zi) % 2 == 0;
        return kit_texture_color<R>(sc, is_even ? t.even : t.odd, u, v, point);
    }
    case 2: {                                                                // ImageTexture, :80-101
        const Image &im = sc.images[(size_t)t.image];
        long width = std::labs((long)t.width), height = std::labs((long)t.height);
        bool invert_x = t.width < 0, invert_y = t.height < 0;
        auto pixel = [&](long i, long j) {
            long ix = t.origin_y + kit_clip(i, height, t.wrap_y != 0) * (invert_y ? -1 : +1);
            long jx = t.origin_x + kit_clip(j, width, t.wrap_x != 0) * (invert_x ? -1 : +1);
            ix = std::min<long>(std::max<long>(ix, 0), im.h - 1);           // the reference asserts here
            jx = std::min<long>(std::max<long>(jx, 0), im.w - 1);
            const uint8_t *px = &im.rgb[3 * ((size_t)ix * im.w + (size_t)jx)];
            return V3<R>(R(px[0]), R(px[1]), R(px[2])) / R(255);
        };
        R i = v * R(height) - R(0.5), j = u * R(width) - R(0.5);
        long i1 = (long)std::floor(i), i2 = i1 + 1, j1 = (long)std::floor(j), j2 = j1 + 1;
        R a = i - R(i1), b = j - R(j1);
        V3<R> c1 = pixel(i1, j1) * (1 - b) + pixel(i1, j2) * b;
        V3<R> c2 = pixel(i2, j1) * (1 - b) + pixel(i2, j2) * b;
        return c1 * (1 - a) + c2 * a;
    }
    case 3: {                                                                // NoiseTexture, :145-148
        R grey = std::fabs(kit_noise_turbulence<R>(sc.kit_perlins[(size_t)t.perlin], R(t.scale) * point, 7));
        return V3<R>(grey, grey, grey);
    }
    case 4: {                                                                // Marble, :160-163
        R grey = (1 + std::sin(R(t.scale) * point.z + 10 * kit_noise_turbulence<R>(sc.kit_perlins[(size_t)t.perlin], point, 7))) * R(0.5);
        return V3<R>(grey, grey, grey);
    }
    default:
        return V3<R>(R(t.color[0]), R(t.color[1]), R(t.color[2]));
    }
}

template <class R> V3<R> material_color(const Scene &sc, const OrcMaterial &m, const HitInfo<R> &hit) {
    switch (m.kind) {
    case orc_material_kind_lambertian_textured:
    case orc_material_kind_metal_textured:
        return sample_image<R>(sc.images[m.texture], hit.u, hit.v);
    case orc_material_kind_lambertian_perlin_noise:
    case orc_material_kind_metal_perlin_noise:
        return perlin_color<R>(sc.perlins[m.texture], hit.point);
    case orc_material_kind_lambertian_kit_texture:
    case orc_material_kind_emissive_kit_texture:
        return kit_texture_color<R>(sc, (int)m.texture, hit.u, hit.v, hit.point);
    default:
        return V3<R>(R(m.albedo[0]), R(m.albedo[1]), R(m.albedo[2]));
    }
}

template <class R> struct ScatterResult { V3<R> emitted, attenuation; Ray<R> scattered; };

// MaterialsImpl.h:147-215.  Where the two reference renderers differ: metal absorbs when the
// fuzzed reflection points strictly below the surface (< 0, MaterialsImpl.h:167) or, under the
// Kit convention, on it too (<= 0, Material.swift:55); the scattered direction is normalised
// either way (MaterialsImpl.h:169; every Kit shape is invariant to the direction's length).
template <class R> bool scatter(const Scene &sc, const OrcMaterial &m, Ray<R> ray, const HitInfo<R> &hit, Rng &rng, ScatterResult<R> &res) {
    res.emitted = V3<R>();
    res.attenuation = V3<R>();
    switch (m.kind) {
    case orc_material_kind_lambertian_colored:
    case orc_material_kind_lambertian_textured:
    case orc_material_kind_lambertian_perlin_noise:
    case orc_material_kind_lambertian_kit_texture:
        while (true) {
            V3<R> d = hit.normal + random_unit_vector_3d<R>(rng);
            R len_sq = length_squared(d);
            if (len_sq > R(kMinVectorLengthSquared)) {
                res.scattered = {hit.point, d / std::sqrt(len_sq)};
                break;
            }
        }
        res.attenuation = material_color(sc, m, hit);
        return true;
    case orc_material_kind_metal_colored:
    case orc_material_kind_metal_textured:
    case orc_material_kind_metal_perlin_noise: {
        V3<R> reflected = reflect(ray.direction, hit.normal) + R(m.fuzz) * random_unit_vector_3d<R>(rng);
        const R side = dot(reflected, hit.normal);
        if (sc.kit_convention ? side <= 0 : side < 0) return false;
        res.attenuation = material_color(sc, m, hit);
        res.scattered = {hit.point, normalize(reflected)};
        return true;
    }
    case orc_material_kind_dielectric: {
        R eta = hit.front ? R(1) / R(m.refraction_index) : R(m.refraction_index);
        V3<R> refracted = refract_or_reflect(ray.direction, hit.normal, eta, rng.random_f<R>());
        res.attenuation = V3<R>(1, 1, 1);
        res.scattered = {hit.point, refracted};
        return true;
    }
    case orc_material_kind_emissive_colored:
        res.emitted = V3<R>(R(m.albedo[0]), R(m.albedo[1]), R(m.albedo[2]));
        return false;
    case orc_material_kind_emissive_kit_texture:
        res.emitted = material_color(sc, m, hit);
        return false;
    case orc_material_kind_isotropic_colored:
        res.scattered = {hit.point, random_unit_vector_3d<R>(rng)};
        res.attenuation = V3<R>(R(m.albedo[0]), R(m.albedo[1]), R(m.albedo[2]));
        return true;
    }
    return false;
}

// ---------------------------------------------------------------- camera and integrator
// Shaders.metal:20-68 (same construction as Camera.swift:29-60).
template <class R> struct Camera {
    R image_w, image_h;
    V3<R> center, viewport_u, viewport_v, viewport_center, defocus_u, defocus_v;
    bool defocus;
    Camera(int w, int h, const OrcCameraConfig &c) {
        image_w = R(w); image_h = R(h);
        R half_fov = R(c.vertical_FOV) * kPi<R>() / 360;
        R viewport_height = 2 * std::tan(half_fov) * R(c.focus_distance);
        R viewport_width = viewport_height * R(w) / R(h);
        V3<R> from(R(c.look_from[0]), R(c.look_from[1]), R(c.look_from[2]));
        V3<R> at(R(c.look_at[0]), R(c.look_at[1]), R(c.look_at[2]));
        V3<R> up(R(c.up[0]), R(c.up[1]), R(c.up[2]));
        center = from;
        V3<R> w_ = normalize(from - at);
        V3<R> u_ = normalize(cross(up, w_));
        V3<R> v_ = cross(w_, u_);
        viewport_u = viewport_width * u_;
        viewport_v = -viewport_height * v_;
        viewport_center = from - R(c.focus_distance) * w_;
        R defocus_radius = R(c.focus_distance) * std::tan(R(c.defocus_angle) * kPi<R>() / 360);
        defocus_u = u_ * defocus_radius;
        defocus_v = v_ * defocus_radius;
        defocus = c.defocus_angle > 0;    // Camera.swift:54-59,108: no draw without defocus
    }
    // Draw order Camera.swift:97-100 / Shaders.metal:50-54: origin, offset x, offset y.
    Ray<R> get_ray(uint32_t gx, uint32_t gy, Rng &rng) const {
        V3<R> origin = center;
        if (defocus) {
            R px, py;
            random_unit_vector_2d<R>(rng, px, py);
            origin = center + px * defocus_u + py * defocus_v;
        }
        R ox = rng.random_f<R>();
        R oy = rng.random_f<R>();
        R sx = (R(gx) + ox) / image_w - R(0.5);
        R sy = (R(gy) + oy) / image_h - R(0.5);
        V3<R> sample = viewport_center + sx * viewport_u + sy * viewport_v;
        return {origin, normalize(sample - origin)};
    }
};

// Shaders.metal:76-86 (including its a = 0.5*y + 1.0).
template <class R> V3<R> background_color(uint32_t mode, V3<R> dir) {
    if (mode == 0) return V3<R>();
    R a = R(0.5) * dir.y + R(1.0);
    return (R(1.0) - a) * V3<R>(1, 1, 1) + a * V3<R>(R(0.5), R(0.7), R(1.0));
}

constexpr double kTMin = 1e-3;   // Camera.swift:117

// Camera.swift:113-131 (recursive form), with the background of Shaders.metal:104 on a
// miss.  At depth 0 the tail contributes nothing while emission gathered at shallower
// bounces is kept (Camera.swift:114-122; Shaders.metal:127-128 would drop it).
template <class R> V3<R> ray_color(const Scene &sc, uint32_t background, Ray<R> ray, int depth, Rng &rng, Counters *counters, double time = 0) {
    if (depth <= 0) return V3<R>();
    TraceCtx<R> ctx{&sc, counters, nullptr, rng.pcg.state, time};
    HitInfo<R> hit;
    if (!scene_hit<R>(ctx, ray, R(kTMin), kInf<R>(), hit)) return background_color<R>(background, ray.direction);
    const OrcMaterial &m = sc.materials[hit.material_offset];
    // the texture-backed RayTracingKit kinds count with the kinds they scatter like
    if (counters) counters->shades[m.kind == orc_material_kind_lambertian_kit_texture ? 1 : (m.kind == orc_material_kind_emissive_kit_texture ? 8 : (m.kind < 10 ? m.kind : 0))]++;
    ScatterResult<R> res;
    bool did = scatter<R>(sc, m, ray, hit, rng, res);
    if (!did) return res.emitted;
    return res.emitted + res.attenuation * ray_color<R>(sc, background, res.scattered, depth - 1, rng, counters, time);
}

}  // namespace orc
