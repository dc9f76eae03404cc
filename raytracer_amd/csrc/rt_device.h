// rt_device.h - device-side building blocks of the path tracer (gfx950).
//
// Stages, in the order a path goes through them:
//   raygen  camera ray of one (pixel, sample)            Shaders.metal:20-68
//   extend  BVH traversal + closed-form primitive tests   RenderableImpl.h, Hittable.swift:637-645
//   shade   hit completion + material scatter             MaterialsImpl.h:147-215
// The kernels in rt_kernels.hip compose them either as separate launches over path
// queues or as wave-wide phases of one persistent kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_shared.h"

#define RT_DEV __device__ __forceinline__

// ------------------------------------------------------------------ small vector type
struct f3 { float x, y, z; };
RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RT_DEV f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV f3 madd(f3 a, float s, f3 b) { return mk3(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }  // a*s + b
RT_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RT_DEV f3 normalize3(f3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }
RT_DEV bool finite(float x) { return fabsf(x) < __builtin_huge_valf(); }
// Natural logarithm of a number in (2^-24, 1] (the free-flight draw of a volume): the hardware's log2 times ln 2,
// about 2 ulp, where the library routine spends two dozen instructions on the last bit.
RT_DEV float log_unit(float x) { return __builtin_amdgcn_logf(x) * 0.69314718056f; }

#define RT_INF __builtin_huge_valf()
#define RT_PI 3.14159265358979323846f

// ------------------------------------------------------------------ random numbers
// PCG32, Impl/RNG.h:31-52.
struct Pcg { uint64_t state, inc; };

RT_DEV uint32_t pcg_next(Pcg &g) {
    uint64_t old = g.state;
    g.state = old * 6364136223846793005ULL + g.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}

RT_DEV Pcg pcg_seed(uint64_t initstate, uint64_t initseq) {   // RNG.h:37-44
    Pcg g;
    g.state = 0u;
    g.inc = (initseq << 1u) | 1u;
    pcg_next(g);
    g.state += initstate;
    pcg_next(g);
    return g;
}

// Uniform in [0,1): the top 24 bits.  (RNG.h:54-56 scales all 32 bits, which can round to
// 1.0f and then feeds log(0) to the volume sampler.)
RT_DEV float rnd(Pcg &g) { return (float)(pcg_next(g) >> 8) * (1.0f / 16777216.0f); }

// WyRand output number k of `seed` (RayTracingKit/PRNG.swift:17-21): the generator's state
// after k+1 steps is seed + (k+1)*C, so every output is addressable by its index.
RT_DEV uint64_t wyrand_at(uint64_t seed, uint64_t k) {
    uint64_t s = seed + (k + 1) * 0xa0761d6478bd642fULL;
    uint64_t t = s ^ 0xe7037ed1a0b428dbULL;
    return __umul64hi(s, t) ^ (s * t);
}

// Generator of one (pixel, sample): per-sample seed = WyRand output `sample` of rng_seed
// (the reference draws a new 64-bit seed for every one-sample pass,
// ContentView.swift:201-202), pixel offsets as Shaders.metal:141-143.
RT_DEV Pcg sample_stream(uint64_t rng_seed, uint32_t sample, uint32_t x, uint32_t y) {
    uint64_t pass = wyrand_at(rng_seed, sample);
    uint32_t lo = (uint32_t)pass, hi = (uint32_t)(pass >> 32);
    return pcg_seed((uint64_t)(uint32_t)(x * 5569u + lo), (uint64_t)(uint32_t)(y * 2707u + hi));
}

// rnd(g) * 2 - 1 in one rounding: the scaling by 2^-24 and the doubling are exact, so this is
// the same float as the two-step form.
RT_DEV float rnd_signed(Pcg &g) { return fmaf((float)(pcg_next(g) >> 8), 1.0f / 8388608.0f, -1.0f); }

#define RT_MIN_LEN_SQ 1e-24f   // RNG.h:13

// RNG.h:59-68: rejection from the square, normalised onto the circle.
RT_DEV void random_unit_vector_2d(Pcg &g, float &ox, float &oy) {
    while (true) {
        float x = rnd_signed(g);
        float y = rnd_signed(g);
        float len_sq = x * x + y * y;
        if (RT_MIN_LEN_SQ < len_sq && len_sq <= 1.0f) {
            float inv = __builtin_amdgcn_rsqf(len_sq);
            ox = x * inv;
            oy = y * inv;
            return;
        }
    }
}

// RayTracingKit/Vector3D.swift:113-127 (RNG.h:70-80 forgets z*z in the length).
RT_DEV f3 random_unit_vector_3d(Pcg &g) {
    while (true) {
        float x = rnd_signed(g);
        float y = rnd_signed(g);
        float z = rnd_signed(g);
        float len_sq = fmaf(x, x, fmaf(y, y, z * z));
        if (RT_MIN_LEN_SQ < len_sq && len_sq <= 1.0f) {
            float inv = __builtin_amdgcn_rsqf(len_sq);
            return mk3(x * inv, y * inv, z * inv);
        }
    }
}

// ------------------------------------------------------------------ ray generation
// Shaders.metal:50-67 with the draw order of Camera.swift:97-100 (origin, offset x,
// offset y) and no draw when there is no defocus (Camera.swift:108).
RT_DEV void camera_ray(const RtDevCamera &c, uint32_t gx, uint32_t gy, Pcg &g, f3 &origin, f3 &dir) {
    origin = ld3(c.center);
    if (c.defocus) {
        float px, py;
        random_unit_vector_2d(g, px, py);
        origin = madd(ld3(c.defocus_u), px, madd(ld3(c.defocus_v), py, origin));
    }
    float off_x = rnd(g);
    float off_y = rnd(g);
    float sx = ((float)gx + off_x) * c.inv_w - 0.5f;
    float sy = ((float)gy + off_y) * c.inv_h - 0.5f;
    f3 sample = madd(ld3(c.viewport_u), sx, madd(ld3(c.viewport_v), sy, ld3(c.viewport_center)));
    dir = normalize3(sample - origin);
}

// Shaders.metal:76-86 (its a = 0.5*y + 1.0 kept as is).
RT_DEV f3 background_color(uint32_t mode, f3 dir) {
    if (mode == 0) return mk3(0, 0, 0);
    float a = fmaf(0.5f, dir.y, 1.0f);
    float w = 1.0f - a;
    return mk3(fmaf(a, 0.5f, w), fmaf(a, 0.7f, w), fmaf(a, 1.0f, w));
}

// ------------------------------------------------------------------ primitive intervals
// Every primitive is a convex volume (the quad a half space): along a ray it is one
// interval [t0, t1].  tag0/tag1 say which surface the ends lie on (cylinder: 0 bottom cap,
// 1 top cap, 2 tube; cuboid: face index left,right,bottom,top,back,front).
struct Span { float t0, t1; uint32_t tag0, tag1; };
RT_DEV Span empty_span() { Span s; s.t0 = RT_INF; s.t1 = -RT_INF; s.tag0 = s.tag1 = 0; return s; }
RT_DEV bool span_valid(const Span &s) { return s.t0 <= s.t1; }

// UNI = the address is the same in every lane of the wave and the scene data is never written while
// a kernel runs: the load is typed as one from constant memory, which lets it go through the scalar
// cache into scalar registers (one request per wave instead of one per lane).
template <bool UNI, class T> RT_DEV T ld(const T *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (UNI) return *(const __attribute__((address_space(4))) T *)(uintptr_t)p;
#endif
    return *p;
}

struct XPrim { f3 c0, c1, c2, t; float p0, p1, p2, p3; };
template <bool UNI = false>
RT_DEV XPrim load_xprim(const RtDevScene &s, uint32_t idx) {
    const float4 *q = (const float4 *)s.xprims + 4u * idx;
    float4 a = ld<UNI>(q), b = ld<UNI>(q + 1), c = ld<UNI>(q + 2), d = ld<UNI>(q + 3);
    XPrim x;
    x.c0 = mk3(a.x, a.y, a.z);
    x.c1 = mk3(a.w, b.x, b.y);
    x.c2 = mk3(b.z, b.w, c.x);
    x.t = mk3(c.y, c.z, c.w);
    x.p0 = d.x; x.p1 = d.y; x.p2 = d.z; x.p3 = d.w;
    return x;
}
RT_DEV f3 rot_t(const XPrim &x, f3 v) { return mk3(dot(x.c0, v), dot(x.c1, v), dot(x.c2, v)); }     // transpose(R) * v
RT_DEV f3 rot(const XPrim &x, f3 v) { return madd(x.c0, v.x, madd(x.c1, v.y, x.c2 * v.z)); }          // R * v

// Impl/RenderableImpl.h:30-56.  The discriminant is taken as a*(r^2 - |oc - (b/a) d|^2):
// the same number as b^2 - a*c without subtracting two large squares.
template <bool UNI = false>
RT_DEV Span sphere_span(const RtDevScene &s, uint32_t idx, f3 o, f3 d) {
    float4 g = ld<UNI>((const float4 *)s.spheres + idx);
    f3 oc = o - mk3(g.x, g.y, g.z);
    float a = dot(d, d);
    float b = dot(d, oc);
    float inv_a = rcp(a);
    float ba = b * inv_a;
    f3 l = madd(d, -ba, oc);
    float h = fmaf(g.w, g.w, -dot(l, l));
    Span sp = empty_span();
    if (h >= 0.0f) {
        float root = fsqrt(h * a);
        sp.t0 = (-b - root) * inv_a;
        sp.t1 = (-b + root) * inv_a;
    }
    return sp;
}

// One slab of a box/cylinder in the local frame: RenderableImpl.h:102-139 and :269-308.
RT_DEV void slab(float size, float origin, float direction, float &tin, float &tout, bool &in_is_low) {
    float inv = rcp(direction);
    float tb = -origin * inv;
    float tt = (size - origin) * inv;
    if (finite(tb) && finite(tt)) {
        in_is_low = !(tb > tt);
        tin = fminf(tb, tt);
        tout = fmaxf(tb, tt);
    } else if (origin >= 0.0f && origin < size) {
        in_is_low = true;
        tin = -RT_INF; tout = RT_INF;
    } else {
        in_is_low = true;
        tin = RT_INF; tout = -RT_INF;
    }
}

// Impl/RenderableImpl.h:89-255 (local frame: axis +y, base at the origin).  Tube
// coefficients from the xz components alone: a = d.x^2 + d.z^2 equals d.d - d.y^2
// (Hittable.swift:264) without the cancellation.
RT_DEV Span cylinder_span(const XPrim &x, f3 o, f3 d) {
    float radius = x.p0, height = x.p1;
    f3 lo = rot_t(x, o - x.t), ld = rot_t(x, d);
    float p_in, p_out;
    bool low;
    slab(height, lo.y, ld.y, p_in, p_out, low);
    uint32_t p_in_tag = low ? 0u : 1u, p_out_tag = low ? 1u : 0u;

    float t_in, t_out;
    float a = fmaf(ld.x, ld.x, ld.z * ld.z);
    float b = fmaf(ld.x, lo.x, ld.z * lo.z);
    float oo = fmaf(lo.x, lo.x, lo.z * lo.z);
    bool parallel = true;
    t_in = RT_INF; t_out = -RT_INF;
    if (a > 0.0f) {
        float inv_a = rcp(a);
        float ba = b * inv_a;
        float lx = fmaf(-ba, ld.x, lo.x), lz = fmaf(-ba, ld.z, lo.z);
        float h = fmaf(radius, radius, -fmaf(lx, lx, lz * lz));
        if (h < 0.0f) {
            parallel = false;      // D < 0: no intersection with the tube
        } else {
            float root = fsqrt(h * a);
            float t1 = (-b - root) * inv_a, t2 = (-b + root) * inv_a;
            if (finite(t1) && finite(t2)) { t_in = t1; t_out = t2; parallel = false; }
        }
    }
    if (parallel) {   // RenderableImpl.h:190-200
        if (oo > radius * radius) { t_in = RT_INF; t_out = -RT_INF; }
        else { t_in = -RT_INF; t_out = RT_INF; }
    }
    Span sp;
    if (p_in > t_in) { sp.t0 = p_in; sp.tag0 = p_in_tag; } else { sp.t0 = t_in; sp.tag0 = 2u; }       // :222
    if (p_out < t_out) { sp.t1 = p_out; sp.tag1 = p_out_tag; } else { sp.t1 = t_out; sp.tag1 = 2u; }  // :223
    return sp;
}

// Impl/RenderableImpl.h:257-334 (box [0,size] in the local frame).
RT_DEV Span cuboid_span(const XPrim &x, f3 o, f3 d) {
    f3 lo = rot_t(x, o - x.t), ld = rot_t(x, d);
    float xi, xo, yi, yo, zi, zo;
    bool xl, yl, zl;
    slab(x.p0, lo.x, ld.x, xi, xo, xl);
    slab(x.p1, lo.y, ld.y, yi, yo, yl);
    slab(x.p2, lo.z, ld.z, zi, zo, zl);
    uint32_t xin = xl ? 0u : 1u, yin = yl ? 2u : 3u, zin = zl ? 4u : 5u;
    Span sp;
    // :331-332, ties go to the later axis
    if (xi > yi) { if (xi > zi) { sp.t0 = xi; sp.tag0 = xin; } else { sp.t0 = zi; sp.tag0 = zin; } }
    else { if (yi > zi) { sp.t0 = yi; sp.tag0 = yin; } else { sp.t0 = zi; sp.tag0 = zin; } }
    if (xo < yo) { if (xo < zo) { sp.t1 = xo; sp.tag1 = xin ^ 1u; } else { sp.t1 = zo; sp.tag1 = zin ^ 1u; } }
    else { if (yo < zo) { sp.t1 = yo; sp.tag1 = yin ^ 1u; } else { sp.t1 = zo; sp.tag1 = zin ^ 1u; } }
    return sp;
}

// Impl/RenderableImpl.h:395-436: the quad as the half space behind its plane, cut to
// the parallelogram.  Record: origin (c0), v x w (c1), w x u (c2), normal (p0..p2), d (p3).
// plain: the quad stands alone (not an operand of a CSG shape, not the boundary of a volume): the far end of its half
// space is then no event of its own - a hit is a finite distance, the range of RayTracingKit is 0.001 ..< infinity
// (Camera.swift:117), where Shaders.metal:187-190 would take t = +inf for a hit and shade the plane point once more for a
// ray that leaves the quad's back - so both ends of the interval are the plane hit.
RT_DEV Span quad_span(const XPrim &x, f3 o, f3 d, float &alpha, float &beta, bool plain = false) {
    f3 normal = mk3(x.p0, x.p1, x.p2);
    float denom = dot(d, normal);
    float t = (x.p3 - dot(o, normal)) * rcp(denom);
    Span sp = empty_span();
    alpha = beta = 0.0f;
    if (finite(t)) {
        f3 p = madd(d, t, o) - x.c0;
        alpha = dot(x.c1, p);          // c1 = v x w, c2 = w x u (taken on the host): the triple products of RenderableImpl.h:418-426
        beta = dot(x.c2, p);
        if (!(alpha < 0.0f || alpha > 1.0f || beta < 0.0f || beta > 1.0f)) {
            sp.t0 = denom < 0.0f ? t : -RT_INF;
            sp.t1 = (denom < 0.0f && !plain) ? RT_INF : t;
        }
    }
    return sp;
}

template <bool UNI = false>
RT_DEV Span prim_span(const RtDevScene &s, uint32_t ref, f3 o, f3 d, bool plain = false) {
    uint32_t kind = (ref >> RT_KIND_SHIFT) & 7u, idx = ref & RT_INDEX_MASK;
    if (kind == RT_KIND_SPHERE) return sphere_span<UNI>(s, idx, o, d);
    XPrim x = load_xprim<UNI>(s, idx);
    if (kind == RT_KIND_CYLINDER) return cylinder_span(x, o, d);
    if (kind == RT_KIND_CUBOID) return cuboid_span(x, o, d);
    float al, be;
    return quad_span(x, o, d, al, be, plain);
}

// ------------------------------------------------------------------ hit completion
struct Surface {
    f3 normal;          // as enumerated by the reference (not flipped against the ray)
    float u, v;
    uint32_t material;
};

// Material of surface `tag` of primitive `ref` (what prim_surface reports, without the geometry).
RT_DEV uint32_t prim_material(const RtDevScene &s, uint32_t ref, uint32_t tag) {
    const uint32_t kind = (ref >> RT_KIND_SHIFT) & 7u, idx = ref & RT_INDEX_MASK;
    if (kind == RT_KIND_SPHERE) return s.sphere_mat[idx];
    const uint32_t *mats = s.xprim_mat + 6u * idx;
    return kind == RT_KIND_QUAD ? mats[0] : mats[tag];        // cylinder: 0 bottom, 1 top, 2 side; cuboid: its six faces
}

// Attributes of the boundary of primitive `ref` at parameter t on surface `tag`:
// normals and texture coordinates of RenderableImpl.h:72-86 (sphere), :113-121,141-146,
// 181-189,203-208 (cylinder), :350-392 (cuboid), :418-426 (quad).
RT_DEV Surface prim_surface(const RtDevScene &s, uint32_t ref, uint32_t tag, f3 o, f3 d, float t, bool want_uv) {
    uint32_t kind = (ref >> RT_KIND_SHIFT) & 7u, idx = ref & RT_INDEX_MASK;
    Surface r;
    r.u = r.v = 0.0f;
    f3 p = madd(d, t, o);
    if (kind == RT_KIND_SPHERE) {
        float4 g = ((const float4 *)s.spheres)[idx];
        r.normal = (p - mk3(g.x, g.y, g.z)) * rcp(g.w);
        r.material = s.sphere_mat[idx];
        if (want_uv) {
            f3 n = r.normal;
            if (s.sphere_rot) {
                const float4 *q = (const float4 *)s.sphere_rot + 3u * idx;
                float4 a = q[0], b = q[1], c = q[2];
                n = mk3(dot(mk3(a.x, a.y, a.z), r.normal), dot(mk3(b.x, b.y, b.z), r.normal), dot(mk3(c.x, c.y, c.z), r.normal));
            }
            // RenderableImpl.h:83 measures the angle from -x; RayTracingKit/Hittable.swift:188 from +x
            // with z mirrored: half a turn apart
            r.u = ((s.kit_convention ? atan2f(-n.z, n.x) : atan2f(n.z, -n.x)) + RT_PI) * (1.0f / (2.0f * RT_PI));
            r.v = acosf(fminf(fmaxf(-n.y, -1.0f), 1.0f)) * (1.0f / RT_PI);
        }
        return r;
    }
    XPrim x = load_xprim(s, idx);
    const uint32_t *mats = s.xprim_mat + 6u * idx;
    if (kind == RT_KIND_CYLINDER) {
        float radius = x.p0, height = x.p1;
        f3 lp = rot_t(x, p - x.t);
        if (tag == 2u) {
            r.normal = rot(x, mk3(lp.x, 0.0f, lp.z) * rcp(radius));
            r.material = mats[2];
            if (want_uv) {
                // RenderableImpl.h:205; the Kit's u = atan2(z, x) / 2pi may be negative (Hittable.swift:281)
                r.u = s.kit_convention ? atan2f(lp.z, lp.x) * (1.0f / (2.0f * RT_PI))
                                       : (atan2f(lp.z, -lp.x) + RT_PI) * (1.0f / (2.0f * RT_PI));
                r.v = lp.y * rcp(height);
            }
        } else {
            float flip = tag == 0u ? -1.0f : 1.0f;
            r.normal = x.c1 * flip;
            r.material = mats[tag];
            float inv2r = rcp(2.0f * radius);
            if (!want_uv) {
            } else if (s.kit_convention) {            // Hittable.swift:245,248: (x, z) on both caps
                r.u = fmaf(lp.x, inv2r, 0.5f);
                r.v = fmaf(lp.z, inv2r, 0.5f);
            } else {                           // RenderableImpl.h:113-121
                r.u = fmaf(lp.z * flip, inv2r, 0.5f);
                r.v = fmaf(lp.x, inv2r, 0.5f);
            }
        }
        return r;
    }
    if (kind == RT_KIND_CUBOID) {
        uint32_t axis = tag >> 1;
        f3 col = axis == 0u ? x.c0 : (axis == 1u ? x.c1 : x.c2);
        r.normal = (tag & 1u) ? col : -col;
        r.material = mats[tag];
        if (!want_uv) return r;
        f3 lp = rot_t(x, p - x.t) * mk3(rcp(x.p0), rcp(x.p1), rcp(x.p2));
        switch (tag) {
        case 0: r.u = 1.0f - lp.z; r.v = lp.y; break;          // left
        case 1: r.u = lp.z; r.v = lp.y; break;                 // right
        case 2: r.u = lp.x; r.v = 1.0f - lp.z; break;          // bottom
        case 3: r.u = lp.x; r.v = lp.z; break;                 // top
        case 4: r.u = 1.0f - lp.x; r.v = lp.y; break;          // back
        default: r.u = lp.x; r.v = lp.y; break;                // front
        }
        return r;
    }
    // quad
    r.normal = mk3(x.p0, x.p1, x.p2);
    r.material = mats[0];
    if (want_uv) {
        f3 q = p - x.c0;
        r.u = dot(x.c1, q);
        r.v = dot(x.c2, q);
    }
    return r;
}

// ------------------------------------------------------------------ CSG: interval-list evaluation
// A CSG shape is flattened on the host to its leaf primitives plus a truth table: bit m
// of `truth` says whether a point that lies inside exactly the leaves of mask m lies
// inside the shape.  Walking the leaf interval ends in increasing t (ties: the leaf with
// the larger index first, which is what the reference's nested merges do,
// RenderableImpl.h:532-548) and looking the mask up after each step yields the shape's own
// entry/exit events: wherever the looked-up bit changes.  This reproduces the depth
// counting of RenderableImpl.h:678-691 for Union / Intersection / Subtract at any nesting.
struct CsgIter {
    float tin[RT_CSG_LEAVES], tout[RT_CSG_LEAVES];
    uint64_t tags;            // 3 bits entry tag + 3 bits exit tag per leaf
    uint32_t entered, exited;
    // The truth table of a shape with at most 5 leaves is one word and stays in a register; a
    // larger one (6-8 leaves: up to 256 bits) is looked up in the shape's record in memory.
    // (Eight words selected by a compare chain made the compiler keep this whole struct in
    // scratch memory: it turned the chain into a select of addresses.)
    uint32_t truth0;
    uint32_t wide_csg;        // index of the CSG record when the table is wider than one word, else RT_REF_NONE
    bool inside;
    // current event
    float t;
    uint32_t leaf, tag;
    bool is_exit, leaf_exit;
};

RT_DEV bool csg_lookup(const RtDevScene &s, const CsgIter &it, uint32_t mask) {
    uint32_t w = it.truth0;
    if (it.wide_csg != RT_REF_NONE) w = s.csg[it.wide_csg].truth[mask >> 5];       // the mask differs from lane to lane
    return (w >> (mask & 31u)) & 1u;
}

// Starts the walk over `shape`: a CSG shape, or a primitive taken as a shape of one leaf (inside
// the shape = inside the leaf).  One loop serves both, so the primitive interval code exists once
// in a kernel however many kinds of object lead here.
// only_first: the caller reads nothing but leaf 0's interval (a plain primitive tested for its two boundaries), so
// the other seven are neither cleared nor written.
template <bool UNI = false>
RT_DEV void csg_begin(const RtDevScene &s, uint32_t shape, f3 o, f3 d, CsgIter &it, bool only_first = false) {
    const bool is_csg = ((shape >> RT_KIND_SHIFT) & 7u) == RT_KIND_CSG;
    const uint32_t idx = shape & RT_INDEX_MASK;
    const RtDevCsg *c = s.csg + (is_csg ? idx : 0u);
    const uint32_t n = is_csg ? ld<UNI>(&c->n_leaves) : 1u;
    it.tags = 0;
    it.entered = it.exited = 0;
    it.truth0 = is_csg ? ld<UNI>(&c->truth[0]) : 2u;         // one leaf: inside exactly when mask = 1
    it.wide_csg = n > 5u ? idx : RT_REF_NONE;
    if (!only_first) {
#pragma unroll
        for (int i = 0; i < RT_CSG_LEAVES; i++) { it.tin[i] = RT_INF; it.tout[i] = RT_INF; }
    }
    for (uint32_t i = 0; i < n; i++) {
        Span sp = prim_span<UNI>(s, is_csg ? ld<UNI>(&c->leaf[i]) : shape, o, d, only_first);
        bool ok = span_valid(sp);
        float a = ok ? sp.t0 : RT_INF, b = ok ? sp.t1 : RT_INF;
        if (only_first) {
            it.tin[0] = a; it.tout[0] = b;
        } else {
#pragma unroll
            for (int j = 0; j < RT_CSG_LEAVES; j++) {
                if (j == (int)i) { it.tin[j] = a; it.tout[j] = b; }
            }
        }
        it.tags |= (uint64_t)(sp.tag0 | (sp.tag1 << 3)) << (6u * i);
        if (!ok) { it.entered |= 1u << i; it.exited |= 1u << i; }     // nothing left to visit
    }
    it.inside = csg_lookup(s, it, 0u);
}

// Advances to the next event of the shape; false when none is left.  An end at +inf is
// never reached (the reference's merge cannot select it either, RenderableImpl.h:538).
RT_DEV bool csg_next(const RtDevScene &s, CsgIter &it) {
    while (true) {
        float best = RT_INF;
        int who = -1;
#pragma unroll
        for (int j = RT_CSG_LEAVES - 1; j >= 0; j--) {
            bool ent = (it.entered >> j) & 1u, ext = (it.exited >> j) & 1u;
            float cand = ext ? RT_INF : (ent ? it.tout[j] : it.tin[j]);
            if (cand < best) { best = cand; who = j; }
        }
        if (who < 0) return false;
        bool leaf_exit = (it.entered >> who) & 1u;
        if (leaf_exit) it.exited |= 1u << who; else it.entered |= 1u << who;
        uint32_t mask = it.entered & ~it.exited;
        bool now = csg_lookup(s, it, mask);
        if (now != it.inside) {
            it.inside = now;
            it.t = best;
            it.leaf = (uint32_t)who;
            it.leaf_exit = leaf_exit;
            it.is_exit = !now;
            it.tag = (uint32_t)(it.tags >> (6u * who + (leaf_exit ? 3u : 0u))) & 7u;
            return true;
        }
    }
}

// A shape of more than RT_CSG_LEAVES leaves (up to 64) is walked with the same rule and the same tie order, but
// nothing per leaf is kept: every step computes the leaves' intervals again and evaluates the shape's expression
// (a postfix program over "inside leaf i" bits) instead of looking a table up.  A step costs as many interval
// tests as the shape has leaves; the state is two 64-bit masks.  Such shapes are rare (the reference's scenes stop
// at five leaves); kernels only look for them in scenes that hold one (RtDevScene::n_wide_csg).
struct WideIter {
    f3 o, d;
    const uint32_t *leaves, *tokens;
    uint32_t n, n_tokens;
    uint64_t entered, exited;
    bool inside;
    // current event
    float t;
    uint32_t leaf, tag;
    bool is_exit, leaf_exit;
};

RT_DEV bool is_wide_csg(const RtDevScene &s, uint32_t shape) {
    return s.n_wide_csg != 0u && ((shape >> RT_KIND_SHIFT) & 7u) == RT_KIND_CSG && s.csg[shape & RT_INDEX_MASK].n_leaves > RT_CSG_LEAVES;
}

RT_DEV bool wide_inside(const WideIter &it, uint64_t mask) {
    uint64_t st = 0;                 // a stack of bits, top in bit 0
    for (uint32_t i = 0; i < it.n_tokens; i++) {
        const uint32_t tok = it.tokens[i];
        if (tok < 64u) { st = (st << 1) | ((mask >> tok) & 1u); continue; }
        const uint64_t b = st & 1u, a = (st >> 1) & 1u;
        const uint64_t r = tok == RT_CSG_TOKEN_UNION ? (a | b) : (tok == RT_CSG_TOKEN_INTERSECTION ? (a & b) : (a & ~b & 1u));
        st = ((st >> 2) << 1) | r;
    }
    return st & 1u;
}

RT_DEV void csg_begin_wide(const RtDevScene &s, uint32_t shape, f3 o, f3 d, WideIter &it) {
    const RtDevCsg &c = s.csg[shape & RT_INDEX_MASK];
    const uint32_t *rec = s.csg_pool + c.leaf[0];
    it.o = o; it.d = d;
    it.n = c.n_leaves;
    it.n_tokens = rec[2];
    it.leaves = rec + RT_CSG_POOL_HEADER;
    it.tokens = it.leaves + it.n;
    it.entered = it.exited = 0;
    it.inside = wide_inside(it, 0);
    it.t = 0.f; it.leaf = it.tag = 0; it.is_exit = it.leaf_exit = false;
}

RT_DEV bool csg_next(const RtDevScene &s, WideIter &it) {
    while (true) {
        float best = RT_INF;
        int who = -1;
        uint32_t tag = 0;
        for (int j = (int)it.n - 1; j >= 0; j--) {
            if ((it.exited >> j) & 1u) continue;
            const Span sp = prim_span(s, it.leaves[j], it.o, it.d);
            if (!span_valid(sp)) { it.entered |= 1ull << j; it.exited |= 1ull << j; continue; }     // nothing to visit
            const bool ent = (it.entered >> j) & 1u;
            const float cand = ent ? sp.t1 : sp.t0;
            if (cand < best) { best = cand; who = j; tag = ent ? sp.tag1 : sp.tag0; }
        }
        if (who < 0) return false;
        const bool leaf_exit = (it.entered >> who) & 1u;
        if (leaf_exit) it.exited |= 1ull << who; else it.entered |= 1ull << who;
        const bool now = wide_inside(it, it.entered & ~it.exited);
        if (now != it.inside) {
            it.inside = now;
            it.t = best;
            it.leaf = (uint32_t)who;
            it.leaf_exit = leaf_exit;
            it.is_exit = !now;
            it.tag = tag;
            return true;
        }
    }
}

// Leaf `leaf` of a CSG shape and whether its normals are turned round (it sits under an odd number of
// subtracted operands, RenderableImpl.h:712).
template <bool WIDE = true>
RT_DEV uint32_t csg_leaf(const RtDevScene &s, uint32_t csg_idx, uint32_t leaf, bool &flip) {
    const RtDevCsg &c = s.csg[csg_idx];
    if (WIDE && s.n_wide_csg != 0u && c.n_leaves > RT_CSG_LEAVES) {
        const uint32_t *rec = s.csg_pool + c.leaf[0];
        const uint64_t flips = rec[0] | ((uint64_t)rec[1] << 32);
        flip = (flips >> leaf) & 1u;
        return rec[RT_CSG_POOL_HEADER + leaf];
    }
    flip = (c.flip_mask >> leaf) & 1u;
    return c.leaf[leaf & (RT_CSG_LEAVES - 1u)];
}

// ------------------------------------------------------------------ motion blur
// The frame of a moving object at `time`: transform.pow(time) = (rotation^time, time *
// translation), RayTracingKit/Transform.swift:82-84,243-250.  Rays go into the frame
// (Hittable.swift:578-592), normals come back out (:70-73); t is the same on both sides.
struct MotionFrame { f3 c0, c1, c2, t; uint32_t base; bool rotates; };
RT_DEV MotionFrame motion_frame(const RtDevScene &s, uint32_t idx, float time) {
    const RtDevMotion m = s.motion[idx];
    MotionFrame f;
    f.t = mk3(m.tx * time, m.ty * time, m.tz * time);
    f.base = m.base;
    // a pure translation (inv_sin = 0: the identity rotation) needs no sine, cosine or matrix:
    // multiplying by the identity changes no bit of a vector
    f.rotates = m.inv_sin != 0.0f;
    f.c0 = mk3(1, 0, 0); f.c1 = mk3(0, 1, 0); f.c2 = mk3(0, 0, 1);
    if (f.rotates) {
        float a = m.half_angle * time;
        float w = cosf(a), k = sinf(a) * m.inv_sin;
        float x = m.qx * k, y = m.qy * k, z = m.qz * k;
        // columns of the rotation matrix of the unit quaternion (w, x, y, z)
        f.c0 = mk3(1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y + w * z), 2.0f * (x * z - w * y));
        f.c1 = mk3(2.0f * (x * y - w * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z + w * x));
        f.c2 = mk3(2.0f * (x * z + w * y), 2.0f * (y * z - w * x), 1.0f - 2.0f * (x * x + y * y));
    }
    return f;
}
RT_DEV f3 frame_in(const MotionFrame &f, f3 v) { return f.rotates ? mk3(dot(f.c0, v), dot(f.c1, v), dot(f.c2, v)) : v; }
RT_DEV f3 frame_out(const MotionFrame &f, f3 v) { return f.rotates ? madd(f.c0, v.x, madd(f.c1, v.y, f.c2 * v.z)) : v; }

// ------------------------------------------------------------------ closest hit of one object
// What extend remembers about the closest hit; `code` lets shade rebuild the surface:
//   primitive        tag
//   CSG              tag | (leaf & 7) << 3 | leaf_exit << 6 | (leaf >> 3) << 28 (only wide shapes have leaves beyond 7)
//   volume (CDV)     bit 31 set = scatter inside the medium; low bits = the base's code for
//                    the entry event (scatter) or the exit event (boundary)
struct Best { float t; uint32_t ref, code; };

#define RT_CODE_SCATTER 0x80000000u
#define RT_CODE_LEAF_EXIT_SHIFT 6
#define RT_CODE_INTERVAL_SHIFT 8        // the volume's interval counter: bits 8..27
#define RT_CODE_LEAF_HIGH_SHIFT 28
template <bool WIDE>
RT_DEV uint32_t code_leaf(uint32_t code) {
    return ((code >> 3) & 7u) | (WIDE ? ((code >> RT_CODE_LEAF_HIGH_SHIFT) & 7u) << 3 : 0u);
}

// Accept rule of Shaders.metal:182-199 for one candidate distance.
// tie_wins: a candidate at exactly the closest distance so far replaces it (set when the object under test comes
// before the holder of that distance in the order the reference would have met them, RayTracingKit/Hittable.swift:662-671).
// (A hit is a finite distance - RayTracingKit's range is 0.001 ..< infinity, Camera.swift:117 - and the only infinite
// interval ends are the open end of a quad's half space and of shapes built from quads: quad_span gives a quad that
// stands alone a closed interval, CSG walks never reach an end at infinity, the volume code checks its exit.)
RT_DEV bool in_range(float t, float tmin, float tmax_incl, bool found, float best, bool tie_wins = false) {
    return t >= tmin && (found ? (t < best || (tie_wins && t == best)) : t <= tmax_incl);
}

// Constant-density volume over a primitive (one interval [t_in, t_out]), RenderableImpl.h:737-777:
// one free-flight draw; a scatter inside the interval, else the interval's exit.
template <bool UNI>
RT_DEV void volume_over_primitive(const RtDevScene &s, uint32_t cdv_idx, uint32_t self, float t_in, float t_out, uint32_t tag_in,
                                  uint32_t tag_out, float neg_inv_density, float tmin, float tmax, uint64_t volume_key,
                                  const float *scripted, bool &found, Best &best, bool tie_wins = false) {
    const float t1 = fmaxf(0.0f, t_in), t2 = t_out;
    if (!(t1 < t2)) return;
    if (found && best.t <= t1) return;      // nothing inside the medium can beat the current hit
    float u;
    if (scripted) u = scripted[0];
    else { Pcg vr = pcg_seed(volume_key, (uint64_t)(ld<UNI>(s.cdv_obj + cdv_idx) + 1u)); u = rnd(vr); }
    const float t = fmaf(log_unit(1.0f - u), neg_inv_density, t1);
    if (!(t <= t2)) return;
    if (in_range(t, tmin, tmax, found, best.t, tie_wins)) { best.t = t; best.ref = self; best.code = RT_CODE_SCATTER | tag_in; found = true; }
    else if (t2 < RT_INF && in_range(t2, tmin, tmax, found, best.t, tie_wins)) { best.t = t2; best.ref = self; best.code = tag_out; found = true; }
}

RT_DEV uint32_t event_code(const CsgIter &it) { return it.tag | (it.leaf << 3) | ((it.leaf_exit ? 1u : 0u) << RT_CODE_LEAF_EXIT_SHIFT); }
template <class It>
RT_DEV uint32_t event_code(const It &it) {
    return it.tag | ((it.leaf & 7u) << 3) | ((it.leaf_exit ? 1u : 0u) << RT_CODE_LEAF_EXIT_SHIFT) | ((it.leaf >> 3) << RT_CODE_LEAF_HIGH_SHIFT);
}

// First boundary of a shape that is an acceptable hit (Shaders.metal:182-199 on the shape's event list).
template <class It>
RT_DEV void walk_boundary(const RtDevScene &s, It &it, uint32_t self, float tmin, float tmax, bool &found, Best &best, bool tie_wins = false) {
    while (csg_next(s, it)) {
        if (in_range(it.t, tmin, tmax, found, best.t, tie_wins)) {
            best.t = it.t; best.ref = self;
            best.code = event_code(it);
            found = true;
            break;
        }
        if (found && it.t >= best.t) break;
    }
}

// Constant-density volume over a shape's event list, RenderableImpl.h:737-777: one free-flight draw per
// interval of the base shape; a scatter inside the interval is followed by the interval's exit.
template <class It>
RT_DEV void walk_volume(const RtDevScene &s, It &it, uint32_t cdv_idx, uint32_t self, float neg_inv_density, float tmin, float tmax,
                        uint64_t volume_key, const float *scripted, uint32_t n_scripted, bool &found, Best &best, bool tie_wins = false) {
    Pcg vr;
    vr.state = vr.inc = 0;
    uint32_t draw = 0, interval = 0;
    while (csg_next(s, it)) {
        float t_entry = it.t;
        uint32_t entry_code = event_code(it);
        float t2 = RT_INF;
        uint32_t exit_code = 0;
        bool has_exit = csg_next(s, it);
        if (has_exit) { t2 = it.t; exit_code = event_code(it); }
        float t1 = fmaxf(0.0f, t_entry);
        if (found && best.t <= t1) return;      // nothing in this or a later interval can beat the current hit
        if (t1 < t2) {
            if (!scripted && draw == 0) vr = pcg_seed(volume_key, (uint64_t)(s.cdv_obj[cdv_idx] + 1u));
            float u = scripted ? scripted[draw % n_scripted] : rnd(vr);
            draw++;
            float t = fmaf(log_unit(1.0f - u), neg_inv_density, t1);
            if (t <= t2) {
                uint32_t iv = interval << RT_CODE_INTERVAL_SHIFT;
                if (in_range(t, tmin, tmax, found, best.t, tie_wins)) { best.t = t; best.ref = self; best.code = RT_CODE_SCATTER | iv | entry_code; found = true; return; }
                if (has_exit && t2 < RT_INF && in_range(t2, tmin, tmax, found, best.t, tie_wins)) { best.t = t2; best.ref = self; best.code = iv | exit_code; found = true; return; }
            }
        }
        if (!has_exit) return;
        interval++;
    }
}

// LEAN: the kernel is short of registers (texture code next to the intersection code), so a volume
// over a primitive goes through the general event walk instead of its own few lines.
// UNI: `ref` is the same in every lane (the list traversal), so the object's records are read once per wave.
// WIDE: the scene holds a CSG shape of more than RT_CSG_LEAVES leaves; only kernels built for such scenes carry that walk.
// DIRECT: plain primitives and volumes over one are tested by their own few lines instead of the general event walk (the
// list traversal, whose kernels have the registers for it).
// tie_wins: see in_range.
template <bool SIMPLE, bool MOTION = false, bool LEAN = false, bool UNI = false, bool WIDE = false, bool DIRECT = UNI>
RT_DEV void test_object(const RtDevScene &s, uint32_t ref, f3 o, f3 d, float tmin, float tmax, uint64_t volume_key,
                        const float *scripted, uint32_t n_scripted, bool &found, Best &best, float time = 0.0f, bool tie_wins = false) {
    uint32_t kind = (ref >> RT_KIND_SHIFT) & 7u, idx = ref & RT_INDEX_MASK;
    const uint32_t self = ref;           // what a hit is recorded under
    if (MOTION && kind == RT_KIND_MOTION) {
        MotionFrame f = motion_frame(s, idx, time);
        o = frame_in(f, o - f.t);
        d = frame_in(f, d);
        ref = f.base;
        kind = (ref >> RT_KIND_SHIFT) & 7u;
        idx = ref & RT_INDEX_MASK;
    }
    if (SIMPLE) {
        Span sp = sphere_span<UNI>(s, idx, o, d);
        if (span_valid(sp)) {
            if (in_range(sp.t0, tmin, tmax, found, best.t, tie_wins)) { best.t = sp.t0; best.ref = self; best.code = 0; found = true; }
            else if (in_range(sp.t1, tmin, tmax, found, best.t, tie_wins)) { best.t = sp.t1; best.ref = self; best.code = 0; found = true; }
        }
        return;
    }
    if (DIRECT && kind <= RT_KIND_QUAD) {
        // the list traversal knows the kind per wave: a plain primitive is tested directly, without
        // setting up the interval walk (which costs as much as a quad test)
        Span sp = prim_span<UNI>(s, ref, o, d, true);
        if (span_valid(sp)) {
            if (in_range(sp.t0, tmin, tmax, found, best.t, tie_wins)) { best.t = sp.t0; best.ref = self; best.code = sp.tag0; found = true; }
            else if (in_range(sp.t1, tmin, tmax, found, best.t, tie_wins)) { best.t = sp.t1; best.ref = self; best.code = sp.tag1; found = true; }
        }
        return;
    }
    // Every kind goes through one walk over the leaf intervals of a shape: a primitive is a shape
    // of one leaf, a volume walks its base shape.
    const bool volume = kind == RT_KIND_CDV;
    uint32_t shape = ref;
    float neg_inv_density = 0.0f;
    if (volume) {
        const RtDevCdv cv = ld<UNI && !MOTION>(s.cdv + idx);
        shape = cv.base;
        neg_inv_density = cv.neg_inv_density;
        if (DIRECT && ((shape >> RT_KIND_SHIFT) & 7u) != RT_KIND_CSG) {
            Span sp = prim_span<UNI>(s, shape, o, d);
            if (span_valid(sp))
                volume_over_primitive<UNI>(s, idx, self, sp.t0, sp.t1, sp.tag0, sp.tag1, neg_inv_density, tmin, tmax, volume_key, scripted, found, best, tie_wins);
            return;
        }
    }
    if (WIDE && is_wide_csg(s, shape)) {
        WideIter w;
        csg_begin_wide(s, shape, o, d, w);
        if (volume) walk_volume(s, w, idx, self, neg_inv_density, tmin, tmax, volume_key, scripted, n_scripted, found, best, tie_wins);
        else walk_boundary(s, w, self, tmin, tmax, found, best, tie_wins);
        return;
    }
    CsgIter it;
    csg_begin<UNI>(s, shape, o, d, it, !volume && kind <= RT_KIND_QUAD);
    if (!volume) {
        if (kind <= RT_KIND_QUAD) {
            // one interval: entry, else exit (Shaders.metal:182-199 on a two-hit enumerator)
            const float t0 = it.tin[0], t1 = it.tout[0];
            if (!(it.entered & 1u)) {
                if (in_range(t0, tmin, tmax, found, best.t, tie_wins)) { best.t = t0; best.ref = self; best.code = (uint32_t)it.tags & 7u; found = true; }
                else if (in_range(t1, tmin, tmax, found, best.t, tie_wins)) { best.t = t1; best.ref = self; best.code = ((uint32_t)it.tags >> 3) & 7u; found = true; }
            }
            return;
        }
        walk_boundary(s, it, self, tmin, tmax, found, best, tie_wins);
        return;
    }
    // Constant-density volume, RenderableImpl.h:737-777: one free-flight draw per interval of
    // the base shape; a scatter inside the interval is followed by the interval's exit.
    if (!LEAN && ((shape >> RT_KIND_SHIFT) & 7u) != RT_KIND_CSG) {
        // a primitive base has one interval: no event walk needed
        if (it.entered & 1u) return;
        volume_over_primitive<UNI>(s, idx, self, it.tin[0], it.tout[0], (uint32_t)it.tags & 7u, ((uint32_t)it.tags >> 3) & 7u, neg_inv_density,
                                   tmin, tmax, volume_key, scripted, found, best, tie_wins);
        return;
    }
    walk_volume(s, it, idx, self, neg_inv_density, tmin, tmax, volume_key, scripted, n_scripted, found, best, tie_wins);
}

// ------------------------------------------------------------------ BVH traversal (extend)
// Node: 16 floats = for the left then the right child (min, max) pairs per axis
// (x0,x1, y0,y1, z0,z1), then left ref, right ref, 2 pad.  The pairs let both slab
// distances of an axis come out of one packed FMA: t = plane * (1/d) + (-o/d).
// A reference with RT_REF_LEAF set names an object (kind | index), otherwise a node;
// RT_REF_NONE doubles as "traversal finished".
// The per-lane stack lives in LDS as stack[entry * RT_BLOCK + lane]: the 64 lanes of a
// wave hit 64 consecutive banks whatever their individual depth.
typedef float v2f __attribute__((ext_vector_type(2)));

struct Trav {
    f3 o, d;
    f3 inv_d, oid;         // 1/d and -o/d
    uint32_t cur;
    uint32_t sp;           // LDS byte address of the lane's next free stack slot (slots lie RT_BLOCK words apart)
    Best best;
    bool found;
};

// The stack pointer is kept as an absolute LDS address, so a push or a pop is the access itself and one add.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
RT_DEV uint32_t lds_address(const uint32_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(uintptr_t)(const lds_u32 *)p;
#else
    return (uint32_t)(uintptr_t)p;
#endif
}
RT_DEV uint32_t lds_load(uint32_t address) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const lds_u32 *)address;
#else
    return address;
#endif
}
RT_DEV void lds_store(uint32_t address, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *(lds_u32 *)address = v;
#else
    (void)address; (void)v;
#endif
}
#define RT_STACK_STEP (RT_BLOCK * 4u)      // bytes between a lane's consecutive stack entries

// `stack`: the lane's entry 0.
RT_DEV void trav_begin(Trav &t, uint32_t root, f3 o, f3 d, float tmax, const uint32_t *stack) {
    t.o = o; t.d = d;
    // A zero direction component would make plane*(1/d) - o*(1/d) an inf - inf; a component
    // of magnitude 2^-80 instead keeps the product form exact in sign and still puts
    // the slab planes ~1e24 units away.
    const float tiny = 8.271806e-25f;
    f3 ds = mk3(fabsf(d.x) > tiny ? d.x : copysignf(tiny, d.x), fabsf(d.y) > tiny ? d.y : copysignf(tiny, d.y),
                fabsf(d.z) > tiny ? d.z : copysignf(tiny, d.z));
    t.inv_d = mk3(rcp(ds.x), rcp(ds.y), rcp(ds.z));
    t.oid = mk3(-o.x * t.inv_d.x, -o.y * t.inv_d.y, -o.z * t.inv_d.z);
    t.cur = root;
    t.sp = lds_address(stack);
    t.best.t = tmax;
    t.best.ref = RT_REF_NONE;
    t.best.code = 0;
    t.found = false;
}

// A kernel that wants more waves per SIMD than a full-depth LDS stack allows keeps only the
// first LDS_ENTRIES entries of each lane's stack in LDS; deeper ones (rare: they need that many
// levels with both children hit) go to a per-thread column in global memory,
// deep[(entry - LDS_ENTRIES) * deep_stride].  LDS_ENTRIES = 0: the whole stack is in LDS.
typedef __attribute__((address_space(1))) uint32_t global_u32;   // typed as global so that the two homes never share a flat access
struct DeepStack {
    global_u32 *column;    // already offset by the thread's index in the grid
    uint32_t stride;       // threads in the grid
    uint32_t lds_limit;    // LDS address just behind the workgroup's LDS entries: a slot address below it is in LDS, and
                           // (address - lds_limit) / RT_STACK_STEP is the entry's index in the column otherwise
};
// STEP: bytes between a lane's consecutive stack entries = 4 x the threads of the workgroup (RT_STACK_STEP for the
// 256-thread workgroups of most kernels).
template <int LDS_ENTRIES, uint32_t STEP = RT_STACK_STEP>
RT_DEV void stack_put(const DeepStack *deep, uint32_t slot, uint32_t v) {
    if (LDS_ENTRIES == 0 || slot < deep->lds_limit) lds_store(slot, v);
    else deep->column[(size_t)((slot - deep->lds_limit) / STEP) * deep->stride] = v;
}
template <int LDS_ENTRIES, uint32_t STEP = RT_STACK_STEP>
RT_DEV uint32_t stack_get(const DeepStack *deep, uint32_t slot) {
    if (LDS_ENTRIES == 0 || slot < deep->lds_limit) return lds_load(slot);
    return deep->column[(size_t)((slot - deep->lds_limit) / STEP) * deep->stride];
}

template <int LDS_ENTRIES = 0, uint32_t STEP = RT_STACK_STEP>
RT_DEV void trav_pop(Trav &t, const uint32_t *stack, const DeepStack *deep = nullptr) {
    if (t.sp == lds_address(stack)) { t.cur = RT_REF_NONE; return; }
    t.sp -= STEP;
    t.cur = stack_get<LDS_ENTRIES, STEP>(deep, t.sp);
}

// Entry distance of the ray into one child box, and whether it is hit within
// [tmin, limit].  fminf/fmaxf drop a NaN operand (0 * inf when an axis-parallel ray lies
// in a box plane), which leaves that axis unconstrained.
RT_DEV float box_entry(v2f tx, v2f ty, v2f tz, float tmin, float limit, bool &hit) {
    float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), tmin));
    float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), limit));
    hit = tn <= tf;
    return tn;
}

// One inner-node step: tests both children, goes to the nearer one, pushes the farther.
template <int LDS_ENTRIES = 0, uint32_t STEP = RT_STACK_STEP>
RT_DEV void trav_node(const float4 *nodes, Trav &t, float tmin, uint32_t *stack, const DeepStack *deep = nullptr) {
    const float4 *n = nodes + 4u * t.cur;
    float4 a = n[0], b = n[1], c = n[2], e = n[3];
    v2f ix = {t.inv_d.x, t.inv_d.x}, iy = {t.inv_d.y, t.inv_d.y}, iz = {t.inv_d.z, t.inv_d.z};
    v2f ox = {t.oid.x, t.oid.x}, oy = {t.oid.y, t.oid.y}, oz = {t.oid.z, t.oid.z};
    v2f lx = {a.x, a.y}, ly = {a.z, a.w}, lz = {b.x, b.y};
    v2f rx = {b.z, b.w}, ry = {c.x, c.y}, rz = {c.z, c.w};
    bool hl, hr;
    // a box farther than the closest hit so far cannot hold a closer one
    float limit = t.best.t;
    float tl = box_entry(lx * ix + ox, ly * iy + oy, lz * iz + oz, tmin, limit, hl);
    float tr = box_entry(rx * ix + ox, ry * iy + oy, rz * iz + oz, tmin, limit, hr);
    uint32_t lref = __float_as_uint(e.x), rref = __float_as_uint(e.y);
    hr = hr && rref != RT_REF_NONE;       // a single-object scene has no right child
    if (hl && hr) {
        bool left_first = tl <= tr;
        stack_put<LDS_ENTRIES, STEP>(deep, t.sp, left_first ? rref : lref);
        t.sp += STEP;
        t.cur = left_first ? lref : rref;
    } else if (hl) {
        t.cur = lref;
    } else if (hr) {
        t.cur = rref;
    } else {
        trav_pop<LDS_ENTRIES, STEP>(t, stack, deep);
    }
}

// The same step on the 32-byte node form: words 0..5 hold, per axis, the two children's min
// planes (left in the low half, right in the high half) and then their max planes; words
// 6, 7 the refs.  Which word holds the near planes depends only on the sign of the ray
// direction, so one select per word replaces the min/max pair per plane pair, and the
// half -> float widening rides inside the FMA (v_fma_mix_f32).
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
// min(a, b) as the one instruction.  fminf() first quiets a possible signalling NaN in an operand the compiler cannot
// see the origin of (the closest distance so far, carried round the loop): one more instruction in every node step.
RT_DEV float min_raw(float a, float b) {
    float r;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
#else
    r = a < b ? a : b;
#endif
    return r;
}
RT_DEV v2f slab_pair(uint32_t packed, float inv, float oid) {
    v2h h = __builtin_bit_cast(v2h, packed);
    v2f r = {fmaf((float)h.x, inv, oid), fmaf((float)h.y, inv, oid)};
    return r;
}
typedef __attribute__((address_space(3))) uint4 lds_u4;
// TOP_BYTES > 0: the first TOP_BYTES of the node array (the nodes with the largest boxes, which the builder of a large
// tree puts first) also sit in LDS at byte address `top_lds`; a lane standing on one of them reads it from there.  The
// vector L1 takes one lane's address per cycle and a 32-byte node is two of them, which is what bounds the walk of a
// large tree; a node read from LDS costs that unit nothing.
template <int LDS_ENTRIES = 0, uint32_t STEP = RT_STACK_STEP, uint32_t TOP_BYTES = 0>
RT_DEV void trav_node16(const uint4 *nodes16, Trav &t, float tmin, uint32_t *stack, const DeepStack *deep = nullptr, uint32_t top_lds = 0) {
    // t.cur is the node's byte offset (references inside this array are stored that way, and so is the start)
    const uint4 *n = (const uint4 *)((const char *)nodes16 + t.cur);
    const bool px = t.inv_d.x >= 0.0f, py = t.inv_d.y >= 0.0f, pz = t.inv_d.z >= 0.0f;
    float limit = t.best.t;
    uint4 a, b;
#if defined(__HIP_DEVICE_COMPILE__)
    if (TOP_BYTES) {
        // The two homes are read by different instructions on purpose: given two pointers to choose from, the compiler
        // merges them into one flat pointer and reads the node with five flat loads (measured: half the speed).
        const bool in_top = t.cur < TOP_BYTES;
        if (!in_top) { a = n[0]; b = n[1]; }
        if (in_top) {
            const uint32_t addr = top_lds + t.cur;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
        }
    } else
#endif
    { a = n[0]; b = n[1]; }
    // (left, right) entry and exit distances per axis; a NaN (inf - inf) is dropped by max3/min3
    v2f nx = slab_pair(px ? a.x : a.y, t.inv_d.x, t.oid.x), fx = slab_pair(px ? a.y : a.x, t.inv_d.x, t.oid.x);
    v2f ny = slab_pair(py ? a.z : a.w, t.inv_d.y, t.oid.y), fy = slab_pair(py ? a.w : a.z, t.inv_d.y, t.oid.y);
    v2f nz = slab_pair(pz ? b.x : b.y, t.inv_d.z, t.oid.z), fz = slab_pair(pz ? b.y : b.x, t.inv_d.z, t.oid.z);
    float tl = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, tmin)), tr = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, tmin));
    float el = fminf(fminf(fx.x, fy.x), min_raw(fz.x, limit)), er = fminf(fminf(fx.y, fy.y), min_raw(fz.y, limit));
    bool hl = tl <= el, hr = tr <= er;
    uint32_t lref = b.z, rref = b.w;
    // (an absent right child - the only node of a one-object scene - needs no test here: its box is the empty one,
    // +inf .. -inf, and with the near and far planes picked by the ray's signs no ray enters it)
    if (hl && hr) {
        bool left_first = tl <= tr;
        stack_put<LDS_ENTRIES, STEP>(deep, t.sp, left_first ? rref : lref);
        t.sp += STEP;
        t.cur = left_first ? lref : rref;
    } else if (hl) {
        t.cur = lref;
    } else if (hr) {
        t.cur = rref;
    } else {
        trav_pop<LDS_ENTRIES, STEP>(t, stack, deep);
    }
}

// One step on the four-children node form (RtDevScene::nodes4): tests the four child boxes, goes to the nearest one
// entered and pushes the others, farthest first.  A ray that walks a large tree pays for every node it fetches with a
// cache line from L2, whatever the node's size: this form fetches half as many nodes for the same box tests.
// TOP_BYTES as in trav_node16.
template <int LDS_ENTRIES = 0, uint32_t STEP = RT_STACK_STEP, uint32_t TOP_BYTES = 0>
RT_DEV void trav_node4(const uint4 *nodes4, Trav &t, float tmin, uint32_t *stack, const DeepStack *deep = nullptr, uint32_t top_lds = 0) {
    const uint4 *n = (const uint4 *)((const char *)nodes4 + t.cur);
    const bool px = t.inv_d.x >= 0.0f, py = t.inv_d.y >= 0.0f, pz = t.inv_d.z >= 0.0f;
    const float limit = t.best.t;
    uint4 x, y, z, r;
#if defined(__HIP_DEVICE_COMPILE__)
    if (TOP_BYTES) {
        const bool in_top = t.cur < TOP_BYTES;
        if (!in_top) { x = n[0]; y = n[1]; z = n[2]; r = n[3]; }
        if (in_top) {
            const uint32_t addr = top_lds + t.cur;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(r) : "v"(addr) : "memory");
        }
    } else
#endif
    { x = n[0]; y = n[1]; z = n[2]; r = n[3]; }
    // per axis: words 0, 1 the min planes of children (0,1) and (2,3), words 2, 3 their max planes
    const v2f nxa = slab_pair(px ? x.x : x.z, t.inv_d.x, t.oid.x), nxb = slab_pair(px ? x.y : x.w, t.inv_d.x, t.oid.x);
    const v2f fxa = slab_pair(px ? x.z : x.x, t.inv_d.x, t.oid.x), fxb = slab_pair(px ? x.w : x.y, t.inv_d.x, t.oid.x);
    const v2f nya = slab_pair(py ? y.x : y.z, t.inv_d.y, t.oid.y), nyb = slab_pair(py ? y.y : y.w, t.inv_d.y, t.oid.y);
    const v2f fya = slab_pair(py ? y.z : y.x, t.inv_d.y, t.oid.y), fyb = slab_pair(py ? y.w : y.y, t.inv_d.y, t.oid.y);
    const v2f nza = slab_pair(pz ? z.x : z.z, t.inv_d.z, t.oid.z), nzb = slab_pair(pz ? z.y : z.w, t.inv_d.z, t.oid.z);
    const v2f fza = slab_pair(pz ? z.z : z.x, t.inv_d.z, t.oid.z), fzb = slab_pair(pz ? z.w : z.y, t.inv_d.z, t.oid.z);
    const float t0 = fmaxf(fmaxf(nxa.x, nya.x), fmaxf(nza.x, tmin)), t1 = fmaxf(fmaxf(nxa.y, nya.y), fmaxf(nza.y, tmin));
    const float t2 = fmaxf(fmaxf(nxb.x, nyb.x), fmaxf(nzb.x, tmin)), t3 = fmaxf(fmaxf(nxb.y, nyb.y), fmaxf(nzb.y, tmin));
    const float e0 = fminf(fminf(fxa.x, fya.x), min_raw(fza.x, limit)), e1 = fminf(fminf(fxa.y, fya.y), min_raw(fza.y, limit));
    const float e2 = fminf(fminf(fxb.x, fyb.x), min_raw(fzb.x, limit)), e3 = fminf(fminf(fxb.y, fyb.y), min_raw(fzb.y, limit));
    // The children entered, nearest first: (entry distance, ref) pairs through a five-exchange network.  The distance is
    // not below tmin > 0, so its bits order as the floats do; a child that is not entered sorts last.
    const uint32_t MISS = 0xFFFFFFFFu;
    uint32_t k0 = t0 <= e0 ? __float_as_uint(t0) : MISS, k1 = t1 <= e1 ? __float_as_uint(t1) : MISS;
    uint32_t k2 = t2 <= e2 ? __float_as_uint(t2) : MISS, k3 = t3 <= e3 ? __float_as_uint(t3) : MISS;
    uint32_t r0 = r.x, r1 = r.y, r2 = r.z, r3 = r.w;
#define RT_ORDER(ka, ra, kb, rb) { const bool sw = kb < ka; const uint32_t k_lo = sw ? kb : ka, k_hi = sw ? ka : kb, r_lo = sw ? rb : ra, r_hi = sw ? ra : rb; \
                                   ka = k_lo; kb = k_hi; ra = r_lo; rb = r_hi; }
    RT_ORDER(k0, r0, k1, r1) RT_ORDER(k2, r2, k3, r3) RT_ORDER(k0, r0, k2, r2) RT_ORDER(k1, r1, k3, r3) RT_ORDER(k1, r1, k2, r2)
#undef RT_ORDER
    if (k0 == MISS) {
        trav_pop<LDS_ENTRIES, STEP>(t, stack, deep);
        return;
    }
    if (k1 != MISS) {
        if (k2 != MISS) {
            if (k3 != MISS) { stack_put<LDS_ENTRIES, STEP>(deep, t.sp, r3); t.sp += STEP; }
            stack_put<LDS_ENTRIES, STEP>(deep, t.sp, r2);
            t.sp += STEP;
        }
        stack_put<LDS_ENTRIES, STEP>(deep, t.sp, r1);
        t.sp += STEP;
    }
    t.cur = r0;
}

// Runs one ray to completion (parity kernels and the wavefront extend stage).
template <bool SIMPLE, bool MOTION = false, bool WIDE = false>
RT_DEV bool trace(const RtDevScene &s, f3 o, f3 d, float tmin, float tmax, uint64_t volume_key,
                  const float *scripted, uint32_t n_scripted, uint32_t *stack, Best &best, float time = 0.0f) {
    Trav t;
    trav_begin(t, s.root, o, d, tmax, stack);
    const float4 *nodes = (const float4 *)s.nodes;
    while (t.cur != RT_REF_NONE) {
        while (!(t.cur & RT_REF_LEAF)) trav_node(nodes, t, tmin, stack);
        if (t.cur != RT_REF_NONE) {
            test_object<SIMPLE, MOTION, false, false, WIDE>(s, t.cur & ~RT_REF_LEAF, o, d, tmin, tmax, volume_key, scripted, n_scripted, t.found, t.best, time);
            trav_pop(t, stack);
        }
    }
    best = t.best;
    return t.found;
}

// ------------------------------------------------------------------ surface of the closest hit
struct HitRecord {
    f3 point, normal;      // normal flipped to oppose the ray (HitTesting.h:24-32)
    float u, v;
    uint32_t material;
    uint32_t object;
    bool front;
};

RT_DEV Surface csg_surface(const RtDevScene &s, uint32_t csg_idx, uint32_t code, f3 o, f3 d, float t, bool want_uv) {
    bool flip;
    const uint32_t leaf_ref = csg_leaf(s, csg_idx, code_leaf<true>(code), flip);
    Surface sf = prim_surface(s, leaf_ref, code & 7u, o, d, t, want_uv);
    if (flip) sf.normal = -sf.normal;      // RenderableImpl.h:712
    return sf;
}

// ------------------------------------------------------------------ materials (shade)
// Material record: 8 floats = kind, texture id, fuzz, refraction index, albedo rgb, pad.
struct Material { uint32_t kind, texture; float fuzz, ior; f3 albedo; };
RT_DEV Material unpack_material(float4 a, float4 b);
RT_DEV Material load_material(const RtDevScene &s, uint32_t id) {
    const float4 *q = (const float4 *)s.materials + 2u * id;
    return unpack_material(q[0], q[1]);
}
RT_DEV Material unpack_material(float4 a, float4 b) {
    Material m;
    m.kind = __float_as_uint(a.x);
    m.texture = __float_as_uint(a.y);
    m.fuzz = a.z;
    m.ior = a.w;
    m.albedo = mk3(b.x, b.y, b.z);
    return m;
}
RT_DEV bool material_wants_uv(uint32_t kind) { return kind == 2u || kind == 5u || kind >= 10u; }

// Surface attributes of the closest hit.  Whatever the object's kind, the hit lies on one leaf
// primitive (named by best.code for CSG shapes), so the attribute code is reached from one place.
// by_material (with m_out): texture coordinates are worked out only when the hit's material reads them (an image or
// one of the Kit's textures) - the arc tangent and arc cosine of a sphere or a tube cost as much as the rest of this
// function - and the material record comes back through m_out.
template <bool MOTION = false, bool WIDE = true>
RT_DEV void complete_hit(const RtDevScene &s, Best best, f3 o, f3 d, bool want_uv, HitRecord &h, float time = 0.0f,
                         bool by_material = false, Material *m_out = nullptr) {
    uint32_t kind = (best.ref >> RT_KIND_SHIFT) & 7u, idx = best.ref & RT_INDEX_MASK;
    h.point = madd(d, best.t, o);
    const f3 world_d = d;
    MotionFrame frame;
    bool moved = false;
    uint32_t object = 0;
    if (MOTION && kind == RT_KIND_MOTION) {
        frame = motion_frame(s, idx, time);
        object = s.motion_obj[idx];
        o = frame_in(frame, o - frame.t);
        d = frame_in(frame, d);
        best.ref = frame.base;
        kind = (best.ref >> RT_KIND_SHIFT) & 7u;
        idx = best.ref & RT_INDEX_MASK;
        moved = true;
    }
    uint32_t shape = best.ref;
    const bool scatter = kind == RT_KIND_CDV && (best.code & RT_CODE_SCATTER);
    if (kind == RT_KIND_CDV) {
        if (!moved) object = s.cdv_obj[idx];
        shape = s.cdv[idx].base;
    }
    const uint32_t skind = (shape >> RT_KIND_SHIFT) & 7u, sidx = shape & RT_INDEX_MASK;
    uint32_t leaf_ref = shape;
    bool flip = false;
    if (skind == RT_KIND_CSG) {
        leaf_ref = csg_leaf<WIDE>(s, sidx, code_leaf<WIDE>(best.code), flip);
        if (kind == RT_KIND_CSG && !moved) object = s.csg_obj[sidx];
    } else if (kind != RT_KIND_CDV && !moved) {
        object = skind == RT_KIND_SPHERE ? s.sphere_obj[sidx] : s.xprim_obj[sidx];
    }
    // A scatter inside a medium takes the ray's own direction for its normal; only its texture coordinates come
    // from the surface (those of the ENTRY hit of the interval, RenderableImpl.h:760-766), so a medium whose
    // material reads none needs neither the entry distance nor the surface.
    bool surface_uv = want_uv || scatter;
    if (by_material) {
        *m_out = load_material(s, prim_material(s, leaf_ref, best.code & 7u));
        surface_uv = want_uv && material_wants_uv(m_out->kind);
    }
    Surface sf;
    if (scatter && !surface_uv) {
        sf.normal = mk3(0.0f, 0.0f, 0.0f);
        sf.u = sf.v = 0.0f;
        sf.material = prim_material(s, leaf_ref, best.code & 7u);
    } else {
        float t_surface = best.t;
        if (scatter) {
            // the entry hit lies on leaf_ref at its entry or exit
            Span sp = prim_span(s, leaf_ref, o, d);
            t_surface = ((best.code >> RT_CODE_LEAF_EXIT_SHIFT) & 1u) ? sp.t1 : sp.t0;
        }
        sf = prim_surface(s, leaf_ref, best.code & 7u, o, d, t_surface, surface_uv);
    }
    if (flip) sf.normal = -sf.normal;
    if (scatter) sf.normal = -d;
    if (MOTION && moved) sf.normal = frame_out(frame, sf.normal);
    h.object = object;
    h.u = sf.u; h.v = sf.v;
    h.material = sf.material;
    if (dot(sf.normal, world_d) > 0.0f) { h.normal = -sf.normal; h.front = false; }
    else { h.normal = sf.normal; h.front = true; }
}

// MaterialsImpl.h:53-58: normalised coordinates, nearest texel, clamp to edge; v = 0 is the
// bottom row (the loader's bottom-left origin), rows are stored top to bottom.
RT_DEV f3 sample_image(const RtDevScene &s, uint32_t tex, float u, float v) {
    RtDevTexture t = s.textures[tex];
    int x = (int)floorf(u * (float)t.width), y = (int)floorf(v * (float)t.height);
    x = min(max(x, 0), (int)t.width - 1);
    y = min(max(y, 0), (int)t.height - 1);
    uint32_t px = s.texels[t.offset + (t.height - 1u - (uint32_t)y) * t.width + (uint32_t)x];
    return mk3((float)(px & 255u), (float)((px >> 8) & 255u), (float)((px >> 16) & 255u)) * (1.0f / 255.0f);
}

RT_DEV float mixf(float a, float b, float t) { return fmaf(b - a, t, a); }

// MaterialsImpl.h:60-121
RT_DEV float perlin_noise(const RtDevPerlin &tex, f3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    int xi0 = ((int)fx) & 255, yi0 = ((int)fy) & 255, zi0 = ((int)fz) & 255;
    int xi1 = (xi0 + 1) & 255, yi1 = (yi0 + 1) & 255, zi1 = (zi0 + 1) & 255;
    float tx = p.x - fx, ty = p.y - fy, tz = p.z - fz;
    float u = tx * tx * tx * (tx * (6.0f * tx - 15.0f) + 10.0f);
    float v = ty * ty * ty * (ty * (6.0f * ty - 15.0f) + 10.0f);
    float w = tz * tz * tz * (tz * (6.0f * tz - 15.0f) + 10.0f);
    int hx0 = tex.permutations[0][xi0], hx1 = tex.permutations[0][xi1];
    int h00 = tex.permutations[1][(hx0 + yi0) & 255], h10 = tex.permutations[1][(hx1 + yi0) & 255];
    int h01 = tex.permutations[1][(hx0 + yi1) & 255], h11 = tex.permutations[1][(hx1 + yi1) & 255];
    float x0 = tx, x1 = tx - 1.0f, y0 = ty, y1 = ty - 1.0f, z0 = tz, z1 = tz - 1.0f;
#define RT_GRAD(h, zi, X, Y, Z) (dot(ld3(tex.vectors[tex.permutations[2][((h) + (zi)) & 255]]), mk3(X, Y, Z)))
    float a = mixf(RT_GRAD(h00, zi0, x0, y0, z0), RT_GRAD(h10, zi0, x1, y0, z0), u);
    float b = mixf(RT_GRAD(h01, zi0, x0, y1, z0), RT_GRAD(h11, zi0, x1, y1, z0), u);
    float c = mixf(RT_GRAD(h00, zi1, x0, y0, z1), RT_GRAD(h10, zi1, x1, y0, z1), u);
    float e = mixf(RT_GRAD(h01, zi1, x0, y1, z1), RT_GRAD(h11, zi1, x1, y1, z1), u);
#undef RT_GRAD
    return mixf(mixf(a, b, v), mixf(c, e, v), w);
}

// MaterialsImpl.h:123-139
RT_DEV f3 perlin_color(const RtDevScene &s, uint32_t id, f3 point) {
    const RtDevPerlin &tex = s.perlin[id];
    float t = 0.0f;
    if (tex.turbulence == 0u) {
        t = 1.0f + perlin_noise(tex, point * tex.frequency);
    } else {
        float f = tex.frequency, weight = 1.0f;
        for (uint32_t i = 0; i < tex.turbulence; i++) {
            t = fmaf(perlin_noise(tex, point * f), weight, t);
            weight *= 0.5f;
            f *= 2.0f;
        }
        t = fabsf(t);
    }
    f3 c0 = ld3(tex.colors[0]), c1 = ld3(tex.colors[1]);
    return madd(c1 - c0, t, c0);
}

// ---- RayTracingKit textures
// PerlinNoise subscript, Texture.swift:184-209: the lattice cell comes from Int(x), which
// truncates towards zero, while the fraction comes from floor (kept as is).
RT_DEV float kit_perlin_noise(const RtDevKitPerlin &n, f3 p) {
    float u = p.x - floorf(p.x), v = p.y - floorf(p.y), w = p.z - floorf(p.z);
    float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
    int i = (int)p.x, j = (int)p.y, k = (int)p.z;
    float result = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
        for (int dj = 0; dj < 2; dj++)
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                uint32_t index = n.perm[0][(i + di) & 255] ^ n.perm[1][(j + dj) & 255] ^ n.perm[2][(k + dk) & 255];
                float weight = (di == 0 ? 1.0f - uu : uu) * (dj == 0 ? 1.0f - vv : vv) * (dk == 0 ? 1.0f - ww : ww);
                result = fmaf(weight, dot(ld3(n.vectors[index]), mk3(u - (float)di, v - (float)dj, w - (float)dk)), result);
            }
    return result;
}
// Texture.swift:211-223
RT_DEV float kit_turbulence(const RtDevKitPerlin &n, f3 p, int octaves) {
    float accum = 0.0f, weight = 1.0f;
    for (int o = 0; o < octaves; o++) {
        accum = fmaf(weight, kit_perlin_noise(n, p), accum);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return accum;
}
// ImageTexture.clip, Texture.swift:103-120 (Swift's % truncates, like C's)
RT_DEV int kit_clip(int index, int size, bool wrap) {
    if (index < 0) return wrap ? size + (index % size) : 0;
    if (index >= size) return wrap ? index % size : size - 1;
    return index;
}
RT_DEV f3 kit_texel(const RtDevScene &s, const RtDevKitTexture &t, const RtDevTexture &img, int i, int j) {
    int row = t.origin_y + kit_clip(i, t.height, t.wrap_y != 0u) * t.step_y;
    int col = t.origin_x + kit_clip(j, t.width, t.wrap_x != 0u) * t.step_x;
    // the reference asserts on indices outside the image (Image.swift:65-68); clamp instead of faulting
    row = min(max(row, 0), (int)img.height - 1);
    col = min(max(col, 0), (int)img.width - 1);
    uint32_t px = s.texels[img.offset + (uint32_t)row * img.width + (uint32_t)col];
    return mk3((float)(px & 255u), (float)((px >> 8) & 255u), (float)((px >> 16) & 255u)) * (1.0f / 255.0f);
}
// Texture subscripts, Texture.swift:24,44-50,80-93,145-148,160-163.  A checker only picks
// one of its two textures, so nesting is a loop, not a recursion.
RT_DEV f3 kit_texture_color(const RtDevScene &s, uint32_t id, float u, float v, f3 point) {
    RtDevKitTexture t = s.kit_textures[id];
    for (int hop = 0; hop < 8 && t.kind == 1u; hop++) {
        int xi = (int)floorf(t.scale * point.x), yi = (int)floorf(t.scale * point.y), zi = (int)floorf(t.scale * point.z);
        bool even = ((xi + yi + zi) % 2) == 0;
        t = s.kit_textures[even ? t.even : t.odd];
    }
    switch (t.kind) {
    case 2: {
        RtDevTexture img = s.textures[t.image];
        float fi = fmaf(v, (float)t.height, -0.5f), fj = fmaf(u, (float)t.width, -0.5f);
        float i1f = floorf(fi), j1f = floorf(fj);
        int i1 = (int)i1f, j1 = (int)j1f;
        float a = fi - i1f, b = fj - j1f;
        f3 c1 = kit_texel(s, t, img, i1, j1) * (1.0f - b) + kit_texel(s, t, img, i1, j1 + 1) * b;
        f3 c2 = kit_texel(s, t, img, i1 + 1, j1) * (1.0f - b) + kit_texel(s, t, img, i1 + 1, j1 + 1) * b;
        return c1 * (1.0f - a) + c2 * a;
    }
    case 3: {
        float grey = fabsf(kit_turbulence(s.kit_perlin[t.perlin], point * t.scale, 7));
        return mk3(grey, grey, grey);
    }
    case 4: {
        float grey = (1.0f + sinf(fmaf(t.scale, point.z, 10.0f * kit_turbulence(s.kit_perlin[t.perlin], point, 7)))) * 0.5f;
        return mk3(grey, grey, grey);
    }
    default:
        return mk3(t.color[0], t.color[1], t.color[2]);
    }
}

// TEXTURES = false compiles the image / Perlin paths out (scenes without such materials).
template <bool TEXTURES>
RT_DEV f3 material_color(const RtDevScene &s, const Material &m, const HitRecord &h) {
    if (TEXTURES) {
        if (m.kind == 2u || m.kind == 5u) return sample_image(s, m.texture, h.u, h.v);
        if (m.kind == 3u || m.kind == 6u) return perlin_color(s, m.texture, h.point);
        if (m.kind >= 10u) return kit_texture_color(s, m.texture, h.u, h.v, h.point);
    }
    return m.albedo;
}

// MaterialsImpl.h:16-24
RT_DEV float schlick(float cos_theta, float eta) {
    float sq = (1.0f - eta) * rcp(1.0f + eta);
    float r0 = sq * sq;
    float x = 1.0f + cos_theta;
    float x2 = x * x;
    return fmaf(1.0f - r0, x2 * x2 * x, r0);
}

// One scattering event, MaterialsImpl.h:147-215.  Returns false when the path ends here
// (emissive surface, or metal absorbing a reflection that points below the surface).
template <bool TEXTURES>
RT_DEV bool scatter(const RtDevScene &s, const Material &m, f3 d, const HitRecord &h, Pcg &g,
                    f3 &emitted, f3 &attenuation, f3 &new_d) {
    emitted = mk3(0, 0, 0);
    attenuation = mk3(0, 0, 0);
    // lambertian, metal and isotropic all start by drawing one unit vector: drawn once
    // here, so that a wave holding a mix of them runs the rejection loop once
    f3 rv = mk3(0, 0, 0);
    if (m.kind != 7u && m.kind != 8u && m.kind != 11u) rv = random_unit_vector_3d(g);
    switch (m.kind) {
    case 1: case 2: case 3: case 10: {      // lambertian
        while (true) {
            f3 v = h.normal + rv;
            float len_sq = dot(v, v);
            if (len_sq > RT_MIN_LEN_SQ) { new_d = v * __builtin_amdgcn_rsqf(len_sq); break; }
            rv = random_unit_vector_3d(g);
        }
        attenuation = material_color<TEXTURES>(s, m, h);
        return true;
    }
    case 4: case 5: case 6: {      // metal
        f3 refl = madd(h.normal, -2.0f * dot(d, h.normal), d);
        refl = madd(rv, m.fuzz, refl);
        // MaterialsImpl.h:167 absorbs below the surface, RayTracingKit/Material.swift:55 on it too
        const float side = dot(refl, h.normal);
        if (s.kit_convention ? side <= 0.0f : side < 0.0f) return false;
        attenuation = material_color<TEXTURES>(s, m, h);
        new_d = normalize3(refl);
        return true;
    }
    case 7: {                      // dielectric, MaterialsImpl.h:26-47,173-180
        float eta = h.front ? rcp(m.ior) : m.ior;
        float u = rnd(g);
        float cos_theta = fmaxf(dot(d, h.normal), -1.0f);
        f3 r_perp = madd(h.normal, -cos_theta, d) * eta;
        float perp_sq = dot(r_perp, r_perp);
        bool refracts = !(perp_sq > 1.0f) && !(u <= 0.0f) && !(u < 1.0f && schlick(cos_theta, eta) > u);
        if (refracts) new_d = madd(h.normal, -fsqrt(1.0f - perp_sq), r_perp);
        else new_d = madd(h.normal, -2.0f * dot(d, h.normal), d);
        attenuation = mk3(1, 1, 1);
        return true;
    }
    case 8:                        // emissive
        emitted = m.albedo;
        return false;
    case 11:                       // Emissive over a texture, RayTracingKit/Material.swift:75-93
        emitted = material_color<TEXTURES>(s, m, h);
        return false;
    default:                       // isotropic
        new_d = rv;
        attenuation = m.albedo;
        return true;
    }
}
