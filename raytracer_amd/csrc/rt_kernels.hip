// rt_kernels.hip - HIP kernels of the path-tracing hot path (gfx950 / MI355X) and their
// launchers.
//
// Two forms of the render loop (Shaders.metal:93-159 is the reference's single kernel):
//
//  * wavefront: k_wf_raygen -> { k_wf_extend -> k_wf_shade } x max_depth -> k_wf_accumulate.
//    Path state is SoA float4 streams in HBM (80 B per path).  Shade compacts the
//    surviving paths into the other queue (eight shards, ballots + one atomic per
//    workgroup), so every extend launch runs on dense waves.  Finished samples are written to a per-sample
//    slot and summed per pixel by the accumulate kernel: no scattered float atomics.
//
//  * fused: k_fused, one persistent kernel.  A wave works on an 8x8 pixel tile for a chunk of
//    samples at a time and runs raygen / extend / shade as wave-wide phases; a lane whose path
//    ended takes the next (pixel, sample) at once - of the current item or, when that is used
//    up, of the next one - so lanes stay busy while path state never leaves registers.  Tile
//    sums are kept in LDS and flushed once per work item.
//
// Both draw sample s of pixel (x, y) from the same generator stream, so they are the same
// estimator sample for sample.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "rt_device.h"
#include "rt_launch.h"

#ifndef RT_WPS0
#define RT_WPS0 8              // waves per SIMD of the spheres-only kernel (64 VGPRs)
#endif
#ifndef RT_WPS_DEEP
#define RT_WPS_DEEP 7          // ... of its deep-tree form: 72 VGPRs, measured +5 % on the million-sphere workload over 8 (6: +-0)
#endif
#ifndef RT_WPS_XPRIM
#define RT_WPS_XPRIM 4         // ... of the tree kernels of scenes with cylinders, boxes, quads, CSG shapes or volumes: 128 VGPRs, no scratch;
                               // left free the textured one takes 137 registers and runs at 3 waves, 14 % slower on the CSG / textured workload
#endif
#ifndef RT_WPS_LIST1
#define RT_WPS_LIST1 5         // ... of the list kernel of untextured scenes
#endif
#ifndef RT_WPS_MOVING_SPHERES
#define RT_WPS_MOVING_SPHERES 6   // ... of the spheres-and-moving-spheres kernel
#endif
#ifndef RT_WIDE_REFILL
#define RT_WIDE_REFILL 24u
#endif
#ifndef RT_WIDE_NODE_MIN
#define RT_WIDE_NODE_MIN 40u
#endif
#ifndef RT_TOP_NODES
#define RT_TOP_NODES 448       // nodes of a deep tree's head kept in LDS by its kernel (14 KB next to a 512-thread workgroup's 32-KB stack: three workgroups per CU)
#endif
#ifndef RT_DOUBLE_STEP
#define RT_DOUBLE_STEP 24u       // lanes on inner nodes from which a second node step follows the first without a new count (+1.6 % on random spheres)
#endif
#define RT_T_MIN 1e-3f          // Camera.swift:117
#define RT_WAVES (RT_BLOCK / 64)

RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_DEV uint32_t prefix_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ------------------------------------------------------------------ parity surface
template <int FEAT>
__global__ __launch_bounds__(RT_BLOCK) void k_first_hit(RtDevScene s, const float *rays, int n, float tmin, float tmax,
                                                         uint64_t volume_key, float time, RtHit *out) {
    __shared__ uint32_t stack[RT_STACK_DEPTH * RT_BLOCK];
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + 6 * (size_t)i;
    f3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);
    Best b;
    constexpr bool MOTION = (FEAT & RT_FEAT_MOTION) != 0;
    bool found = trace<!(FEAT & RT_FEAT_XPRIM), MOTION, (FEAT & RT_FEAT_WIDE_CSG) != 0>(s, o, d, tmin, tmax, volume_key, nullptr, 0, stack + threadIdx.x, b, time);
    RtHit h;
    h.t = 0.f;
    h.point[0] = h.point[1] = h.point[2] = 0.f;
    h.normal[0] = h.normal[1] = h.normal[2] = 0.f;
    h.uv[0] = h.uv[1] = 0.f;
    h.object = -1;
    h.material_offset = 0;
    h.front_face = 0;
    if (found) {
        HitRecord hr;
        complete_hit<MOTION, (FEAT & RT_FEAT_WIDE_CSG) != 0>(s, b, o, d, true, hr, time);
        h.t = b.t;
        h.point[0] = hr.point.x; h.point[1] = hr.point.y; h.point[2] = hr.point.z;
        h.normal[0] = hr.normal.x; h.normal[1] = hr.normal.y; h.normal[2] = hr.normal.z;
        h.uv[0] = hr.u; h.uv[1] = hr.v;
        h.object = (int32_t)hr.object;
        h.material_offset = hr.material;
        h.front_face = hr.front ? 1 : 0;
    }
    out[i] = h;
}

RT_DEV void put_event(RtEvent *ev, int32_t &n, float t, bool is_exit, const Surface *sf, f3 o, f3 d,
                      const MotionFrame *frame = nullptr) {
    if (n >= RT_MAX_EVENTS) return;
    RtEvent e;
    e.t = t;
    e.is_exit = is_exit ? 1 : 0;
    e.pad = 0;
    e.point[0] = e.point[1] = e.point[2] = 0.f;
    e.normal[0] = e.normal[1] = e.normal[2] = 0.f;
    e.uv[0] = e.uv[1] = 0.f;
    e.material_offset = 0;
    if (sf && finite(t)) {
        f3 p = madd(d, t, o);
        e.point[0] = p.x; e.point[1] = p.y; e.point[2] = p.z;
        f3 nn = frame ? frame_out(*frame, sf->normal) : sf->normal;
        e.normal[0] = nn.x; e.normal[1] = nn.y; e.normal[2] = nn.z;
        e.uv[0] = sf->u; e.uv[1] = sf->v;
        e.material_offset = sf->material;
    }
    ev[n++] = e;
}

// Surface of a shape's boundary event: `code` names the face and, for a CSG shape, the leaf.
RT_DEV Surface shape_surface(const RtDevScene &s, uint32_t shape, uint32_t code, f3 o, f3 d, float t, bool want_uv) {
    if (((shape >> RT_KIND_SHIFT) & 7u) == RT_KIND_CSG) return csg_surface(s, shape & RT_INDEX_MASK, code, o, d, t, want_uv);
    return prim_surface(s, shape, code & 7u, o, d, t, want_uv);
}

// A primitive's interval as an event list of two (the protocol of the CSG walks).
struct PrimIter {
    Span sp;
    int stage;               // 0: entry next, 1: exit next, 2: done
    float t;
    uint32_t leaf, tag;
    bool is_exit, leaf_exit;
};
RT_DEV bool csg_next(const RtDevScene &, PrimIter &it) {
    if (it.stage > 1) return false;
    it.is_exit = it.leaf_exit = it.stage == 1;
    it.t = it.is_exit ? it.sp.t1 : it.sp.t0;
    it.tag = it.is_exit ? it.sp.tag1 : it.sp.tag0;
    it.leaf = 0;
    it.stage++;
    return true;
}

template <class It>
RT_DEV void emit_shape_events(const RtDevScene &s, It &it, uint32_t shape, f3 o, f3 d, RtEvent *ev, int32_t &cnt, f3 wo, f3 wd,
                              const MotionFrame *fr) {
    while (csg_next(s, it)) {
        uint32_t code = event_code(it) & ~(1u << RT_CODE_LEAF_EXIT_SHIFT);
        Surface sf;
        if (finite(it.t)) sf = shape_surface(s, shape, code, o, d, it.t, true);
        put_event(ev, cnt, it.t, it.is_exit, &sf, wo, wd, fr);
    }
}

template <class It>
RT_DEV void emit_volume_events(const RtDevScene &s, It &it, const RtDevCdv &cv, f3 o, f3 d, const float *scripted, int n_scripted,
                               uint64_t volume_key, uint32_t object, RtEvent *ev, int32_t &cnt) {
    Pcg vr = pcg_seed(volume_key, (uint64_t)(object + 1u));
    uint32_t draw = 0;
    while (csg_next(s, it)) {
        const float t_entry = it.t;
        const uint32_t entry_code = event_code(it) & ~(1u << RT_CODE_LEAF_EXIT_SHIFT);
        float t_exit = RT_INF;
        uint32_t exit_code = 0;
        const bool has_exit = csg_next(s, it);
        if (has_exit) { t_exit = it.t; exit_code = event_code(it) & ~(1u << RT_CODE_LEAF_EXIT_SHIFT); }
        float t1 = fmaxf(0.f, t_entry);
        if (t1 < t_exit) {
            float u = scripted ? scripted[draw % (uint32_t)n_scripted] : rnd(vr);
            draw++;
            float t = fmaf(log_unit(1.f - u), cv.neg_inv_density, t1);
            if (t <= t_exit) {
                Surface en = shape_surface(s, cv.base, entry_code, o, d, t_entry, true);
                en.normal = -d;
                put_event(ev, cnt, t, false, &en, o, d);
                if (has_exit) {
                    Surface ex;
                    if (finite(t_exit)) ex = shape_surface(s, cv.base, exit_code, o, d, t_exit, true);
                    put_event(ev, cnt, t_exit, true, &ex, o, d);
                }
            }
        }
        if (!has_exit) break;
    }
}

// All boundary crossings of one shape, in order: the enumerator protocol of
// Impl/RenderableImpl.h as a list.
__global__ __launch_bounds__(RT_BLOCK) void k_intervals(RtDevScene s, uint32_t ref, const float *rays, int n,
                                                         const float *scripted, int n_scripted, uint64_t volume_key,
                                                         uint32_t object, float time, RtEvent *events, int32_t *counts) {
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + 6 * (size_t)i;
    f3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);
    const f3 wo = o, wd = d;           // events report world-space points
    RtEvent *ev = events + (size_t)i * RT_MAX_EVENTS;
    int32_t cnt = 0;
    uint32_t kind = (ref >> RT_KIND_SHIFT) & 7u, idx = ref & RT_INDEX_MASK;
    MotionFrame frame;
    const MotionFrame *fr = nullptr;
    if (kind == RT_KIND_MOTION) {
        frame = motion_frame(s, idx, time);
        fr = &frame;
        o = frame_in(frame, o - frame.t);
        d = frame_in(frame, d);
        ref = frame.base;
        kind = (ref >> RT_KIND_SHIFT) & 7u;
        idx = ref & RT_INDEX_MASK;
    }
    if (kind <= RT_KIND_QUAD) {
        Span sp = prim_span(s, ref, o, d);
        if (span_valid(sp)) {
            Surface a, b;
            if (finite(sp.t0)) a = prim_surface(s, ref, sp.tag0, o, d, sp.t0, true);
            if (finite(sp.t1)) b = prim_surface(s, ref, sp.tag1, o, d, sp.t1, true);
            put_event(ev, cnt, sp.t0, false, &a, wo, wd, fr);
            put_event(ev, cnt, sp.t1, true, &b, wo, wd, fr);
        }
    } else if (kind == RT_KIND_CSG) {
        if (is_wide_csg(s, ref)) {
            WideIter w;
            csg_begin_wide(s, ref, o, d, w);
            emit_shape_events(s, w, ref, o, d, ev, cnt, wo, wd, fr);
        } else {
            CsgIter it;
            csg_begin(s, ref, o, d, it);
            emit_shape_events(s, it, ref, o, d, ev, cnt, wo, wd, fr);
        }
    } else {
        // ConstantDensityVolume<Impl>::HitEnumerator, RenderableImpl.h:727-802
        const RtDevCdv cv = s.cdv[idx];
        if (((cv.base >> RT_KIND_SHIFT) & 7u) != RT_KIND_CSG) {
            PrimIter p;
            p.sp = prim_span(s, cv.base, o, d);
            p.stage = span_valid(p.sp) ? 0 : 2;
            emit_volume_events(s, p, cv, o, d, scripted, n_scripted, volume_key, object, ev, cnt);
        } else if (is_wide_csg(s, cv.base)) {
            WideIter w;
            csg_begin_wide(s, cv.base, o, d, w);
            emit_volume_events(s, w, cv, o, d, scripted, n_scripted, volume_key, object, ev, cnt);
        } else {
            CsgIter it;
            csg_begin(s, cv.base, o, d, it);
            emit_volume_events(s, it, cv, o, d, scripted, n_scripted, volume_key, object, ev, cnt);
        }
    }
    counts[i] = cnt;
}

__global__ __launch_bounds__(RT_BLOCK) void k_camera_rays(RtDevCamera cam, RtDevRender rc, int x0, int y0, int w, int h,
                                                           float *out) {
    uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    uint32_t total = (uint32_t)w * (uint32_t)h * rc.spp;
    if (i >= total) return;
    uint32_t pix = i / rc.spp, sidx = i % rc.spp;
    uint32_t x = (uint32_t)x0 + pix % (uint32_t)w, y = (uint32_t)y0 + pix / (uint32_t)w;
    uint64_t seed = ((uint64_t)rc.seed_hi << 32) | rc.seed_lo;
    Pcg g = sample_stream(seed, rc.first_sample + sidx, x, y);
    f3 o, d;
    camera_ray(cam, x, y, g, o, d);
    float *r = out + 6 * (size_t)i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
}

// Camera.swift:92, Vector3D.swift:185-191,208-211 in float: sum / n, square root, min with 1,
// floor(nextDown(256) * x).  Every operation is a correctly rounded single one (contraction off;
// `/` and sqrtf are the IEEE ones under hipcc's defaults), so the bytes are those of a plain IEEE
// evaluation of the same expressions.  A negative or NaN sum (the Kit's precondition would trap)
// gives 0.
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(const float *accum, float n_samples, int npix, uint8_t *rgbx) {
#pragma clang fp contract(off)
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= npix) return;
    uint32_t px = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float mean = accum[3 * (size_t)i + c] / n_samples;
        float v = mean > 0.f ? fminf(sqrtf(mean), 1.f) : 0.f;
        uint32_t q = (uint32_t)floorf(255.99998f * v);
        px |= (q > 255u ? 255u : q) << (8 * c);
    }
    ((uint32_t *)rgbx)[i] = px;
}

// One scattering event per query: the shade stage's material step alone.
__global__ __launch_bounds__(RT_BLOCK) void k_scatter(RtDevScene s, const RtScatterQuery *queries, int n, uint64_t seed,
                                                       RtScatterResult *out) {
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const RtScatterQuery q = queries[i];
    HitRecord h;
    h.point = ld3(q.point);
    h.normal = ld3(q.normal);
    h.u = q.uv[0]; h.v = q.uv[1];
    h.material = q.material_offset;
    h.object = 0;
    h.front = q.front_face != 0;
    Material m = load_material(s, q.material_offset);
    Pcg g = sample_stream(seed, q.rng_sample, q.rng_x, q.rng_y);
    f3 emitted, atten, nd = mk3(0, 0, 0);
    bool did = scatter<true>(s, m, ld3(q.direction), h, g, emitted, atten, nd);
    RtScatterResult r;
    r.scattered = did ? 1 : 0;
    r.direction[0] = nd.x; r.direction[1] = nd.y; r.direction[2] = nd.z;
    r.attenuation[0] = atten.x; r.attenuation[1] = atten.y; r.attenuation[2] = atten.z;
    r.emitted[0] = emitted.x; r.emitted[1] = emitted.y; r.emitted[2] = emitted.z;
    r.pad[0] = r.pad[1] = 0;
    out[i] = r;
}

// Raw generator words for bit-exact comparison with a host implementation.
__global__ __launch_bounds__(RT_BLOCK) void k_rng_words(const uint32_t *streams, int n, int words, uint32_t *out, uint32_t *wy_out) {
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = streams + 8 * (size_t)i;
    const uint64_t seed = ((uint64_t)r[1] << 32) | r[0];
    Pcg g = sample_stream(seed, r[2], r[3], r[4]);
    for (int k = 0; k < words; k++) out[(size_t)i * words + k] = pcg_next(g);
    const uint64_t w = wyrand_at(seed, r[2]);
    wy_out[2 * (size_t)i] = (uint32_t)w;
    wy_out[2 * (size_t)i + 1] = (uint32_t)(w >> 32);
}

// Shaders.metal:151-158.  Every float operation is a correctly rounded single one
// (contraction off; `/` and sqrtf are the IEEE ones under hipcc's defaults, unlike the
// __fsqrt_rn / rcp shortcuts), so the 10-bit words are the ones a plain IEEE evaluation of
// the reference's expressions gives.
__global__ __launch_bounds__(RT_BLOCK) void k_progressive_present(const float *pass_sum, float spp, float t, int npix,
                                                                  uint32_t *acc, uint32_t *bgra) {
#pragma clang fp contract(off)
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= npix) return;
    const float half_below = 0.49999997f;           // nextafter(0.5f, 0)
    const float keep = 1.0f - t;
    uint32_t word = acc[i], out_word = 0, px = 0xFF000000u;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float color = fminf(pass_sum[3 * (size_t)i + c] / spp, 1.0f);
        float old_color = ((float)((word >> (10 * c)) & 1023u) + half_below) / 1024.0f;
        float total = old_color * keep + color * t;
        float scaled = total * 1024.0f;
        uint32_t q = scaled > 0.0f ? (uint32_t)scaled : 0u;
        out_word |= (q > 1023u ? 1023u : q) << (10 * c);
        float shown = sqrtf(total);
        shown = (shown < 0.0f || shown != shown) ? 0.0f : fminf(shown, 1.0f);
        px |= (uint32_t)rintf(shown * 255.0f) << (8 * (2 - c));      // b, g, r, a in memory
    }
    acc[i] = out_word;
    bgra[i] = px;
}

// ------------------------------------------------------------------ one bounce of shading
// Shaders.metal:102-119 for one path: adds emission / background, scatters, and says
// whether the path goes on.  Depth exhaustion keeps what was gathered so far
// (Camera.swift:114-122) instead of zeroing it (Shaders.metal:127-128).
template <int FEAT>
RT_DEV bool shade_path(const RtDevScene &s, uint32_t background, uint32_t max_depth, bool found, const Best &b,
                       f3 &o, f3 &d, f3 &thr, f3 &rad, uint32_t &depth, Pcg &g, float time = 0.0f) {
    if (!found) {
        rad = rad + thr * background_color(background, d);
        return false;
    }
    HitRecord h;
    uint32_t kind = (b.ref >> RT_KIND_SHIFT) & 7u;
    bool maybe_uv = (FEAT & RT_FEAT_TEXTURE) != 0;
    Material m;
    if (!(FEAT & RT_FEAT_XPRIM)) {
        // spheres only: geometry and the sphere's own copy of its material record are
        // independent loads (one memory round trip)
        // A moving sphere (scenes of spheres and moving spheres only) is its base sphere seen from
        // the frame of its transform at the sample's time; the normal comes back out of the frame.
        uint32_t ref = b.ref;
        f3 fo = o, fd = d;
        MotionFrame frame;
        const bool moving = (FEAT & RT_FEAT_MOTION) && kind == RT_KIND_MOTION;
        if ((FEAT & RT_FEAT_MOTION) && moving) {
            frame = motion_frame(s, b.ref & RT_INDEX_MASK, time);
            fo = frame_in(frame, o - frame.t);
            fd = frame_in(frame, d);
            ref = frame.base;
        }
        uint32_t idx = ref & RT_INDEX_MASK;
        float4 g4 = ((const float4 *)s.spheres)[idx];
        const float4 *mr = (const float4 *)s.sphere_mrec + 2u * idx;
        float4 m0 = mr[0], m1 = mr[1];
        m = unpack_material(m0, m1);
        h.point = madd(d, b.t, o);
        f3 n = (madd(fd, b.t, fo) - mk3(g4.x, g4.y, g4.z)) * rcp(g4.w);
        if ((FEAT & RT_FEAT_MOTION) && moving) n = frame_out(frame, n);
        h.material = 0;
        h.u = h.v = 0.f;
        if (dot(n, d) > 0.f) { h.normal = -n; h.front = false; } else { h.normal = n; h.front = true; }
        if (maybe_uv && material_wants_uv(m.kind)) {
            Surface sf = prim_surface(s, ref, 0, fo, fd, b.t, true);
            h.u = sf.u; h.v = sf.v;
        }
    } else {
        complete_hit<(FEAT & RT_FEAT_MOTION) != 0, (FEAT & RT_FEAT_WIDE_CSG) != 0>(s, b, o, d, maybe_uv, h, time, true, &m);
    }
    f3 emitted, atten, nd;
    bool did = scatter<(FEAT & RT_FEAT_TEXTURE) != 0>(s, m, d, h, g, emitted, atten, nd);
    rad = rad + thr * emitted;
    if (!did) return false;
    thr = thr * atten;
    o = h.point;
    d = nd;
    depth++;
    return depth < max_depth;
}

// ------------------------------------------------------------------ wavefront pipeline
// The path queues are split into RT_WF_SHARDS shards, each with its own live counter on
// its own 128-byte line: one returning atomic per workgroup per 256 paths on one of eight
// words, instead of one per wave on a single word (a single word takes ~88 returning
// atomics per microsecond, which would cap the first bounces at ~6 Gpaths/s).
// Workgroup b works on shard b % 8 (the blocks that share b % 8 share an XCD under
// round-robin placement, so a shard's counter and queue stay in one L2; that is a speed
// matter only).  Paths are dealt to the shards in rounds of 256 so that every shard holds
// an even sample of the image.
#define RT_WF_SHARDS 8u
#define RT_WF_COUNTER_STRIDE 32u     // uint32 words between counters (128 B)

struct WfQueues {
    // (no radiance stream: a path gathers light only where it ends - at an emitter, which does not scatter, or in the
    // background - so the radiance of every path still in a queue is zero)
    float4 *ray_o[2], *ray_d[2], *thr[2];
    uint4 *rng[2];
    float4 *hit, *sample_rad;
    uint32_t *counts;        // [queue][shard], RT_WF_COUNTER_STRIDE apart
    uint32_t shard_cap;      // slots per shard
};
RT_DEV uint32_t *wf_counter(const WfQueues &q, int queue, uint32_t shard) {
    return q.counts + ((uint32_t)queue * RT_WF_SHARDS + shard) * RT_WF_COUNTER_STRIDE;
}

#define RT_WF_NEXT_OFFSET (2u * RT_WF_SHARDS * RT_WF_COUNTER_STRIDE)       // words: the shards' `next` counters follow the live counters
#define RT_WF_REFILL 24u
#define RT_WF_CHUNK 512u               // paths a wave takes from its shard per atomic
RT_DEV uint32_t *wf_next(const WfQueues &q, uint32_t shard) { return q.counts + RT_WF_NEXT_OFFSET + shard * RT_WF_COUNTER_STRIDE; }

// Ray generation: path i of the batch is sample i / npix of pixel i % npix.
__global__ __launch_bounds__(RT_BLOCK) void k_wf_raygen(WfQueues q, RtDevCamera cam, RtDevRender rc, uint32_t batch_spp) {
    uint32_t npix = rc.width * rc.height, total = npix * batch_spp;
    uint64_t seed = ((uint64_t)rc.seed_hi << 32) | rc.seed_lo;
    if (blockIdx.x == 0 && threadIdx.x < RT_WF_SHARDS) {
        uint32_t sh = threadIdx.x, full = total >> 8, rem = total & 255u;
        uint32_t cnt = (full / RT_WF_SHARDS) * 256u + (sh < full % RT_WF_SHARDS ? 256u : 0u) + (sh == full % RT_WF_SHARDS ? rem : 0u);
        *wf_counter(q, 0, sh) = cnt;
        *wf_counter(q, 1, sh) = 0;
        *wf_next(q, sh) = 0;
    }
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < total; i += gridDim.x * RT_BLOCK) {
        uint32_t pix = i % npix, sidx = i / npix;
        uint32_t x = pix % rc.width, y = pix / rc.width;
        Pcg g = sample_stream(seed, rc.first_sample + sidx, x, y);
        f3 o, d;
        camera_ray(cam, x, y, g, o, d);
        float time = rc.animated ? rnd(g) : 0.f;     // Camera.swift:89: after the ray, one per sample
        uint32_t chunk = i >> 8;
        uint32_t slot = (chunk % RT_WF_SHARDS) * q.shard_cap + (chunk / RT_WF_SHARDS) * 256u + (i & 255u);
        q.ray_o[0][slot] = make_float4(o.x, o.y, o.z, __uint_as_float(i));
        q.ray_d[0][slot] = make_float4(d.x, d.y, d.z, __uint_as_float(0u));
        q.thr[0][slot] = make_float4(1.f, 1.f, 1.f, time);
        q.rng[0][slot] = make_uint4((uint32_t)g.state, (uint32_t)(g.state >> 32), (uint32_t)g.inc, (uint32_t)(g.inc >> 32));
    }
}

// Extend: closest hit of every live path of queue `cur`.  Waves are persistent over their shard of the queue: a lane
// whose ray has found its closest hit stores it and, once `WF_REFILL` lanes of the wave are free, the free lanes take
// the shard's next paths (one atomic per wave on the shard's `next` counter), so the node steps and leaf tests of a wave
// run on rays that are still walking instead of on the longest walk of a fixed set of 64.  The walk itself is the fused
// kernel's: per-lane resumable traversal, stack in LDS, node steps and leaf tests as wave-wide phases, 32-byte nodes
// with byte-offset references where the scene has them.
// STACK: stack entries per lane (16 for trees of at most 16 levels: 16 KB of LDS per workgroup instead of 24).
template <int FEAT, bool NODE16, int STACK>
__global__ __launch_bounds__(RT_BLOCK) void k_wf_extend(RtDevScene s, WfQueues q, int cur) {
    __shared__ uint32_t stack_mem[STACK * RT_BLOCK];
    constexpr bool SIMPLE = !(FEAT & RT_FEAT_XPRIM);
    constexpr bool MOTION = (FEAT & RT_FEAT_MOTION) != 0;
    constexpr bool WIDE = (FEAT & RT_FEAT_WIDE_CSG) != 0;
    const uint32_t shard = blockIdx.x % RT_WF_SHARDS;
    const uint32_t count = *wf_counter(q, cur, shard);
    if (blockIdx.x / RT_WF_SHARDS == 0 && threadIdx.x == 0) *wf_counter(q, cur ^ 1, shard) = 0;   // shade appends to the other queue
    uint32_t *next = wf_next(q, shard);
    const uint32_t lane = lane_id();
    uint32_t *stack = stack_mem + threadIdx.x;
    const float4 *nodes = (const float4 *)s.nodes;
    const uint4 *nodes16 = (const uint4 *)s.nodes16;
    const uint32_t root = (NODE16 && !(s.root & RT_REF_LEAF)) ? s.root * 32u : s.root;
    const uint32_t node_min = 8u, leaf_max = SIMPLE ? 12u : 65u, heavy_min = 8u;
    Trav t;
    trav_begin(t, RT_REF_NONE, mk3(0, 0, 0), mk3(0, 0, 1), RT_INF, stack);
    // (Measured and left out: reading a lane's next ray from the queue while it still walks the current one.  Loads
    // return in order, so the first node fetch after a hand-out waited for the queue read all the same: 2.70 -> 2.36
    // Gsamples/s on the random-spheres workload.)
    // A wave takes a chunk of consecutive paths of its shard per atomic and deals them to its lanes over several
    // hand-outs: one word serves about 90 returning atomics per microsecond, and an atomic per hand-out (some 30 rays)
    // was 19 million of them per pass on eight words - 26 ms, more than the walk itself.
    // (a chunk is a quarter of a wave's even share of the shard, between 64 and RT_WF_CHUNK paths: small queues still
    // spread over all the waves)
    const uint32_t waves_on_shard = (gridDim.x / RT_WF_SHARDS) * RT_WAVES;
    const uint32_t chunk = min(RT_WF_CHUNK, max(64u, (count / (waves_on_shard * 4u) + 63u) & ~63u));
    uint32_t slot = 0, mine_next = 0, mine_end = 0;
    bool busy = false, more = count != 0u;
    uint64_t key = 0;
    float time = 0.f;
    while (true) {
        if (busy && t.cur == RT_REF_NONE) {
            q.hit[slot] = make_float4(t.best.t, __uint_as_float(t.found ? t.best.ref : RT_REF_NONE), __uint_as_float(t.best.code), 0.f);
            busy = false;
        }
        const uint64_t free_mask = __ballot(!busy);
        if (free_mask && mine_next == mine_end && more) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(next, chunk);
            base = __builtin_amdgcn_readfirstlane(base);
            mine_next = min(base, count);
            mine_end = min(base + chunk, count);
            if (base + chunk >= count) more = false;           // the shard has nothing beyond this chunk
        }
        if (free_mask && mine_next < mine_end) {
            const uint32_t n_free = (uint32_t)__popcll(free_mask);
            if (!busy) {
                const uint32_t j = mine_next + prefix_rank(free_mask);
                if (j < mine_end) {
                    slot = shard * q.shard_cap + j;
                    const float4 ro = q.ray_o[cur][slot], rd = q.ray_d[cur][slot];
                    if (FEAT & RT_FEAT_XPRIM) {
                        const uint4 r = q.rng[cur][slot];
                        key = ((uint64_t)r.y << 32) | r.x;
                    }
                    if (MOTION) time = q.thr[cur][slot].w;
                    trav_begin(t, root, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), RT_INF, stack);
                    busy = true;
                }
            }
            mine_next = min(mine_next + n_free, mine_end);
        }
        if (__ballot(busy) == 0) {
            if (mine_next == mine_end && !more) break;         // nothing in flight, nothing left to take
            continue;
        }
        while (true) {
            while (true) {
                const bool inner = !(t.cur & RT_REF_LEAF);
                const uint32_t n_inner = (uint32_t)__popcll(__ballot(inner));
                if (n_inner == 0) break;
                const bool waits = !inner && t.cur != RT_REF_NONE;
                uint32_t n_ready = (uint32_t)__popcll(__ballot(waits));
                if (!SIMPLE) {
                    const uint32_t n_heavy = (uint32_t)__popcll(__ballot(waits && ((t.cur >> RT_KIND_SHIFT) & 7u) >= RT_KIND_CSG));
                    if (n_heavy < heavy_min) n_ready -= n_heavy;
                }
                if ((n_inner < node_min && n_ready) || n_ready >= leaf_max) break;
                if (inner) {
                    if (NODE16) trav_node16(nodes16, t, RT_T_MIN, stack);
                    else trav_node(nodes, t, RT_T_MIN, stack);
                }
                if (n_inner >= RT_DOUBLE_STEP) {
                    if (!(t.cur & RT_REF_LEAF)) {
                        if (NODE16) trav_node16(nodes16, t, RT_T_MIN, stack);
                        else trav_node(nodes, t, RT_T_MIN, stack);
                    }
                }
            }
            bool at_leaf = (t.cur & RT_REF_LEAF) && t.cur != RT_REF_NONE;
            if (!SIMPLE) {
                const bool heavy = at_leaf && ((t.cur >> RT_KIND_SHIFT) & 7u) >= RT_KIND_CSG;
                const uint32_t n_heavy = (uint32_t)__popcll(__ballot(heavy));
                if (n_heavy && n_heavy < heavy_min) {
                    const bool others = __ballot(!(t.cur & RT_REF_LEAF) || (at_leaf && !heavy)) != 0;
                    if (others) at_leaf = at_leaf && !heavy;
                }
            }
            if (at_leaf) {
                test_object<SIMPLE, MOTION, false, false, WIDE>(s, t.cur & ~RT_REF_LEAF, t.o, t.d, RT_T_MIN, RT_INF, key, nullptr, 0, t.found, t.best, time);
                trav_pop(t, stack);
            }
            const uint32_t n_active = (uint32_t)__popcll(__ballot(t.cur != RT_REF_NONE));
            // back to the hand-out once enough lanes are free; when the shard is used up, only when the wave has finished
            if (n_active == 0 || ((more || mine_next < mine_end) && 64u - n_active >= RT_WF_REFILL)) break;
        }
    }
}

// Shade: scatter every live path of queue `cur`; survivors are appended, densely, to the
// same shard of the other queue (wave ballots, one atomic per workgroup per 256 paths);
// finished samples go to their slot.
template <int FEAT>
__global__ __launch_bounds__(RT_BLOCK) void k_wf_shade(RtDevScene s, WfQueues q, int cur, uint32_t background,
                                                        uint32_t max_depth) {
    __shared__ uint32_t wave_count[RT_WAVES];
    __shared__ uint32_t block_base;
    const uint32_t lane = lane_id(), wid = threadIdx.x >> 6;
    const uint32_t shard = blockIdx.x % RT_WF_SHARDS, lb = blockIdx.x / RT_WF_SHARDS, nb = gridDim.x / RT_WF_SHARDS;
    const uint32_t count = *wf_counter(q, cur, shard);
    const int nxt = cur ^ 1;
    if (lb == 0 && threadIdx.x == 0) *wf_next(q, shard) = 0;          // the next extend launch hands out from the start again
    for (uint32_t jb = lb * RT_BLOCK; jb < count; jb += nb * RT_BLOCK) {     // uniform over the workgroup
        uint32_t j = jb + threadIdx.x, i = shard * q.shard_cap + j;
        bool survive = false;
        f3 o, d, thr, rad;
        float time = 0.f;
        uint32_t pid = 0, depth = 0;
        Pcg g;
        g.state = g.inc = 0;
        if (j < count) {
            float4 ro = q.ray_o[cur][i], rd = q.ray_d[cur][i], t4 = q.thr[cur][i], h4 = q.hit[i];
            uint4 rr = q.rng[cur][i];
            o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z);
            thr = mk3(t4.x, t4.y, t4.z); rad = mk3(0.f, 0.f, 0.f);
            time = t4.w;
            pid = __float_as_uint(ro.w); depth = __float_as_uint(rd.w);
            g.state = ((uint64_t)rr.y << 32) | rr.x;
            g.inc = ((uint64_t)rr.w << 32) | rr.z;
            Best b;
            b.t = h4.x; b.ref = __float_as_uint(h4.y); b.code = __float_as_uint(h4.z);
            survive = shade_path<FEAT>(s, background, max_depth, b.ref != RT_REF_NONE, b, o, d, thr, rad, depth, g, time);
            if (!survive) q.sample_rad[pid] = make_float4(rad.x, rad.y, rad.z, 0.f);
        }
        uint64_t mask = __ballot(survive);
        if (lane == 0) wave_count[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (uint32_t w = 0; w < RT_WAVES; w++) tot += wave_count[w];
            block_base = tot ? atomicAdd(wf_counter(q, nxt, shard), tot) : 0u;
        }
        __syncthreads();
        if (survive) {
            uint32_t before = 0;
            for (uint32_t w = 0; w < wid; w++) before += wave_count[w];
            uint32_t k = shard * q.shard_cap + block_base + before + prefix_rank(mask);
            q.ray_o[nxt][k] = make_float4(o.x, o.y, o.z, __uint_as_float(pid));
            q.ray_d[nxt][k] = make_float4(d.x, d.y, d.z, __uint_as_float(depth));
            q.thr[nxt][k] = make_float4(thr.x, thr.y, thr.z, time);
            q.rng[nxt][k] = make_uint4((uint32_t)g.state, (uint32_t)(g.state >> 32), (uint32_t)g.inc, (uint32_t)(g.inc >> 32));
        }
        __syncthreads();          // wave_count / block_base are rewritten by the next round
    }
}

// Tail: once only a handful of paths are left (deep chains inside glass: a few thousand of the 16 million a batch starts
// with), a launch per stage and bounce costs more than the work in it - an extend launch cannot finish in less than the
// 18 us one ray's chain of dependent fetches takes.  This kernel takes what is left in queue `cur` and runs every path to
// its end, extend and shade alternating in the lane, so the remaining bounces are one launch.
#define RT_WF_TAIL_PATHS 65536u        // live paths at or below which the host hands the rest of a batch to the tail kernel
template <int FEAT>
__global__ __launch_bounds__(RT_BLOCK) void k_wf_tail(RtDevScene s, WfQueues q, int cur, uint32_t background, uint32_t max_depth) {
    __shared__ uint32_t stack[RT_STACK_DEPTH * RT_BLOCK];
    const uint32_t shard = blockIdx.x % RT_WF_SHARDS, lb = blockIdx.x / RT_WF_SHARDS, nb = gridDim.x / RT_WF_SHARDS;
    const uint32_t count = *wf_counter(q, cur, shard);
    for (uint32_t j = lb * RT_BLOCK + threadIdx.x; j < count; j += nb * RT_BLOCK) {
        const uint32_t i = shard * q.shard_cap + j;
        const float4 ro = q.ray_o[cur][i], rd = q.ray_d[cur][i], t4 = q.thr[cur][i];
        const uint4 rr = q.rng[cur][i];
        f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z), thr = mk3(t4.x, t4.y, t4.z), rad = mk3(0.f, 0.f, 0.f);
        const float time = t4.w;
        const uint32_t pid = __float_as_uint(ro.w);
        uint32_t depth = __float_as_uint(rd.w);
        Pcg g;
        g.state = ((uint64_t)rr.y << 32) | rr.x;
        g.inc = ((uint64_t)rr.w << 32) | rr.z;
        while (true) {
            Best b;
            const bool found = trace<!(FEAT & RT_FEAT_XPRIM), (FEAT & RT_FEAT_MOTION) != 0, (FEAT & RT_FEAT_WIDE_CSG) != 0>(
                s, o, d, RT_T_MIN, RT_INF, g.state, nullptr, 0, stack + threadIdx.x, b, time);
            if (!shade_path<FEAT>(s, background, max_depth, found, b, o, d, thr, rad, depth, g, time)) break;
        }
        q.sample_rad[pid] = make_float4(rad.x, rad.y, rad.z, 0.f);
    }
}

// Accumulate: per pixel, the batch's samples in sample order, then into the accumulator.
__global__ __launch_bounds__(RT_BLOCK) void k_wf_accumulate(const float4 *sample_rad, uint32_t npix, uint32_t batch_spp,
                                                             float *accum) {
    uint32_t p = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (p >= npix) return;
    float r = 0.f, g = 0.f, b = 0.f;
    for (uint32_t k = 0; k < batch_spp; k++) {
        float4 v = sample_rad[(size_t)k * npix + p];
        r += v.x; g += v.y; b += v.z;
    }
    accum[3 * (size_t)p + 0] += r;
    accum[3 * (size_t)p + 1] += g;
    accum[3 * (size_t)p + 2] += b;
}

// Diagnostic build only: the arithmetic of one object test priced with the constants the roofline uses
// (sphere 30, cylinder 95, cuboid 75, quad 45, 6 per merged CSG event - two events per leaf -, 20 per volume
// interval), so that the phase counters give the flops a kernel executes on scenes of any kind of object.
RT_DEV uint32_t prim_flops(uint32_t ref) {
    const uint32_t kind = (ref >> RT_KIND_SHIFT) & 7u;
    return kind == RT_KIND_SPHERE ? 30u : (kind == RT_KIND_CYLINDER ? 95u : (kind == RT_KIND_CUBOID ? 75u : 45u));
}
RT_DEV uint32_t object_flops(const RtDevScene &s, uint32_t ref) {
    uint32_t kind = (ref >> RT_KIND_SHIFT) & 7u, idx = ref & RT_INDEX_MASK, f = 0;
    if (kind == RT_KIND_MOTION) { ref = s.motion[idx].base; kind = (ref >> RT_KIND_SHIFT) & 7u; idx = ref & RT_INDEX_MASK; f += 33u; }
    if (kind == RT_KIND_CDV) { ref = s.cdv[idx].base; kind = (ref >> RT_KIND_SHIFT) & 7u; idx = ref & RT_INDEX_MASK; f += 20u; }
    if (kind != RT_KIND_CSG) return f + prim_flops(ref);
    const RtDevCsg &c = s.csg[idx];
    const uint32_t n = c.n_leaves;
    const uint32_t *leaves = n > RT_CSG_LEAVES ? s.csg_pool + c.leaf[0] + RT_CSG_POOL_HEADER : c.leaf;
    for (uint32_t i = 0; i < n; i++) f += prim_flops(leaves[i]) + 12u;
    return f;
}

// ------------------------------------------------------------------ fused persistent kernel
// Lanes whose traversal has ended wait until at least `refill` (default 48) lanes are finished or free, then the
// wave leaves the traversal loop, shades them (which usually starts their next bounce) and
// hands fresh samples to lanes whose path ended; the others keep their traversal state
// across the excursion.

// The camera lives in the workspace (RT_FUSED_CAMERA_OFFSET) instead of the kernel's argument
// registers: only ray generation reads it, and two dozen scalar registers held for the whole
// kernel made the compiler spill scalars inside the traversal loops.
#define RT_FUSED_CAMERA_OFFSET 2048u
#define RT_FUSED_DIAG_OFFSET 1024u   // the diagnostic build's phase counters
#define RT_ITEM_COUNTERS 8u
#define RT_ITEM_COUNTER_STRIDE 32u    // words between item counters (128 bytes): the first kilobyte of the workspace
#define RT_FUSED_DEEP_OFFSET 4096u
__global__ void k_fused_setup(uint32_t *workspace, RtDevCamera cam) {
    for (uint32_t i = threadIdx.x; i < RT_FUSED_CAMERA_OFFSET / 4u; i += blockDim.x) workspace[i] = 0u;   // item counters, diagnostics
    if (threadIdx.x == 0) *(RtDevCamera *)((char *)workspace + RT_FUSED_CAMERA_OFFSET) = cam;
}

// PROF = true is a diagnostic build: it counts, per phase, wave-level executions, the lanes
// that took part, and shader cycles, into the workspace header (RT_FUSED_DIAG_OFFSET).  Its run time
// means nothing (the stamps serialise the phases).
// STACK: LDS stack entries per lane; WPS: waves per SIMD the register allocation is bounded
// for.  Spheres-only scenes run at 8 waves per SIMD (64 VGPRs, 19 KB of LDS per workgroup: 16
// stack entries per lane; a deeper tree keeps the entries beyond those in global memory),
// textured sphere scenes at 5 (96 VGPRs, 27 KB), scenes with cylinders, boxes, quads, CSG
// shapes or volumes at 4 (128 VGPRs: at 96 their leaf tests spill and run at half the rate).
// DEEP: the tree may be deeper than STACK; entries from STACK up live in global memory behind
// the workspace header (RT_FUSED_DEEP_OFFSET), one column per thread of the grid.
// LIST: the scene is small enough to carry its objects as a plain list (RtDevScene::object_list);
// the extend phase then tests object 0, 1, 2 ... in turn, the same object in every lane: no tree, no
// stack, no divergence between kinds of object, and every lane with a path takes part.
// BLOCK: threads per workgroup.  TOP: nodes at the head of the 32-byte node array that every workgroup also keeps in
// LDS (the builder of a deep tree puts the nodes with the largest boxes there); a larger workgroup shares one copy
// among more waves.
// ARITY 4: the tree is read in its four-children form (RtDevScene::nodes4; TOP then counts 32-byte halves of its nodes).
template <int FEAT, bool PROF, int STACK, int WPS, bool NODE16, bool DEEP = false, bool LIST = false, int BLOCK = RT_BLOCK, int TOP = 0, int ARITY = 2>
__global__ __launch_bounds__(BLOCK, WPS) void k_fused(RtDevScene s, const RtDevCamera *camp, uint32_t background, RtDevRender rc, float *accum,
                                                     uint32_t *counter, uint32_t n_items, uint32_t tiles_x,
                                                     uint32_t n_tiles, uint32_t chunk_spp, uint32_t knobs, uint32_t heavy_and_counters) {
    __shared__ uint32_t stack_mem[STACK * BLOCK];
    constexpr int LDS_ENTRIES = DEEP ? STACK : 0;
    constexpr uint32_t STEP = (uint32_t)BLOCK * 4u;       // bytes between a lane's consecutive stack entries
    DeepStack deep_cols;
    deep_cols.column = (global_u32 *)(uintptr_t)(counter + RT_FUSED_DEEP_OFFSET / 4u) + blockIdx.x * BLOCK + threadIdx.x;
    deep_cols.stride = gridDim.x * BLOCK;
    deep_cols.lds_limit = lds_address(stack_mem) + (uint32_t)LDS_ENTRIES * STEP;
    const DeepStack *deep = DEEP ? &deep_cols : nullptr;
    __shared__ float tile_acc[BLOCK / 64][192];
    __shared__ __attribute__((aligned(16))) uint4 top_nodes[TOP ? 2 * TOP : 1];
    uint32_t top_lds = 0;
    if (TOP) {
        const uint32_t n_top = ARITY == 4 ? min(2u * TOP, s.n_nodes4 * 4u) : min((uint32_t)TOP, s.n_nodes) * 2u;
        const uint4 *head = (const uint4 *)(ARITY == 4 ? s.nodes4 : s.nodes16);
        for (uint32_t i = threadIdx.x; i < 2u * TOP; i += BLOCK) top_nodes[i] = i < n_top ? head[i] : make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        top_lds = lds_address((const uint32_t *)top_nodes);
    }
    constexpr bool SIMPLE = !(FEAT & RT_FEAT_XPRIM);
    const uint32_t refill = knobs & 255u, node_min = (knobs >> 8) & 255u;
    // Ray generation with a defocus disk (a rejection loop) costs about as much as six node steps:
    // it waits until 16 lanes are free.  Without the disk it is cheap and free lanes are not left idle.
    const uint32_t raygen_min = (knobs >> 18) & 127u;
    const uint32_t leaf_max = (knobs >> 25) & 127u;
    const uint32_t counters_log2 = heavy_and_counters >> 8, heavy_min = heavy_and_counters & 127u;
    const uint32_t lane = lane_id();
    uint32_t *stack = stack_mem + threadIdx.x;
    float *acc = tile_acc[threadIdx.x >> 6];
    const uint64_t seed = ((uint64_t)rc.seed_hi << 32) | rc.seed_lo;
    const float4 *nodes = (const float4 *)s.nodes;
    const uint4 *nodes16 = (const uint4 *)(ARITY == 4 ? s.nodes4 : s.nodes16);
    unsigned long long pc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // {execs, lanes, cycles} x {shade, raygen, node, leaf}, flops of the object tests, box tests of the list form
    unsigned long long stamp = 0;
#define RT_PROF_BEGIN() do { if (PROF) stamp = __builtin_amdgcn_s_memtime(); } while (0)
#define RT_PROF_END(k, cond) do { if (PROF) { unsigned long long m_ = __ballot(cond); if (m_) { pc[3 * (k)] += 1; pc[3 * (k) + 1] += __popcll(m_); pc[3 * (k) + 2] += __builtin_amdgcn_s_memtime() - stamp; } } } while (0)
    // adds a per-lane count over the lanes for which `cond` holds (the wave's total goes to lane 0's counter)
#define RT_PROF_SUM(k, cond, value) do { if (PROF) { uint32_t v_ = (cond) ? (value) : 0u; for (int o_ = 32; o_ > 0; o_ >>= 1) v_ += __shfl_xor(v_, o_); pc[k] += v_; } } while (0)
    // Work items (an 8x8 tile x a chunk of samples) are streamed: when the current item has
    // handed out its last sample the wave takes the next one at once and the free lanes go on
    // with it, while the lanes still walking paths of earlier items finish those on the side.
    // The tile sums in LDS always belong to the current item: they are flushed when the wave
    // moves on, and a straggler adds its radiance to memory itself ("direct").
    // (tx0, ty0): the tile's corner in the accumulator; gy0: its first row in the image (they
    // differ when the accumulator holds one shard's bands only)
    uint32_t next = 0, total = 0, tx0 = 0, ty0 = 0, gy0 = 0, s_begin = 0;
    bool more = true;                  // this workgroup's item counter has not run out yet
    bool has_path = false, direct = false;
    f3 thr = mk3(0, 0, 0), rad = mk3(0, 0, 0);
    uint32_t depth = 0, pxy = 0;       // accumulator pixel of the lane's path: x | y << 16
    uint64_t key = 0;
    float time = 0.f;                  // the sample's time; only scenes with moving objects draw one
    Pcg g;
    g.state = g.inc = 0;
    Trav t;
    trav_begin(t, RT_REF_NONE, mk3(0, 0, 0), mk3(0, 0, 1), RT_INF, stack);
    acc[lane] = 0.f; acc[lane + 64] = 0.f; acc[lane + 128] = 0.f;
    while (true) {
        // ---- shade phase: lanes whose traversal ended
        bool start = false;
        f3 o = t.o, d = t.d;
        RT_PROF_BEGIN();
        const bool shading = has_path && t.cur == RT_REF_NONE;
        // (Measured and left out: holding lanes whose hit wants a texture evaluated from noise tables or an image - five
        // octaves of Perlin noise cost some 600 instructions, usually for one or two lanes of the wave - until 4 / 8 / 16
        // of them can be shaded together: 5.18 -> 5.10 / 5.01 / 4.77 Gsamples/s on the CSG and textures workload, 6.19 ->
        // 6.15 / 6.12 / 5.94 on the Kit's first world.  A held lane is a lane that walks no ray.)
        if (shading) {
            bool alive = shade_path<FEAT>(s, background, rc.max_depth, t.found, t.best, o, d, thr, rad, depth, g, time);
            if (alive) {
                start = true;
            } else {
                // accumulate phase
                const uint32_t x = pxy & 0xFFFFu, y = pxy >> 16;
                if (direct) {
                    float *dst = accum + 3 * ((size_t)y * rc.width + x);
                    atomicAdd(dst + 0, rad.x);
                    atomicAdd(dst + 1, rad.y);
                    atomicAdd(dst + 2, rad.z);
                } else {
                    // LDS atomics: lanes of the wave may share a pixel
                    const uint32_t pix = (x - tx0) + 8u * (y - ty0);
                    atomicAdd(&acc[3u * pix + 0u], rad.x);
                    atomicAdd(&acc[3u * pix + 1u], rad.y);
                    atomicAdd(&acc[3u * pix + 2u], rad.z);
                }
                has_path = false;
            }
        }
        RT_PROF_END(0, shading);
        // ---- the current item is used up: flush its sums and take the next one
        if (next >= total && more) {
            if (total) {
                uint32_t x = tx0 + (lane & 7u), y = ty0 + (lane >> 3);
                if (x < rc.width && gy0 + (lane >> 3) < rc.height) {
                    // other chunks of the same tile may be in flight elsewhere: atomics
                    float *dst = accum + 3 * ((size_t)y * rc.width + x);
                    atomicAdd(dst + 0, acc[3u * lane + 0u]);
                    atomicAdd(dst + 1, acc[3u * lane + 1u]);
                    atomicAdd(dst + 2, acc[3u * lane + 2u]);
                }
                acc[lane] = 0.f; acc[lane + 64] = 0.f; acc[lane + 128] = 0.f;
            }
            direct = direct || has_path;
            // Items are dealt by up to RT_ITEM_COUNTERS counters on separate cache lines, counter c dealing items c,
            // c + n, c + 2n ...: one word takes about 90 returning atomics per microsecond, which a launch of many
            // small items (a few samples per pixel) would wait on.  A workgroup stays with one counter; eight
            // neighbouring workgroups (one per XCD under round-robin placement) share it, so every counter is served by
            // an even sample of the chip and, holding an equal share of interleaved items, they run out together.
            const uint32_t which_counter = (blockIdx.x >> 3) & ((1u << counters_log2) - 1u);
            uint32_t item = 0;
            if (lane == 0) item = atomicAdd(counter + RT_ITEM_COUNTER_STRIDE * which_counter, 1u);
            item = (__builtin_amdgcn_readfirstlane(item) << counters_log2) | which_counter;
            next = total = 0;
            if (item >= n_items) {
                more = false;                    // every wave gets here: the grid drains
            } else {
                const uint32_t tile = item % n_tiles, chunk = item / n_tiles;
                s_begin = chunk * chunk_spp;
                total = (min(rc.spp, s_begin + chunk_spp) - s_begin) * 64u;
                const uint32_t tile_row = tile / tiles_x;
                tx0 = (tile % tiles_x) * 8u;
                ty0 = tile_row * 8u;
                gy0 = (tile_row * rc.band_count + rc.band_index) * 8u;
            }
        }
        RT_PROF_BEGIN();
        const bool generating = !has_path && next < total;
        // ---- raygen phase: lanes without a path take the item's next samples
        const uint64_t need = __ballot(!has_path);
        if (next < total && ((uint32_t)__popcll(need) >= raygen_min || __popcll(need) == 64)) {
            // The camera is read here, by scalar loads (it is constant while the kernel runs): the
            // pointer is taken afresh each time so that the two dozen values live in scalar
            // registers during ray generation only.
            const RtDevCamera *cp = camp;
            asm volatile("" : "+s"(cp));
            const __attribute__((address_space(4))) uint32_t *cw = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)cp;
            RtDevCamera cam_now;
#pragma unroll
            for (uint32_t i = 0; i < sizeof(RtDevCamera) / 4u; i++) ((uint32_t *)&cam_now)[i] = cw[i];
            if (!has_path) {
                uint32_t k = next + prefix_rank(need);
                if (k < total) {
                    const uint32_t row = (k >> 3) & 7u;
                    uint32_t x = tx0 + (k & 7u), y = gy0 + row;
                    if (x < rc.width && y < rc.height) {
                        g = sample_stream(seed, rc.first_sample + s_begin + (k >> 6), x, y);
                        camera_ray(cam_now, x, y, g, o, d);
                        // the all-features kernel serves still scenes too: it asks the render record
                        if ((FEAT & RT_FEAT_MOTION) && (!(FEAT & RT_FEAT_WIDE_CSG) || rc.animated)) time = rnd(g);
                        thr = mk3(1, 1, 1);
                        rad = mk3(0, 0, 0);
                        depth = 0;
                        pxy = x | ((ty0 + row) << 16);      // where the sample is accumulated
                        direct = false;
                        has_path = true;
                        start = true;
                    }
                }
            }
            next += (uint32_t)__popcll(need);
        }
        RT_PROF_END(1, generating);
        if (start) {
            key = g.state;
            // the 32-byte node form walks byte offsets
            trav_begin(t, ARITY == 4 ? s.root4 : (NODE16 && !(s.root & RT_REF_LEAF)) ? s.root * 32u : s.root, o, d, RT_INF, stack);
        }
        const uint32_t n_paths = (uint32_t)__popcll(__ballot(has_path));
        if (n_paths == 0) {
            if (!more && next >= total) break;       // nothing left to hand out, nothing in flight
            continue;
        }
        // ---- extend phase
        // (Measured and left out: noting, per lane, the objects whose box the ray enters and then letting every lane test
        // one of its own noted objects per round - two fog boxes in one round of volume code instead of two.  The records
        // then come through per-lane vector loads, three dependent ones deep (list entry, volume, primitive), where the
        // uniform form reads them through the scalar cache: Cornell with smoke 8.26 -> 6.98 Gsamples/s, the Kit's first
        // world 5.85 -> 5.59.)
        if (LIST) {
            // Array<any Hittable>.hit, RayTracingKit/Hittable.swift:662-671: closest hit over the list,
            // the range shrinking as hits are found.  The object is the same in all lanes (its record
            // comes through scalar loads); boxes are only tested where the object's own test costs more.
            RT_PROF_BEGIN();
            const bool walking = t.cur != RT_REF_NONE;
            const uint32_t n_listed = s.n_listed;
            for (uint32_t k = 0; k < n_listed; k++) {
                const __attribute__((address_space(4))) uint32_t *rec =
                    (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)(s.object_list + 8u * k);
                const uint32_t ref = rec[0];
                const uint32_t okind = (ref >> RT_KIND_SHIFT) & 7u;
                bool wanted = walking;
                if (okind != RT_KIND_SPHERE && okind != RT_KIND_QUAD) {
                    const float x0 = __uint_as_float(rec[1]), x1 = __uint_as_float(rec[2]), y0 = __uint_as_float(rec[3]);
                    const float y1 = __uint_as_float(rec[4]), z0 = __uint_as_float(rec[5]), z1 = __uint_as_float(rec[6]);
                    v2f tx = {fmaf(x0, t.inv_d.x, t.oid.x), fmaf(x1, t.inv_d.x, t.oid.x)};
                    v2f ty = {fmaf(y0, t.inv_d.y, t.oid.y), fmaf(y1, t.inv_d.y, t.oid.y)};
                    v2f tz = {fmaf(z0, t.inv_d.z, t.oid.z), fmaf(z1, t.inv_d.z, t.oid.z)};
                    bool in_box;
                    (void)box_entry(tx, ty, tz, RT_T_MIN, t.best.t, in_box);
                    RT_PROF_SUM(13, walking, 1u);
                    wanted = wanted && in_box;
                }
                RT_PROF_SUM(12, wanted, object_flops(s, ref));
                if (wanted)
                    test_object<SIMPLE, (FEAT & RT_FEAT_MOTION) != 0, (FEAT & RT_FEAT_TEXTURE) != 0, true>(s, ref, t.o, t.d, RT_T_MIN, RT_INF, key, nullptr, 0, t.found, t.best, time);
            }
            if (walking) t.cur = RT_REF_NONE;
            RT_PROF_END(3, walking);
            continue;
        }
        while (true) {
            // node steps go on while enough lanes sit on inner nodes; once fewer than
            // node_min do and some lane waits at a leaf, or leaf_max lanes wait there, the
            // leaf tests get their turn.  Heavy leaves (CSG shapes, volumes, moving objects: an
            // order of magnitude more instructions than a primitive, and rare) are held back
            // until heavy_min lanes wait at one or nothing else can move: the wave then runs
            // that code once for many lanes instead of once per lane.
            while (true) {
                const bool inner = !(t.cur & RT_REF_LEAF);
                const uint32_t n_inner = (uint32_t)__popcll(__ballot(inner));
                if (n_inner == 0) break;
                const bool waits = !inner && t.cur != RT_REF_NONE;
                uint32_t n_ready = (uint32_t)__popcll(__ballot(waits));
                if (!SIMPLE) {
                    const uint32_t n_heavy = (uint32_t)__popcll(__ballot(waits && ((t.cur >> RT_KIND_SHIFT) & 7u) >= RT_KIND_CSG));
                    if (n_heavy < heavy_min) n_ready -= n_heavy;
                }
                if ((n_inner < node_min && n_ready) || n_ready >= leaf_max) break;
                RT_PROF_BEGIN();
                if (inner) {
                    if (ARITY == 4) trav_node4<LDS_ENTRIES, STEP, TOP * 32u>(nodes16, t, RT_T_MIN, stack, deep, top_lds);
                    else if (NODE16) trav_node16<LDS_ENTRIES, STEP, TOP * 32u>(nodes16, t, RT_T_MIN, stack, deep, top_lds);
                    else trav_node<LDS_ENTRIES, STEP>(nodes, t, RT_T_MIN, stack, deep);
                }
                RT_PROF_END(2, inner);
                if (n_inner >= RT_DOUBLE_STEP) {
                    // most lanes are on inner nodes: a second step before the counts are taken again
                    RT_PROF_BEGIN();
                    const bool inner2 = !(t.cur & RT_REF_LEAF);
                    if (inner2) {
                        if (ARITY == 4) trav_node4<LDS_ENTRIES, STEP, TOP * 32u>(nodes16, t, RT_T_MIN, stack, deep, top_lds);
                        else if (NODE16) trav_node16<LDS_ENTRIES, STEP, TOP * 32u>(nodes16, t, RT_T_MIN, stack, deep, top_lds);
                        else trav_node<LDS_ENTRIES, STEP>(nodes, t, RT_T_MIN, stack, deep);
                    }
                    RT_PROF_END(2, inner2);
                }
            }
            RT_PROF_BEGIN();
            bool at_leaf = (t.cur & RT_REF_LEAF) && t.cur != RT_REF_NONE;
            if (!SIMPLE) {
                const bool heavy = at_leaf && ((t.cur >> RT_KIND_SHIFT) & 7u) >= RT_KIND_CSG;
                const uint32_t n_heavy = (uint32_t)__popcll(__ballot(heavy));
                if (n_heavy && n_heavy < heavy_min) {
                    // held back unless nothing else can move: no lane on an inner node, none at a light leaf
                    const bool others = __ballot(!(t.cur & RT_REF_LEAF) || (at_leaf && !heavy)) != 0;
                    if (others) at_leaf = at_leaf && !heavy;
                }
            }
            RT_PROF_SUM(12, at_leaf, object_flops(s, t.cur & ~RT_REF_LEAF));
            if (at_leaf) {
                test_object<SIMPLE, (FEAT & RT_FEAT_MOTION) != 0, (FEAT & RT_FEAT_TEXTURE) != 0, false, (FEAT & RT_FEAT_WIDE_CSG) != 0>(s, t.cur & ~RT_REF_LEAF, t.o, t.d, RT_T_MIN, RT_INF, key, nullptr, 0, t.found, t.best, time);
                trav_pop<LDS_ENTRIES, STEP>(t, stack, deep);
            }
            RT_PROF_END(3, at_leaf);
            uint32_t n_active = (uint32_t)__popcll(__ballot(t.cur != RT_REF_NONE));
            if (n_active == 0 || 64u - n_active >= refill) break;     // finished or free lanes
        }
    }
    if (PROF && lane == 0) {
        unsigned long long *out = (unsigned long long *)(counter + RT_FUSED_DIAG_OFFSET / 4u);
        for (int k = 0; k < 14; k++) atomicAdd(out + k, pc[k]);
    }
#undef RT_PROF_BEGIN
#undef RT_PROF_END
#undef RT_PROF_SUM
}

// ------------------------------------------------------------------ launchers
#define RT_WF_HEADER 4096u          // the shard counters
static uint32_t wf_shard_cap(uint32_t capacity) {
    uint32_t per = (capacity + RT_WF_SHARDS - 1u) / RT_WF_SHARDS;
    return (per + 255u) / 256u * 256u + 256u;      // whole rounds of 256, plus the ragged last round
}
size_t rt_wavefront_bytes(uint32_t capacity) {
    size_t slots = (size_t)wf_shard_cap(capacity) * RT_WF_SHARDS;
    return RT_WF_HEADER + slots * (2 * 4 * 16 + 16) + (size_t)capacity * 16;      // two queues of four 16-byte streams, the hits, the finished samples
}

void rt_wavefront_carve(void *workspace, uint32_t capacity, RtWavefrontBuffers &b) {
    char *p = (char *)workspace;
    b.counts = (uint32_t *)p;
    p += RT_WF_HEADER;
    b.shard_cap = wf_shard_cap(capacity);
    size_t stride = (size_t)b.shard_cap * RT_WF_SHARDS * 16;
    for (int k = 0; k < 2; k++) {
        b.ray_o[k] = (float4 *)p; p += stride;
        b.ray_d[k] = (float4 *)p; p += stride;
        b.thr[k] = (float4 *)p; p += stride;
        b.rng[k] = (uint4 *)p; p += stride;
    }
    b.hit = (float4 *)p; p += stride;
    b.sample_rad = (float4 *)p;
    b.capacity = capacity;
}

static inline int blocks_for(size_t n) { return (int)((n + RT_BLOCK - 1) / RT_BLOCK); }

hipError_t rt_launch_first_hit(const RtDevScene &s, uint32_t feat, const float *rays, int n, float tmin, float tmax,
                               uint64_t volume_key, float time, RtHit *out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (feat & RT_FEAT_WIDE_CSG)
        k_first_hit<15><<<blocks_for(n), RT_BLOCK, 0, stream>>>(s, rays, n, tmin, tmax, volume_key, time, out);
    else if (feat & RT_FEAT_MOTION)
        k_first_hit<7><<<blocks_for(n), RT_BLOCK, 0, stream>>>(s, rays, n, tmin, tmax, volume_key, time, out);
    else if (feat & RT_FEAT_XPRIM)
        k_first_hit<3><<<blocks_for(n), RT_BLOCK, 0, stream>>>(s, rays, n, tmin, tmax, volume_key, time, out);
    else
        k_first_hit<2><<<blocks_for(n), RT_BLOCK, 0, stream>>>(s, rays, n, tmin, tmax, volume_key, time, out);
    return hipGetLastError();
}

hipError_t rt_launch_intervals(const RtDevScene &s, uint32_t ref, const float *rays, int n, const float *scripted,
                               int n_scripted, uint64_t volume_key, uint32_t object, float time, RtEvent *events,
                               int32_t *counts, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    k_intervals<<<blocks_for(n), RT_BLOCK, 0, stream>>>(s, ref, rays, n, scripted, n_scripted, volume_key, object, time, events, counts);
    return hipGetLastError();
}

hipError_t rt_launch_camera_rays(const RtDevCamera &cam, const RtDevRender &rc, int x0, int y0, int w, int h,
                                 float *rays_out, hipStream_t stream) {
    size_t total = (size_t)w * h * rc.spp;
    if (total == 0) return hipSuccess;
    k_camera_rays<<<blocks_for(total), RT_BLOCK, 0, stream>>>(cam, rc, x0, y0, w, h, rays_out);
    return hipGetLastError();
}

hipError_t rt_launch_resolve(const float *accum, float n_samples, int width, int height, uint8_t *rgbx, hipStream_t stream) {
    int npix = width * height;
    if (npix <= 0) return hipSuccess;
    k_resolve<<<blocks_for(npix), RT_BLOCK, 0, stream>>>(accum, n_samples, npix, rgbx);
    return hipGetLastError();
}

hipError_t rt_launch_scatter(const RtDevScene &s, const RtScatterQuery *queries, int n, uint64_t seed, RtScatterResult *out,
                             hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    k_scatter<<<blocks_for(n), RT_BLOCK, 0, stream>>>(s, queries, n, seed, out);
    return hipGetLastError();
}

hipError_t rt_launch_rng_words(const uint32_t *streams, int n, int words_per_stream, uint32_t *out, uint32_t *wyrand_out,
                               hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    k_rng_words<<<blocks_for(n), RT_BLOCK, 0, stream>>>(streams, n, words_per_stream, out, wyrand_out);
    return hipGetLastError();
}

hipError_t rt_launch_progressive_present(const float *pass_sum, uint32_t spp, uint32_t pass_counter, int width, int height,
                                         uint32_t *acc, uint8_t *bgra, hipStream_t stream) {
    int npix = width * height;
    if (npix <= 0) return hipSuccess;
    k_progressive_present<<<blocks_for(npix), RT_BLOCK, 0, stream>>>(pass_sum, (float)spp, 1.0f / (float)pass_counter, npix, acc,
                                                                     (uint32_t *)bgra);
    return hipGetLastError();
}

// Header (item counters, diagnostics, camera) + the deep-stack columns of the largest grid.
size_t rt_fused_workspace_bytes(int cus) {
    return RT_FUSED_DEEP_OFFSET + (size_t)cus * 8u * RT_BLOCK * (RT_WIDE_STACK_DEPTH - 16) * 4u;
}
// The least a deep tree needs for the kernels that keep 16 stack entries per lane in LDS: columns for the binary form.
static size_t fused_workspace_bytes_binary(int cus) {
    return RT_FUSED_DEEP_OFFSET + (size_t)cus * 8u * RT_BLOCK * (RT_STACK_DEPTH - 16) * 4u;
}

int rt_current_device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}

template <int FEAT>
static hipError_t wavefront_batch(const RtDevScene &s, const RtDevCamera &cam, const RtDevRender &rc, uint32_t batch_spp,
                                  const WfQueues &q, float *accum, hipStream_t stream, RtKernelTimer *timer, int grid, RtWfProbe *probe,
                                  bool shallow) {
    uint32_t npix = rc.width * rc.height;
    k_wf_raygen<<<grid, RT_BLOCK, 0, stream>>>(q, cam, rc, batch_spp);
    const bool probing = probe && probe->init();
    const uint32_t first_probe = 1;
    // Paths only ever leave the queues, so the live count read RT_WF_LAG bounces ago bounds this bounce's.
    uint32_t live_bound = npix * batch_spp;
    const int full_grid = grid;
    for (uint32_t bounce = 0; bounce < rc.max_depth; bounce++) {
        int cur = (int)(bounce & 1u);
        if (live_bound <= RT_WF_TAIL_PATHS) {
            // few paths left (the count is RT_WF_LAG bounces old, so fewer still): the rest of their bounces in one launch
            k_wf_tail<FEAT><<<std::min(full_grid, (int)(RT_WF_TAIL_PATHS / RT_BLOCK)), RT_BLOCK, 0, stream>>>(s, q, cur, cam.background, rc.max_depth);
            break;
        }
        if (timer) timer->begin(stream);
        // (a tree of at most 16 levels: 16 KB of LDS per workgroup, ten of them to a CU when the registers allow)
        if (s.nodes16 && shallow) k_wf_extend<FEAT, true, 16><<<grid / 8 * 10, RT_BLOCK, 0, stream>>>(s, q, cur);
        else if (s.nodes16) k_wf_extend<FEAT, true, RT_STACK_DEPTH><<<grid, RT_BLOCK, 0, stream>>>(s, q, cur);
        else k_wf_extend<FEAT, false, RT_STACK_DEPTH><<<grid, RT_BLOCK, 0, stream>>>(s, q, cur);
        if (timer) timer->end(stream);
        k_wf_shade<FEAT><<<grid, RT_BLOCK, 0, stream>>>(s, q, cur, cam.background, rc.max_depth);
        if (!probing || bounce + 1 >= rc.max_depth) continue;
        // the queues drain geometrically: the live count of the queue just filled goes to the host without a stream
        // synchronisation, and is looked at RT_WF_LAG bounces later (see RtWfProbe)
        if (bounce >= first_probe) {
            const uint32_t slot = bounce % RT_WF_PROBES;
            hipError_t e = hipMemcpyAsync(probe->host + slot * 256u, q.counts + (uint32_t)(cur ^ 1) * RT_WF_SHARDS * RT_WF_COUNTER_STRIDE, 1024, hipMemcpyDeviceToHost, stream);
            if (e != hipSuccess) return e;
            e = hipEventRecord(probe->ev[slot], stream);
            if (e != hipSuccess) return e;
        }
        if (bounce >= first_probe + RT_WF_LAG) {
            const uint32_t slot = (bounce - RT_WF_LAG) % RT_WF_PROBES;
            hipError_t e = hipEventSynchronize(probe->ev[slot]);
            if (e != hipSuccess) return e;
            uint32_t live = 0;
            for (uint32_t sh = 0; sh < RT_WF_SHARDS; sh++) live += probe->host[slot * 256u + sh * RT_WF_COUNTER_STRIDE];
            if (live == 0) break;
            live_bound = live;
        }
    }
    k_wf_accumulate<<<blocks_for(npix), RT_BLOCK, 0, stream>>>(q.sample_rad, npix, batch_spp, accum);
    return hipGetLastError();
}

hipError_t rt_render_wavefront(const RtDevScene &s, uint32_t feat, const RtDevCamera &cam, RtDevRender rc, float *accum,
                               void *workspace, size_t workspace_bytes, hipStream_t stream, RtKernelTimer *timer, int cus, RtWfProbe *probe,
                               uint32_t bvh_depth) {
    const bool shallow = bvh_depth <= 16;
    uint32_t npix = rc.width * rc.height;
    if (npix == 0 || rc.spp == 0) return hipSuccess;
    if (workspace_bytes < rt_wavefront_bytes(npix)) return hipErrorInvalidValue;
    uint32_t max_batch = RT_WAVEFRONT_BATCH_SPP;
    while (max_batch > 1 && rt_wavefront_bytes(npix * max_batch) > workspace_bytes) max_batch--;
    RtWavefrontBuffers b;
    rt_wavefront_carve(workspace, npix * max_batch, b);
    WfQueues q;
    for (int k = 0; k < 2; k++) { q.ray_o[k] = b.ray_o[k]; q.ray_d[k] = b.ray_d[k]; q.thr[k] = b.thr[k]; q.rng[k] = b.rng[k]; }
    q.hit = b.hit; q.sample_rad = b.sample_rad; q.counts = b.counts; q.shard_cap = b.shard_cap;
    if (timer) timer->name = "k_wf_extend";
    uint32_t total_spp = rc.spp, first = rc.first_sample;
    for (uint32_t s0 = 0; s0 < total_spp; s0 += max_batch) {
        uint32_t batch = std::min(max_batch, total_spp - s0);
        RtDevRender r = rc;
        r.first_sample = first + s0;
        r.spp = batch;
        int grid = std::min(blocks_for((size_t)npix * batch), cus * 8);
        grid = (grid + (int)RT_WF_SHARDS - 1) / (int)RT_WF_SHARDS * (int)RT_WF_SHARDS;
        hipError_t e;
        switch ((feat & RT_FEAT_WIDE_CSG) ? 15u : (feat & 7u)) {
        case 15: e = wavefront_batch<15>(s, cam, r, batch, q, accum, stream, timer, grid, probe, shallow); break;
        case 0: e = wavefront_batch<0>(s, cam, r, batch, q, accum, stream, timer, grid, probe, shallow); break;
        case 1: e = wavefront_batch<1>(s, cam, r, batch, q, accum, stream, timer, grid, probe, shallow); break;
        case 2: e = wavefront_batch<2>(s, cam, r, batch, q, accum, stream, timer, grid, probe, shallow); break;
        case 3: e = wavefront_batch<3>(s, cam, r, batch, q, accum, stream, timer, grid, probe, shallow); break;
        case 5: e = wavefront_batch<5>(s, cam, r, batch, q, accum, stream, timer, grid, probe, shallow); break;
        default: e = wavefront_batch<7>(s, cam, r, batch, q, accum, stream, timer, grid, probe, shallow); break;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int FEAT, int STACK, int WPS, bool NODE16, bool DEEP = false, bool LIST = false, int BLOCK = RT_BLOCK, int TOP = 0, int ARITY = 2>
static hipError_t fused_launch(const RtDevScene &s, const RtDevCamera &cam, const RtDevRender &rc, float *accum,
                               uint32_t *counter, uint32_t knobs, uint32_t knobs2, hipStream_t stream, RtKernelTimer *timer, int cus,
                               bool prepare_only) {
    // tile rows of this shard (all of them when band_count is 1)
    const uint32_t all_rows = (rc.height + 7u) / 8u;
    const uint32_t tiles_y = rc.band_index < all_rows ? (all_rows - rc.band_index + rc.band_count - 1u) / rc.band_count : 0u;
    const uint32_t tiles_x = (rc.width + 7u) / 8u, n_tiles = tiles_x * tiles_y;
    if (n_tiles == 0 && !prepare_only) return hipSuccess;
    // workgroups per compute unit of this instantiation: asked once per device (at commit, so
    // that a render call makes no runtime query)
    // (host threads may commit or render at the same time: the table is atomic, and two threads that both find it
    // empty store the same answer.  The diagnostic build shares the entry of the plain one; it may hold fewer
    // workgroups per CU, which only means some of the grid waits its turn - workgroups do not depend on one another.)
    static std::atomic<int> occ_of_device[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int occ = (dev >= 0 && dev < 64) ? occ_of_device[dev].load(std::memory_order_relaxed) : 0;
    if (occ < 1) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fused<FEAT, false, STACK, WPS, NODE16, DEEP, LIST, BLOCK, TOP, ARITY>, BLOCK, 0) != hipSuccess || occ < 1) occ = 1;
        if (dev >= 0 && dev < 64) occ_of_device[dev].store(occ, std::memory_order_relaxed);
    }
    if (prepare_only) return hipSuccess;
    static const bool show_occupancy = getenv("RT_SHOW_OCCUPANCY") != nullptr;       // diagnostic: what the runtime answered
    if (show_occupancy) fprintf(stderr, "k_fused<feat %d, stack %d, wps %d, block %d, top %d, arity %d>: %d workgroups per CU\n", FEAT, STACK, WPS, BLOCK, TOP, ARITY, occ);
    if (DEEP) occ = std::min(occ, 2048 / BLOCK);        // the deep-stack columns are sized for 2048 threads per CU
    int grid = cus * occ;
    // Work items = tiles x sample chunks.  Large chunks leave the waves finishing at different
    // times at the end of the launch (with 3 chunks per tile that tail was 15 % of a 1024-spp
    // launch); small ones only cost a flush of the tile sums each, since a wave streams from one
    // item into the next: chunks of about 32 samples per pixel, more of them while that gives
    // fewer than 96 items per resident wave, never below 4 samples per pixel.
    const uint32_t target = 32u, min_chunk = 4u;
    constexpr uint32_t WAVES = BLOCK / 64;
    uint32_t want_items = (uint32_t)grid * WAVES * 96u;
    uint32_t n_chunks = std::max((rc.spp + target / 2u) / target, (want_items + n_tiles - 1u) / n_tiles);
    n_chunks = std::max(1u, std::min(n_chunks, std::max(1u, rc.spp / min_chunk)));
    uint32_t chunk_spp = (rc.spp + n_chunks - 1u) / n_chunks;
    n_chunks = (rc.spp + chunk_spp - 1u) / chunk_spp;
    uint32_t n_items = n_tiles * n_chunks;
    grid = (int)std::min<uint32_t>((uint32_t)grid, (n_items + WAVES - 1u) / WAVES);
    // item counters: as many as there are workgroups to serve them, at most RT_ITEM_COUNTERS, a power of two
    uint32_t counters_log2 = 0;
    while ((2u << counters_log2) <= RT_ITEM_COUNTERS && (16u << counters_log2) <= (uint32_t)grid) counters_log2++;
    const bool prof = (knobs & 0x10000u) != 0;
    // refill threshold left at 0 by the caller: 48 finished or free lanes before a wave leaves its traversal loop; the
    // kernel that reads the top of the tree from LDS does better with fuller waves (28: +12 % over 48 there, where the
    // plain deep-tree kernel loses 2 % with 32)
    if ((knobs & 255u) == 0u) knobs |= ARITY == 4 ? RT_WIDE_REFILL : TOP ? 28u : 48u;
    // node_min left at 255: a wave goes on with node steps while 8 lanes stand on inner nodes; the four-children step is
    // three times as long and is not run for so few
    if (((knobs >> 8) & 255u) == 255u) knobs = (knobs & ~0xFF00u) | ((ARITY == 4 ? RT_WIDE_NODE_MIN : 8u) << 8);
    k_fused_setup<<<1, 256, 0, stream>>>(counter, cam);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const RtDevCamera *camp = (const RtDevCamera *)((const char *)counter + RT_FUSED_CAMERA_OFFSET);
    if (timer) timer->begin(stream);
    if (prof)
        k_fused<FEAT, true, STACK, WPS, NODE16, DEEP, LIST, BLOCK, TOP, ARITY><<<grid, BLOCK, 0, stream>>>(s, camp, cam.background, rc, accum, counter, n_items, tiles_x, n_tiles, chunk_spp, knobs, (knobs2 & 127u) | (counters_log2 << 8));
    else
        k_fused<FEAT, false, STACK, WPS, NODE16, DEEP, LIST, BLOCK, TOP, ARITY><<<grid, BLOCK, 0, stream>>>(s, camp, cam.background, rc, accum, counter, n_items, tiles_x, n_tiles, chunk_spp, knobs, (knobs2 & 127u) | (counters_log2 << 8));
    if (timer) timer->end(stream);
    return hipGetLastError();
}

hipError_t rt_render_fused(const RtDevScene &s, uint32_t feat, uint32_t bvh_depth, const RtDevCamera &cam, RtDevRender rc,
                           uint32_t knobs, float *accum, void *workspace, size_t workspace_bytes, hipStream_t stream,
                           RtKernelTimer *timer, int cus, bool prepare_only, uint32_t knobs2) {
    if (!prepare_only) {
        if (rc.width == 0 || rc.height == 0 || rc.spp == 0) return hipSuccess;
        if (workspace_bytes < RT_FUSED_HEADER || rc.band_count == 0 || rc.band_index >= rc.band_count) return hipErrorInvalidValue;
    }
    if (timer) timer->name = "k_fused";
    uint32_t *counter = (uint32_t *)workspace;
    const bool n16 = (knobs & 0x20000u) != 0 && s.nodes16 != nullptr;
#define RT_FUSED(F, ST, W) (n16 ? fused_launch<F, ST, W, true>(s, cam, rc, accum, counter, knobs, knobs2, stream, timer, cus, prepare_only) \
                                : fused_launch<F, ST, W, false>(s, cam, rc, accum, counter, knobs, knobs2, stream, timer, cus, prepare_only))
    // a scene with a wide CSG shape (rare) has one kernel: every feature, the tree
    if (feat & RT_FEAT_WIDE_CSG) return RT_FUSED(15, RT_STACK_DEPTH, 4);
    if (s.object_list && s.n_listed && !(knobs2 & 0x100u)) {
        // a handful of objects: the list, not the tree (one stack entry is declared and never used).  Untextured
        // scenes run 5 waves per SIMD (96 registers; 4 waves: -8 %, 6 waves with 32 spilled registers: +-0);
        // with texture code next to it 4 waves win (5: -13 % on the Kit's second world)
#define RT_FUSED_LIST(F, W) fused_launch<F, 1, W, false, false, true>(s, cam, rc, accum, counter, knobs, knobs2, stream, timer, cus, prepare_only)
        switch (feat & 7u) {
        case 0: return RT_FUSED_LIST(0, RT_WPS0);
        case 1: return RT_FUSED_LIST(1, RT_WPS_LIST1);
        case 2: return RT_FUSED_LIST(2, 5);
        case 3: return RT_FUSED_LIST(3, 4);
        case 4: case 5: return RT_FUSED_LIST(5, 4);
        default: return RT_FUSED_LIST(7, 4);
        }
#undef RT_FUSED_LIST
    }
    switch (feat & 7u) {
    case 0:
        if (bvh_depth <= 16) return RT_FUSED(0, 16, RT_WPS0);
        // a deep tree (a million spheres) waits on memory, not on the ALUs: many waves per SIMD with
        // the rare deep stack entries in global memory beat 5 waves with the whole stack in LDS
        if (workspace_bytes >= fused_workspace_bytes_binary(cus) || prepare_only) {
            // With half-precision nodes: 512-thread workgroups that keep the head of the node array (the largest boxes:
            // the builder of a deep tree puts them there) in LDS, 6 waves per SIMD, walking the four-children form
            // (1.46 Gsamples/s on a million spheres at 4K).  For comparison, from the environment: RT_DEEP_BINARY=1 walks
            // the 32-byte two-children nodes in the same way (0.98), RT_DEEP_PLAIN=1 those without the LDS copy, 7 waves
            // per SIMD (0.93).
            static const bool plain = getenv("RT_DEEP_PLAIN") != nullptr;
            static const bool binary = getenv("RT_DEEP_BINARY") != nullptr;
            hipError_t e;
            if (n16 && !plain && !binary && s.nodes4 && (workspace_bytes >= rt_fused_workspace_bytes(cus) || prepare_only)) {
                e = fused_launch<0, 16, 6, true, true, false, 512, RT_TOP_NODES, 4>(s, cam, rc, accum, counter, knobs, knobs2, stream, timer, cus, prepare_only);
            } else if (n16 && !plain) e = fused_launch<0, 16, 6, true, true, false, 512, RT_TOP_NODES>(s, cam, rc, accum, counter, knobs, knobs2, stream, timer, cus, prepare_only);
            else if (n16) e = fused_launch<0, 16, RT_WPS_DEEP, true, true>(s, cam, rc, accum, counter, knobs, knobs2, stream, timer, cus, prepare_only);
            else e = fused_launch<0, 16, RT_WPS_DEEP, false, true>(s, cam, rc, accum, counter, knobs, knobs2, stream, timer, cus, prepare_only);
            if (!prepare_only) return e;
        }
        return RT_FUSED(0, RT_STACK_DEPTH, 5);
    case 1: return RT_FUSED(1, RT_STACK_DEPTH, RT_WPS_XPRIM);
    case 2: return RT_FUSED(2, RT_STACK_DEPTH, 5);
    case 3: return RT_FUSED(3, RT_STACK_DEPTH, RT_WPS_XPRIM);
    case 4:
        // spheres and moving spheres only: the spheres kernel with the frame change in front of its sphere test
        if (bvh_depth <= 16) return RT_FUSED(4, 16, RT_WPS_MOVING_SPHERES);
        return RT_FUSED(5, RT_STACK_DEPTH, 4);
    case 5: return RT_FUSED(5, RT_STACK_DEPTH, 4);
    default: return RT_FUSED(7, RT_STACK_DEPTH, 4);
    }
#undef RT_FUSED
}

// ------------------------------------------------------------------ render without a workspace argument
// The fused kernel's scratch (item counters, diagnostics, camera, deep-stack columns) for callers
// that cannot allocate device memory: one array per GPU, part of the loaded code object, held
// until the process ends.  `scratch` is a host-side stand-in for the symbol, never a device
// address: the address of each device's copy comes from hipGetSymbolAddress with that device
// current (rt_static_scratch_init), and only that pointer is handed to kernels and memsets.
#define RT_STATIC_SCRATCH_BYTES (RT_FUSED_DEEP_OFFSET + (size_t)RT_STATIC_MAX_CUS * 8u * RT_BLOCK * (RT_WIDE_STACK_DEPTH - 16) * 4u)
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];
static std::atomic<void *> scratch_of_device[RT_STATIC_MAX_DEVICES];     // contexts may be created from several host threads

hipError_t rt_static_scratch_init(int device) {
    if (device < 0 || device >= RT_STATIC_MAX_DEVICES) return hipErrorInvalidDevice;
    if (scratch_of_device[device].load(std::memory_order_acquire)) return hipSuccess;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    scratch_of_device[device].store(ptr, std::memory_order_release);      // every thread resolves the same address
    return hipSuccess;
}

hipError_t rt_render_fused_static(const RtDevScene &s, uint32_t feat, uint32_t bvh_depth, const RtDevCamera &cam, RtDevRender rc,
                                  uint32_t knobs, float *accum, hipStream_t stream, int cus, int device, uint32_t knobs2) {
    void *ptr = (device >= 0 && device < RT_STATIC_MAX_DEVICES) ? scratch_of_device[device].load(std::memory_order_acquire) : nullptr;
    if (!ptr) return hipErrorNotInitialized;
    // a device with more compute units than the columns were sized for runs the grid of RT_STATIC_MAX_CUS units
    return rt_render_fused(s, feat, bvh_depth, cam, rc, knobs, accum, ptr, RT_STATIC_SCRATCH_BYTES, stream,
                           nullptr, std::min(cus, RT_STATIC_MAX_CUS), false, knobs2);
}
