// oracle_capi.cpp - extern "C" surface of the CPU oracle (test infrastructure only;
// see oracle_api.h for who may load it).
#include "oracle.hpp"

#include <atomic>
#include <thread>

using namespace orc;

struct OrcScene { Scene s; double query_time = 0; };

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return -1; }

// All orc_* functions get C linkage from their declarations in oracle_api.h.

const char *orc_last_error(void) { return g_err.c_str(); }

OrcScene *orc_scene_create(void) { return new OrcScene(); }
void orc_scene_destroy(OrcScene *s) { delete s; }

int orc_add_material(OrcScene *s, const OrcMaterial *m) {
    if (m->kind < 1 || m->kind > orc_material_kind_emissive_kit_texture) return fail("material kind out of range");
    s->s.materials.push_back(*m);
    return (int)s->s.materials.size() - 1;
}

int orc_add_texture_rgb8(OrcScene *s, int w, int h, const uint8_t *rgb) {
    if (w <= 0 || h <= 0) return fail("empty texture");
    Image im;
    im.w = w; im.h = h;
    im.rgb.assign(rgb, rgb + (size_t)3 * w * h);
    s->s.images.push_back(std::move(im));
    return (int)s->s.images.size() - 1;
}

int orc_add_kit_perlin(OrcScene *s, const OrcKitPerlin *p) {
    s->s.kit_perlins.push_back(*p);
    return (int)s->s.kit_perlins.size() - 1;
}
int orc_add_kit_texture(OrcScene *s, const OrcKitTexture *t) {
    s->s.kit_textures.push_back(*t);
    return (int)s->s.kit_textures.size() - 1;
}
int orc_kit_texture_color(const OrcScene *s, int texture, int precision, double u, double v, const double point[3], double out_rgb[3]) {
    if (texture < 0 || texture >= (int)s->s.kit_textures.size()) return fail("bad texture id");
    if (precision == ORC_F32) {
        V3<float> c = kit_texture_color<float>(s->s, texture, (float)u, (float)v, V3<float>((float)point[0], (float)point[1], (float)point[2]));
        out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
    } else {
        V3<double> c = kit_texture_color<double>(s->s, texture, u, v, V3<double>(point[0], point[1], point[2]));
        out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
    }
    return 0;
}
int orc_add_perlin(OrcScene *s, const OrcPerlin *p) {
    s->s.perlins.push_back(*p);
    return (int)s->s.perlins.size() - 1;
}

static void set_xf(Shape &sh, const OrcTransform &t) {
    for (int i = 0; i < 9; i++) sh.rot[i] = t.rotation[i];
    for (int i = 0; i < 3; i++) sh.trans[i] = t.translation[i];
}
static int push(OrcScene *s, Shape &&sh) {
    s->s.committed = false;
    sh.kit_uv = s->s.kit_convention;
    s->s.shapes.push_back(std::move(sh));
    return (int)s->s.shapes.size() - 1;
}

int orc_shape_sphere(OrcScene *s, const OrcSphere *p) {
    Shape sh; sh.kind = SPHERE; set_xf(sh, p->transform); sh.radius = p->radius; sh.mat[0] = p->material_offset;
    return push(s, std::move(sh));
}
// n plain spheres (identity rotation) as objects in one call: center_radius n x 4 floats.
int orc_add_sphere_objects(OrcScene *s, int n, const float *center_radius, const uint32_t *material_offsets) {
    int first = (int)s->s.objects.size();
    s->s.shapes.reserve(s->s.shapes.size() + (size_t)n);
    for (int i = 0; i < n; i++) {
        Shape sh; sh.kind = SPHERE;
        for (int a = 0; a < 3; a++) sh.trans[a] = center_radius[4 * (size_t)i + a];
        sh.radius = center_radius[4 * (size_t)i + 3];
        sh.mat[0] = material_offsets[i];
        s->s.objects.push_back(push(s, std::move(sh)));
    }
    return first;
}
int orc_shape_cylinder(OrcScene *s, const OrcCylinder *p) {
    Shape sh; sh.kind = CYLINDER; set_xf(sh, p->transform); sh.radius = p->radius; sh.height = p->height;
    sh.mat[0] = p->bottom_material_offset; sh.mat[1] = p->top_material_offset; sh.mat[2] = p->side_material_offset;
    return push(s, std::move(sh));
}
int orc_shape_cuboid(OrcScene *s, const OrcCuboid *p) {
    Shape sh; sh.kind = CUBOID; set_xf(sh, p->transform);
    for (int i = 0; i < 3; i++) sh.size[i] = p->size[i];
    for (int i = 0; i < 6; i++) sh.mat[i] = p->material_offset[i];
    return push(s, std::move(sh));
}
int orc_shape_quad(OrcScene *s, const OrcQuad *p) {
    Shape sh; sh.kind = QUAD;
    for (int i = 0; i < 3; i++) {
        sh.q_origin[i] = p->origin[i]; sh.q_u[i] = p->u[i]; sh.q_v[i] = p->v[i];
        sh.q_w[i] = p->w[i]; sh.q_normal[i] = p->normal[i];
    }
    sh.q_d = p->d; sh.mat[0] = p->material_offset;
    return push(s, std::move(sh));
}
int orc_shape_csg(OrcScene *s, int op, int n, const int *children) {
    if (n < 1) return fail("composition needs children");
    Shape sh; sh.kind = COMPOSITION;
    for (int i = 0; i < n; i++) {
        if (children[i] < 0 || children[i] >= (int)s->s.shapes.size()) return fail("bad child shape id");
        sh.children.push_back(children[i]);
    }
    // RenderableImpl.h:515-522
    sh.subtract = op == ORC_OP_SUBTRACT;
    sh.min_count = op == ORC_OP_INTERSECTION ? n : 1;
    return push(s, std::move(sh));
}
int orc_shape_cdv(OrcScene *s, int base, float density) {
    if (base < 0 || base >= (int)s->s.shapes.size()) return fail("bad base shape id");
    Shape sh; sh.kind = CDV; sh.base = base; sh.density = density;
    return push(s, std::move(sh));
}
int orc_shape_motion(OrcScene *s, int base, const double quat_wxyz[4], const double translation[3]) {
    if (base < 0 || base >= (int)s->s.shapes.size()) return fail("bad base shape id");
    if (s->s.shapes[base].kind == CDV || s->s.shapes[base].kind == MOTION) return fail("motion blur takes a primitive or a composition");
    Shape sh; sh.kind = MOTION; sh.base = base;
    for (int i = 0; i < 4; i++) sh.m_quat[i] = quat_wxyz[i];
    for (int i = 0; i < 3; i++) sh.m_trans[i] = translation[i];
    s->s.animated = true;
    return push(s, std::move(sh));
}
void orc_set_query_time(OrcScene *s, double time) { s->query_time = time; }
void orc_set_convention(OrcScene *s, int kit) {
    s->s.kit_convention = kit != 0;
    for (Shape &sh : s->s.shapes) sh.kit_uv = kit != 0;
}
int orc_shape_bounds(OrcScene *s, int shape, double out_min_max[6]) {
    if (shape < 0 || shape >= (int)s->s.shapes.size()) return fail("bad shape id");
    Box b = shape_box(s->s, shape);
    for (int a = 0; a < 3; a++) { out_min_max[a] = b.mn[a]; out_min_max[3 + a] = b.mx[a]; }
    return 0;
}
void orc_animation_box_of_point(const double quat_wxyz[4], const double translation[3], const double p[3], double out_min_max[6]) {
    Box b = animation_box_of_point(Quat{quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]}, translation, p);
    for (int a = 0; a < 3; a++) { out_min_max[a] = b.mn[a]; out_min_max[3 + a] = b.mx[a]; }
}
void orc_transform_pow(const double quat_wxyz[4], const double translation[3], double t, const double p[3], double out[3]) {
    Quat q = quat_pow(Quat{quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]}, t);
    quat_rotate(q, p, out);
    for (int a = 0; a < 3; a++) out[a] += t * translation[a];
}
int orc_add_object(OrcScene *s, int shape) {
    if (shape < 0 || shape >= (int)s->s.shapes.size()) return fail("bad shape id");
    s->s.committed = false;
    s->s.objects.push_back(shape);
    return (int)s->s.objects.size() - 1;
}
int orc_commit(OrcScene *s) { commit(s->s); return 0; }

template <class R>
static int enumerate_t(const Scene &sc, double time, int shape, const double o[3], const double d[3],
                       const float *script, int n_script, OrcEvent *out, int max_out) {
    Rng scripted;
    scripted.script = script; scripted.n_script = n_script;
    TraceCtx<R> ctx{&sc, nullptr, script ? &scripted : nullptr, 0, time};
    Ray<R> ray{V3<R>(R(o[0]), R(o[1]), R(o[2])), V3<R>(R(d[0]), R(d[1]), R(d[2]))};
    auto e = make_enum<R>(ctx, shape, ray, 0);
    int n = 0;
    for (; e->hasNext() && n < max_out; e->move(), n++) {
        OrcEvent &ev = out[n];
        ev.t = (double)e->t();
        ev.is_exit = e->isExit();
        ev.material_offset = 0;
        ev.pad = 0;
        for (int i = 0; i < 3; i++) ev.point[i] = ev.normal[i] = 0;
        ev.uv[0] = ev.uv[1] = 0;
        if (std::isfinite(ev.t)) {
            V3<R> p = e->point(), nn = e->normal();
            ev.point[0] = p.x; ev.point[1] = p.y; ev.point[2] = p.z;
            ev.normal[0] = nn.x; ev.normal[1] = nn.y; ev.normal[2] = nn.z;
            R u, v;
            e->texture_coordinates(u, v);
            ev.uv[0] = u; ev.uv[1] = v;
            ev.material_offset = e->material_offset();
        }
    }
    return n;
}

int orc_enumerate(const OrcScene *s, int shape, int precision, const double origin[3], const double direction[3],
                  const float *scripted_rng, int n_scripted, OrcEvent *out, int max_out) {
    if (shape < 0 || shape >= (int)s->s.shapes.size()) return fail("bad shape id");
    if (precision == ORC_F32) return enumerate_t<float>(s->s, s->query_time, shape, origin, direction, scripted_rng, n_scripted, out, max_out);
    return enumerate_t<double>(s->s, s->query_time, shape, origin, direction, scripted_rng, n_scripted, out, max_out);
}

template <class R>
static void first_hit_t(const Scene &sc, double time, int n, const double *rays, double tmin, double tmax, uint64_t cdv_state, OrcHit *out) {
    for (int i = 0; i < n; i++) {
        const double *r = rays + 6 * (size_t)i;
        Ray<R> ray{V3<R>(R(r[0]), R(r[1]), R(r[2])), V3<R>(R(r[3]), R(r[4]), R(r[5]))};
        TraceCtx<R> ctx{&sc, nullptr, nullptr, cdv_state, time};
        HitInfo<R> h;
        OrcHit &o = out[i];
        std::memset(&o, 0, sizeof(o));
        o.object = -1;
        if (scene_hit<R>(ctx, ray, R(tmin), R(tmax), h)) {
            o.t = h.t;
            o.point[0] = h.point.x; o.point[1] = h.point.y; o.point[2] = h.point.z;
            o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
            o.uv[0] = h.u; o.uv[1] = h.v;
            o.material_offset = h.material_offset;
            o.object = h.object;
            o.front_face = h.front ? 1 : 0;
        }
    }
}

int orc_first_hit(const OrcScene *s, int precision, int n, const double *rays, double tmin, double tmax,
                  uint64_t cdv_state, OrcHit *out) {
    if (!s->s.committed) return fail("scene not committed");
    if (precision == ORC_F32) first_hit_t<float>(s->s, s->query_time, n, rays, tmin, tmax, cdv_state, out);
    else first_hit_t<double>(s->s, s->query_time, n, rays, tmin, tmax, cdv_state, out);
    return 0;
}

template <class R>
static void scatter_t(const Scene &sc, int n, const OrcScatterQuery *queries, uint64_t seed, OrcScatterResult *out) {
    for (int i = 0; i < n; i++) {
        const OrcScatterQuery &q = queries[i];
        HitInfo<R> hit;
        hit.point = V3<R>(R(q.point[0]), R(q.point[1]), R(q.point[2]));
        hit.normal = V3<R>(R(q.normal[0]), R(q.normal[1]), R(q.normal[2]));
        hit.front = q.front_face != 0;
        hit.material_offset = q.material_offset;
        hit.u = R(q.uv[0]); hit.v = R(q.uv[1]);
        Ray<R> ray{hit.point, V3<R>(R(q.direction[0]), R(q.direction[1]), R(q.direction[2]))};
        Rng rng;
        rng.pcg = sample_stream(seed, q.sample, q.x, q.y);
        ScatterResult<R> res;
        res.scattered = ray;
        bool did = scatter<R>(sc, sc.materials[q.material_offset], ray, hit, rng, res);
        OrcScatterResult &o = out[i];
        std::memset(&o, 0, sizeof(o));
        o.scattered = did ? 1 : 0;
        if (did) { o.direction[0] = res.scattered.direction.x; o.direction[1] = res.scattered.direction.y; o.direction[2] = res.scattered.direction.z; }
        o.attenuation[0] = res.attenuation.x; o.attenuation[1] = res.attenuation.y; o.attenuation[2] = res.attenuation.z;
        o.emitted[0] = res.emitted.x; o.emitted[1] = res.emitted.y; o.emitted[2] = res.emitted.z;
    }
}

int orc_scatter(const OrcScene *s, int precision, int n, const OrcScatterQuery *queries, uint64_t rng_seed, OrcScatterResult *out) {
    for (int i = 0; i < n; i++)
        if (queries[i].material_offset >= s->s.materials.size()) return fail("bad material offset");
    if (precision == ORC_F32) scatter_t<float>(s->s, n, queries, rng_seed, out);
    else scatter_t<double>(s->s, n, queries, rng_seed, out);
    return 0;
}

template <class R>
static void camera_rays_t(const OrcCameraConfig *cc, int width, int height, const OrcRenderConfig *rc,
                          int x0, int y0, int w, int h, double *out) {
    Camera<R> cam(width, height, *cc);
    size_t k = 0;
    for (int y = y0; y < y0 + h; y++)
        for (int x = x0; x < x0 + w; x++)
            for (uint32_t s = 0; s < rc->samples_per_pixel; s++) {
                Rng rng;
                rng.pcg = sample_stream(rc->rng_seed, rc->first_sample + s, (uint32_t)x, (uint32_t)y);
                Ray<R> r = cam.get_ray((uint32_t)x, (uint32_t)y, rng);
                out[k++] = r.origin.x; out[k++] = r.origin.y; out[k++] = r.origin.z;
                out[k++] = r.direction.x; out[k++] = r.direction.y; out[k++] = r.direction.z;
            }
}

int orc_camera_rays(const OrcCameraConfig *cc, int precision, int width, int height, const OrcRenderConfig *rc,
                    int x0, int y0, int w, int h, double *out) {
    if (precision == ORC_F32) camera_rays_t<float>(cc, width, height, rc, x0, y0, w, h, out);
    else camera_rays_t<double>(cc, width, height, rc, x0, y0, w, h, out);
    return 0;
}

// Row-parallel like Camera.swift:62-80 (one task per row); workers pull rows from a
// shared counter.  Unlike Camera.swift:82-83 the generator is per (pixel, sample), so the
// image does not depend on the number of workers.
template <class R>
static void render_t(const Scene &sc, const OrcCameraConfig *cc, const OrcRenderConfig *rc, int width, int height,
                     int x0, int y0, int w, int h, int threads, double *out_sum, OrcCounters *counters) {
    Camera<R> cam(width, height, *cc);
    std::atomic<int> next_row(0);
    std::vector<Counters> per_thread((size_t)threads);
    auto worker = [&](int tid) {
        Counters *cnt = counters ? &per_thread[(size_t)tid] : nullptr;
        while (true) {
            int row = next_row.fetch_add(1);
            if (row >= h) break;
            int y = y0 + row;
            for (int col = 0; col < w; col++) {
                int x = x0 + col;
                V3<R> color;    // accumulates in R like Camera.swift:86-91
                for (uint32_t s = 0; s < rc->samples_per_pixel; s++) {
                    Rng rng;
                    rng.pcg = sample_stream(rc->rng_seed, rc->first_sample + s, (uint32_t)x, (uint32_t)y);
                    Ray<R> ray = cam.get_ray((uint32_t)x, (uint32_t)y, rng);
                    // Camera.swift:89: one time per sample, after the ray; only drawn when something moves
                    double time = sc.animated ? rng.random_f<double>() : 0.0;
                    color = color + ray_color<R>(sc, cc->background, ray, (int)rc->max_depth, rng, cnt, time);
                    if (cnt) cnt->samples++;
                }
                double *o = out_sum + 3 * ((size_t)row * w + col);
                o[0] = color.x; o[1] = color.y; o[2] = color.z;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto &t : pool) t.join();
    if (counters) {
        Counters total;
        for (auto &c : per_thread) total.add(c);
        counters->samples = total.samples; counters->rays = total.rays; counters->hits = total.hits;
        counters->box_tests = total.box_tests;
        for (int i = 0; i < 4; i++) counters->prim_tests[i] = total.prim_tests[i];
        counters->csg_events = total.csg_events; counters->cdv_intervals = total.cdv_intervals;
        for (int i = 0; i < 10; i++) counters->shades[i] = total.shades[i];
    }
}

int orc_render(const OrcScene *s, const OrcCameraConfig *cc, const OrcRenderConfig *rc, int precision,
               int width, int height, int x0, int y0, int w, int h, int threads, double *out_sum, OrcCounters *counters) {
    if (!s->s.committed) return fail("scene not committed");
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    if (precision == ORC_F32) render_t<float>(s->s, cc, rc, width, height, x0, y0, w, h, threads, out_sum, counters);
    else render_t<double>(s->s, cc, rc, width, height, x0, y0, w, h, threads, out_sum, counters);
    return threads;
}

void orc_pcg32(uint64_t initstate, uint64_t initseq, int n, uint32_t *out) {
    Pcg32 g(initstate, initseq);
    for (int i = 0; i < n; i++) out[i] = g.next_u32();
}
void orc_wyrand(uint64_t seed, int n, uint64_t *out) {
    for (int i = 0; i < n; i++) out[i] = wyrand_at(seed, (uint64_t)i);
}
void orc_sample_stream_words(uint64_t rng_seed, uint32_t sample, uint32_t x, uint32_t y, int n, uint32_t *out) {
    Pcg32 g = sample_stream(rng_seed, sample, x, y);
    for (int i = 0; i < n; i++) out[i] = g.next_u32();
}
void orc_sample_stream(uint64_t rng_seed, uint32_t sample, uint32_t x, uint32_t y, uint64_t out_state_inc[2]) {
    Pcg32 g = sample_stream(rng_seed, sample, x, y);
    out_state_inc[0] = g.state;
    out_state_inc[1] = g.inc;
}


// Camera.swift:92 + Vector3D.swift:185-191,208-211 in float, one operation at a time: sum / n,
// square root, min with 1, floor(nextDown(256) * x); written as RGBX8.
void orc_resolve_rgb8(const float *sums, uint32_t n_samples, int n_pixels, uint8_t *rgbx) {
    const float scale = std::nextafter(256.0f, 0.0f);
    for (int i = 0; i < n_pixels; i++) {
        for (int c = 0; c < 3; c++) {
            float mean = sums[3 * (size_t)i + c] / (float)n_samples;
            float v = mean > 0.0f ? std::fmin(std::sqrt(mean), 1.0f) : 0.0f;
            float q = std::floor(scale * v);
            rgbx[4 * (size_t)i + c] = (uint8_t)(q > 255.0f ? 255.0f : q);
        }
        rgbx[4 * (size_t)i + 3] = 0;
    }
}

// Shaders.metal:151-158, one float operation at a time (this file is built with
// -ffp-contract=off, so nothing fuses).
void orc_progressive_present(const float *pass_sum, uint32_t samples_per_pixel, uint32_t pass_counter,
                             uint32_t *acc, uint8_t *bgra, int n_pixels) {
    const float half_below = std::nextafter(0.5f, 0.0f);
    const float t = 1.0f / (float)pass_counter;
    const float keep = 1.0f - t;
    for (int i = 0; i < n_pixels; i++) {
        uint32_t word = acc[i], out_word = 0;
        uint8_t px[3];
        for (int c = 0; c < 3; c++) {
            float color = pass_sum[3 * (size_t)i + c] / (float)samples_per_pixel;        // :151
            color = std::fmin(color, 1.0f);                                              // :152
            float old_color = ((float)((word >> (10 * c)) & 1023u) + half_below) / 1024.0f;   // :153
            float total = old_color * keep + color * t;                                  // :155
            float scaled = total * 1024.0f;
            uint32_t q = scaled > 0.0f ? (uint32_t)scaled : 0u;                          // :156 (ushort conversion truncates)
            out_word |= (q > 1023u ? 1023u : q) << (10 * c);
            float shown = std::sqrt(total);                                              // :158
            shown = shown < 0.0f || shown != shown ? 0.0f : (shown > 1.0f ? 1.0f : shown);
            px[c] = (uint8_t)std::nearbyint(shown * 255.0f);
        }
        acc[i] = out_word;
        bgra[4 * (size_t)i + 0] = px[2];
        bgra[4 * (size_t)i + 1] = px[1];
        bgra[4 * (size_t)i + 2] = px[0];
        bgra[4 * (size_t)i + 3] = 255;
    }
}
