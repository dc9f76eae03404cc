// rt_launch.h - host-callable launchers of the HIP kernels (implemented in rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/raytracer_amd.h"
#include "rt_shared.h"

// Scene features that select a kernel specialisation.
enum : uint32_t {
    RT_FEAT_XPRIM = 1u,      // anything other than plain spheres and moving plain spheres in the BVH
    RT_FEAT_TEXTURE = 2u,    // image or Perlin materials
    RT_FEAT_MOTION = 4u,     // moving objects: every sample draws a time
    RT_FEAT_WIDE_CSG = 8u,   // a CSG shape of more than RT_CSG_LEAVES leaves: one kernel with every feature (15)
};

// HIP events around every launch of the render loop's dominant kernel, on the stream the
// kernel runs on, when timing was asked for (RT_OPTION_KERNEL_TIMING).  Pairs are kept until
// collect() sums them (after synchronising); the pool is bounded: once RT_TIMER_MAX_EVENTS are
// in flight the recorded pairs are collected before the next one is taken, so a caller that
// never reads the time cannot grow it.
#define RT_TIMER_MAX_EVENTS 512
struct RtKernelTimer {
    std::vector<hipEvent_t> pool;      // start0, stop0, start1, stop1, ...
    size_t used = 0;
    double total_ms = 0.0;
    uint64_t launches = 0;
    const char *name = "";
    bool enabled = false;
    bool open = false;                 // begin() recorded a start that end() has to close

    void begin(hipStream_t st) {
        open = false;
        if (!enabled) return;
        if (used + 2 > RT_TIMER_MAX_EVENTS) collect();
        if (used + 2 > pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess) { enabled = false; return; }
            if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); enabled = false; return; }
            pool.push_back(a);
            pool.push_back(b);
        }
        open = hipEventRecord(pool[used], st) == hipSuccess;
    }
    void end(hipStream_t st) {
        if (!open) return;
        open = false;
        if (hipEventRecord(pool[used + 1], st) == hipSuccess) used += 2;
    }
    void collect() {
        for (size_t i = 0; i + 1 < used; i += 2) {
            float ms = 0.f;
            if (hipEventSynchronize(pool[i + 1]) == hipSuccess && hipEventElapsedTime(&ms, pool[i], pool[i + 1]) == hipSuccess) {
                total_ms += ms;
                launches++;
            }
        }
        used = 0;
    }
    void reset() { collect(); total_ms = 0.0; launches = 0; }
    ~RtKernelTimer() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

// How the wavefront loop learns that its queues have run empty without ever draining the stream: after the shade
// launch of a bounce the eight live counters of the next queue are copied to pinned host memory and an event is
// recorded; RT_WF_LAG bounces later the host waits for that event only - the bounces launched since are still queued, so
// the GPU stays busy - and stops launching once the copy says that no path is left.  Created on first use, owned by
// the scene (one render call at a time per scene, as everywhere in this interface).
#define RT_WF_LAG 2
#define RT_WF_PROBES 4
struct RtWfProbe {
    uint32_t *host = nullptr;          // RT_WF_PROBES x 256 words of pinned memory: one queue's counters per slot
    hipEvent_t ev[RT_WF_PROBES] = {};
    bool ready = false, failed = false;

    bool init() {
        if (ready || failed) return ready;
        if (hipHostMalloc((void **)&host, RT_WF_PROBES * 1024, hipHostMallocDefault) != hipSuccess) { host = nullptr; failed = true; return false; }
        for (int i = 0; i < RT_WF_PROBES; i++)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { failed = true; return false; }
        ready = true;
        return true;
    }
    ~RtWfProbe() {
        for (int i = 0; i < RT_WF_PROBES; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
        if (host) (void)hipHostFree(host);
    }
};

// Layout of the wavefront pipeline's workspace for `capacity` paths.
struct RtWavefrontBuffers {
    float4 *ray_o[2], *ray_d[2], *thr[2];
    uint4 *rng[2];
    float4 *hit;
    float4 *sample_rad;
    uint32_t *counts;         // live paths per [queue][shard]
    uint32_t capacity, shard_cap;
};

size_t rt_wavefront_bytes(uint32_t capacity);
void rt_wavefront_carve(void *workspace, uint32_t capacity, RtWavefrontBuffers &out);

hipError_t rt_launch_first_hit(const RtDevScene &s, uint32_t feat, const float *rays, int n, float tmin, float tmax,
                               uint64_t volume_key, float time, RtHit *out, hipStream_t stream);
hipError_t rt_launch_intervals(const RtDevScene &s, uint32_t ref, const float *rays, int n, const float *scripted,
                               int n_scripted, uint64_t volume_key, uint32_t object, float time, RtEvent *events,
                               int32_t *counts, hipStream_t stream);
hipError_t rt_launch_camera_rays(const RtDevCamera &cam, const RtDevRender &rc, int x0, int y0, int w, int h,
                                 float *rays_out, hipStream_t stream);
hipError_t rt_launch_resolve(const float *accum, float inv_samples, int width, int height, uint8_t *rgbx, hipStream_t stream);
hipError_t rt_launch_scatter(const RtDevScene &s, const RtScatterQuery *queries, int n, uint64_t seed, RtScatterResult *out,
                             hipStream_t stream);
hipError_t rt_launch_rng_words(const uint32_t *streams, int n, int words_per_stream, uint32_t *out, uint32_t *wyrand_out,
                               hipStream_t stream);

hipError_t rt_launch_progressive_present(const float *pass_sum, uint32_t spp, uint32_t pass_counter, int width, int height,
                                         uint32_t *acc, uint8_t *bgra, hipStream_t stream);

// spp samples per pixel starting at rc.first_sample, added to accum.  `cus` is the compute-unit
// count of the device the scene lives on (grids and the deep-stack columns are sized by it).
hipError_t rt_render_wavefront(const RtDevScene &s, uint32_t feat, const RtDevCamera &cam, RtDevRender rc,
                               float *accum, void *workspace, size_t workspace_bytes, hipStream_t stream,
                               RtKernelTimer *timer, int cus, RtWfProbe *probe = nullptr, uint32_t bvh_depth = RT_STACK_DEPTH);
// knobs2: bits 0-6 = heavy_min (lanes that must wait at a CSG shape / volume / moving object before
// the wave runs that code; 0 or 1 = never held back).
// prepare_only: no launch; makes the runtime queries a later call of the same scene would make
// (workgroups per compute unit of the kernel variants the scene selects), so that render calls
// themselves query nothing.
hipError_t rt_render_fused(const RtDevScene &s, uint32_t feat, uint32_t bvh_depth, const RtDevCamera &cam, RtDevRender rc,
                           uint32_t knobs, float *accum, void *workspace, size_t workspace_bytes, hipStream_t stream,
                           RtKernelTimer *timer, int cus, bool prepare_only = false, uint32_t knobs2 = 0);

// The fused kernel on the library's own per-device scratch array (no workspace argument, no
// allocation, no timing events).  rt_static_scratch_init(device) resolves the array's address on
// `device`, which must be current; once per device, before the first render on it.
#define RT_STATIC_MAX_CUS 256          // the deep-stack columns of the array cover 8 workgroups on each of 256 units
#define RT_STATIC_MAX_DEVICES 64
hipError_t rt_static_scratch_init(int device);
hipError_t rt_render_fused_static(const RtDevScene &s, uint32_t feat, uint32_t bvh_depth, const RtDevCamera &cam, RtDevRender rc,
                                  uint32_t knobs, float *accum, hipStream_t stream, int cus, int device, uint32_t knobs2);

#define RT_WAVEFRONT_BATCH_SPP 8u      // samples per pixel resident in the path queues at once
#define RT_FUSED_HEADER 4096u          // the least workspace the fused kernel runs with
size_t rt_fused_workspace_bytes(int cus);      // what it wants: header + deep-stack columns
int rt_current_device_cus();                    // compute units of the current device (256 if unknown)
