"""Renderer: the host object that owns a flattened scene and drives the render loop.

Mirrors `Renderer` of the reference app (MetalRayTracer/ContentView.swift:103-235): built
once from a Scene (flatten + acceleration structure, :122-170), then `draw` adds one pass
to a per-pixel accumulator (:182-224).  Device memory and streams come from torch; every
computation happens in the native library.
"""
import ctypes as C

import numpy as np
import torch

from . import abi
from ._native import check, lib
from .scene import RenderConfig

HIT_DTYPE = np.dtype([("t", "f4"), ("point", "f4", 3), ("normal", "f4", 3), ("uv", "f4", 2),
                      ("object", "i4"), ("material_offset", "u4"), ("front_face", "i4")])
SCATTER_QUERY_DTYPE = np.dtype([("direction", "f4", 3), ("normal", "f4", 3), ("point", "f4", 3), ("uv", "f4", 2),
                                ("front_face", "i4"), ("material_offset", "u4"), ("rng_sample", "u4"), ("rng_x", "u4"), ("rng_y", "u4")])
SCATTER_RESULT_DTYPE = np.dtype([("scattered", "i4"), ("direction", "f4", 3), ("attenuation", "f4", 3), ("emitted", "f4", 3),
                                 ("pad", "u4", 2)])
assert SCATTER_QUERY_DTYPE.itemsize == 64 and SCATTER_RESULT_DTYPE.itemsize == 48
EVENT_DTYPE = np.dtype([("t", "f4"), ("point", "f4", 3), ("normal", "f4", 3), ("uv", "f4", 2),
                        ("material_offset", "u4"), ("is_exit", "i4"), ("pad", "i4")])
assert HIT_DTYPE.itemsize == 48 and EVENT_DTYPE.itemsize == 48


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _no_stream:
    """`with` block that leaves the current stream alone."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class Context:
    def __init__(self, device_ordinal=0):
        self.device_ordinal = device_ordinal
        self.device = torch.device("cuda", device_ordinal)
        torch.cuda.set_device(self.device)
        self.h = C.c_void_p()
        check(lib.rt_context_create(device_ordinal, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            lib.rt_context_destroy(self.h)
            self.h = None


class NativeScene:
    """Encoder sink that forwards the C records to the library (see scene.SceneEncoder)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = C.c_void_p()
        check(lib.rt_scene_create(ctx.h, C.byref(self.h)))
        self.committed = False

    def __del__(self):
        if getattr(self, "h", None):
            lib.rt_scene_destroy(self.h)
            self.h = None

    # ---- sink protocol
    def add_material(self, rec):
        return check(lib.rt_add_material(self.h, C.byref(rec), C.sizeof(rec)))

    def add_dummy_materials(self, n):
        for _ in range(n):
            self.add_material(abi.RtColoredLambertianMaterial(abi.material_kind_lambertian_colored, abi.F3(.5, .5, .5)))

    def add_texture_rgb8(self, w, h, pixels):
        buf = np.ascontiguousarray(pixels, dtype=np.uint8)
        return check(lib.rt_add_texture_rgb8(self.h, w, h, buf.ctypes.data_as(C.c_void_p)))

    def add_kit_perlin(self, rec):
        return check(lib.rt_add_kit_perlin(self.h, C.byref(rec)))

    def add_kit_texture(self, rec):
        return check(lib.rt_add_kit_texture(self.h, C.byref(rec)))

    def shape_sphere(self, rec):
        return check(lib.rt_shape_sphere(self.h, C.byref(rec)))

    def shape_cylinder(self, rec):
        return check(lib.rt_shape_cylinder(self.h, C.byref(rec)))

    def shape_cuboid(self, rec):
        return check(lib.rt_shape_cuboid(self.h, C.byref(rec)))

    def shape_quad(self, rec):
        return check(lib.rt_shape_quad(self.h, C.byref(rec)))

    def shape_csg(self, op, ids):
        arr = (C.c_int * len(ids))(*ids)
        return check(lib.rt_shape_csg(self.h, op, len(ids), arr))

    def shape_cdv(self, base, density):
        return check(lib.rt_shape_cdv(self.h, base, density))

    def shape_motion(self, base, rec):
        return check(lib.rt_shape_motion(self.h, base, C.byref(rec)))

    def bounds(self, shape):
        out = (C.c_float * 6)()
        check(lib.rt_shape_bounds(self.h, shape, out))
        return np.array(out[:3]), np.array(out[3:])

    def set_time(self, time):
        check(lib.rt_scene_set_time(self.h, float(time)))

    def add_object(self, shape):
        return check(lib.rt_add_object(self.h, shape))

    def add_spheres_bulk(self, centers, radii, mat_ids):
        cr = np.ascontiguousarray(np.concatenate([centers, radii[:, None]], axis=1), dtype=np.float32)
        m = np.ascontiguousarray(mat_ids, dtype=np.uint32)
        return check(lib.rt_add_sphere_objects(self.h, len(m), cr.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p)))

    def set_option(self, option, value):
        check(lib.rt_scene_set_option(self.h, option, value))

    def set_convention(self, convention):
        self.set_option(abi.RT_OPTION_CONVENTION, convention)

    def commit(self):
        check(lib.rt_scene_commit(self.h))
        self.committed = True

    def stats(self):
        st = abi.RtSceneStats()
        check(lib.rt_scene_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    # ---- queries
    def _rays_dev(self, rays):
        if isinstance(rays, torch.Tensor):
            t = rays.to(device=self.ctx.device, dtype=torch.float32).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(rays, dtype=np.float32)).to(self.ctx.device)
        return t.reshape(-1, 6)

    def first_hit(self, rays, tmin=1e-3, tmax=float("inf"), volume_key=0):
        """Closest hit of each ray; returns a numpy structured array (HIT_DTYPE)."""
        r = self._rays_dev(rays)
        n = r.shape[0]
        out = torch.zeros(max(n, 1) * 48, dtype=torch.uint8, device=self.ctx.device)
        check(lib.rt_first_hit(self.h, r.data_ptr(), n, tmin, tmax, volume_key, out.data_ptr(), _stream()))
        torch.cuda.synchronize(self.ctx.device)
        return out.cpu().numpy()[: n * 48].view(HIT_DTYPE)

    def scatter(self, queries, rng_seed):
        """One scattering event per query (SCATTER_QUERY_DTYPE array) -> SCATTER_RESULT_DTYPE array."""
        q = np.ascontiguousarray(queries, dtype=SCATTER_QUERY_DTYPE)
        n = len(q)
        dev = torch.from_numpy(q.view(np.uint8).reshape(-1)).to(self.ctx.device)
        out = torch.zeros(max(n, 1) * 48, dtype=torch.uint8, device=self.ctx.device)
        check(lib.rt_scatter(self.h, dev.data_ptr(), n, rng_seed, out.data_ptr(), _stream()))
        torch.cuda.synchronize(self.ctx.device)
        return out.cpu().numpy()[: n * 48].view(SCATTER_RESULT_DTYPE)

    def intervals(self, shape, rays, scripted=None, volume_key=0):
        """Ordered boundary crossings of `shape` along each ray: list of event arrays."""
        r = self._rays_dev(rays)
        n = r.shape[0]
        ev = torch.zeros(max(n, 1) * abi.RT_MAX_EVENTS * 48, dtype=torch.uint8, device=self.ctx.device)
        cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=self.ctx.device)
        s_ptr, s_n = None, 0
        if scripted:
            s = torch.tensor(list(scripted), dtype=torch.float32, device=self.ctx.device)
            s_ptr, s_n = s.data_ptr(), len(scripted)
        check(lib.rt_intervals(self.h, shape, r.data_ptr(), n, s_ptr, s_n, volume_key, ev.data_ptr(), cnt.data_ptr(),
                               _stream()))
        torch.cuda.synchronize(self.ctx.device)
        evs = ev.cpu().numpy().view(EVENT_DTYPE).reshape(-1, abi.RT_MAX_EVENTS)
        cnts = cnt.cpu().numpy()
        return [evs[i, : cnts[i]] for i in range(n)]


def rng_words(ctx, streams, words_per_stream):
    """Raw PCG32 words of the device's (pixel, sample) streams and the WyRand per-sample seeds.
    streams: rows of (rng_seed, sample, x, y).  Returns (uint32 [n, words], uint64 [n])."""
    rec = np.zeros((len(streams), 8), dtype=np.uint32)
    for i, (seed, sample, x, y) in enumerate(streams):
        rec[i, :5] = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, sample, x, y)
    n = len(streams)
    dev = torch.from_numpy(rec).to(ctx.device)
    out = torch.zeros((max(n, 1), max(words_per_stream, 1)), dtype=torch.int32, device=ctx.device)
    wy = torch.zeros((max(n, 1), 2), dtype=torch.int32, device=ctx.device)
    check(lib.rt_rng_words(ctx.h, dev.data_ptr(), n, words_per_stream, out.data_ptr(), wy.data_ptr(), _stream()))
    torch.cuda.synchronize(ctx.device)
    words = out.cpu().numpy().view(np.uint32)[:n, :words_per_stream]
    w = wy.cpu().numpy().view(np.uint32)[:n].astype(np.uint64)
    return words, w[:, 0] | (w[:, 1] << np.uint64(32))


def camera_rays(ctx, camera, config, width, height, window=None):
    x0, y0, w, h = window if window else (0, 0, width, height)
    n = w * h * config.samples_per_pixel
    out = torch.zeros((max(n, 1), 6), dtype=torch.float32, device=ctx.device)
    cc, rc = camera.impl, config.impl
    check(lib.rt_camera_rays(ctx.h, C.byref(cc), C.byref(rc), config.first_sample, width, height, x0, y0, w, h,
                             out.data_ptr(), _stream()))
    torch.cuda.synchronize(ctx.device)
    return out[:n].cpu().numpy()


class Renderer:
    """ContentView.swift:103-235.  `draw` = one pass of `samples_per_pixel` samples added to
    the float accumulator; the accumulator resets when the size changes (:172-180)."""

    def __init__(self, scene, width, height, pipeline=abi.RT_PIPELINE_FUSED, device_ordinal=0, ctx=None, dedup=True,
                 bvh=abi.RT_BVH_SAH, refill=None, node_min=None, nodes=abi.RT_NODES_AUTO, timing=False, profile_phases=False,
                 bands=None, native=None, heavy_min=None, traversal=None):
        """`native`: an already committed NativeScene of `scene` to render from (a large scene is
        flattened and its tree built once, then shared by several renderers)."""
        self.ctx = ctx or (native.ctx if native is not None else Context(device_ordinal))
        self.scene = scene
        self.camera = scene.camera
        self.native = native if native is not None else NativeScene(self.ctx)
        if native is None:
            scene.encode(self.native, dedup=dedup)
            self.native.set_option(abi.RT_OPTION_BVH_BUILDER, bvh)
            self.native.set_option(abi.RT_OPTION_NODE_FORMAT, nodes)
        if native is not None and (refill or node_min is not None or heavy_min is not None or traversal is not None
                                   or timing or profile_phases):
            # these are properties of the committed scene, shared by every renderer on it: setting them here would
            # silently change the kernels of the others
            raise ValueError("refill / node_min / heavy_min / traversal / timing / profile_phases belong to the scene: "
                             "set them with native.set_option() before sharing it, not per Renderer")
        if refill:
            self.native.set_option(abi.RT_OPTION_REFILL_THRESHOLD, refill)
        if node_min is not None:
            self.native.set_option(abi.RT_OPTION_NODE_MIN, node_min)
        if heavy_min is not None:
            self.native.set_option(abi.RT_OPTION_HEAVY_MIN, heavy_min)
        if traversal is not None:
            self.native.set_option(abi.RT_OPTION_TRAVERSAL, traversal)
        if timing:
            self.native.set_option(abi.RT_OPTION_KERNEL_TIMING, 1)
        if profile_phases:
            self.native.set_option(abi.RT_OPTION_PROFILE_PHASES, 1)
        if native is None:
            self.native.commit()
        self.pipeline = pipeline
        # bands = (index, count): this renderer owns every count-th band of 8 pixel rows, starting
        # at band `index`; its accumulator holds just those bands (image-space sharding)
        self.bands = bands
        if bands is not None and pipeline != abi.RT_PIPELINE_FUSED:
            raise ValueError("image-space sharding runs on the fused pipeline")
        self.pass_counter = 0
        self.samples = 0
        self.resize(width, height)

    def resize(self, width, height):
        self.width, self.height = width, height
        rows = height
        if self.bands is not None:
            # every shard's buffer has the rows of shard 0 (the largest), so that equal-sized buffers can be gathered
            rows = check(lib.rt_band_rows(height, 0, self.bands[1]))
        self.accum = torch.zeros((rows, width, 3), dtype=torch.float32, device=self.ctx.device)
        # the size follows the compute-unit count of the current device: ask with the scene's device current
        with torch.cuda.device(self.ctx.device):
            ws = lib.rt_render_workspace_size(self.pipeline, width, height)
        self.workspace = torch.empty(max(ws, 256), dtype=torch.uint8, device=self.ctx.device)
        self.pass_counter = 0
        self.samples = 0

    def reset(self):
        self.accum.zero_()
        self.pass_counter = 0
        self.samples = 0

    def draw(self, samples_per_pixel=1, max_depth=10, rng_seed=0, first_sample=None):
        """Adds one pass.  `first_sample` defaults to the number of samples already held, so
        successive passes continue one stream of samples."""
        self.pass_counter += 1
        first = self.samples if first_sample is None else first_sample
        rc = RenderConfig(samples_per_pixel, max_depth, self.pass_counter, rng_seed, first).impl
        cc = self.camera.impl
        if self.bands is None:
            check(lib.rt_render(self.native.h, C.byref(cc), C.byref(rc), first, self.pipeline, self.width, self.height,
                                self.accum.data_ptr(), self.workspace.data_ptr(), self.workspace.numel(), _stream()))
        else:
            check(lib.rt_render_bands(self.native.h, C.byref(cc), C.byref(rc), first, self.bands[0], self.bands[1],
                                      self.width, self.height, self.accum.data_ptr(), self.workspace.data_ptr(),
                                      self.workspace.numel(), _stream()))
        self.samples += samples_per_pixel

    def mean(self):
        """Per-pixel mean radiance, [H, W, 3] float32 on the device."""
        return self.accum / max(self.samples, 1)

    def image_rgb8(self):
        out = torch.empty((self.height, self.width, 4), dtype=torch.uint8, device=self.ctx.device)
        check(lib.rt_resolve_rgb8(self.ctx.h, self.accum.data_ptr(), max(self.samples, 1), self.width, self.height,
                                  out.data_ptr(), _stream()))
        torch.cuda.synchronize(self.ctx.device)
        return out[..., :3].cpu().numpy()

    def kernel_time(self, reset=False):
        """(total ms, launches, kernel name) of the dominant kernel since the last reset; needs timing=True."""
        ms, n, name = C.c_double(), C.c_uint64(), C.c_char_p()
        check(lib.rt_kernel_time(self.native.h, 1 if reset else 0, C.byref(ms), C.byref(n), C.byref(name)))
        return ms.value, n.value, (name.value or b"").decode()


class ProgressiveView:
    """The interactive loop of the reference app (ContentView.swift:103-235) without the
    window: every frame adds one pass of 1 sample per pixel at depth 10 with a fresh seed,
    blends it into a 10-bit running mean and produces the bgra8 drawable; the pass counter
    restarts when the camera or the size changes and the view pauses at 200 passes."""

    MAX_PASSES = 200          # ContentView.swift:188
    SAMPLES_PER_PASS = 1      # ContentView.swift:202
    MAX_DEPTH = 10

    def __init__(self, scene, width, height, seed=0, overlap=False, **renderer_args):
        """overlap: successive passes alternate between two streams (one of each priority class, so that they sit on
        different hardware queues), each with its own pass-sum buffer and workspace: pass n + 1 starts on the compute
        units that pass n's last, long paths leave idle, and the present steps stay in pass order (an event per pass).
        The words and pixels are the same either way.  Off by default: the reference's loop is one pass at a time.
        (Three passes in flight - two streams of one priority class - reached 3554 frames/s once and hung the next run;
        not offered.)"""
        from .scene import WyRand
        self.renderer = Renderer(scene, width, height, **renderer_args)
        self.seeds = WyRand(seed)             # stands in for SystemRandomNumberGenerator (:201)
        self.paused = False
        self.overlap = bool(overlap)
        self._alloc()

    def _alloc(self):
        r = self.renderer
        dev = r.ctx.device
        self.acc = torch.zeros((r.height, r.width), dtype=torch.int32, device=dev)        # rgb10a2Uint words
        self.drawable = torch.zeros((r.height, r.width, 4), dtype=torch.uint8, device=dev)  # bgra8Unorm
        self.pass_counter = 0
        self.paused = False
        # per pass slot: stream, pass sums, workspace, event recorded after the slot's present
        self._slots = []
        for k in range(2 if self.overlap else 1):
            # one stream of each priority class: the two then sit on different hardware queues (two streams of one class
            # may share a queue, and passes on a shared queue do not overlap)
            stream = torch.cuda.Stream(device=dev, priority=-k) if self.overlap else None
            sums = r.accum if k == 0 else torch.zeros_like(r.accum)
            work = r.workspace if k == 0 else torch.empty_like(r.workspace)
            self._slots.append((stream, sums, work, torch.cuda.Event()))
        self._last = None                      # slot of the latest pass
        if self.overlap:
            for stream, _, _, _ in self._slots:
                stream.wait_stream(torch.cuda.current_stream(dev))      # after the allocations above

    # ---- Renderer.scene didSet / mtkView(_:drawableSizeWillChange:)  (:107-113, :172-180)
    def set_camera(self, camera):
        if camera != self.renderer.camera:
            self.renderer.camera = camera
            self.set_needs_redraw()

    def resize(self, width, height):
        self.wait()
        self.renderer.resize(width, height)
        self._alloc()

    def set_needs_redraw(self):
        self.pass_counter = 0
        self.paused = False

    def wait(self):
        """Makes the caller's current stream wait for the latest pass (its drawable and running mean)."""
        if self.overlap and self._last is not None:
            torch.cuda.current_stream(self.renderer.ctx.device).wait_event(self._slots[self._last][3])

    # ---- Renderer.draw(in:)  (:182-224)
    def draw(self, rng_seed=None):
        """One frame.  Returns False (and does nothing) while paused."""
        if self.paused:
            return False
        self.pass_counter += 1
        if self.pass_counter >= self.MAX_PASSES:
            self.paused = True
        r = self.renderer
        seed = self.seeds.next() if rng_seed is None else rng_seed
        slot = self.pass_counter % len(self._slots)
        stream, sums, work, done = self._slots[slot]
        rc = RenderConfig(self.SAMPLES_PER_PASS, self.MAX_DEPTH, self.pass_counter, seed, 0).impl
        cc = r.camera.impl
        with torch.cuda.stream(stream) if stream is not None else _no_stream():
            sums.zero_()
            check(lib.rt_render(r.native.h, C.byref(cc), C.byref(rc), 0, r.pipeline, r.width, r.height, sums.data_ptr(),
                                work.data_ptr(), work.numel(), _stream()))
            if self.overlap and self._last is not None:
                torch.cuda.current_stream().wait_event(self._slots[self._last][3])      # the running mean is updated in pass order
            check(lib.rt_progressive_present(r.ctx.h, sums.data_ptr(), self.SAMPLES_PER_PASS, self.pass_counter,
                                             self.acc.data_ptr(), self.drawable.data_ptr(), r.width, r.height, _stream()))
            done.record()
        r.accum = sums                          # this pass alone, as before
        self._last = slot
        return True

    def running_mean(self):
        """The 10-bit accumulator decoded as the kernel reads it back (Shaders.metal:153)."""
        w = self.acc.cpu().numpy().astype(np.uint32)
        rgb = np.stack([(w >> (10 * c)) & 1023 for c in range(3)], axis=-1).astype(np.float32)
        return (rgb + np.nextafter(np.float32(0.5), np.float32(0))) / np.float32(1024)
