"""Integer, byte and index work of the HIP path is bit-exact (GPU needed): the device's random
generators word for word, the 8-bit output stage byte for byte, the reference's record bytes
through the C ABI, and image-space sharding pixel for pixel."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_bridge as ob
from raytracer_amd import abi, scenes
from raytracer_amd.scene import ColoredLambertian, RenderConfig, Scene, CameraConfig, Sphere
from test_abi_and_host import reference_sphere_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from raytracer_amd.renderer import Context
    return Context(0)


def pcg32_python(initstate, initseq, n):
    """Impl/RNG.h:31-52 in plain integers (a third implementation next to the oracle's C++)."""
    mask = (1 << 64) - 1
    mult = 6364136223846793005
    state, inc = 0, ((initseq << 1) | 1) & mask
    state = (state * mult + inc) & mask
    state = (state + initstate) & mask
    state = (state * mult + inc) & mask
    out = []
    for _ in range(n):
        old = state
        state = (old * mult + inc) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        out.append(((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF)
    return out


def test_device_generators_word_for_word(ctx):
    """PCG32 (Impl/RNG.h:31-52) and WyRand (RayTracingKit/PRNG.swift:17-21) as the kernels run them:
    raw 32-bit words of (seed, sample, x, y) streams and the per-sample 64-bit seeds equal the
    oracle's and a plain-integer evaluation, bit for bit."""
    from raytracer_amd.renderer import rng_words
    rs = np.random.RandomState(5)
    streams = [(0, 0, 0, 0), (1, 0, 0, 0), (42, 54, 0, 0), (0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 65535, 65535),
               (0x9E3779B97F4A7C15, 1023, 1919, 1079), (7, 8191, 3839, 2159)]
    streams += [(int(rs.randint(0, 2 ** 62)) * 4 + int(rs.randint(0, 4)), int(rs.randint(0, 2 ** 31)), int(rs.randint(0, 65536)),
                 int(rs.randint(0, 65536))) for _ in range(250)]
    n_words = 24
    words, seeds = rng_words(ctx, streams, n_words)
    for i, (seed, sample, x, y) in enumerate(streams):
        per_sample = ob.wyrand(seed, sample + 1)[sample] if sample < 4096 else None
        if per_sample is not None:
            assert int(seeds[i]) == per_sample, (i, hex(int(seeds[i])), hex(per_sample))
        # WyRand output number `sample` in plain integers
        m64 = (1 << 64) - 1
        s = (seed + (sample + 1) * 0xA0761D6478BD642F) & m64
        prod = s * (s ^ 0xE7037ED1A0B428DB)
        assert int(seeds[i]) == ((prod >> 64) ^ prod) & m64
        expect = ob.sample_stream_words(seed, sample, x, y, n_words)
        assert [int(v) for v in words[i]] == expect, (i, seed, sample, x, y)
        lo, hi = int(seeds[i]) & 0xFFFFFFFF, int(seeds[i]) >> 32
        assert expect == pcg32_python((x * 5569 + lo) & 0xFFFFFFFF, (y * 2707 + hi) & 0xFFFFFFFF, n_words)
    # the canonical PCG32 demo stream (seed 42, sequence 54) through the oracle, for the record
    assert ob.pcg32(42, 54, 4) == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293]
    assert pcg32_python(42, 54, 4) == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293]


def test_resolve_rgb8_byte_for_byte(ctx):
    """Camera.swift:92 + Vector3D.swift:185-211: mean, square root, clamp, floor(nextDown(256) x).
    The kernel's bytes equal the oracle's float evaluation exactly - on a rendered image and on
    values chosen to sit on quantisation edges, zero, above one, negative and NaN."""
    from raytracer_amd._native import check, lib
    from raytracer_amd.renderer import Renderer, _stream
    scene = scenes.simple_balls()
    r = Renderer(scene, 64, 36, ctx=ctx)
    r.draw(4, 10, 1)
    img = r.image_rgb8()
    ref = ob.resolve_rgb8(r.accum.cpu().numpy().reshape(-1, 3), 4).reshape(36, 64, 3)
    assert np.array_equal(img, ref)
    # edge values: every 8-bit level's lower edge and its float neighbours, for several sample counts
    for n in (1, 3, 7, 100, 1024):
        levels = np.arange(0, 257, dtype=np.float64) / np.nextafter(np.float32(256), np.float32(0))
        edge = (levels ** 2 * n).astype(np.float32)
        vals = np.concatenate([edge, np.nextafter(edge, np.float32(-1)), np.nextafter(edge, np.float32(1e9)),
                               np.array([0, -0.0, -1, 1e-30, 1e30, np.inf, np.nan, n, n * 1.0000001], dtype=np.float32)])
        vals = np.resize(vals, ((len(vals) + 2) // 3) * 3).astype(np.float32)
        sums = torch.from_numpy(vals.reshape(-1, 3)).to(ctx.device)
        npx = sums.shape[0]
        out = torch.zeros((npx, 4), dtype=torch.uint8, device=ctx.device)
        check(lib.rt_resolve_rgb8(ctx.h, sums.data_ptr(), n, npx, 1, out.data_ptr(), _stream()))
        torch.cuda.synchronize()
        got = out.cpu().numpy()[:, :3]
        assert np.array_equal(got, ob.resolve_rgb8(vals.reshape(-1, 3), n)), n


def test_reference_record_bytes_give_the_same_hit(ctx):
    """The 80 bytes of a reference `Sphere` (simd layout, Renderable.h:13-25) and the 32 bytes of a
    `ColoredLambertianMaterial` handed over raw give the hits of the same scene described through
    the host API."""
    from raytracer_amd._native import check, lib
    from raytracer_amd.renderer import NativeScene
    from common import random_rays
    m = ColoredLambertian((0.25, 0.5, 0.75))
    q = np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], dtype=np.float32)     # a rotation: columns
    spheres = [((1.0, 2.0, 3.0), 0.5), ((-1.0, 0.5, 0.0), 1.25), ((0.0, -2.0, 1.0), 0.75)]
    raw = NativeScene(ctx)
    mat = bytes([1, 0, 0, 0]) + bytes(12) + np.array([0.25, 0.5, 0.75, 0], dtype=np.float32).tobytes()
    assert check(lib.rt_add_material(raw.h, mat, len(mat))) == 0
    for c, rad in spheres:
        b = reference_sphere_bytes(c, rad, 0, columns=[tuple(float(v) for v in col) for col in q])
        sid = check(lib.rt_shape_sphere(raw.h, C.cast(C.c_char_p(b), C.POINTER(abi.RtSphere))))
        raw.add_object(sid)
    raw.commit()
    from raytracer_amd.scene import raw_transform
    host = NativeScene(ctx)
    Scene(CameraConfig(), [Sphere(radius=rad, material=m, transform=raw_transform(q, c)) for c, rad in spheres]).encode(host)
    host.commit()
    rays = random_rays(4000, 3, center=(0, 0, 1), radius=6.0, aim_radius=2.5)
    a, b = raw.first_hit(rays), host.first_hit(rays)
    assert (a["object"] >= 0).sum() > 500
    assert a.tobytes() == b.tobytes()         # every field of every hit, uv (which depends on the rotation) included


@pytest.mark.parametrize("size,count", [((96, 54), 2), ((131, 77), 3), ((64, 8), 4), ((40, 100), 8)])
def test_band_shards_put_together_are_the_image(ctx, size, count):
    """Image-space sharding: every shard renders only its bands of 8 rows into its own band buffer;
    the shards assembled (no arithmetic) equal the image one call renders - same random streams
    per (pixel, sample), so equal up to the order of the float sums within a pixel."""
    from raytracer_amd import dist as rdist
    from raytracer_amd._native import lib
    from raytracer_amd.renderer import Renderer
    scene = scenes.simple_balls()
    w, h = size
    whole = Renderer(scene, w, h, ctx=ctx)
    whole.draw(6, 12, 9)
    parts = []
    for index in range(count):
        assert lib.rt_band_rows(h, index, count) == rdist.band_rows(h, index, count)
        r = Renderer(scene, w, h, ctx=ctx, bands=(index, count))
        assert r.accum.shape[0] == rdist.band_rows(h, 0, count)
        r.draw(6, 12, 9)
        parts.append(r.accum)
    torch.cuda.synchronize()
    image = rdist.assemble_bands(torch.stack(parts), h)
    assert image.shape == whole.accum.shape
    scale = max(1.0, whole.accum.abs().max().item())
    assert (image - whole.accum).abs().max().item() <= 1e-5 * scale
    # rows beyond a shard's own bands stay untouched (zero)
    for index, part in enumerate(parts):
        assert part[rdist.band_rows(h, index, count):].abs().max().item() == 0 if rdist.band_rows(h, index, count) < part.shape[0] else True
    assert lib.rt_band_rows(h, count, count) < 0 and lib.rt_band_rows(h, 0, 0) < 0
