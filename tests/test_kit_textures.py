"""RayTracingKit textures (Texture.swift), texture-backed materials (Material.swift:17-41,
75-93) and the PPM writer (Image.swift:86-117).  The reference has no test of these, so the
oracle is checked against values worked here straight from the reference's expressions
(parity unpinned), and the HIP path against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_bridge as ob
from raytracer_amd import abi, image, scenes
from raytracer_amd.scene import (BilinearImageTexture, CameraConfig, CheckerTexture, ColoredLambertian, Cuboid, Cylinder,
                                 ImageTexture, KitEmissive, KitLambertian, KitPerlinNoise, Marble, NoiseTexture, Quad,
                                 RenderConfig, Scene, SceneEncoder, SolidColor, Sphere, Transform, WyRand)


def tiny_image():
    px = np.array([[[0, 0, 0], [255, 0, 0], [10, 20, 30]],
                   [[0, 255, 0], [0, 0, 255], [40, 50, 60]]], dtype=np.uint8)      # 3 wide, 2 high, rows top to bottom
    return ImageTexture("tiny", px)


def bilinear_by_hand(tex, u, v):
    """Texture.swift:80-120 written out with Python integers and floats."""
    px = tex.image.pixels.astype(np.float64) / 255.0
    width, height = abs(tex.width), abs(tex.height)
    sx, sy = (-1 if tex.width < 0 else 1), (-1 if tex.height < 0 else 1)

    def clip(index, size, wrap):
        if index < 0:
            return size + int(np.fmod(index, size)) if wrap else 0
        if index >= size:
            return index % size if wrap else size - 1
        return index

    def get(i, j):
        return px[tex.origin_y + clip(i, height, tex.wrap_y) * sy, tex.origin_x + clip(j, width, tex.wrap_x) * sx]
    i, j = v * height - 0.5, u * width - 0.5
    i1, j1 = int(np.floor(i)), int(np.floor(j))
    a, b = i - i1, j - j1
    c1 = get(i1, j1) * (1 - b) + get(i1, j1 + 1) * b
    c2 = get(i1 + 1, j1) * (1 - b) + get(i1 + 1, j1 + 1) * b
    return c1 * (1 - a) + c2 * a


def test_oracle_image_texture_windows():
    img = tiny_image()
    cases = [BilinearImageTexture(img),                                   # whole image, v up
             BilinearImageTexture(img, 0, 0, 3, 2),                       # rows downwards
             BilinearImageTexture(img, 2, 1, -2, -2, wrap_x=False, wrap_y=False),   # mirrored window, clamped
             BilinearImageTexture(img, 1, 0, 2, 2, wrap_x=True, wrap_y=False)]
    rs = np.random.RandomState(2)
    for tex in cases:
        orc = ob.OracleScene()
        tid = tex.encode(SceneEncoder(orc))
        for u, v in list(rs.uniform(-0.2, 1.2, size=(60, 2))) + [(0.0, 0.0), (1.0, 1.0), (0.5, 0.5)]:
            got = orc.kit_texture_color(tid, u, v, (0, 0, 0))
            assert np.allclose(got, bilinear_by_hand(tex, u, v), atol=1e-12), (tex.width, tex.height, u, v)
    # a texel centre returns the texel: default window, v = 0.25 is the centre of the bottom row
    orc = ob.OracleScene()
    tid = BilinearImageTexture(img).encode(SceneEncoder(orc))
    assert np.allclose(orc.kit_texture_color(tid, 0.5, 0.25, (0, 0, 0)), [0, 0, 1])
    assert np.allclose(orc.kit_texture_color(tid, 0.5, 0.75, (0, 0, 0)), [1, 0, 0])


def test_oracle_checker_solid_noise_marble():
    orc = ob.OracleScene()
    enc = SceneEncoder(orc)
    rng = WyRand(9)
    noise = KitPerlinNoise(rng)
    checker = CheckerTexture(0.5, (1, 0, 0), CheckerTexture(0.25, (0, 1, 0), SolidColor((0, 0, 1))))
    cid = checker.encode(enc)
    # Texture.swift:44-50: floor(p / scale) summed; even -> first texture
    assert np.allclose(orc.kit_texture_color(cid, 0, 0, (0.1, 0.1, 0.1)), [1, 0, 0])
    assert np.allclose(orc.kit_texture_color(cid, 0, 0, (-0.1, 0.1, 0.1)), [0, 0, 1])      # cell -1: odd; inner cell -1: odd
    assert np.allclose(orc.kit_texture_color(cid, 0, 0, (0.6, 0.1, 0.1)), [0, 1, 0])       # outer odd, inner cell 2: even
    nid = NoiseTexture(noise, 2.0).encode(enc)
    mid = Marble(noise, 1.5).encode(enc)

    def turb(p, n=7):                 # Texture.swift:184-223
        def single(p):
            u, v, w = p - np.floor(p)
            h = lambda x: x * x * (3 - 2 * x)
            i, j, k = (int(c) for c in p)              # truncation, as Int()
            out = 0.0
            for di in range(2):
                for dj in range(2):
                    for dk in range(2):
                        idx = int(noise.perms[0][(i + di) & 255]) ^ int(noise.perms[1][(j + dj) & 255]) ^ int(noise.perms[2][(k + dk) & 255])
                        wgt = (h(u) if di else 1 - h(u)) * (h(v) if dj else 1 - h(v)) * (h(w) if dk else 1 - h(w))
                        out += wgt * float(np.dot(noise.vectors[idx].astype(np.float64), [u - di, v - dj, w - dk]))
            return out
        acc, wgt, q = 0.0, 1.0, np.array(p, dtype=np.float64)
        for _ in range(n):
            acc += wgt * single(q)
            wgt *= 0.5
            q = q * 2.0
        return acc
    for p in [(0.3, 1.7, 2.2), (-1.4, 0.2, 5.9), (10.25, -3.5, 0.75)]:
        grey = abs(turb(2.0 * np.array(p)))
        assert np.allclose(orc.kit_texture_color(nid, 0, 0, p), [grey] * 3, atol=1e-12)
        m = (1 + np.sin(1.5 * p[2] + 10 * turb(p))) * 0.5
        assert np.allclose(orc.kit_texture_color(mid, 0, 0, p), [m] * 3, atol=1e-12)


def test_ppm_writer(tmp_path):
    rgb = np.array([[[255, 0, 1], [2, 3, 4]], [[5, 6, 7], [8, 9, 10]], [[11, 12, 13], [14, 15, 16]]], dtype=np.uint8)
    assert image.ppm_text(rgb) == "P3\n2 3\n255\n255 0 1\n2 3 4\n5 6 7\n8 9 10\n11 12 13\n14 15 16\n"   # Image.swift:86-97
    path = os.path.join(str(tmp_path), "results", "light.ppm")        # the directory is created, Image.swift:100-101
    image.write_ppm(path, rgb)
    assert np.array_equal(image.read_ppm(path), rgb)


def test_host_texture_errors_without_a_gpu():
    from raytracer_amd._native import lib
    h = C.c_void_p()
    assert lib.rt_scene_create(None, C.byref(h)) == 0
    px = tiny_image().pixels
    img = lib.rt_add_texture_rgb8(h, 3, 2, px.ctypes.data_as(C.c_void_p))
    t = abi.RtKitTexture()
    t.kind, t.image, t.origin_x, t.origin_y, t.width, t.height = abi.RT_KIT_TEXTURE_IMAGE, img, 0, 1, 3, -2
    assert lib.rt_add_kit_texture(h, C.byref(t)) == 0
    t.width = 4
    assert lib.rt_add_kit_texture(h, C.byref(t)) < 0 and b"outside" in lib.rt_last_error()
    t.width, t.height = 3, 0
    assert lib.rt_add_kit_texture(h, C.byref(t)) < 0
    c = abi.RtKitTexture()
    c.kind, c.scale, c.even, c.odd = abi.RT_KIT_TEXTURE_CHECKER, 1.0, 0, 5
    assert lib.rt_add_kit_texture(h, C.byref(c)) < 0                   # children must exist already
    c.odd = 0
    assert lib.rt_add_kit_texture(h, C.byref(c)) == 1
    n = abi.RtKitTexture()
    n.kind, n.perlin = abi.RT_KIT_TEXTURE_MARBLE, 0
    assert lib.rt_add_kit_texture(h, C.byref(n)) < 0                   # no noise tables yet
    n.kind = 9
    assert lib.rt_add_kit_texture(h, C.byref(n)) < 0
    m = abi.RtKitTexturedMaterial(abi.material_kind_lambertian_kit_texture, 7)
    assert lib.rt_add_material(h, C.byref(m), C.sizeof(m)) < 0
    m.texture = 1
    assert lib.rt_add_material(h, C.byref(m), C.sizeof(m)) == 0
    lib.rt_scene_destroy(h)


def test_kit_worlds_flatten_on_the_oracle():
    for make in (scenes.kit_world1, scenes.kit_world2):
        scene = make()
        orc = ob.oracle_scene_from(scene)
        img, _, _ = orc.render(scene.camera, RenderConfig(1, 6, 1, 2, 0), 32, 18)
        assert np.isfinite(img).all() and img.max() > 0


# ---------------------------------------------------------------- HIP path against the oracle
@pytest.fixture(scope="module")
def ctx():
    from raytracer_amd.renderer import Context
    return Context(0)


def emissive_gallery():
    """Every texture kind as an emitter on a quad, a sphere, a cylinder and a box: at depth 1
    a pixel is the texture value at the first hit."""
    rng = WyRand(21)
    img = ImageTexture.procedural("gallery", 64, 48, 5)
    noise = KitPerlinNoise(rng)
    textures = [
        BilinearImageTexture(img),
        BilinearImageTexture(img, 40, 10, -24, 30, wrap_x=False, wrap_y=True),
        CheckerTexture(0.3, (0.9, 0.1, 0.1), CheckerTexture(0.1, (0.1, 0.9, 0.1), Marble(noise, 2.0))),
        NoiseTexture(noise, 3.0),
        Marble(noise, 4.0),
    ]
    objs = []
    for k, tex in enumerate(textures):
        x = 2.4 * (k - 2)
        e = KitEmissive(tex)
        # (the quad stays off z = 0: on a lattice plane of the noise and the checker the cell is decided by rounding)
        objs += [Quad((x - 1, 1.2, 0.13), (2, 0, 0), (0, 1.6, 0), e),
                 Sphere(radius=0.8, material=e, transform=Transform.translate(x, 0.2, 0.5) * Transform.rotation_deg(30 * k, "y")),
                 Cylinder(Transform.translate(x - 0.6, -2.4, 0.4) * Transform.rotation_deg(20, "x"), 0.4, 1.4, material=e),
                 Cuboid(Transform.translate(x + 0.1, -2.4, 0.2) * Transform.rotation_deg(35, "y"), (0.7, 1.2, 0.7), e)]
    cam = CameraConfig(vertical_fov=45, look_from=(0, 0, 9), look_at=(0, 0, 0), background=abi.background_lighting_none)
    return Scene(cam, objs)


@pytest.mark.gpu
def test_texture_values_at_first_hits(ctx):
    import torch
    from raytracer_amd.renderer import Renderer
    scene = emissive_gallery()
    w, h, spp, seed = 320, 180, 4, 13
    ref, _, _ = ob.oracle_scene_from(scene).render(scene.camera, RenderConfig(spp, 1, 1, seed, 0), w, h, precision=ob.ORC_F64)
    for pipeline in (abi.RT_PIPELINE_FUSED, abi.RT_PIPELINE_WAVEFRONT):
        r = Renderer(scene, w, h, pipeline=pipeline, ctx=ctx)
        r.draw(spp, 1, seed)
        torch.cuda.synchronize()
        got = r.accum.cpu().numpy().astype(np.float64)
        close = np.abs(got - ref) <= 2e-3 * spp
        lit = ref.sum(axis=2) > 0
        print("pipeline", pipeline, "lit pixels", int(lit.sum()), "equal to 2e-3:", close[lit].mean())
        assert lit.sum() > 0.3 * w * h
        # float32 coordinates move a sample across a checker or texel edge now and then
        assert close[lit].mean() > 0.995


@pytest.mark.gpu
@pytest.mark.parametrize("make", [scenes.kit_world1, scenes.kit_world2])
def test_kit_worlds_render_like_the_oracle(ctx, make):
    import torch
    from raytracer_amd.renderer import Renderer
    from test_gpu_render import psnr
    scene = make()
    w, h, spp, depth, seed = 128, 72, 16, 12, 4
    ref, _, _ = ob.oracle_scene_from(scene).render(scene.camera, RenderConfig(spp, depth, 1, seed, 0), w, h, precision=ob.ORC_F64)
    ref = ref / spp
    imgs = []
    for pipeline in (abi.RT_PIPELINE_FUSED, abi.RT_PIPELINE_WAVEFRONT):
        r = Renderer(scene, w, h, pipeline=pipeline, ctx=ctx)
        r.draw(spp, depth, seed)
        torch.cuda.synchronize()
        img = r.mean().cpu().numpy().astype(np.float64)
        imgs.append(img)
        p = psnr(img, ref)
        print(make.__name__, "pipeline", pipeline, "PSNR %.1f dB" % p, "equal to 1e-3: %.3f" % (np.abs(img - ref) <= 1e-3).mean())
        assert p >= 30.0
    assert psnr(imgs[0], imgs[1]) >= 60.0
    out = Renderer(scene, 64, 36, ctx=ctx)
    out.draw(4, 8, 1)
    rgb = out.image_rgb8()
    assert image.read_ppm_text(image.ppm_text(rgb)).shape == (36, 64, 3)


# ---------------------------------------------------------------- image files
def test_image_loader_decodes_png_and_jpeg_and_falls_back(tmp_path):
    """TextureLoader.doLoad (MetalRayTracer/Texture.swift:29-41) / Image.load (Image.swift:120-129)."""
    from PIL import Image as PILImage
    rs = np.random.RandomState(1)
    px = rs.randint(0, 256, size=(12, 20, 3)).astype(np.uint8)
    PILImage.fromarray(px).save(os.path.join(str(tmp_path), "exact.png"))
    t = ImageTexture.load("exact.png", [str(tmp_path)])
    assert np.array_equal(t.pixels, px)                       # lossless, rows top to bottom
    smooth = np.zeros((32, 48, 3), dtype=np.uint8)
    smooth[..., 0] = np.linspace(0, 255, 48)[None, :]
    smooth[..., 1] = np.linspace(0, 255, 32)[:, None]
    PILImage.fromarray(smooth).save(os.path.join(str(tmp_path), "smooth.jpg"), quality=95)
    j = ImageTexture.load("smooth.jpg", ["/nonexistent", str(tmp_path)])
    assert j.pixels.shape == (32, 48, 3) and np.abs(j.pixels.astype(int) - smooth.astype(int)).mean() < 3
    missing = ImageTexture.load("nope.jpg", [str(tmp_path)])
    assert missing.pixels.shape == (1, 1, 3) and list(missing.pixels[0, 0]) == [255, 0, 255]     # magenta placeholder, :43-54
    open(os.path.join(str(tmp_path), "broken.jpg"), "wb").write(b"not a jpeg")
    assert ImageTexture.load("broken.jpg", [str(tmp_path)]).pixels.shape == (1, 1, 3)
    assert ImageTexture.bundled("nope.jpg", 16, 8, 3, str(tmp_path)).pixels.shape == (8, 16, 3)


REF_TEXTURES = "/root/reference/RayTracing/textures"


@pytest.mark.skipif(not os.path.isdir(REF_TEXTURES), reason="the reference's bundled JPEGs are not on this machine")
def test_kit_world1_with_the_bundled_jpegs():
    """With the reference's own earthmap.jpg / barrel.jpg the barrel windows of
    RayTracing/main.swift:53-55 stay inside the 512x512 image and the scene flattens."""
    scene = scenes.kit_world1(texture_dir=REF_TEXTURES)
    orc = ob.oracle_scene_from(scene)
    img, _, _ = orc.render(scene.camera, RenderConfig(1, 6, 1, 2, 0), 32, 18)
    assert np.isfinite(img).all() and img.max() > 0
    from raytracer_amd._native import lib
    h = C.c_void_p()
    assert lib.rt_scene_create(None, C.byref(h)) == 0

    class Sink:
        def __getattr__(self, fn):
            def call(*args):
                if fn == "add_material":
                    rc = lib.rt_add_material(h, C.byref(args[0]), C.sizeof(args[0]))
                elif fn == "add_texture_rgb8":
                    rc = lib.rt_add_texture_rgb8(h, args[0], args[1], np.ascontiguousarray(args[2]).ctypes.data_as(C.c_void_p))
                elif fn == "shape_csg":
                    arr = (C.c_int * len(args[1]))(*args[1])
                    rc = lib.rt_shape_csg(h, args[0], len(args[1]), arr)
                elif fn == "add_object":
                    rc = lib.rt_add_object(h, args[0])
                elif fn == "set_convention":
                    rc = lib.rt_scene_set_option(h, abi.RT_OPTION_CONVENTION, args[0])
                else:
                    rc = getattr(lib, "rt_" + fn)(h, C.byref(args[0]))
                assert rc >= 0, (fn, lib.rt_last_error())
                return rc
            return call
    scene.encode(Sink())
    lib.rt_scene_destroy(h)
