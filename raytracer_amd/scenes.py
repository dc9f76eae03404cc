"""Scenes: the reference's demo scenes (data, MetalRayTracer/Scene.swift:18-257 and
RayTracing/main.swift:11-104) and the seeded synthetic workloads the benchmark runs.

The reference draws its random numbers from the system generator; every scene here
takes a seed and draws from WyRand (RayTracingKit/PRNG.swift) so that it is the same
scene everywhere.  The reference's JPEG textures are replaced by procedural RGB8 images.
"""
import math

import numpy as np

from . import abi
from .scene import (CameraConfig, ColoredEmissive, ColoredIsotropic, ColoredLambertian, ColoredMetal,
                    ConstantDensityVolume, Cuboid, Cylinder, Dielectric, ImageTexture, Intersection,
                    PerlinNoiseLambertian, PerlinNoiseTexture, Quad, Scene, Sphere, Subtract,
                    TexturedLambertian, Transform, Union, WyRand, u64_to_unit, wyrand_block, MotionBlur,
                    BilinearImageTexture, CheckerTexture, KitEmissive, KitLambertian, KitPerlinNoise, Marble, NoiseTexture, Transformed)
from .simd import Quat, normalize, vec3

T = Transform


def simple_balls():
    """Scene.swift:18-32."""
    ground = ColoredLambertian((0.8, 0.8, 0.0))
    center = ColoredLambertian((0.1, 0.2, 0.5))
    left = Dielectric(1.50)
    right = ColoredMetal((0.8, 0.6, 0.2), 1.0)
    return Scene(CameraConfig(), [
        Sphere((-1.0, 0.0, -1), 0.5, left),
        Sphere((0.0, 0.0, -1.2), 0.5, center),
        Sphere((+1.0, 0.0, -1), 0.5, right),
        Sphere((0, -100.5, -1), 100, ground),
    ])


def single_cylinder():
    """Scene.swift:34-47."""
    ground = ColoredLambertian((0.5, 0.8, 0.0))
    return Scene(CameraConfig(), [
        Cylinder(T.translate(0, 0, -1) * T.rotation_deg(15, "x"), 0.2, 0.5, ground),
    ])


def simple_cylinders():
    """Scene.swift:49-78."""
    ground = ColoredLambertian((0.8, 0.8, 0.0))
    center = ColoredLambertian((0.1, 0.2, 0.5))
    left = Dielectric(1.50)
    right = ColoredMetal((0.8, 0.6, 0.2), 0.1)
    return Scene(CameraConfig(), [
        Cylinder(T.translate(-0.5, 0, -1) * T.rotation_deg(90, "x") * T.translate(0, -0.3, 0), 0.2, 0.6, left),
        Cylinder(T.translate(0, -0.2, -1.2), 0.2, 0.4, center),
        Cylinder(T.translate(+0.5, 0, -1) * T.rotation_deg(90, "z") * T.translate(0, -0.2, 0), 0.2, 0.4, right),
        Sphere((0, -100.5, -1), 100, ground),
    ])


def textured_cylinders(seed=5, texture_dir=None):
    """Scene.swift:80-119; the bundled JPEGs from `texture_dir` (MetalRayTracer/Textures) when
    given, procedural images of the same sizes otherwise."""
    ground = ColoredLambertian((1.8, 1.8, 1.0))
    left = Dielectric(1.50)
    right = ColoredMetal((0.8, 0.6, 0.2), 0.1)
    tex = lambda name, w, h, k: TexturedLambertian(ImageTexture.bundled(name, w, h, seed + k, texture_dir))
    return Scene(CameraConfig(), [
        Cylinder(T.translate(-0.5, 0, -1) * T.rotation_deg(90, "x") * T.translate(0, -0.3, 0), 0.2, 0.6, left),
        Cylinder(T.translate(0, -0.2, -1.2), 0.2, 0.4,
                 bottom=tex("barrel-bottom.jpg", 192, 192, 0), top=tex("barrel-top.jpg", 192, 192, 1),
                 side=tex("barrel-side.jpg", 512, 320, 2)),
        Sphere((0, 0.8, -1.2), 0.3, tex("earthmap.jpg", 1024, 512, 3)),
        Cylinder(T.translate(+0.5, 0, -1) * T.rotation_deg(90, "z") * T.translate(0, -0.2, 0), 0.2, 0.4, right),
        Sphere((0, -100.5, -1), 100, ground),
    ])


def _glass():
    """Scene.swift:128-141."""
    return Subtract(
        Cylinder.between((0, 0, 0), (0, 12, 0), 4, material=Dielectric(1.5)),
        Cylinder.between((0, 1, 0), (0, 13, 0), 3.5, material=Dielectric(1.5)),
    )


def pencil_in_glass():
    """Scene.swift:121-176."""
    ground = ColoredLambertian((0.8, 0.8, 0.0))
    center = ColoredLambertian((0.1, 0.2, 0.5))
    water = Cylinder.between((0, 1.001, 0), (0, 8, 0), 3.5 - 0.001, material=Dielectric(1.33))
    alpha = np.float32(math.pi / 6)
    p1 = vec3(-3.4 * math.cos(alpha), 1.1, -3.4 * math.sin(alpha))
    p2 = vec3(0, 12, 3.4)
    p3 = p1 + (p2 - p1) * np.float32(13.0 / 11.0)
    pencil = Cylinder.between(p1, p3, 0.1, material=center)
    return Scene(CameraConfig(vertical_fov=60, look_from=(-15, 14, 0), look_at=(8, 3, 0)), [
        Sphere((0, -500.5, -1), 500.0, ground),
        _glass(),
        water,
        pencil,
    ])


def _composition_object():
    """Scene.swift:181-198."""
    red = ColoredLambertian((1, 0, 0))
    green = ColoredLambertian((0, 1, 0))
    blue = ColoredLambertian((0, 0, 1))
    box = Cuboid(T.translate(-1, -1, -1), (2, 2, 2), red)
    sphere = Sphere((0, 0, 0), 1.3, blue)
    cyl1 = Cylinder(T.translate(0, -2, 0), 0.55, 4, green)
    cyl2 = Cylinder(T.rotation_deg(90, "x") * T.translate(0, -2, 0), 0.55, 4, green)
    cyl3 = Cylinder(T.rotation_deg(90, "z") * T.translate(0, -2, 0), 0.55, 4, green)
    return Subtract(Intersection(box, sphere), Union(cyl1, cyl2, cyl3))


def composition_demo():
    """Scene.swift:178-211."""
    ground = ColoredLambertian((0.5, 0.5, 0.5))
    return Scene(CameraConfig(vertical_fov=90, look_from=(3, 3, 3), look_at=(0, 0, 0)), [
        Sphere((0, -501, -1), 500.0, ground),
        _composition_object(),
    ])


def quads(seed=9):
    """Scene.swift:213-257 (the app's default scene)."""
    left = ColoredLambertian((1.0, 0.2, 0.2))
    back = ColoredEmissive((4.0, 4.0, 4.0))
    right = ColoredLambertian((0.2, 0.2, 1.0))
    upper = ColoredLambertian((1.0, 0.5, 0.0))
    lower = ColoredLambertian((0.2, 0.8, 0.8))
    rng = WyRand(seed)
    return Scene(CameraConfig(vertical_fov=80, look_from=(0, 0, 9), background=abi.background_lighting_none), [
        Quad((-3, -2, 5), (0, 0, -4), (0, 4, 0), left),
        Quad((-2, -2, 0), (4, 0, 0), (0, 4, 0), back),
        Quad((3, -2, 1), (0, 0, 4), (0, 4, 0), right),
        Quad((-2, 3, 1), (4, 0, 0), (0, 0, 4), upper),
        Quad((-2, -3, 5), (4, 0, 0), (0, 0, -4), lower),
        Sphere((0, 0, 2), 1, PerlinNoiseLambertian(
            PerlinNoiseTexture.generate((0, 0, 0), (1, 1, 1), frequency=4, turbulence=5, rng=rng))),
        ConstantDensityVolume(Cuboid(T.translate(-3, -3, 0), (6, 3, 5), ColoredIsotropic((1.0, 1.0, 0.0))), 0.5),
    ])


def world1(seed=21):
    """RayTracing/main.swift:11-104 (the CLI's default world and camera) expressed with the
    Metal-side types: the lens is an Intersection of two glass spheres whose centres carry
    the Kit's Transformed(...) placement; image textures are procedural; the barrels keep
    their three placements."""
    ground = ColoredLambertian((0.5, 0.5, 0.5))
    glass = Dielectric(1.5)
    place = T.translate(-0.5, 0, 0) * T.rotation_deg(-45, "y")
    lens = Intersection(Sphere(place.act((0, 1, -9.915)), 10.0, glass), Sphere(place.act((0, 1, 9.915)), 10.0, glass))
    earth = TexturedLambertian(ImageTexture.procedural("earthmap.jpg", 256, 128, seed))
    earth_tr = T.translate(-4, 1, 0) * T.rotation_deg(23, "z") * T.rotation_deg(190, "y")
    barrel = dict(bottom=TexturedLambertian(ImageTexture.procedural("barrel-bottom", 64, 64, seed + 1)),
                  top=TexturedLambertian(ImageTexture.procedural("barrel-top", 64, 64, seed + 2)),
                  side=TexturedLambertian(ImageTexture.procedural("barrel-side", 256, 128, seed + 3)))
    objs = [
        Sphere((0, -1000, 0), 1000, ground),
        lens,
        Sphere(radius=1.0, material=earth, transform=earth_tr),
        Quad((-5, 10, 5), (10, 0, 0), (0, 0, -10), ColoredEmissive((4.0, 4.0, 4.0))),
        Sphere((4, 1, 0), 1.0, ColoredMetal((0.7, 0.6, 0.5), 0.0)),
        Cylinder(T.translate(0, 0, 3), 0.5, 2, **barrel),
        Cylinder(T.translate(-1.2, 0, 2.7) * T.rotation_deg(115, "y"), 0.5, 2, **barrel),
        Cylinder(T.translate(-1.7, 0.5, 3.5) * T.rotation_deg(-30, "y") * T.rotation_deg(-90, "z")
                 * T.rotation_deg(90, "y"), 0.5, 2, **barrel),
        Quad((-1, 0, -2), (0, 0.8, 0), (5.5, 0, 0), ColoredLambertian((0.7, 0.8, 0))),
        Quad((-1, 0, -2), (0, 0, 0.8), (5.5, 0, 0), ColoredLambertian((0.0, 1.0, 0.8))),
    ]
    cam = CameraConfig(vertical_fov=20, look_from=(-10, 6, 8), look_at=(0, 0, 0), defocus_angle=0.6,
                       focus_distance=10, background=abi.background_lighting_none)
    return Scene(cam, objs)


# ---------------------------------------------------------------------------- synthetic workloads
def random_spheres(seed=1):
    """The 'random spheres' workload: ground sphere, a 22x22 grid of small spheres with
    jittered centres (80 % lambertian, 15 % metal, 5 % glass) and three large spheres;
    sky background, 0.6 degree defocus at distance 10."""
    rng = WyRand(seed)
    objs = [Sphere((0, -1000, 0), 1000, ColoredLambertian((0.5, 0.5, 0.5)))]
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose = rng.uniform()
            c = (a + 0.9 * rng.uniform(), 0.2, b + 0.9 * rng.uniform())
            if math.dist(c, (4, 0.2, 0)) <= 0.9:
                continue
            if choose < 0.8:
                alb = [rng.uniform() * rng.uniform() for _ in range(3)]
                mat = ColoredLambertian(alb)
            elif choose < 0.95:
                alb = [rng.uniform(0.5, 1) for _ in range(3)]
                mat = ColoredMetal(alb, rng.uniform(0, 0.5))
            else:
                mat = Dielectric(1.5)
            objs.append(Sphere(c, 0.2, mat))
    objs.append(Sphere((0, 1, 0), 1.0, Dielectric(1.5)))
    objs.append(Sphere((-4, 1, 0), 1.0, ColoredLambertian((0.4, 0.2, 0.1))))
    objs.append(Sphere((4, 1, 0), 1.0, ColoredMetal((0.7, 0.6, 0.5), 0.0)))
    cam = CameraConfig(vertical_fov=20, look_from=(13, 2, 3), look_at=(0, 0, 0), defocus_angle=0.6,
                       focus_distance=10.0, background=abi.background_lighting_sky)
    return Scene(cam, objs)


def bouncing_spheres(seed=1):
    """The random-spheres workload with its lambertian spheres in motion: each moves
    straight up by up to half a unit over the exposure, a few metal ones also spin about
    a tilted axis while they drift."""
    base = random_spheres(seed)
    rng = WyRand(seed ^ 0xB0B)
    objs = []
    for k, obj in enumerate(base.objects):
        small = isinstance(obj, Sphere) and obj.radius < 0.5
        if small and isinstance(obj.material, ColoredLambertian):
            objs.append(MotionBlur(Transform.translate(0, 0.5 * rng.uniform(), 0), obj))
        elif small and isinstance(obj.material, ColoredMetal) and k % 3 == 0:
            spin = Transform(Quat.from_angle_axis(rng.uniform(0.2, 1.2), normalize(vec3(0.3, 1.0, 0.2))),
                             (0.3 * rng.uniform(), 0, 0.3 * rng.uniform()))
            objs.append(MotionBlur(spin, obj))
        else:
            objs.append(obj)
    return Scene(base.camera, objs)


def kit_world1(seed=3, texture_dir=None):
    """makeWorld1 of the command-line renderer (RayTracing/main.swift:11-104) with the Kit's
    own textures: an earth-like image on a tilted sphere, three barrels whose caps and
    side are windows of one 512x512 image (bilinear, the side window mirrored in v), a
    glass lens as an intersection of two spheres, a ceiling light, a mirror ball and two
    coloured quads.  `texture_dir` (RayTracing/textures) supplies the bundled JPEGs; procedural
    images of the same sizes stand in without it."""
    earth = ImageTexture.bundled("earthmap.jpg", 1024, 512, seed, texture_dir)
    barrel = ImageTexture.bundled("barrel.jpg", 512, 512, seed + 1, texture_dir)
    glass = Dielectric(1.5)
    lens = Transformed(Transform.translate(-0.5, 0, 0) * Transform.rotation_deg(-45, "y"),
                       Intersection(Sphere((0, 1, -9.915), 10.0, glass), Sphere((0, 1, 9.915), 10.0, glass)))
    top = KitLambertian(BilinearImageTexture(barrel, 128, 320, 192, 192))
    bottom = KitLambertian(BilinearImageTexture(barrel, 320, 320, 192, 192))
    side = KitLambertian(BilinearImageTexture(barrel, 0, 320, 512, -320))
    T = Transform
    barrels = [Cylinder(tr, 0.5, 2.0, bottom=bottom, top=top, side=side) for tr in (
        T.translate(0, 0, 3),
        T.translate(-1.2, 0, 2.7) * T.rotation_deg(115, "y"),
        T.translate(-1.7, 0.5, 3.5) * T.rotation_deg(-30, "y") * T.rotation_deg(-90, "z") * T.rotation_deg(90, "y"))]
    objs = [
        Sphere((0, -1000, 0), 1000, ColoredLambertian((0.5, 0.5, 0.5))),
        lens,
        Sphere(radius=1.0, material=KitLambertian(BilinearImageTexture(earth)),
               transform=T.translate(-4, 1, 0) * T.rotation_deg(23, "z") * T.rotation_deg(190, "y")),
        Quad((-5, 10, 5), (10, 0, 0), (0, 0, -10), ColoredEmissive((4, 4, 4))),
        Sphere((4, 1, 0), 1.0, ColoredMetal((0.7, 0.6, 0.5), 0.0)),
    ] + barrels + [
        Quad((-1, 0, -2), (0, 0.8, 0), (5.5, 0, 0), ColoredLambertian((0.7, 0.8, 0.0))),
        Quad((-1, 0, -2), (0, 0, 0.8), (5.5, 0, 0), ColoredLambertian((0.0, 1.0, 0.8))),
    ]
    cam = CameraConfig(vertical_fov=20, look_from=(-10, 6, 8), look_at=(0, 0, 0), defocus_angle=0.6,
                       focus_distance=10, background=abi.background_lighting_none)
    return Scene(cam, objs, convention=abi.RT_CONVENTION_KIT)


def kit_world2(seed=4):
    """makeWorld2 (RayTracing/main.swift:106-125): a noise-textured ground and a marble
    sphere, plus a checkered block lit by a textured light so that every Kit texture kind
    is on screen."""
    rng = WyRand(seed)
    noise_tex = NoiseTexture(KitPerlinNoise(rng), 1.0)
    marble = Marble(KitPerlinNoise(rng), 1.0)
    checker = CheckerTexture(0.32, (0.2, 0.3, 0.1), CheckerTexture(0.08, (0.9, 0.9, 0.9), marble))
    objs = [
        Sphere((0, -1000, 0), 1000, KitLambertian(noise_tex)),
        Sphere((0, 2, 0), 2, KitLambertian(marble)),
        Cuboid(Transform.translate(2.5, 0, 1.5), (1.2, 1.2, 1.2), KitLambertian(checker)),
        Quad((-2, 6, -2), (4, 0, 0), (0, 0, 4), KitEmissive(CheckerTexture(1.0, (6, 6, 6), (6, 3, 1)))),
    ]
    cam = CameraConfig(vertical_fov=20, look_from=(13, 2, 3), look_at=(0, 0, 0), background=abi.background_lighting_sky)
    return Scene(cam, objs, convention=abi.RT_CONVENTION_KIT)


def cornell_smoke():
    """Cornell box with two smoke-filled boxes: six quads spanning 555^3, an emissive
    ceiling quad, two rotated cuboids wrapped in constant-density volumes."""
    red = ColoredLambertian((0.65, 0.05, 0.05))
    white = ColoredLambertian((0.73, 0.73, 0.73))
    green = ColoredLambertian((0.12, 0.45, 0.15))
    light = ColoredEmissive((7, 7, 7))
    objs = [
        Quad((555, 0, 0), (0, 555, 0), (0, 0, 555), green),
        Quad((0, 0, 0), (0, 555, 0), (0, 0, 555), red),
        Quad((113, 554, 127), (330, 0, 0), (0, 0, 305), light),
        Quad((0, 555, 0), (555, 0, 0), (0, 0, 555), white),
        Quad((0, 0, 0), (555, 0, 0), (0, 0, 555), white),
        Quad((0, 0, 555), (555, 0, 0), (0, 555, 0), white),
        ConstantDensityVolume(Cuboid(T.translate(265, 0, 295) * T.rotation_deg(15, "y"), (165, 330, 165),
                                     ColoredIsotropic((0, 0, 0))), 0.01),
        ConstantDensityVolume(Cuboid(T.translate(130, 0, 65) * T.rotation_deg(-18, "y"), (165, 165, 165),
                                     ColoredIsotropic((1, 1, 1))), 0.01),
    ]
    cam = CameraConfig(vertical_fov=40, look_from=(278, 278, -800), look_at=(278, 278, 0),
                       background=abi.background_lighting_none)
    return Scene(cam, objs)


def csg_textured(seed=7):
    """CSG Subtract / Intersection / Union, cylinders and cuboids under rigid transforms,
    one image-textured sphere and one Perlin sphere: the reference's compositionDemo
    object and pencilInGlass glass on a ground sphere, with a grid of 64 extra transformed
    cuboids and cylinders."""
    rng = WyRand(seed)
    objs = [Sphere((0, -501, -1), 500.0, ColoredLambertian((0.5, 0.5, 0.5))), _composition_object()]
    # the glass, scaled into the scene by placing it to the side
    objs.append(Subtract(
        Cylinder(T.translate(4, -1, -3), 1.0, 3.0, Dielectric(1.5)),
        Cylinder(T.translate(4, -0.75, -3), 0.85, 3.0, Dielectric(1.5))))
    palette = [ColoredLambertian((0.8, 0.3, 0.3)), ColoredLambertian((0.3, 0.8, 0.3)), ColoredLambertian((0.3, 0.3, 0.8)),
               ColoredMetal((0.8, 0.8, 0.8), 0.05), ColoredMetal((0.9, 0.7, 0.3), 0.3), Dielectric(1.5),
               ColoredLambertian((0.9, 0.9, 0.2)), ColoredLambertian((0.2, 0.9, 0.9))]
    for i in range(8):
        for j in range(8):
            x = -10.5 + 3.0 * i + rng.uniform(-0.3, 0.3)
            z = -14.0 + 2.0 * j + rng.uniform(-0.3, 0.3)
            if abs(x) < 2.5 and abs(z) < 2.5:
                z -= 6.0
            mat = palette[(i * 8 + j) % len(palette)]
            tr = (T.translate(x, -1.0, z) * T.rotation_deg(rng.uniform(0, 360), "y")
                  * T.rotation_deg(rng.uniform(-20, 20), "x"))
            if (i + j) % 2 == 0:
                s = (rng.uniform(0.4, 0.9), rng.uniform(0.4, 1.4), rng.uniform(0.4, 0.9))
                objs.append(Cuboid(tr, s, mat))
            else:
                objs.append(Cylinder(tr, rng.uniform(0.25, 0.5), rng.uniform(0.5, 1.5), mat))
    objs.append(Sphere(radius=0.9, transform=T.translate(-3.5, -0.1, 1.0) * T.rotation_deg(30, "y"),
                       material=TexturedLambertian(ImageTexture.procedural("checker-gradient", 512, 256, seed))))
    objs.append(Sphere((3.2, -0.2, 1.5), 0.8, PerlinNoiseLambertian(
        PerlinNoiseTexture.generate((0.1, 0.1, 0.1), (1, 1, 1), frequency=4, turbulence=5, rng=WyRand(11)))))
    cam = CameraConfig(vertical_fov=50, look_from=(5, 4, 7), look_at=(0, -0.3, -1),
                       background=abi.background_lighting_sky)
    return Scene(cam, objs)


class SphereCloud:
    """n random spheres as arrays (a scene of a million Python objects would be slow to
    build): centres uniform in [-100,100]^3, radius uniform in [0.05,0.25], material
    i mod 64 from a 64-entry palette (70 % lambertian, 20 % metal, 10 % glass)."""

    def __init__(self, n=1_000_000, seed=42):
        self.n = n
        r = u64_to_unit(wyrand_block(seed, 4 * n)).reshape(n, 4)
        self.centers = (r[:, :3] * 200.0 - 100.0).astype(np.float32)
        self.radii = (0.05 + 0.2 * r[:, 3]).astype(np.float32)
        rng = WyRand(seed ^ 0x5EED)
        self.palette = []
        for k in range(64):
            u = k / 64.0
            if u < 0.7:
                self.palette.append(ColoredLambertian([rng.uniform() * rng.uniform() for _ in range(3)]))
            elif u < 0.9:
                self.palette.append(ColoredMetal([rng.uniform(0.5, 1) for _ in range(3)], rng.uniform(0, 0.5)))
            else:
                self.palette.append(Dielectric(1.5))
        self.camera = CameraConfig(vertical_fov=40, look_from=(0, 0, -260), look_at=(0, 0, 0),
                                   background=abi.background_lighting_sky)

    def encode(self, sink, dedup=True):
        from .scene import SceneEncoder
        enc = SceneEncoder(sink, dedup=dedup)
        mats = [enc.encode(m) for m in self.palette]
        mat_ids = np.array(mats, dtype=np.uint32)[np.arange(self.n) % 64]
        sink.add_spheres_bulk(self.centers, self.radii, mat_ids)
        return enc


WORKLOADS = {
    # name: (scene factory, width, height, spp, depth, seed)
    "cfg1_random_spheres_400x225": (random_spheres, 400, 225, 16, 50, 1),
    "cfg2_random_spheres_1080p": (random_spheres, 1920, 1080, 1024, 50, 1),
    "cfg3_cornell_smoke_1080p": (cornell_smoke, 1920, 1080, 4096, 50, 3),
    "cfg4_csg_textured_1080p": (csg_textured, 1920, 1080, 2048, 50, 7),
    "cfg5_sphere_cloud_4k": (SphereCloud, 3840, 2160, 8192, 50, 42),
    "f2_bouncing_spheres_1080p": (bouncing_spheres, 1920, 1080, 256, 50, 1),
    "f3_kit_world1_1080p": (kit_world1, 1920, 1080, 100, 50, 3),
    "f3_kit_world2_1080p": (kit_world2, 1920, 1080, 100, 50, 4),
}


def make_workload(name):
    """(scene, width, height, spp, depth, seed) of a named measurement workload."""
    make, w, h, spp, depth, seed = WORKLOADS[name]
    if make is SphereCloud:
        scene = make(1_000_000, seed)
    elif make is cornell_smoke:
        scene = make()
    else:
        scene = make(seed)
    return scene, w, h, spp, depth, seed


DEMO_SCENES = {
    "simpleBalls": simple_balls,
    "singleCylinder": single_cylinder,
    "simpleCylinders": simple_cylinders,
    "texturedCylinders": textured_cylinders,
    "pencilInGlass": pencil_in_glass,
    "compositionDemo": composition_demo,
    "quads": quads,
    "world1": world1,
}
