"""Host-side scene API: the reference's Swift scene description, mirrored type for type.

Reference: MetalRayTracer/Renderable.swift (Transform, Renderable), Primitives.swift
(Sphere, Cylinder, Cuboid, Quad), CSG.swift (Subtract, Union, Intersection),
ConstantDensityVolume.swift, Material.swift (nine material types), Texture.swift
(ImageTexture, PerlinNoiseTexture.generate), Config.swift (CameraConfig, RenderConfig),
Scene.swift (Scene).  `as_impl` plays the role of `asImpl(&encoder)`: it writes the C
records of include/raytracer_amd.h into an encoder, which forwards them to the native
library (or, in the tests, to the CPU oracle).
"""
import ctypes as C
import math

import numpy as np

from . import abi
from .simd import Quat, vec3, cross, dot, length, normalize, f32

AXES = {"x": (1.0, 0.0, 0.0), "y": (0.0, 1.0, 0.0), "z": (0.0, 0.0, 1.0)}


def _f3(v):
    return abi.F3(float(v[0]), float(v[1]), float(v[2]))


# ---------------------------------------------------------------------------- RNG
class WyRand:
    """RayTracingKit/PRNG.swift:10-22.  Used to seed every synthetic scene."""

    MASK = (1 << 64) - 1

    def __init__(self, seed):
        self.state = seed & self.MASK

    def next(self):
        self.state = (self.state + 0xA0761D6478BD642F) & self.MASK
        m = self.state * (self.state ^ 0xE7037ED1A0B428DB)
        return ((m >> 64) ^ m) & self.MASK

    def uniform(self, lo=0.0, hi=1.0):
        """53-bit uniform in [lo, hi)."""
        return lo + (hi - lo) * ((self.next() >> 11) * (1.0 / 9007199254740992.0))

    def shuffle(self, seq):
        """Fisher-Yates, in place."""
        for i in range(len(seq) - 1, 0, -1):
            j = self.next() % (i + 1)
            seq[i], seq[j] = seq[j], seq[i]


def wyrand_block(seed, n):
    """First n WyRand outputs of `seed` as a numpy uint64 array (vectorised 64x64->128)."""
    k = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + k * np.uint64(0xA0761D6478BD642F)
        t = s ^ np.uint64(0xE7037ED1A0B428DB)
        m32 = np.uint64(0xFFFFFFFF)
        a0, a1 = s & m32, s >> np.uint64(32)
        b0, b1 = t & m32, t >> np.uint64(32)
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = (p00 >> np.uint64(32)) + (p01 & m32) + (p10 & m32)
        lo = (p00 & m32) | ((mid & m32) << np.uint64(32))
        hi = p11 + (p01 >> np.uint64(32)) + (p10 >> np.uint64(32)) + (mid >> np.uint64(32))
    return hi ^ lo


def u64_to_unit(a):
    return (a >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


# ---------------------------------------------------------------------------- transform
class Transform:
    """Renderable.swift:34-70: rigid transform = unit quaternion + translation."""

    def __init__(self, rotation=None, translation=(0.0, 0.0, 0.0)):
        self.rotation = rotation if rotation is not None else Quat.identity()
        self.translation = vec3(translation)
        self.matrix = None      # explicit 3x3 columns; overrides the quaternion when set

    @staticmethod
    def identity():
        return Transform()

    @staticmethod
    def rotation_deg(degrees, axis):
        ax = AXES[axis] if isinstance(axis, str) else axis
        return Transform(rotation=Quat.from_angle_axis(f32(degrees) * f32(math.pi) / f32(180), ax))

    @staticmethod
    def translate(x, y=None, z=None):
        return Transform(translation=vec3(x, y, z))

    def __mul__(self, rhs):
        return Transform(self.rotation * rhs.rotation, self.rotation.act(rhs.translation) + self.translation)

    @property
    def inverse(self):
        r = self.rotation.inverse
        return Transform(r, -r.act(self.translation))

    def act(self, p):
        return self.rotation.act(p) + self.translation

    @property
    def as_impl(self):
        t = abi.RtTransform()
        cols = self.matrix if self.matrix is not None else self.rotation.matrix_columns()
        for c in range(3):
            t.rotation.columns[c] = _f3(cols[c])
        t.translation = _f3(self.translation)
        return t


def raw_transform(columns, translation=(0, 0, 0)):
    """A transform given directly as a 3x3 matrix (its columns): what the reference's tests
    build with simd_matrix3x3(...)."""
    t = Transform(translation=translation)
    t.matrix = np.asarray(columns, dtype=np.float32).reshape(3, 3)
    return t


# ---------------------------------------------------------------------------- textures
class ImageTexture:
    """Texture.swift:57-59.  The reference names a bundled JPEG; no JPEG decoder is in
    scope here, so a texture carries its 8-bit RGB pixels (rows top to bottom).  A name
    without pixels gives the loader's 1x1 magenta placeholder (Texture.swift:43-54)."""

    def __init__(self, name, pixels=None):
        self.name = name
        if pixels is None:
            pixels = np.array([[[255, 0, 255]]], dtype=np.uint8)
        self.pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        assert self.pixels.ndim == 3 and self.pixels.shape[2] == 3

    @staticmethod
    def load(name, bundle_dirs=()):
        """TextureLoader.doLoad, Texture.swift:29-41: the file `name` from the first directory
        that has it, decoded to 8-bit RGB (the reference decodes with MTKTextureLoader /
        CoreGraphics, here PIL does); the placeholder when no directory has it or it does not
        decode.  Rows are kept top to bottom; the bottom-left origin the loader asks for is
        what the samplers apply."""
        import os
        for d in bundle_dirs:
            path = os.path.join(d, name)
            if os.path.isfile(path):
                try:
                    from PIL import Image
                    with Image.open(path) as im:
                        return ImageTexture(name, np.asarray(im.convert("RGB"), dtype=np.uint8))
                except Exception:      # the reference falls back to the placeholder on any loader error (:38-40)
                    break
        return ImageTexture(name)

    @staticmethod
    def bundled(name, width, height, seed, texture_dir=None):
        """The bundled image `name` when `texture_dir` holds it, a procedural stand-in of the
        given size otherwise (the reference's JPEGs do not travel with this repository)."""
        if texture_dir:
            t = ImageTexture.load(name, [texture_dir])
            if t.pixels.shape[:2] != (1, 1):
                return t
        return ImageTexture.procedural(name, width, height, seed)

    @staticmethod
    def procedural(name, width, height, seed):
        """Checker-gradient RGB8 image from WyRand(seed): synthetic stand-in for the JPEGs."""
        r = wyrand_block(seed, 64)
        cols = ((r[:24] >> np.uint64(24)) & np.uint64(0xFF)).astype(np.uint8).reshape(8, 3)
        y, x = np.mgrid[0:height, 0:width]
        cell = ((x * 16 // width) + (y * 8 // height)) % 8
        img = cols[cell].astype(np.float32)
        grad = 0.5 + 0.5 * (x / max(1, width - 1))
        img = np.clip(img * grad[..., None], 0, 255)
        return ImageTexture(name, img.astype(np.uint8))

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, o):
        return isinstance(o, ImageTexture) and o.name == self.name


class PerlinNoiseTexture:
    """Materials.h:38-45 + Texture.swift:62-104 (generate)."""

    def __init__(self, colors, frequency, turbulence, vectors, permutations):
        self.colors = np.asarray(colors, dtype=np.float32).reshape(2, 3)
        self.frequency = float(frequency)
        self.turbulence = int(turbulence)
        self.vectors = np.asarray(vectors, dtype=np.float32).reshape(256, 3)
        self.permutations = np.asarray(permutations, dtype=np.uint8).reshape(3, 256)

    @staticmethod
    def generate(color0, color1, frequency=1.0, turbulence=0, rng=None):
        """Texture.swift:63-104: 256 random vectors inside the unit ball (NOT normalised)
        and three shuffled permutation tables.  The reference draws from the system
        generator; here the caller's seeded WyRand."""
        rng = rng or WyRand(0)
        vectors = []
        while len(vectors) < 256:
            v = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)]
            if v[0] * v[0] + v[1] * v[1] + v[2] * v[2] <= 1:
                vectors.append(v)
        perms = []
        for _ in range(3):
            p = list(range(256))
            rng.shuffle(p)
            perms.append(p)
        return PerlinNoiseTexture([color0, color1], frequency, turbulence, vectors, perms)

    @property
    def as_impl(self):
        t = abi.RtPerlinNoiseTexture()
        for i in range(2):
            t.colors[i] = _f3(self.colors[i])
        t.frequency = self.frequency
        t.turbulence = self.turbulence
        padded = np.zeros((256, 4), dtype=np.float32)
        padded[:, :3] = self.vectors
        C.memmove(t.vectors, padded.tobytes(), 256 * 16)
        C.memmove(t.permutations, self.permutations.tobytes(), 768)
        return t


# ---------------------------------------------------------------------------- RayTracingKit textures
class KitTexture:
    """RayTracingKit/Texture.swift:9-11.  `encode` registers the texture (children first)
    with the sink and returns its handle."""

    def encode(self, encoder):
        key = id(self)
        if key not in encoder._kit:      # the object is kept with its handle: ids of dead objects get reused
            encoder._kit[key] = (encoder.sink.add_kit_texture(self.record(encoder)), self)
        return encoder._kit[key][0]


class SolidColor(KitTexture):
    """Texture.swift:13-25."""

    def __init__(self, albedo):
        self.albedo = vec3(albedo)

    def record(self, encoder):
        t = abi.RtKitTexture()
        t.kind = abi.RT_KIT_TEXTURE_SOLID
        t.color = (C.c_float * 3)(*[float(c) for c in self.albedo])
        return t


class CheckerTexture(KitTexture):
    """Texture.swift:27-51: colours or textures for the even and odd cells."""

    def __init__(self, scale, even, odd):
        self.scale = float(scale)
        self.even = even if isinstance(even, KitTexture) else SolidColor(even)
        self.odd = odd if isinstance(odd, KitTexture) else SolidColor(odd)

    def record(self, encoder):
        t = abi.RtKitTexture()
        t.kind = abi.RT_KIT_TEXTURE_CHECKER
        t.scale = self.scale
        t.even = self.even.encode(encoder)
        t.odd = self.odd.encode(encoder)
        return t


class BilinearImageTexture(KitTexture):
    """ImageTexture of the Kit, Texture.swift:53-120: a window of an image (negative width /
    height run towards smaller indices), bilinear, wrapped or clamped.  `image` is an
    ImageTexture (name + RGB8 rows, top to bottom)."""

    def __init__(self, image, origin_x=None, origin_y=None, width=None, height=None, wrap_x=True, wrap_y=True):
        h, w, _ = image.pixels.shape
        self.image = image
        if origin_x is None:          # init(image:), Texture.swift:64-66
            origin_x, origin_y, width, height = 0, h - 1, w, -h
        self.origin_x, self.origin_y, self.width, self.height = int(origin_x), int(origin_y), int(width), int(height)
        self.wrap_x, self.wrap_y = bool(wrap_x), bool(wrap_y)

    def record(self, encoder):
        t = abi.RtKitTexture()
        t.kind = abi.RT_KIT_TEXTURE_IMAGE
        t.image = encoder.load_texture(self.image)
        t.origin_x, t.origin_y, t.width, t.height = self.origin_x, self.origin_y, self.width, self.height
        t.wrap_x, t.wrap_y = int(self.wrap_x), int(self.wrap_y)
        return t


class KitPerlinNoise:
    """PerlinNoise, Texture.swift:166-182: 256 vectors with components in [-1, 1] and three
    shuffled byte tables (the caller's seeded generator stands in for the system one)."""

    def __init__(self, rng):
        self.vectors = np.array([[rng.uniform(-1, 1) for _ in range(3)] for _ in range(256)], dtype=np.float32)
        self.perms = []
        for _ in range(3):
            p = list(range(256))
            rng.shuffle(p)
            self.perms.append(np.array(p, dtype=np.uint8))

    def encode(self, encoder):
        key = id(self)
        if key not in encoder._kit:
            rec = abi.RtKitPerlin()
            C.memmove(rec.vectors, self.vectors.tobytes(), 256 * 12)
            for name, p in zip(("perm_x", "perm_y", "perm_z"), self.perms):
                C.memmove(getattr(rec, name), p.tobytes(), 256)
            encoder._kit[key] = (encoder.sink.add_kit_perlin(rec), self)
        return encoder._kit[key][0]


class NoiseTexture(KitTexture):
    """Texture.swift:136-149."""
    KIND = abi.RT_KIT_TEXTURE_NOISE

    def __init__(self, noise, scale):
        self.noise = noise
        self.scale = float(scale)

    def record(self, encoder):
        t = abi.RtKitTexture()
        t.kind = self.KIND
        t.scale = self.scale
        t.perlin = self.noise.encode(encoder)
        return t


class Marble(NoiseTexture):
    """Texture.swift:151-164."""
    KIND = abi.RT_KIT_TEXTURE_MARBLE


# ---------------------------------------------------------------------------- materials
class Material:
    """Material.swift:8-19."""

    def as_impl(self, encoder):
        raise NotImplementedError


class ColoredLambertian(Material):
    def __init__(self, albedo):
        self.albedo = vec3(albedo)

    def as_impl(self, encoder):
        return abi.RtColoredLambertianMaterial(abi.material_kind_lambertian_colored, _f3(self.albedo))


class TexturedLambertian(Material):
    def __init__(self, albedo):
        self.albedo = albedo

    def as_impl(self, encoder):
        tid = encoder.load_texture(self.albedo)
        return abi.RtTexturedLambertianMaterial(abi.material_kind_lambertian_textured, abi.RtImageTexture(tid))


class KitLambertian(Material):
    """Lambertian(texture:), RayTracingKit/Material.swift:17-41."""

    def __init__(self, texture):
        self.texture = texture

    def as_impl(self, encoder):
        return abi.RtKitTexturedMaterial(abi.material_kind_lambertian_kit_texture, self.texture.encode(encoder))


class KitEmissive(Material):
    """Emissive(texture:), RayTracingKit/Material.swift:75-93."""

    def __init__(self, texture):
        self.texture = texture

    def as_impl(self, encoder):
        return abi.RtKitTexturedMaterial(abi.material_kind_emissive_kit_texture, self.texture.encode(encoder))


class PerlinNoiseLambertian(Material):
    def __init__(self, albedo):
        self.albedo = albedo

    def as_impl(self, encoder):
        return abi.RtPerlinNoiseLambertianMaterial(abi.material_kind_lambertian_perlin_noise, self.albedo.as_impl)


class ColoredMetal(Material):
    def __init__(self, albedo, fuzz):
        assert 0 <= fuzz <= 1
        self.albedo = vec3(albedo)
        self.fuzz = float(fuzz)

    def as_impl(self, encoder):
        return abi.RtColoredMetalMaterial(abi.material_kind_metal_colored, _f3(self.albedo), self.fuzz)


class TexturedMetal(Material):
    def __init__(self, albedo, fuzz):
        assert 0 <= fuzz <= 1
        self.albedo = albedo
        self.fuzz = float(fuzz)

    def as_impl(self, encoder):
        tid = encoder.load_texture(self.albedo)
        return abi.RtTexturedMetalMaterial(abi.material_kind_metal_textured, abi.RtImageTexture(tid), self.fuzz)


class PerlinNoiseMetal(Material):
    def __init__(self, albedo, fuzz):
        assert 0 <= fuzz <= 1
        self.albedo = albedo
        self.fuzz = float(fuzz)

    def as_impl(self, encoder):
        return abi.RtPerlinNoiseMetalMaterial(abi.material_kind_metal_perlin_noise, self.albedo.as_impl, self.fuzz)


class Dielectric(Material):
    def __init__(self, refraction_index):
        assert refraction_index > 0
        self.refraction_index = float(refraction_index)

    def as_impl(self, encoder):
        return abi.RtDielectricMaterial(abi.material_kind_dielectric, self.refraction_index)


class ColoredEmissive(Material):
    def __init__(self, albedo):
        self.albedo = vec3(albedo)

    def as_impl(self, encoder):
        return abi.RtColoredEmissiveMaterial(abi.material_kind_emissive_colored, _f3(self.albedo))


class ColoredIsotropic(Material):
    def __init__(self, albedo):
        self.albedo = vec3(albedo)

    def as_impl(self, encoder):
        return abi.RtColoredIsotropicMaterial(abi.material_kind_isotropic_colored, _f3(self.albedo))


class MaterialOffset(Material):
    """A raw material handle, for callers that manage the material table themselves (the
    reference's tests pass bare offsets such as 42)."""

    def __init__(self, offset):
        self.offset = int(offset)


# ---------------------------------------------------------------------------- encoder
class SceneEncoder:
    """MaterialEncoder + RenderableGrouper (Material.swift:33-62, SceneBuffers.swift:113-131)
    in front of a sink that receives the C records.  Identical material records share one
    handle (the reference writes one record per use)."""

    def __init__(self, sink, dedup=True):
        self.sink = sink
        self.dedup = dedup
        self._materials = {}
        self._textures = {}
        self._kit = {}           # RayTracingKit textures and noise tables already registered

    def load_texture(self, tex):
        if tex.name not in self._textures:
            h, w, _ = tex.pixels.shape
            self._textures[tex.name] = self.sink.add_texture_rgb8(w, h, tex.pixels)
        return self._textures[tex.name]

    def encode(self, material):
        if isinstance(material, MaterialOffset):
            return material.offset
        rec = material.as_impl(self)
        key = bytes(rec)
        if self.dedup and key in self._materials:
            return self._materials[key]
        mid = self.sink.add_material(rec)
        self._materials[key] = mid
        return mid


# ---------------------------------------------------------------------------- renderables
class AABB:
    """MTLAxisAlignedBoundingBox helpers, Renderable.swift:105-155."""

    def __init__(self, lo=None, hi=None):
        self.min = np.full(3, np.inf, dtype=np.float32) if lo is None else vec3(lo)
        self.max = np.full(3, -np.inf, dtype=np.float32) if hi is None else vec3(hi)

    @staticmethod
    def unlimited():
        return AABB(np.full(3, -np.inf), np.full(3, np.inf))

    def add(self, p):
        self.min = np.minimum(self.min, vec3(p))
        self.max = np.maximum(self.max, vec3(p))

    def unite(self, b):
        self.add(b.min)
        self.add(b.max)

    def intersect(self, b):
        self.min = np.maximum(self.min, b.min)
        self.max = np.minimum(self.max, b.max)


class Renderable:
    """Renderable.swift:72-87."""

    def as_impl(self, encoder):
        """Writes the shape's records; returns the shape id."""
        raise NotImplementedError

    @property
    def bounding_box(self):
        raise NotImplementedError


class Sphere(Renderable):
    """Primitives.swift:12-43."""

    def __init__(self, center=None, radius=1.0, material=None, transform=None):
        self.transform = transform if transform is not None else Transform(translation=center)
        self.radius = float(radius)
        self.material = material

    @property
    def bounding_box(self):
        r = vec3(self.radius, self.radius, self.radius)
        c = self.transform.translation
        return AABB(c - r, c + r)

    def as_impl(self, encoder):
        m = encoder.encode(self.material)
        return encoder.sink.shape_sphere(abi.RtSphere(self.transform.as_impl, self.radius, m))


class Cylinder(Renderable):
    """Primitives.swift:49-146."""

    def __init__(self, transform=None, radius=1.0, height=1.0, material=None, bottom=None, top=None, side=None):
        self.transform = transform if transform is not None else Transform()
        self.radius = float(radius)
        self.height = float(height)
        self.bottom = bottom if bottom is not None else material
        self.top = top if top is not None else material
        self.side = side if side is not None else material

    @staticmethod
    def between(bottom_center, top_center, radius, v_right=(1, 0, 0), material=None, bottom=None, top=None, side=None):
        """Primitives.swift:85-110: axis from two points."""
        bc, tc = vec3(bottom_center), vec3(top_center)
        v = tc - bc
        height = length(v)
        v = v / height
        w = normalize(cross(vec3(v_right), v))
        q1 = Quat.from_to(v, (0, 1, 0))
        w1 = q1.act(w)
        q2 = Quat.from_to(w1, (0, 0, 1))
        q = (q2 * q1).inverse
        return Cylinder(Transform(q, bc), radius, float(height), material, bottom, top, side)

    @property
    def bounding_box(self):
        n = self.transform.rotation.act((0, 1, 0))
        rp = np.sqrt(np.maximum(0, 1.0 - n * n)).astype(np.float32) * f32(self.radius)
        bc = self.transform.translation
        tc = bc + f32(self.height) * n
        b = AABB()
        for p in (bc + rp, bc - rp, tc + rp, tc - rp):
            b.add(p)
        return b

    def as_impl(self, encoder):
        b = encoder.encode(self.bottom)
        t = encoder.encode(self.top)
        s = encoder.encode(self.side)
        return encoder.sink.shape_cylinder(abi.RtCylinder(self.transform.as_impl, self.radius, self.height, b, t, s))


class Cuboid(Renderable):
    """Primitives.swift:152-194.  `materials` (six, Face order left,right,bottom,top,back,
    front) is what the C record allows; the Swift type takes a single material."""

    def __init__(self, transform=None, size=(1, 1, 1), material=None, materials=None):
        self.transform = transform if transform is not None else Transform()
        self.size = vec3(size)
        self.materials = list(materials) if materials is not None else [material] * 6

    @property
    def bounding_box(self):
        b = AABB()
        for i in range(8):
            p = vec3(self.size[0] if i & 1 else 0, self.size[1] if i & 2 else 0, self.size[2] if i & 4 else 0)
            b.add(self.transform.act(p))
        return b

    def as_impl(self, encoder):
        mats = (C.c_size_t * 6)(*[encoder.encode(m) for m in self.materials])
        return encoder.sink.shape_cuboid(abi.RtCuboid(self.transform.as_impl, _f3(self.size), mats))


class Quad(Renderable):
    """Primitives.swift:202-275."""

    def __init__(self, origin=(0, 0, 0), u=(1, 0, 0), v=(0, 1, 0), material=None, transform=None):
        tr = transform if transform is not None else Transform()
        self.origin = tr.act(vec3(origin))
        self.u = tr.rotation.act(vec3(u))
        self.v = tr.rotation.act(vec3(v))
        self.material = material

    @property
    def bounding_box(self):
        b = AABB()
        for p in (self.origin, self.origin + self.u, self.origin + self.v, self.origin + self.u + self.v):
            b.add(p)
        return b

    def as_impl(self, encoder):
        m = encoder.encode(self.material)
        n = cross(self.u, self.v)
        len_sq = dot(n, n)
        w = n / len_sq
        normal = n / f32(math.sqrt(float(len_sq)))
        d = dot(self.origin, normal)
        return encoder.sink.shape_quad(abi.RtQuad(_f3(self.origin), _f3(self.u), _f3(self.v), _f3(w), _f3(normal),
                                                  float(d), m))


def _csg_operand(shape):
    """CSG operands and volume bases are primitives or CSG shapes (Impl/RenderableImpl.h:506-523,
    717-725): a moving or fog-filled operand has no meaning there, and the native library
    refuses it too."""
    if isinstance(shape, (MotionBlur, ConstantDensityVolume)):
        raise TypeError("%s cannot be a CSG operand or a volume base; wrap the whole shape instead" % type(shape).__name__)
    return shape


class Subtract(Renderable):
    """CSG.swift:17-35."""

    def __init__(self, lhs, rhs):
        self.lhs, self.rhs = _csg_operand(lhs), _csg_operand(rhs)

    @property
    def bounding_box(self):
        return self.lhs.bounding_box

    def as_impl(self, encoder):
        ids = [self.lhs.as_impl(encoder), self.rhs.as_impl(encoder)]
        return encoder.sink.shape_csg(abi.RT_CSG_SUBTRACT, ids)


class Union(Renderable):
    """CSG.swift:53-77."""

    def __init__(self, *items):
        self.items = tuple(_csg_operand(it) for it in items)

    @property
    def bounding_box(self):
        b = AABB()
        for it in self.items:
            b.unite(it.bounding_box)
        return b

    def as_impl(self, encoder):
        return encoder.sink.shape_csg(abi.RT_CSG_UNION, [it.as_impl(encoder) for it in self.items])


class Intersection(Renderable):
    """CSG.swift:95-119."""

    def __init__(self, *items):
        self.items = tuple(_csg_operand(it) for it in items)

    @property
    def bounding_box(self):
        b = AABB.unlimited()
        for it in self.items:
            b.intersect(it.bounding_box)
        return b

    def as_impl(self, encoder):
        return encoder.sink.shape_csg(abi.RT_CSG_INTERSECTION, [it.as_impl(encoder) for it in self.items])


class ConstantDensityVolume(Renderable):
    """ConstantDensityVolume.swift:17-33."""

    def __init__(self, base, density):
        self.base = _csg_operand(base)
        self.density = float(density)

    @property
    def bounding_box(self):
        return self.base.bounding_box

    def as_impl(self, encoder):
        return encoder.sink.shape_cdv(self.base.as_impl(encoder), self.density)


def Transformed(transform, base):
    """Transformed of the Kit (RayTracingKit/Hittable.swift:563-602) for a rigid transform
    that does not change with time: the transform is folded into the shape's own records
    (a rotated sphere keeps its texture orientation, a quad moves its vectors, a CSG shape
    or a volume moves its parts), so nothing new reaches the device."""
    if isinstance(base, Sphere):
        return Sphere(radius=base.radius, material=base.material, transform=transform * base.transform)
    if isinstance(base, Cylinder):
        return Cylinder(transform * base.transform, base.radius, base.height, bottom=base.bottom, top=base.top, side=base.side)
    if isinstance(base, Cuboid):
        return Cuboid(transform * base.transform, base.size, materials=base.materials)
    if isinstance(base, Quad):
        return Quad(base.origin, base.u, base.v, base.material, transform=transform)
    if isinstance(base, Subtract):
        return Subtract(Transformed(transform, base.lhs), Transformed(transform, base.rhs))
    if isinstance(base, Union):
        return Union(*[Transformed(transform, it) for it in base.items])
    if isinstance(base, Intersection):
        return Intersection(*[Transformed(transform, it) for it in base.items])
    if isinstance(base, ConstantDensityVolume):
        return ConstantDensityVolume(Transformed(transform, base.base), base.density)
    if isinstance(base, MotionBlur):
        # the motion happens in the outer frame: T (M^t x) = (T M T^-1)^t (T x)
        return MotionBlur(transform * base.transform * transform.inverse, Transformed(transform, base.base))
    raise TypeError("cannot transform %r" % (base,))


class MotionBlur(Renderable):
    """RayTracingKit/Hittable.swift:528-555: `base` seen through transform.pow(time) for the
    sample's time in [0, 1).  The quaternion is kept with w >= 0 like the Kit's
    (Transform.swift:198-206), so the motion takes the short way round."""

    def __init__(self, transform, base):
        self.transform = transform
        self.base = base

    def pow(self, t):
        """transform.pow(t), Transform.swift:82-84,243-250."""
        q = self.transform.rotation
        w, v = float(q.r), q.v.astype(np.float64)
        if w < 0:
            w, v = -w, -v
        if w >= 1.0:
            rot = Quat.identity()
        else:
            a = math.acos(w)
            k = math.sin(a * t) / math.sqrt(1 - w * w)
            rot = Quat(v[0] * k, v[1] * k, v[2] * k, math.cos(a * t))
        return Transform(rot, self.transform.translation.astype(np.float64) * t)

    @property
    def bounding_box(self):
        """A box around the base box's corners at 129 times, widened by the largest step
        between them: encloses the exact sweep (Transform.swift:93-154, which the native
        library restates for the BVH; rt_shape_bounds returns that one)."""
        base = self.base.bounding_box
        corners = [np.array([(base.max if i & 1 else base.min)[0], (base.max if i & 2 else base.min)[1],
                             (base.max if i & 4 else base.min)[2]], dtype=np.float64) for i in range(8)]
        b = AABB()
        step = 0.0
        prev = None
        for k in range(129):
            tr = self.pow(k / 128.0)
            pts = [tr.act(c) for c in corners]
            if prev is not None:
                step = max(step, max(float(np.linalg.norm(p - q)) for p, q in zip(pts, prev)))
            prev = pts
            for p in pts:
                b.add(p)
        b.min = b.min - np.float32(step)
        b.max = b.max + np.float32(step)
        return b

    def as_impl(self, encoder):
        q = self.transform.rotation
        w, v = float(q.r), [float(c) for c in q.v]
        if w < 0:
            w, v = -w, [-c for c in v]
        rec = abi.RtMotion((C.c_float * 4)(w, *v), (C.c_float * 3)(*[float(c) for c in self.transform.translation]))
        return encoder.sink.shape_motion(self.base.as_impl(encoder), rec)


# ---------------------------------------------------------------------------- config
class CameraConfig:
    """Config.swift:9-70: plain fields plus the orbit accessor the app's sliders drive."""

    def __init__(self, vertical_fov=90.0, look_from=(0, 0, 0), look_at=(0, 0, -1), up=(0, 1, 0),
                 defocus_angle=0.0, focus_distance=None, background=abi.background_lighting_sky):
        self.vertical_fov = float(vertical_fov)
        self.look_from = vec3(look_from)
        self.look_at = vec3(look_at)
        self.up = vec3(up)
        self.defocus_angle = float(defocus_angle)
        self.focus_distance = float(focus_distance) if focus_distance is not None else float(
            length(self.look_from - self.look_at))
        self.background = int(background)

    # Config.swift:43-59: the camera position as yaw / pitch (degrees) and distance around look_at
    @property
    def relative_position(self):
        v = (self.look_from - self.look_at).astype(np.float32)
        d = np.float32(np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
        yaw = np.float32(np.arctan2(v[2], v[0])) * np.float32(180) / np.float32(np.pi)
        pitch = np.float32(np.arctan2(v[1], np.sqrt(np.float32(v[0] * v[0] + v[2] * v[2])))) * np.float32(180) / np.float32(np.pi)
        return float(yaw), float(pitch), float(d)

    @relative_position.setter
    def relative_position(self, ypd):
        yaw, pitch, distance = (np.float32(x) for x in ypd)
        rad = np.float32(np.pi) / np.float32(180)
        y = np.sin(pitch * rad)
        h = np.cos(pitch * rad)
        x = h * np.cos(yaw * rad)
        z = h * np.sin(yaw * rad)
        self.look_from = (self.look_at + np.array([x, y, z], dtype=np.float32) * distance).astype(np.float32)

    # Config.swift:99-106 (the background is not part of the comparison there either)
    def __eq__(self, other):
        return (isinstance(other, CameraConfig) and self.vertical_fov == other.vertical_fov
                and np.array_equal(self.look_from, other.look_from) and np.array_equal(self.look_at, other.look_at)
                and np.array_equal(self.up, other.up) and self.defocus_angle == other.defocus_angle
                and self.focus_distance == other.focus_distance)

    __hash__ = None

    def copy(self):
        return CameraConfig(self.vertical_fov, self.look_from.copy(), self.look_at.copy(), self.up.copy(),
                            self.defocus_angle, self.focus_distance, self.background)

    @property
    def impl(self):
        return abi.RtCameraConfig(self.vertical_fov, _f3(self.look_from), _f3(self.look_at), _f3(self.up),
                                  self.defocus_angle, self.focus_distance, self.background)


class RenderConfig:
    """Config.swift:72-88."""

    def __init__(self, samples_per_pixel=10, max_depth=10, pass_counter=1, rng_seed=0, first_sample=0):
        self.samples_per_pixel = int(samples_per_pixel)
        self.max_depth = int(max_depth)
        self.pass_counter = int(pass_counter)
        self.first_sample = int(first_sample)
        self.rng_seed = int(rng_seed) & ((1 << 64) - 1)

    @property
    def impl(self):
        """The reference's record (Config.h:28-33); first_sample travels as an argument."""
        return abi.RtRenderConfig(self.samples_per_pixel, self.max_depth, self.pass_counter, self.rng_seed)


class Scene:
    """Scene.swift:9-16."""

    def __init__(self, camera, objects, convention=abi.RT_CONVENTION_METAL):
        self.camera = camera
        self.objects = list(objects)
        # which of the reference's two renderers decides where they disagree (texture coordinates of
        # spheres and cylinders, the metal absorb rule): scenes of the RayTracingKit CLI use the Kit's
        self.convention = convention

    def encode(self, sink, dedup=True):
        """SceneBuffers.init's loop over objects (SceneBuffers.swift:24-27)."""
        if self.convention != abi.RT_CONVENTION_METAL:
            sink.set_convention(self.convention)
        enc = SceneEncoder(sink, dedup=dedup)
        for obj in self.objects:
            sink.add_object(obj.as_impl(enc))
        return enc
