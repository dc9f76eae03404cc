"""Single-precision vector / quaternion helpers standing in for Apple's `simd`, which the
reference's Swift host uses (MetalRayTracer/Renderable.swift:28-70, Primitives.swift:94-105).

Apple's simd is not part of the reference sources; what is restated here is its published
behaviour: `simd_quatf(angle:axis:)`, Hamilton product, `act`, `simd_float3x3(quat)` and
`simd_quatf(from:to:)`.  The reference's own tests pin the composed rotation matrices
indirectly (MetalRayTracerTests/SphereTests.swift:91-121, CuboidTests.swift:104-135).
"""
import math
import numpy as np

f32 = np.float32


def vec3(x, y=None, z=None):
    if y is None:
        a = np.asarray(x, dtype=np.float32).reshape(3)
        return a.copy()
    return np.array([x, y, z], dtype=np.float32)


def dot(a, b):
    return f32(a[0] * b[0] + a[1] * b[1] + a[2] * b[2])


def cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]],
                    dtype=np.float32)


def length(a):
    return f32(math.sqrt(float(dot(a, a))))


def normalize(a):
    return (np.asarray(a, dtype=np.float32) / length(a)).astype(np.float32)


class Quat:
    """Unit quaternion (ix, iy, iz, r), like simd_quatf."""

    __slots__ = ("v", "r")

    def __init__(self, ix=0.0, iy=0.0, iz=0.0, r=1.0):
        self.v = np.array([ix, iy, iz], dtype=np.float32)
        self.r = f32(r)

    @staticmethod
    def identity():
        return Quat(0, 0, 0, 1)

    @staticmethod
    def from_angle_axis(angle, axis):
        axis = np.asarray(axis, dtype=np.float32)
        s = f32(math.sin(float(angle) / 2))
        c = f32(math.cos(float(angle) / 2))
        return Quat(axis[0] * s, axis[1] * s, axis[2] * s, c)

    @staticmethod
    def from_to(a, b):
        """Rotation taking unit vector a to unit vector b (simd_quatf(from:to:))."""
        a = np.asarray(a, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        c = np.cross(a, b)
        s = float(np.linalg.norm(c))
        d = float(np.dot(a, b))
        if s < 1e-12:
            if d > 0:
                return Quat.identity()
            t = np.array([1.0, 0, 0]) if abs(a[0]) < 0.9 else np.array([0, 1.0, 0])
            axis = np.cross(a, t)
            axis /= np.linalg.norm(axis)
            return Quat(axis[0], axis[1], axis[2], 0.0)
        return Quat.from_angle_axis(math.atan2(s, d), c / s)

    def __mul__(self, o):
        a, b = self, o
        ar, br = float(a.r), float(b.r)
        av = a.v.astype(np.float64)
        bv = b.v.astype(np.float64)
        r = ar * br - float(np.dot(av, bv))
        v = ar * bv + br * av + np.cross(av, bv)
        return Quat(v[0], v[1], v[2], r)

    @property
    def inverse(self):
        return Quat(-self.v[0], -self.v[1], -self.v[2], self.r)

    def act(self, p):
        p = np.asarray(p, dtype=np.float64)
        v = self.v.astype(np.float64)
        r = float(self.r)
        t = 2.0 * np.cross(v, p)
        return (p + r * t + np.cross(v, t)).astype(np.float32)

    def matrix_columns(self):
        """simd_float3x3(quat): columns c0, c1, c2 (float32)."""
        x, y, z = (float(c) for c in self.v)
        w = float(self.r)
        c0 = [1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)]
        c1 = [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)]
        c2 = [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]
        return np.array([c0, c1, c2], dtype=np.float32)
