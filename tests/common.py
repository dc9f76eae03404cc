"""Helpers shared by the tests: golden-fixture loading and comparison rules."""
import json
import os

import numpy as np

from raytracer_amd.scene import (ConstantDensityVolume, Cuboid, Cylinder, Intersection, MaterialOffset, Sphere,
                                 Subtract, Union, raw_transform)
from raytracer_amd.simd import Quat

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

AX = {"x": (1.0, 0, 0), "y": (0, 1.0, 0), "z": (0, 0, 1.0)}


def load_metal_tests():
    with open(os.path.join(GOLDEN, "metal_tests.json")) as f:
        return json.load(f)


def rotation_columns(spec):
    """[[axis, angle], ...] -> columns of simd_matrix3x3(q_first * ... * q_last)."""
    q = Quat.identity()
    for axis, angle in (spec or []):
        q = q * Quat.from_angle_axis(np.float32(angle), AX[axis])
    return q.matrix_columns()


def shape_from_json(d):
    """Golden-fixture shape description -> Renderable, with the tests' bare material offsets."""
    k = d["kind"]
    if k == "csg":
        kids = [shape_from_json(c) for c in d["children"]]
        if d["op"] == "subtract":
            return Subtract(kids[0], kids[1])
        return (Union if d["op"] == "union" else Intersection)(*kids)
    if k == "cdv":
        return ConstantDensityVolume(shape_from_json(d["base"]), d["density"])
    tr = raw_transform(rotation_columns(d["rotation"]), d["translation"])
    if k == "sphere":
        return Sphere(radius=d["radius"], material=MaterialOffset(d["material"]), transform=tr)
    if k == "cylinder":
        m = [MaterialOffset(x) for x in d["materials"]]
        return Cylinder(tr, d["radius"], d["height"], bottom=m[0], top=m[1], side=m[2])
    if k == "cuboid":
        return Cuboid(tr, d["size"], materials=[MaterialOffset(x) for x in d["materials"]])
    raise ValueError(k)


def check_event(ev, exp, tol, where):
    """ev: object with t/point/normal/uv/material_offset/is_exit; exp: fixture hit."""
    assert abs(ev.t - exp["t"]) <= tol, (where, "t", ev.t, exp["t"])
    if exp["point"] is not None:
        assert max(abs(a - b) for a, b in zip(ev.point, exp["point"])) <= tol, (where, "point", list(ev.point))
    if exp["normal"] is not None:
        assert max(abs(a - b) for a, b in zip(ev.normal, exp["normal"])) <= tol, (where, "normal", list(ev.normal))
    if exp["material"] is not None:
        assert ev.material_offset == exp["material"], (where, "material", ev.material_offset)
    if exp["uv"] is not None:
        assert max(abs(a - b) for a, b in zip(ev.uv, exp["uv"])) <= tol, (where, "uv", list(ev.uv))
    if exp["u_valid"]:
        assert -tol < ev.uv[0] < 1 + tol, (where, "u", ev.uv[0])
    if exp["v"] is not None:
        assert abs(ev.uv[1] - exp["v"]) <= tol, (where, "v", ev.uv[1])
    if exp["is_exit"] is not None:
        assert bool(ev.is_exit) == exp["is_exit"], (where, "is_exit")


def random_rays(n, seed, center=(0, 0, 0), radius=3.0, aim_radius=1.5):
    """Rays from random points on a sphere shell aimed at random points near the centre."""
    rs = np.random.RandomState(seed)
    o = rs.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * radius * rs.uniform(0.2, 1.5, size=(n, 1))
    tgt = rs.uniform(-aim_radius, aim_radius, size=(n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o + np.asarray(center), d], axis=1).astype(np.float32)
    # re-normalise in float32 so that |d| = 1 to rounding
    dd = rays[:, 3:].astype(np.float64)
    rays[:, 3:] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    return rays
