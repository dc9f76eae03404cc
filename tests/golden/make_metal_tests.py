"""Writes tests/golden/metal_tests.json: the known-answer vectors of the reference's own
unit tests (MetalRayTracerTests/SphereTests.swift, CylinderTests.swift, CuboidTests.swift,
SubtractTests.swift, ConstantDensityVolumeTests.swift), transcribed as data: the shape
under test, the ray, and the ordered hits each test asserts.

Nothing of the reference is imported or executed: the numbers below are the literals of
those tests (file:line given per case).  Rotations the tests build with
simd_matrix3x3(q2 * q1) are described by their angle/axis factors; the loader composes
them with raytracer_amd.simd.  `null` in an expected hit means "not asserted by the
test".  Tolerance: the tests' defaultTestAccuracy = 2^-20 (Utils.swift:9-13).
"""
import json
import math
import os

S = math.sqrt(1.0 / 3.0)
I = None  # identity rotation

cases = []


def case(name, cite, shape, ray, hits, rng=None):
    cases.append({"name": name, "cite": cite, "shape": shape, "ray": ray, "rng": rng, "hits": hits})


def hit(t, p=None, n=None, m=None, uv=None, u_valid=False, v=None, is_exit=None):
    return {"t": t, "point": p, "normal": n, "material": m, "uv": uv, "u_valid": u_valid, "v": v, "is_exit": is_exit}


# ---- SphereTests.swift:15-89 (no transform), radius 0.5, material 42
sph = {"kind": "sphere", "rotation": I, "translation": [0, 0, 0], "radius": 0.5, "material": 42}
case("sphere_axis_y", "SphereTests.swift:25-44", sph, [[0, -5, 0], [0, 1, 0]], [
    hit(4.5, [0, -0.5, 0], [0, -1, 0], 42, u_valid=True, v=0.0),
    hit(5.5, [0, 0.5, 0], [0, 1, 0], 42, u_valid=True, v=1.0)])
case("sphere_miss", "SphereTests.swift:46-49", sph, [[1, -5, 1], [0, 1, 0]], [])
case("sphere_axis_x", "SphereTests.swift:51-68", sph, [[2, 0, 0], [-1, 0, 0]], [
    hit(1.5, [0.5, 0, 0], [1, 0, 0], 42, [1.0, 0.5]),
    hit(2.5, [-0.5, 0, 0], [-1, 0, 0], 42, [0.5, 0.5])])
case("sphere_diagonal", "SphereTests.swift:70-88", sph, [[-1, 1, -1], [S, -S, S]], [
    hit(math.sqrt(3) - 0.5, [-S / 2, S / 2, -S / 2], [-S, S, -S], 42, [0.375, 0.6959132760153036]),
    hit(math.sqrt(3) + 0.5, [S / 2, -S / 2, S / 2], [S, -S, S], 42, [0.875, 0.3040867239846964])])

# ---- SphereTests.swift:91-121: rotation q2*q1, q1 = pi/3 about y, q2 = pi/4 about z
ROT_21 = [["z", math.pi / 4], ["y", math.pi / 3]]   # product order: first factor applied last
sph_t = {"kind": "sphere", "rotation": ROT_21, "translation": [1, 2, 3], "radius": 0.5, "material": 42}
case("sphere_transformed", "SphereTests.swift:103-120", sph_t, [[1, 2.3, -3], [0, 0, 1]], [
    hit(5.6, [1, 2.3, 2.6], [0, 0.6, -0.8], 42, [0.005726772, 0.6394671]),
    hit(6.4, [1, 2.3, 3.4], [0, 0.6, 0.8], 42, [0.66093993, 0.6394671])])

# ---- CylinderTests.swift:11-143: r 0.5, h 2, materials bottom 5 / top 7 / side 9
cyl = {"kind": "cylinder", "rotation": I, "translation": [0, 0, 0], "radius": 0.5, "height": 2, "materials": [5, 7, 9]}
case("cyl_axis", "CylinderTests.swift:24-43", cyl, [[0, -5, 0], [0, 1, 0]], [
    hit(5.0, [0, 0, 0], [0, -1, 0], 5, [0.5, 0.5], is_exit=False),
    hit(7.0, [0, 2, 0], [0, 1, 0], 7, [0.5, 0.5], is_exit=True)])
case("cyl_axis_plus", "CylinderTests.swift:45-64", cyl, [[0.25, -5, 0.25], [0, 1, 0]], [
    hit(5.0, [0.25, 0, 0.25], [0, -1, 0], 5, [0.25, 0.75], is_exit=False),
    hit(7.0, [0.25, 2, 0.25], [0, 1, 0], 7, [0.75, 0.75], is_exit=True)])
case("cyl_axis_minus", "CylinderTests.swift:66-85", cyl, [[-0.25, -5, -0.25], [0, 1, 0]], [
    hit(5.0, [-0.25, 0, -0.25], [0, -1, 0], 5, [0.75, 0.25], is_exit=False),
    hit(7.0, [-0.25, 2, -0.25], [0, 1, 0], 7, [0.25, 0.25], is_exit=True)])
case("cyl_axis_miss", "CylinderTests.swift:87-90", cyl, [[1, -5, 1], [0, 1, 0]], [])
case("cyl_side_x", "CylinderTests.swift:92-111", cyl, [[-5, 1, 0], [1, 0, 0]], [
    hit(4.5, [-0.5, 1, 0], [-1, 0, 0], 9, [0.5, 0.5], is_exit=False),
    hit(5.5, [0.5, 1, 0], [1, 0, 0], 9, [1.0, 0.5], is_exit=True)])
case("cyl_side_z", "CylinderTests.swift:113-132", cyl, [[0, 0.5, 5], [0, 0, -1]], [
    hit(4.5, [0, 0.5, 0.5], [0, 0, 1], 9, [0.75, 0.25], is_exit=False),
    hit(5.5, [0, 0.5, -0.5], [0, 0, -1], 9, [0.25, 0.25], is_exit=True)])
case("cyl_below", "CylinderTests.swift:134-137", cyl, [[-5, -1, 0], [1, 0, 0]], [])
case("cyl_above", "CylinderTests.swift:139-142", cyl, [[-5, 3, 0], [1, 0, 0]], [])

# ---- CylinderTests.swift:145-188: rotated pi/3 about z
cyl_t = dict(cyl, rotation=[["z", math.pi / 3]])
case("cyl_rotated", "CylinderTests.swift:160-187", cyl_t, [[0, -5, 0], [0, 1, 0]], [
    hit(5.0, [0, 0, 0], [0.866025, -0.5, 0], 5, [0.5, 0.5], is_exit=False),
    hit(5.57735, [0, 0.57735, 0], [0.5, 0.866025, 0], 9, [1.0, 0.14433756], is_exit=True)])

# ---- CuboidTests.swift:10-102: size (2,3,4), materials left 5, right 7, bottom 9, top 11, back 13, front 15
box = {"kind": "cuboid", "rotation": I, "translation": [0, 0, 0], "size": [2, 3, 4], "materials": [5, 7, 9, 11, 13, 15]}
case("box_x", "CuboidTests.swift:20-37", box, [[-1, 1.2, 2.8], [1, 0, 0]], [
    hit(1, [0, 1.2, 2.8], [-1, 0, 0], 5, [0.3, 0.4]), hit(3, [2, 1.2, 2.8], [1, 0, 0], 7, [0.7, 0.4])])
case("box_z", "CuboidTests.swift:39-56", box, [[1.2, 2.7, -1], [0, 0, 1]], [
    hit(1, [1.2, 2.7, 0], [0, 0, -1], 13, [0.4, 0.9]), hit(5, [1.2, 2.7, 4], [0, 0, 1], 15, [0.6, 0.9])])
case("box_y", "CuboidTests.swift:58-75", box, [[1.4, -1, 1.6], [0, 1, 0]], [
    hit(1, [1.4, 0, 1.6], [0, -1, 0], 9, [0.7, 0.6]), hit(4, [1.4, 3, 1.6], [0, 1, 0], 11, [0.7, 0.4])])
for i, (o, d) in enumerate([([-1, -1, -1], [1, 0, 0]), ([-1, -1, -1], [0, 1, 0]), ([-1, -1, -1], [0, 0, 1]),
                            ([10, 10, 10], [-1, 0, 0]), ([10, 10, 10], [0, -1, 0]), ([10, 10, 10], [0, 0, -1])]):
    case("box_miss_%d" % i, "CuboidTests.swift:77-101", box, [o, d], [])

# ---- CuboidTests.swift:104-135
box_t = dict(box, rotation=ROT_21, translation=[1, 2, 3])
case("box_transformed", "CuboidTests.swift:117-134", box_t, [[1.4, -1, 1.6], [0, 1, 0]], [
    hit(3.7430952, [1.4, 2.7430952, 1.6], [-0.61237234, -0.61237246, -0.5], 13, [0.19170964, 0.08086828]),
    hit(4.827569, [1.4, 3.827569, 1.6], [0.35355335, 0.3535534, -0.8660254], 7, [0.16602547, 0.3364812])])

# ---- SubtractTests.swift:10-33: A = cyl r2 h4 (5/7/9) at origin; B = cyl r1 h6 (11/13/15) at (0,1,0)
cyl_a = {"kind": "cylinder", "rotation": I, "translation": [0, 0, 0], "radius": 2, "height": 4, "materials": [5, 7, 9]}
cyl_b = {"kind": "cylinder", "rotation": I, "translation": [0, 1, 0], "radius": 1, "height": 6, "materials": [11, 13, 15]}
diff = {"kind": "csg", "op": "subtract", "children": [cyl_a, cyl_b]}
case("subtract_ABAB", "SubtractTests.swift:35-54", diff, [[0, -1, 0], [0, 1, 0]], [
    hit(1, [0, 0, 0], [0, -1, 0], 5, [0.5, 0.5], is_exit=False),
    hit(2.0, [0, 1, 0], [0, 1, 0], 11, [0.5, 0.5], is_exit=True)])
case("subtract_BABA", "SubtractTests.swift:56-75", diff, [[0, 10, 0], [0, -1, 0]], [
    hit(9, [0, 1, 0], [0, 1, 0], 11, [0.5, 0.5], is_exit=False),
    hit(10, [0, 0, 0], [0, -1, 0], 5, [0.5, 0.5], is_exit=True)])
case("subtract_ABBA", "SubtractTests.swift:77-112", diff, [[0, 2, -5], [0, 0, 1]], [
    hit(3, [0, 2, -2], [0, 0, -1], 9, [0.25, 0.5], is_exit=False),
    hit(4, [0, 2, -1], [0, 0, 1], 15, [0.25, 1 / 6.0], is_exit=True),
    hit(6, [0, 2, 1], [0, 0, -1], 15, [0.75, 1 / 6.0], is_exit=False),
    hit(7, [0, 2, 2], [0, 0, 1], 9, [0.75, 0.5], is_exit=True)])
case("subtract_AA", "SubtractTests.swift:114-133", diff, [[1.5, 2, 5], [0, 0, -1]], [
    hit(3.6771245, [1.5, 2, 1.3228755], [0.75, 0, 0.6614377], 9, [0.88497335, 0.5], is_exit=False),
    hit(6.3228755, [1.5, 2, -1.3228755], [0.75, 0, -0.6614377], 9, [0.11502672, 0.5], is_exit=True)])
case("subtract_BB", "SubtractTests.swift:135-138", diff, [[0, 5, -5], [0, 0, 1]], [])

# ---- SubtractTests.swift:140-174: Subtract<Intersection<Cuboid, Sphere>, Cylinder>
combo = {"kind": "csg", "op": "subtract", "children": [
    {"kind": "csg", "op": "intersection", "children": [
        {"kind": "cuboid", "rotation": I, "translation": [-1, -1, -1], "size": [2, 2, 2], "materials": [11, 12, 13, 14, 15, 16]},
        {"kind": "sphere", "rotation": I, "translation": [0, 0, 0], "radius": 1.2, "material": 21}]},
    {"kind": "cylinder", "rotation": I, "translation": [0, -2, 0], "radius": 0.5, "height": 4, "materials": [31, 32, 33]}]}
case("subtract_combo", "SubtractTests.swift:140-174", combo, [[0, 5, 0], [0, -1, 0]], [])

# ---- ConstantDensityVolumeTests.swift:10-61: ConstantDensityVolume<Cuboid(2,3,4)>, density 0.5
fog = {"kind": "cdv", "density": 0.5, "base": box}
fog_ray = [[3, 1.5, 3], [-0.70710678, 0, -0.70710678]]
case("fog_half", "ConstantDensityVolumeTests.swift:23-41", fog, fog_ray, [
    hit(2.800508, [1.0197418, 1.5, 1.0197418], [0.70710677, 0.0, 0.70710677], 7, [0.5, 0.5]),
    hit(4.242641, [0, 1.5, 0], [0, 0, -1], 13, [1.0, 0.5])], rng=[0.5])
case("fog_threshold", "ConstantDensityVolumeTests.swift:43-47", fog, fog_ray, [], rng=[0.756884])
case("fog_next_down_1", "ConstantDensityVolumeTests.swift:49-53", fog, fog_ray, [], rng=[0.99999994])
case("fog_zero", "ConstantDensityVolumeTests.swift:55-60", fog, fog_ray, [hit(1.41421356)], rng=[0.0])
# the last case only asserts hasNext and t of the first hit
cases[-1]["first_only"] = True

out = {
    "tolerance": 2.0 ** -20,
    "pcg32_demo": {"cite": "PCG32 reference stream for (initstate 42, initseq 54); Impl/RNG.h:37-52 is that generator",
                   "initstate": 42, "initseq": 54,
                   "expected_hex": ["a15c02b7", "7b47f409", "ba1d3330", "83d2f293", "bfa4784b", "cbed606e"]},
    "cases": cases,
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "metal_tests.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(cases), "cases")
