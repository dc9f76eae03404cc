"""Algorithmic work per sample of the path-tracing hot path.

The operation counts N (box tests, primitive tests per kind, CSG merge events, volume
intervals, shading events per material kind, rays, hits) come from an instrumented run
of the CPU float restatement on the same scene and seed, committed as a fixture per
workload (tests/golden/counters_*.json).  The per-operation constants below are FP32
flops with FMA = 2:

  box slab test 24 | sphere 30 | cuboid 75 | cylinder 95 | quad 45 | CSG merge event 6 |
  volume interval 20 | shade lambertian / metal / dielectric / isotropic / emissive
  40 / 45 / 60 / 30 / 6 | ray generation 60 | accumulate 6 | hit completion 20.

Bytes are the minimal records each test has to read: box 32, sphere 20, cuboid 84,
cylinder 68, quad 68, plus 12 per sample for the accumulator.
"""

F_BOX, F_SPHERE, F_CYLINDER, F_CUBOID, F_QUAD = 24, 30, 95, 75, 45
F_CSG_EVENT, F_CDV = 6, 20
F_SHADE = {1: 40, 2: 40, 3: 40, 4: 45, 5: 45, 6: 45, 7: 60, 8: 6, 9: 30}
F_RAYGEN, F_ACCUM, F_HIT = 60, 6, 20
B_BOX, B_SPHERE, B_CYLINDER, B_CUBOID, B_QUAD, B_ACCUM = 32, 20, 68, 84, 68, 12


def per_sample(counters):
    """counters: dict as produced by the oracle (OrcCounters.as_dict).  Returns
    (flops per sample, bytes per sample, flops per sample spent in traversal/intersection)."""
    n = float(counters["samples"])
    sph, cyl, cub, quad = counters["prim_tests"]
    trace = (counters["box_tests"] * F_BOX + sph * F_SPHERE + cyl * F_CYLINDER + cub * F_CUBOID + quad * F_QUAD
             + counters["csg_events"] * F_CSG_EVENT + counters["cdv_intervals"] * F_CDV)
    shade = sum(counters["shades"][k] * f for k, f in F_SHADE.items()) + counters["hits"] * F_HIT
    flops = (trace + shade) / n + F_RAYGEN + F_ACCUM
    nbytes = (counters["box_tests"] * B_BOX + sph * B_SPHERE + cyl * B_CYLINDER + cub * B_CUBOID + quad * B_QUAD) / n + B_ACCUM
    return flops, nbytes, trace / n
