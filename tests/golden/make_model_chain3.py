"""Writes tests/golden/model_chain3.json: a small 3-body PhysModel file of our own (not a reference asset) that exercises
the model builder on shapes the hand model does not have: a hexahedron, a triangular prism (3-sided faces) and an
8-sided two-ring cage.  The expected build (model_chain3.htfx) comes from the reference's own PhysModel constructor:

    python tests/golden/make_model_chain3.py [out.json]
    oracle/_ref/ref_harness modelfile tests/golden/model_chain3.json tests/golden/model_chain3.htfx
"""
import json
import math
import os

import numpy as np

rng = np.random.RandomState(20240607)


def jig(v, s=0.002):
    return [float("%.6g" % (x + rng.uniform(-s, s))) for x in v]


def box(rx, ry, z0, z1):
    v = [(-rx, -ry, z0), (rx, -ry, z0), (rx, ry, z0), (-rx, ry, z0), (-rx, -ry, z1), (rx, -ry, z1), (rx, ry, z1), (-rx, ry, z1)]
    f = [[3, 2, 1, 0], [4, 5, 6, 7], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7]]
    return {"faces": f, "verts": [jig(p) for p in v]}


def prism(r, z0, z1):
    ring = [(r * math.cos(a), r * math.sin(a)) for a in (0.3, 0.3 + 2.1, 0.3 + 4.2)]
    v = [(x, y, z0) for x, y in ring] + [(x, y, z1) for x, y in ring]
    f = [[2, 1, 0], [3, 4, 5], [0, 1, 4, 3], [1, 2, 5, 4], [2, 0, 3, 5]]
    return {"faces": f, "verts": [jig(p) for p in v]}


def rings(n, r0, r1, z0, z1, z2):
    v = []
    for z, r in ((z0, r0), (z1, r1), (z2, r0 * 0.8)):
        v += [(r * math.cos(2 * math.pi * k / n), 0.7 * r * math.sin(2 * math.pi * k / n), z) for k in range(n)]
    f = [list(range(n - 1, -1, -1)), [2 * n + k for k in range(n)]]
    for layer in (0, 1):
        for k in range(n):
            a, b = layer * n + k, layer * n + (k + 1) % n
            f.append([a, b, b + n, a + n])
    return {"faces": f, "verts": [jig(p, 0.001) for p in v]}


model = {
    "controlcages": [box(0.03, 0.02, -0.01, 0.09), prism(0.025, 0.0, 0.06), rings(8, 0.015, 0.018, 0.0, 0.03, 0.055)],
    "joints": [
        {"jointframe": [0, 0, 0, 1], "p0": [0.004, -0.002, 0.085], "p1": [0, 0, 0.001], "rangemax": [60, 25, 0], "rangemin": [-40, -25, 0], "rbi0": 0, "rbi1": 1},
        {"jointframe": [0, 0.0871557, 0, 0.996195], "p0": [0.001, 0.0005, 0.058], "p1": [0, 0, -0.002], "rangemax": [90, 0, 0], "rangemin": [-10, 0, 0], "rbi0": 1, "rbi1": 2},
    ],
}
import sys
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_chain3.json")
with open(out, "w") as fp:
    json.dump(model, fp, indent=1)
print(out)
