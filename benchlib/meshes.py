"""The reference's robot meshes as bench / test inputs (fixture data under tests/golden/; the product package never opens them)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_mesh(stem, bound_radius=None, npz=None):
    """(V, F) of one of the reference's robot meshes kept as a fixture (tests/golden/ref_meshes.npz, written by
    tests/golden/make_golden_meshes.py from src/plan_manager/shapes/<stem>.obj: 'Trefoil' 12 000 faces, 'Lthick' 20 faces),
    centred on its bounding box and - SURVEY.md 8(d): "scaled to the same bound" - scaled so that its farthest vertex lies
    `bound_radius` from the origin."""
    npz = npz or os.path.join(ROOT, "tests", "golden", "ref_meshes.npz")
    z = np.load(npz)
    V = z[stem + "_V"].astype(np.float64).copy(); F = z[stem + "_F"].astype(np.int32).copy()
    V -= 0.5 * (V.min(0) + V.max(0))
    if bound_radius:
        V *= bound_radius / np.linalg.norm(V, axis=1).max()
    return V, F
