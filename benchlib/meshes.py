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


# ---- adversarial robot meshes for the distance lattice (tests/test_gpu_mesh.py, tools/lattice_stress.py) --------------------------------
def box_mesh(lo, hi, n=1, outward=True):
    """Axis-aligned box [lo, hi] as 12 n^2 triangles with shared vertices; outward = counter-clockwise seen from outside."""
    lo = np.asarray(lo, dtype=np.float64); hi = np.asarray(hi, dtype=np.float64)
    V, F, index = [], [], {}

    def vid(p):
        key = tuple(np.round(p, 12))
        if key not in index:
            index[key] = len(V); V.append(p)
        return index[key]
    for axis in range(3):
        u, v = (axis + 1) % 3, (axis + 2) % 3
        for side in (0, 1):
            for i in range(n):
                for j in range(n):
                    def P(a, b):
                        p = np.zeros(3); p[axis] = hi[axis] if side else lo[axis]
                        p[u] = lo[u] + (hi[u] - lo[u]) * a / n; p[v] = lo[v] + (hi[v] - lo[v]) * b / n
                        return p
                    q = [vid(P(i, j)), vid(P(i + 1, j)), vid(P(i + 1, j + 1)), vid(P(i, j + 1))]
                    tris = [(q[0], q[1], q[2]), (q[0], q[2], q[3])]          # normal along +axis
                    for t in tris:
                        F.append(t if (side == 1) == outward else t[::-1])
    return np.array(V), np.array(F, dtype=np.int32)


def merge_meshes(*meshes):
    Vs, Fs, off = [], [], 0
    for V, F in meshes:
        Vs.append(np.asarray(V, dtype=np.float64)); Fs.append(np.asarray(F, dtype=np.int32) + off); off += len(V)
    return np.vstack(Vs), np.vstack(Fs).astype(np.int32)


def adversarial_mesh(name, n=3):
    """(V, F, expect_solid): meshes the distance lattice must either handle bit for bit or refuse.  n: subdivisions per box edge."""
    h = np.array([0.8, 0.5, 0.35])
    if name == "thin shell 1 mm":             # a hollow body: outer sheet outwards, inner sheet inwards, 1 mm of wall
        return merge_meshes(box_mesh(-h, h, n), box_mesh(-h + 1e-3, h - 1e-3, n, outward=False)) + (1,)
    if name == "shell in shell":              # two closed sheets of ONE orientation, one inside the other: winding number 2 inside the inner one
        return merge_meshes(box_mesh(-h, h, n), box_mesh(-0.55 * h, 0.55 * h, n)) + (0,)
    if name == "shell in shell, 2 mm apart":  # ... the doubly covered region's boundary a hair inside the outer sheet
        return merge_meshes(box_mesh(-h, h, n), box_mesh(-h + 2e-3, h - 2e-3, n)) + (0,)
    if name == "two sheets a cell apart":     # two bodies separated by about one lattice cell
        g = 0.015
        return merge_meshes(box_mesh(-h, [-g, h[1], h[2]], n), box_mesh([g, -h[1], -h[2]], h, n)) + (1,)
    if name == "overlapping boxes":           # two bodies sharing a slab thinner than a lattice cell: winding number 2 in it
        return merge_meshes(box_mesh(-h, [0.004, h[1], h[2]], n), box_mesh([-0.004, -h[1], -h[2]], h, n)) + (0,)
    if name == "inverted box":
        return box_mesh(-h, h, n, outward=False) + (0,)
    if name == "open box":                    # one face missing
        V, F = box_mesh(-h, h, n)
        return V, F[: len(F) - 2 * n * n], 0
    if name == "mybox.obj":
        return reference_mesh("mybox", 0.9) + (1,)
    if name == "Lthick.obj x 0.1":            # (relative to the 0.83 m bound the bench scales the reference's meshes to)
        return reference_mesh("Lthick", 0.083) + (1,)
    if name == "Lthick.obj x 10":
        return reference_mesh("Lthick", 8.3) + (1,)
    if name in ("box.obj", "kuang.obj", "drone.obj", "icosahedron.obj"):
        return reference_mesh(name[:-4], 0.9) + (None,)
    raise KeyError(name)
