#!/usr/bin/env python
"""Generates tests/golden/ref_meshes.npz: the two robot meshes SURVEY.md 8(d) names for configs[4] - shapes/Trefoil.obj (12 000
faces) and shapes/Lthick.obj (20 faces, concave) of the reference's src/plan_manager - and its small robot meshes mybox / box / icosahedron / kuang / drone.obj (12-96 faces) - as the PRODUCT's own reader
(isdf_read_obj = igl::read_triangle_mesh's view of the file) delivers them, untransformed.  They cannot travel to the GPU box
with /root/reference.  tests/test_fixtures.py holds the reader against an independent parse of the files.
Run in the build container: python tests/golden/make_golden_meshes.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

PM = "/root/reference/src/plan_manager"
fx = graft.load_package().fixtures
out = {}
for stem in ("Trefoil", "Lthick", "mybox", "box", "icosahedron", "kuang", "drone"):
    V, F = fx.read_obj(os.path.join(PM, "shapes", stem + ".obj"))
    # independent parse: "v x y z" / "f a b c" lines (both files are plain triangle soups without normals / texture indices)
    Vi, Fi = [], []
    for line in open(os.path.join(PM, "shapes", stem + ".obj")):
        t = line.split()
        if t and t[0] == "v":
            Vi.append([float(x) for x in t[1:4]])
        elif t and t[0] == "f":
            Fi.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    assert np.array_equal(V, np.array(Vi)) and np.array_equal(F, np.array(Fi, dtype=np.int32)), stem
    out[stem + "_V"] = V
    out[stem + "_F"] = F.astype(np.int32)
    print(stem, V.shape, F.shape, "bbox", V.min(0), V.max(0), "max |v|", np.linalg.norm(V, axis=1).max())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_meshes.npz"), **out)
print(os.path.getsize(os.path.join(ROOT, "tests", "golden", "ref_meshes.npz")), "bytes")
