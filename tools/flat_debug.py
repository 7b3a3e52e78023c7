#!/usr/bin/env python
"""Developer tool: the flat evaluation of a small mesh against the quad walks on the adversarial test's world - where do t* / gradients differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as graft
from benchlib.meshes import adversarial_mesh
from common import small_world, traj
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
name = sys.argv[1] if len(sys.argv) > 1 else "mybox.obj"
V, F, _ = adversarial_mesh(name)
occ, esdf, res = small_world(pkg, seed=14)
T, cm = traj(pkg, occ, res, N=5, seed=23)
way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
out = {}
for form, flat, wg in (("flat", "1", "1"), ("quad", "0", "1"), ("wave", "0", "0")):
    os.environ["ISDF_MESH_FLAT"] = flat; os.environ["ISDF_MESH_WG"] = wg
    e = pkg.Engine(synth.default_config(capi.V1_SWEPT, safety_hor=0.3)); e.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e.set_shape(synth.make_mesh_shape(V, F))
    per = []
    for i in range(len(pts)):          # one point at a time: which POINT differs
        e.set_points(pts[i:i + 1]); ts = -np.ones(1)
        c, gT, gC = e.eval_single(T, cm, tstar=ts)
        c4, gT4, gC4 = e.eval_swept_at_tstar_host(T, cm, ts)      # the per-lane walks at the same t* (isdf_eval_swept_at_tstar)
        per.append((c, ts[0], gT.copy(), gC.copy(), gC4.copy()))
    out[form] = per
    print(form, e.mesh_info())
for other in ("quad", "wave"):
    nd = 0
    for i, (a, b) in enumerate(zip(out["flat"], out[other])):
        if a[0] != b[0] or a[1] != b[1] or not np.array_equal(a[2], b[2]) or not np.array_equal(a[3], b[3]):
            nd += 1
            if nd <= 8:
                print(f"   per-lane walks at t*: == flat {np.array_equal(a[4], a[3])} ({np.abs(a[4]-a[3]).max():.2e}), == {other} {np.array_equal(b[4], b[3])} ({np.abs(b[4]-b[3]).max():.2e})")
                print(f"point {i} {pts[i]}: flat cost {a[0]!r} t* {a[1]!r} | {other} cost {b[0]!r} t* {b[1]!r} | max |dgC| {np.abs(a[3] - b[3]).max():.3e} max |dgT| {np.abs(a[2] - b[2]).max():.3e}")
    print(f"flat vs {other}: {nd} of {len(pts)} points differ")
