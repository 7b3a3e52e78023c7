#!/usr/bin/env python
"""Developer tool: the swept-volume sweep with a mesh robot, with and without the distance lattice (ISDF_NO_F32_FILTER=1 builds none),
over random maps / trajectories / meshes / bands: t*, cost and gradients must be the same BITS (what the lattice leaves out is never
looked at).  usage: lattice_stress.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
from benchlib.meshes import reference_mesh, adversarial_mesh, box_mesh, merge_meshes
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2025)
bad = 0; active = 0
for case in range(n_cases):
    res = float(rng.choice([0.2, 0.25, 0.4]))
    dims = (int(rng.integers(40, 72)), int(rng.integers(40, 72)), int(rng.integers(24, 40)))
    occ = synth.random_box_map(dims, res=res, occupancy=float(rng.uniform(0.05, 0.2)), seed=int(rng.integers(1, 10 ** 6)), edge=(0.4, 2.0))
    ext = np.array(occ.shape) * res
    N = int(rng.integers(3, 8))
    T, Cf = synth.random_trajectory(ext, N, seed=int(rng.integers(1, 10 ** 6)), piece_T=float(rng.uniform(0.6, 1.6)), jitter=0.5, margin=1.5, occ=occ, res=res)
    cm = synth.colmajor(Cf)
    kind = case % 8
    if kind == 0: V, F = synth.blob_mesh(24, 16, scale=float(rng.uniform(0.5, 1.1)))
    elif kind == 1: V, F = synth.blob_mesh(12, 8, scale=float(rng.uniform(0.5, 1.1)))
    elif kind == 2: V, F = reference_mesh("Trefoil", float(rng.uniform(0.5, 1.0)))
    elif kind == 3:       # the reference's small robot meshes (triangle soups), random scale
        V, F = reference_mesh(str(rng.choice(["Lthick", "mybox", "box", "kuang", "drone", "icosahedron"])), float(rng.uniform(0.08, 2.0)))
    elif kind == 4:       # hollow shell with a random wall (0.5 mm ... 5 cm)
        h = rng.uniform(0.3, 0.9, 3); w = float(10 ** rng.uniform(-3.3, -1.3)); n = int(rng.integers(1, 5))
        V, F = merge_meshes(box_mesh(-h, h, n), box_mesh(-h + w, h - w, n, outward=False))
    elif kind == 5:       # two bodies a random gap apart (0.2 mm ... 5 cm)
        h = rng.uniform(0.3, 0.9, 3); g = float(10 ** rng.uniform(-3.7, -1.3)); n = int(rng.integers(1, 5))
        V, F = merge_meshes(box_mesh(-h, [-g, h[1], h[2]], n), box_mesh([g, -h[1], -h[2]], h, n))
    elif kind == 6:       # meshes that must be REFUSED a lattice (nested, overlapping, inverted, open) - on / off trivially equal, and checked to be refused
        V, F, _ = adversarial_mesh(str(rng.choice(["shell in shell", "shell in shell, 2 mm apart", "overlapping boxes", "inverted box", "open box"])), n=int(rng.integers(1, 5)))
    else:                 # a blob squeezed flat along one axis (slivers, a thin body)
        V, F = synth.blob_mesh(16, 10, scale=float(rng.uniform(0.5, 1.0))); V = V * np.array([1.0, 1.0, float(rng.uniform(0.02, 0.3))])
    safety = float(rng.uniform(0.08, 0.6))
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=float(rng.uniform(1.0, 3.0)))
    if len(pts) == 0:
        continue
    out = {}
    for off in ("0", "1"):
        os.environ["ISDF_NO_F32_FILTER"] = off
        cfg = synth.default_config(capi.V1_SWEPT, safety_hor=safety)
        e = pkg.Engine(cfg); e.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e.set_shape(synth.make_mesh_shape(V, F)); e.set_points(pts)
        ts = -np.ones(len(pts)); steps = []
        for _ in range(2):
            c, gT, gC = e.eval_single(T, cm, tstar=ts)
            steps.append((c, gT.copy(), gC.copy(), ts.copy()))
        out[off] = (steps, e.stats(), e.mesh_info())
    same = all(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) for a, b in zip(out["0"][0], out["1"][0]))
    if kind == 6 and out["0"][2]["lattice"] != (0, 0, 0):
        same = False; print("   a mesh that does not bound a solid was given a lattice:", out["0"][2])
    bad += 0 if same else 1
    with_lattice = globals().get("with_lattice", 0) + (1 if out["0"][2]["lattice"] != (0, 0, 0) else 0)
    active += out["0"][1]["grad_pairs"]
    if n_cases <= 200 or not same or case % 50 == 0: print(f"case {case:2d}: {len(F):5d} faces, {len(pts):5d} points, band {safety:.2f}, res {res}: active {out['0'][1]['grad_pairs']:5d}  {'same bits' if same else 'DIFFERENT'}", flush=True)
print(f"{n_cases} cases ({globals().get('with_lattice', 0)} with a lattice), {active} active points in all, {bad} with a difference")
sys.exit(1 if bad else 0)
