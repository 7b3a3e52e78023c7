#!/usr/bin/env python
"""Developer tool (needs oracle/_ref): the oracle's A* restatement - and, with a GPU, the product's isdf_frontend_astar_search - against the
reference's own AstarPathSearcher on random worlds, robots, kernel sizes and attitude grids.  usage: tools/astar_fuzz.py [n_worlds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); orc = graft.load_oracle(); capi, synth = pkg.capi, pkg.synth
n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(2026)
names = ["Box", "RoundedCone", "Torus_big", "Table", "CSG", "Cappedtorus", "TwistBox", "WireframeBox"]
tot = found = failed = tilted = dev_failed = 0
import torch
GPU = torch.cuda.is_available()
t0 = time.time()
for wi in range(n_worlds):
    name = names[wi % len(names)]
    dims = (int(rng.integers(24, 48)), int(rng.integers(24, 48)), int(rng.integers(8, 20)))
    res = float(rng.choice([0.25, 0.5, 0.375]))          # binary fractions: the reference sizes its grid by ceil((max - min) / res)
    occupancy = float(rng.uniform(0.06, 0.22)) * (1.0 if name == "Box" else 0.6)
    occ = synth.random_box_map(dims, res=res, occupancy=occupancy, seed=1000 + wi, edge=(res, 5 * res))
    kw = dict(params=(float(rng.uniform(0.5, 1.0)), 0.3, 0.2)) if name == "Box" else {}
    shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, float(rng.uniform(-20, 20)), -8, 25), **kw)
    ang = float(rng.choice([9.0, 7.5, 15.0]))
    fe = capi.frontend_config(kernel_size=int(rng.choice([9, 11, 13, 15])), max_roll=45.0, max_pitch=float(rng.choice([45.0, 30.0])), ang_res=ang, safeh=0.0)
    cfg = synth.default_config(capi.V1_SWEPT)
    origin = tuple(float(v) * 0.125 for v in rng.integers(-40, 40, 3))
    o = orc.Oracle(cfg, threads=8); o.set_grid(occ, origin, res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.frontend_build(fe)
    w = orc.RefWorld(cfg, name, shape, res); w.set_occupancy(occ, origin, res); w.frontend_build(fe, name)
    eng = None
    if GPU:
        eng = pkg.Engine(cfg); eng.set_grid(occ, origin, res, capi.GRID_OCCUPANCY); eng.set_shape(shape); eng.frontend_build(fe)
    good = np.argwhere((o.frontend_cspace(occ.shape) != 0).any(axis=-1))
    if len(good) < 2:
        continue
    for q in range(12):
        a, b = good[rng.choice(len(good), 2, replace=False)]
        s, g = (a + rng.random(3)) * res + origin, (b + rng.random(3)) * res + origin
        x_r, rp_r, checks_r = w.astar(s, g)
        x_o, rp_o, st = o.frontend_astar(s, g)
        tot += 1
        ok = st["checks"] == checks_r and (x_r is None) == (x_o is None) and (x_r is None or (np.array_equal(x_o, x_r) and np.array_equal(rp_o, rp_r)))
        if not ok:
            failed += 1
            print(f"MISMATCH world {wi} ({name}) search {q}: checks {st['checks']} vs {checks_r}", flush=True)
        if eng is not None:
            x_d, rp_d, _, r = eng.frontend_astar(s, g)
            if not (r.checks == checks_r and (x_r is None) == (x_d is None) and (x_r is None or (np.array_equal(x_d, x_r) and np.array_equal(rp_d, rp_r)))):
                dev_failed += 1
                print(f"DEVICE MISMATCH world {wi} ({name}) search {q}: checks {r.checks} vs {checks_r}", flush=True)
        if x_r is not None:
            found += 1; tilted += int((rp_r != 0).any())
    print(f"world {wi:2d} {name:13s} dims {dims} res {res} k {fe.kernel_size} ang {ang}: searches so far {tot}, found {found}, with attitudes off level {tilted}, mismatches {failed}", flush=True)
print(f"# {tot} searches on {n_worlds} worlds in {time.time() - t0:.0f} s: {found} paths, {tilted} with attitudes other than level, {failed} mismatches (oracle vs the reference's own code: cells, (roll, pitch) per node, neighbour tests)"
      + (f"; device (isdf_frontend_astar_search) vs the reference's own code: {dev_failed} mismatches" if GPU else ""))
