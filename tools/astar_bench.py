#!/usr/bin/env python
"""Developer tool: the SE(3) A* of the front end (isdf_frontend_astar_search over the device-built configuration space) on a
demo-sized world against the reference's own AstarPathSearcher (oracle/_ref/libref_sweeps.so, when built) and the oracle's
restatement: same searches, path and attitudes compared node for node, wall time per search."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
orc = graft.load_oracle()
res = 0.2
dims = (256, 256, 64)                      # 51.2 x 51.2 x 12.8 m: the demos' worlds are 50 x 50 m
occ = synth.random_box_map(dims, res=res, occupancy=float(os.environ.get("ASTAR_OCC", "0.12")), seed=12345)
rng = np.random.default_rng(0)
have_ref = os.path.exists(orc.REF_SWEEPS) and os.environ.get("ASTAR_SKIP_REF") != "1"
for name, cls, shape, k in (("box 3.2x0.6x0.6 m", "Box", synth.bench_box_shape(), 21),):
    cfg = synth.default_config(capi.V1_SWEPT)
    fe = capi.frontend_config(kernel_size=k, max_roll=45.0, max_pitch=45.0, ang_res=9.0, safeh=0.0)
    eng = pkg.Engine(cfg); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape)
    t0 = time.perf_counter(); eng.frontend_build(fe); tb = time.perf_counter() - t0
    table, ms = eng.frontend_cspace()
    fits = (table != 0).any(axis=-1)
    print(f"{name}, k={k}, 11x11 attitudes, map {dims} at {res} m ({occ.mean():.3f} occupied, some attitude fits in {fits.mean():.3f} of the voxels): "
          f"isdf_frontend_build {tb*1e3:.1f} ms, configuration-space kernel {ms:.2f} ms", flush=True)
    good = np.argwhere(fits)
    pairs = []
    while len(pairs) < 8:
        a, b = good[rng.choice(len(good), 2, replace=False)]
        if np.abs(a - b).max() >= 150:
            pairs.append(((a + 0.5) * res, (b + 0.5) * res))
    w = None
    if have_ref:
        t0 = time.perf_counter()
        w = orc.RefWorld(cfg, cls, shape, res); w.set_occupancy(occ, (0, 0, 0), res); w.frontend_build(fe, cls)
        t_refbuild = time.perf_counter() - t0
        print(f"  reference's own code: kernels + generateMapKernel {t_refbuild*1e3:.0f} ms", flush=True)
    o = orc.Oracle(cfg, threads=8); o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.frontend_build(fe)
    tot_p = tot_r = 0.0
    for q, (s, g) in enumerate(pairs):
        t0 = time.perf_counter(); xyz, rp, rot, r = eng.frontend_astar(s, g); tp = time.perf_counter() - t0
        line = (f"  search {q}: path {r.n_path:4d} nodes, {r.expansions:7d} expansions, {r.checks:9d} neighbour tests | device table (once) "
                f"{r.table_ms:7.1f} ms, search {r.search_ms:8.2f} ms, call {tp*1e3:8.2f} ms")
        if q < 3 and os.environ.get("ASTAR_SKIP_REF") != "1":
            t0 = time.perf_counter(); x0, rp0, st = o.frontend_astar(s, g); to = time.perf_counter() - t0
            same = (xyz is None and x0 is None) or (xyz is not None and x0 is not None and np.array_equal(xyz, x0) and np.array_equal(rp, rp0))
            line += f" | oracle {to*1e3:9.1f} ms identical {same and st['checks'] == r.checks}"
        if w is not None:
            t0 = time.perf_counter(); xr, rpr, checks = w.astar(s, g); tr = time.perf_counter() - t0
            same = (xyz is None and xr is None) or (xyz is not None and xr is not None and np.array_equal(xyz, xr) and np.array_equal(rp, rpr))
            line += f" | reference {tr*1e3:9.1f} ms (x{tr / max(r.search_ms * 1e-3, 1e-9):7.1f}) identical {same and checks == r.checks}"
            if q > 0:           # (the reference's first call also allocates its node map)
                tot_p += r.search_ms * 1e-3; tot_r += tr
        print(line, flush=True)
    if w is not None and tot_p > 0:
        print(f"  searches 1..{len(pairs) - 1}: device-table A* {tot_p*1e3:.1f} ms, reference {tot_r*1e3:.1f} ms, x{tot_r / tot_p:.1f}", flush=True)
