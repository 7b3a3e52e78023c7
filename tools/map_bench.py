#!/usr/bin/env python
"""Developer tool: time the device map products (ESDF build, point-cloud ingestion, constraint-point gather) on the
bench-sized maps and check them against the oracle (the CPU leg is the oracle's restatement of generateESDF3d)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
orc = graft.load_oracle()

for n, res in ((256, 0.2), (512, 0.1)):
    occ = synth.random_box_map((n,) * 3, res=res, occupancy=0.15, seed=12345)
    eng = pkg.Engine(synth.default_config(capi.V3_ESDF_TILE))
    eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    eng.generate_esdf()                     # warm-up (allocations)
    t0 = time.perf_counter(); eng.generate_esdf(); t1 = time.perf_counter()
    e, _, _ = eng.get_grid(capi.GRID_ESDF)
    line = f"ESDF {n}^3: device {1e3*(t1-t0):8.2f} ms  ({occ.size*4/ (t1-t0)/1e9:6.1f} GB/s of fp32 output)"
    if n == 256:
        c0 = time.perf_counter(); want = orc.build_esdf(occ, res); c1 = time.perf_counter()
        line += f" | oracle CPU {1e3*(c1-c0):8.1f} ms | bit-exact: {np.array_equal(e, want.astype(np.float32))}"
    print(line, flush=True)
    # point cloud ingestion: 4 points per occupied voxel
    ii = np.argwhere(occ)
    rng = np.random.default_rng(1)
    P = ((ii[:, None, :] + rng.uniform(0.05, 0.95, (ii.shape[0], 4, 3))) * res).reshape(-1, 3).astype(np.float32)
    t0 = time.perf_counter(); dims = eng.set_pointcloud(P, res, 1, bmin=np.zeros(3), bmax=np.full(3, n * res)); t1 = time.perf_counter()
    occ2, _, _ = eng.get_grid(capi.GRID_OCCUPANCY)
    print(f"point cloud {P.shape[0]} pts -> occupancy {dims}: {1e3*(t1-t0):8.2f} ms (incl. H2D), equal to source map: {np.array_equal(occ2, occ)}", flush=True)
    way = np.stack([np.linspace(5, n * res - 5, 39)] * 3, axis=1)
    t0 = time.perf_counter(); M = eng.gather_points(way, 1.4); t1 = time.perf_counter()
    print(f"gather 39 waypoints: M = {M}, {1e3*(t1-t0):8.2f} ms", flush=True)
