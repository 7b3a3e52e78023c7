#!/usr/bin/env python
"""Developer tool: front end (kernel convolution) on the bench map - table build time and checkKernelValue queries per second
on the device vs the oracle's OpenMP restatement (= the reference's code path, tests/test_oracle_ref.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
orc = graft.load_oracle()
res = 0.2
occ = synth.random_box_map((256,) * 3, res=res, occupancy=0.15, seed=12345)
rng = np.random.default_rng(0)
for name, shape, k in (("box 3.2x0.6x0.6 m", synth.bench_box_shape(), 21), ("rounded cone", synth.bench_rounded_cone_shape(), 21)):
    cfg = synth.default_config(capi.V1_SWEPT)
    fe = capi.frontend_config(kernel_size=k, max_roll=45.0, max_pitch=45.0, ang_res=9.0, safeh=0.0)
    eng = pkg.Engine(cfg); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape)
    eng.frontend_build(fe)
    t0 = time.perf_counter(); eng.frontend_build(fe); tb = time.perf_counter() - t0
    o = orc.Oracle(cfg, threads=30); o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape)
    t0 = time.perf_counter(); o.frontend_build(fe); tbo = time.perf_counter() - t0
    n = 2_000_000
    # the queries an A* makes: free voxels (the search never expands into occupied ones), parents on the attitude grid
    free = np.argwhere(occ == 0)
    idx = free[rng.integers(0, len(free), n)].astype(np.int32)
    fr = rng.choice(np.arange(-45.0, 45.1, 9.0), n); fp = rng.choice(np.arange(-45.0, 45.1, 9.0), n)
    eng.frontend_check(idx[:1000], fr[:1000], fp[:1000])
    t0 = time.perf_counter(); ok, cr, cp, ki = eng.frontend_check(idx, fr, fp); td = time.perf_counter() - t0
    m = 200_000
    t0 = time.perf_counter(); ok0, cr0, cp0, ki0 = o.frontend_check(idx[:m], fr[:m], fp[:m]); tc = time.perf_counter() - t0
    same = np.array_equal(ok[:m], ok0) and np.array_equal(ki[:m], ki0)
    print(f"{name:20s} k={k} 11x11 attitudes, 256^3 map: build device {tb*1e3:7.1f} ms | oracle {tbo*1e3:8.1f} ms || check device {n/td:10.4g} queries/s (host in/out) | "
          f"oracle 30 threads {m/tc:10.4g} queries/s | x{(n/td)/(m/tc):6.1f} | feasible {ok.mean():.3f}, level attitude {np.mean(ki[ok==1]==60):.3f} | identical: {same}", flush=True)
    tab, ms = eng.frontend_cspace(download=False)
    t0 = time.perf_counter(); tab, ms2 = eng.frontend_cspace(download=True); tdl = time.perf_counter() - t0
    print(f"{'':20s} whole configuration space (256^3 voxels x 121 attitudes = {256**3*121:.3g} kernelConv): kernel {ms:8.2f} ms (= {256**3*121/ms*1e3:.3g} conv/s), "
          f"with download of the 268 MB table {tdl*1e3:8.1f} ms; free voxel-attitudes {np.unpackbits(tab.view(np.uint8)).sum() / (256**3 * 121):.3f}", flush=True)
