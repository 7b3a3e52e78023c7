#!/usr/bin/env python
"""Developer tool: the MESH robot kind on the device - shape plugin throughput (isdf_shape_eval), the V3 integral sweep
(C2 geometry, k = 21) and the V1 swept-volume sweep with meshes of 20 / 720 / 4992 faces, next to the oracle
(brute-force closest point + the vendored HDK winding number) at 30 threads on a sample."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
orc = graft.load_oracle()
from bench import build_workload
res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)
for name, (V, F) in (("L prism", synth.l_prism_mesh()), ("blob", synth.blob_mesh(24, 16, scale=0.9)), ("blob", synth.blob_mesh(64, 40, scale=0.9))):
    nF = len(F)
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    Q = rng.uniform(-2.5, 2.5, (400000, 3))
    eng.shape_eval(Q[:1000])
    t0 = time.perf_counter(); s, g = eng.shape_eval(Q); dt = time.perf_counter() - t0
    o = orc.Oracle(cfg, threads=30); o.set_shape(shape)
    nq = 20000 if nF < 1000 else 4000
    t0 = time.perf_counter(); s0, g0 = o.shape_eval(Q[:nq]); dc = time.perf_counter() - t0
    print(f"mesh {name} {nF:5d} faces: shape_eval (sdf + grad, host in/out) {len(Q)/dt:10.4g} points/s | oracle {nq/dc:10.4g} points/s (1 thread: the call is serial) | max |dsdf| {np.abs(s[:nq]-s0).max():.2e}", flush=True)
    # V3 integral sweep, C2 geometry
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    d_out = torch.zeros(eng.out_stride(40), dtype=torch.float64, device=dev)
    for _ in range(3):
        eng.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); t0 = time.perf_counter(); steps = 20
    for _ in range(steps):
        eng.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    out = d_out.cpu().numpy()
    line = f"  V3 C2 (2600 evals, k=21): {dt*1e6:9.1f} us/step = {2600/dt:10.4g} evals/s  stats {eng.stats()}"
    if nF <= 1000:
        o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
        t0 = time.perf_counter(); c0, gT0, gC0, st0 = o.eval(T, cm); dc = time.perf_counter() - t0
        ref = np.concatenate([[c0], gT0, gC0])
        line += f" | oracle 30 threads {dc*1e3:8.1f} ms = {2600/dc:9.4g} evals/s | rel err {np.linalg.norm(out-ref)/np.linalg.norm(ref):.2e}"
    print(line, flush=True)
    # V1 swept-volume sweep
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
    e1 = pkg.Engine(cfg1); e1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e1.set_shape(shape)
    way = cm.reshape(3, -1).T.reshape(40, 6, 3)[1:, 0, :]
    M = e1.gather_points(way, 1.4)
    for _ in range(2):
        e1.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); t0 = time.perf_counter(); steps = 5
    for _ in range(steps):
        e1.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"  V1 M={M}: {dt*1e3:9.3f} ms/step = {M/dt:10.4g} points/s  stats {e1.stats()}", flush=True)
