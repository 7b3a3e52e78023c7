#!/usr/bin/env python
"""Developer tool: V1 swept-volume sweep, device vs the oracle's OpenMP restatement on the same box (bench map, 40 pieces)."""
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
shapes = (("rounded cone", synth.bench_rounded_cone_shape()), ("box", synth.bench_box_shape()))
if "--only-cone" in sys.argv:
    shapes = shapes[:1]
if "--mesh720" in sys.argv:
    shapes = (("mesh 720 faces", synth.make_mesh_shape(*synth.blob_mesh(24, 16, scale=0.9))),)
if "--mesh20" in sys.argv:
    shapes = (("mesh 20 faces", synth.make_mesh_shape(*synth.l_prism_mesh())),)
for shape_name, shape in shapes:
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
    eng = pkg.Engine(cfg1); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape)
    way = cm.reshape(3, -1).T.reshape(40, 6, 3)[1:, 0, :]
    M = eng.gather_points(way, 1.4)
    pts = eng.get_points()
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    d_out = torch.zeros(eng.out_stride(40), dtype=torch.float64, device=dev)
    for _ in range(3):
        eng.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); eng.profile_enable(1); t0 = time.perf_counter()
    for _ in range(20 if 'mesh' not in shape_name else 4):
        eng.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (20 if 'mesh' not in shape_name else 4)
    n, kms = eng.profile_read()
    out = d_out.cpu().numpy()
    if "--no-oracle" in sys.argv:
        print(f"V1 {shape_name:13s} M={M}: device {dt*1e3:7.3f} ms/step (sweep kernel {kms:7.3f} ms) = {M/dt:10.4g} points/s | stats {eng.stats()}", flush=True)
        continue
    o = orc.Oracle(cfg1, threads=30); o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.set_points(pts)
    c0, gT0, gC0, st0 = o.eval(T, cm)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        o.eval(T, cm)
    dc = (time.perf_counter() - t0) / reps
    ref = np.concatenate([[c0], gT0, gC0])
    print(f"V1 {shape_name:13s} M={M}: device {dt*1e3:7.3f} ms/step (sweep kernel {kms:7.3f} ms) = {M/dt:10.4g} points/s | oracle 30 threads {dc*1e3:8.1f} ms/step = {M/dc:10.4g} points/s | "
          f"x{dc/dt:6.1f} | rel err {np.linalg.norm(out-ref)/np.linalg.norm(ref):.2e} | stats {eng.stats()}", flush=True)
