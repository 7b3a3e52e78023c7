#!/usr/bin/env python
"""Developer tool: integral-sweep throughput (C2 and C3 geometry) for the robot shapes of the reference's shipped configs
(scaled to fit the k = 21 tile), i.e. which sweep_kernel instantiation serves them and how fast it is."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
res = 0.2
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
Ts, Cs = [], []
for b in range(128):
    ext = np.array(occ.shape) * res
    Tb, Cf = synth.random_trajectory(ext, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    Ts.append(Tb); Cs.append(synth.colmajor(Cf))
Ts = np.concatenate(Ts); Cs = np.concatenate(Cs)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
shapes = {
    "Torus (R 1.2, r 0.25)": synth.make_shape("Torus", params=(1.2, 0.25), bound_radius=1.45, bbox=((0, 0, 0), (1.45, 0.25, 1.45))),
    "CappedCone (h 0.9, r 0.6/0.25)": synth.make_shape("CappedCone", params=(0.0, 0.0, -0.9, 0.0, 0.0, 0.9, 0.6, 0.25), bound_radius=1.1, bbox=((0, 0, 0), (0.6, 0.6, 0.9))),
    "BendLinear (1.3, 0.2)": synth.make_shape("BendLinear", params=(1.3, 0.2), bound_radius=1.6, bbox=((0, 0, 0), (1.6, 1.6, 1.6))),
    "SmoothIntersection (x0.4)": synth.make_shape("SmoothIntersection", params=(1.2, 1.2, 0.2, 0.4, 0.1), bound_radius=1.7, bbox=((0, 0, 0), (1.2, 1.2, 1.2))),
    "Box": synth.bench_box_shape(),
    "RoundedCone": synth.bench_rounded_cone_shape(),
}
for name, shape in shapes.items():
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    out = []
    for n_traj, TT, CC, steps in ((1, T, cm, 100), (128, Ts, Cs, 10)):
        d_T = torch.from_numpy(TT).to(dev); d_C = torch.from_numpy(CC).to(dev)
        d_out = torch.zeros(n_traj * eng.out_stride(40), dtype=torch.float64, device=dev)
        for _ in range(3):
            eng.eval_device(n_traj, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            eng.eval_device(n_traj, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        out.append((dt, n_traj * 2600 / dt))
    s = eng.stats()
    # V1 swept-volume sweep with the same robot
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
    e1 = pkg.Engine(cfg1); e1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e1.set_shape(shape)
    way = cm.reshape(3, -1).T.reshape(40, 6, 3)[1:, 0, :]
    M = e1.gather_points(way, 1.4)
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    d_o1 = torch.zeros(e1.out_stride(40), dtype=torch.float64, device=dev)
    for _ in range(2):
        e1.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_o1.data_ptr(), 0, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        e1.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_o1.data_ptr(), 0, st)
    torch.cuda.synchronize(); dv1 = (time.perf_counter() - t0) / 10
    print(f"{'':34s} V1 M={M}: {dv1*1e3:7.3f} ms/step = {M/dv1:10.4g} points/s", flush=True)
    print(f"{name:34s} C2 {out[0][0]*1e6:8.1f} us/step {out[0][1]:10.4g} evals/s | C3 {out[1][0]*1e6:9.1f} us/step {out[1][1]:10.4g} evals/s | pairs/eval {s['pairs']/s['units']:.1f}", flush=True)
