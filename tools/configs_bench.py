#!/usr/bin/env python
"""Developer tool: device-resident throughput of the integral sweep on the other BASELINE.json configurations that fit
one GPU (C3: batch of 128 trajectories; C4: one 400-piece trajectory) next to C2, and of the V1 swept-volume sweep."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload

res = 0.2
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream


def run(name, eng, n_traj, N, Ts, Cs, units, steps=50):
    d_T = torch.from_numpy(Ts).to(dev); d_C = torch.from_numpy(Cs).to(dev)
    d_out = torch.zeros(n_traj * eng.out_stride(N), dtype=torch.float64, device=dev)
    for _ in range(5):
        eng.eval_device(n_traj, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.eval_device(n_traj, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:58s} {units:8d} evals/step  {dt*1e6:10.1f} us/step  {units/dt:12.4g} evals/s   {eng.stats()}", flush=True)


occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
for shape_name, shape in (("box", synth.bench_box_shape()), ("rounded cone", synth.bench_rounded_cone_shape())):
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    run(f"C2  1 x 40 pieces, {shape_name}", eng, 1, 40, T, cm, 2600, steps=200)
    # C3: 128 different trajectories (different start/goal seeds) on the shared map
    Ts, Cs = [], []
    for b in range(128):
        ext = np.array(occ.shape) * res
        Tb, Cf = synth.random_trajectory(ext, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
        Ts.append(Tb); Cs.append(synth.colmajor(Cf))
    run(f"C3  128 x 40 pieces, {shape_name}", eng, 128, 40, np.concatenate(Ts), np.concatenate(Cs), 128 * 2600, steps=20)
    _, _, T4, cm4 = build_workload(pkg, 400, 256, res)
    run(f"C4  1 x 400 pieces (one GPU), {shape_name}", eng, 1, 400, T4, cm4, 26000, steps=50)

# V1: the reference's live sweep, obstacle points gathered on the device around the trajectory's waypoints
cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
eng = pkg.Engine(cfg1); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(synth.bench_rounded_cone_shape())
way = cm.reshape(3, -1).T.reshape(40, 6, 3)[1:, 0, :]
M = eng.gather_points(way, 1.4)
run(f"V1  swept-volume sweep, 40 pieces (40 s), M = {M} points", eng, 1, 40, T, cm, M, steps=20)
