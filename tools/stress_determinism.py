#!/usr/bin/env python
"""Developer tool: the per-step output must be bitwise identical over thousands of back-to-back steps (exercises the
last-block cost sum of tail_kernel under uneven load) - bench workload and a 128-trajectory batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload

res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(synth.bench_box_shape())
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for n_traj in (1, 128):
    N = T.size
    Ts = np.tile(T, n_traj); Cs = np.concatenate([cm * (1 + 1e-3 * b) for b in range(n_traj)])
    d_T = torch.from_numpy(Ts).to(dev); d_C = torch.from_numpy(Cs).to(dev)
    outs = [torch.zeros(n_traj * eng.out_stride(N), dtype=torch.float64, device=dev) for _ in range(2)]
    steps = 3000 if n_traj == 1 else 60
    eng.eval_device(n_traj, N, d_T.data_ptr(), d_C.data_ptr(), outs[0].data_ptr(), 0, st)
    torch.cuda.synchronize()
    ref = outs[0].clone()
    bad = 0
    for k in range(steps):
        eng.eval_device(n_traj, N, d_T.data_ptr(), d_C.data_ptr(), outs[1].data_ptr(), 0, st)
        if k % 10 == 9 or n_traj > 1:
            torch.cuda.synchronize()
            bad += int(not torch.equal(outs[1], ref))
            outs[1].zero_()
    print(f"n_traj={n_traj}: {steps} steps, mismatching snapshots: {bad}, cost[0]={float(ref[0]):.12g}", flush=True)
