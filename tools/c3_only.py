#!/usr/bin/env python
"""Developer tool: the batch configuration C3 alone (128 trajectories x 40 pieces, box robot) - a clean target for rocprofv3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
res = 0.2
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
ts_ = torch.cuda.Stream(); st = ts_.cuda_stream       # (not the null stream: it synchronises implicitly with every other stream)
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
eng = pkg.Engine(cfg, devices=[0] * int(os.environ.get('C3_SPLIT', '2')) if '--split' in sys.argv else None); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(synth.bench_box_shape())
ext = np.array(occ.shape) * res
Ts, Cs = [], []
for b in range(128):
    Tb, Cf = synth.random_trajectory(ext, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    Ts.append(Tb); Cs.append(synth.colmajor(Cf))
Ts = np.concatenate(Ts); Cs = np.concatenate(Cs)
d_T = torch.from_numpy(Ts).to(dev); d_C = torch.from_numpy(Cs).to(dev)
d_out = torch.zeros(128 * eng.out_stride(40), dtype=torch.float64, device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
for _ in range(steps):
    eng.eval_device(128, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    eng.eval_device(128, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
torch.cuda.synchronize()
import hashlib
print(f"C3: {(time.perf_counter() - t0) / steps * 1e6:.1f} us/step   sha1(out) {hashlib.sha1(d_out.cpu().numpy().tobytes()).hexdigest()[:16]}")
