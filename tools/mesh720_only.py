import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
V, F = synth.blob_mesh(24, 16, scale=0.9)
shape = synth.make_mesh_shape(V, F)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
d_out = torch.zeros(eng.out_stride(40), dtype=torch.float64, device=dev)
for _ in range(6):
    eng.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
torch.cuda.synchronize()
