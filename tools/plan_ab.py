import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import __graft_entry__ as graft
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
from lpt_experiment import step_us
res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
for shape_name in ("box", "cone"):
    shape = synth.bench_box_shape() if shape_name == "box" else synth.bench_rounded_cone_shape()
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
    N = T.size
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
    for rep in range(2):
        for fl, what in ((0, "plan built every step"), (2, "plan_piece skipped (orders frozen)"), (6, "plan_piece and class stores skipped"), (4, "class stores skipped")):
            os.environ["ISDF_DEBUG_FLAGS"] = str(fl)
            print(f"{shape_name}: {what:40s} {step_us(eng, N, d_T, d_C, d_out, st):7.2f} us/step", flush=True)
