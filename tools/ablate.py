#!/usr/bin/env python
"""Ablation timer for the V3 tile kernel on the bench workload (developer tool, not part of the product).
Prints mean kernel time (HIP events around tile_sweep_kernel) for several configurations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch

pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bench import build_workload

def run(name, esdf, T, cm, shape, steps=30, **cfgkw):
    res = 0.2
    kw = dict(kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    kw.update(cfgkw)
    cfg = synth.default_config(capi.V3_ESDF_TILE, **kw)
    eng = pkg.Engine(cfg)
    eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    N = T.size
    dev = torch.device("cuda", 0)
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5): eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); eng.profile_enable(4, secondary=True)
    t0 = time.perf_counter()
    for _ in range(steps): eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    n, ms = eng.profile_read(); ms2 = eng.profile_read_secondary()
    print(f"{name:36s} scan {ms*1e3:8.2f} us  exact {ms2*1e3:8.2f} us  step {(t1-t0)/steps*1e6:9.2f} us   stats {eng.stats()}", flush=True)

if __name__ == "__main__":
    npieces = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    occ, esdf, T, cm = build_workload(pkg, npieces, 256, 0.2)
    box = synth.bench_box_shape()
    run("full (box)", esdf, T, cm, box)
    run("no collision term (pose+tail only)", esdf, T, cm, box, enable_pos=0)
    run("nothing qualifies (stage only)", esdf, T, cm, box, occ_thresh=-1.0)
    run("safety_hor=-10 (pairs, no grads)", esdf, T, cm, box, safety_hor=-10.0)
    run("cull on", esdf, T, cm, box, enable_cull=1)
    run("rounded cone", esdf, T, cm, synth.bench_rounded_cone_shape())
    run("k=13", esdf, T, cm, box, kernel_size=13)
    nob = synth.make_shape("Box", params=(1.6, 0.3, 0.3), grad_mode=capi.GRAD_CENTRAL)
    run("box without bbox (no row pruning)", esdf, T, cm, nob)
