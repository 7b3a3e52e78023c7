#!/usr/bin/env python
"""Developer tool: isdf_eval through the compiled binding on a ctx over 1 / 2 / 8 devices (this GPU listed repeatedly): what the
one-process multi-device step costs the calling thread.  ISDF_MULTI_NO_PULL=1: the peer-copy form."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
import bench
res = 0.2
occ, esdf, T, cm = bench.build_workload(pkg, 40, 256, res)
shape = synth.bench_box_shape()
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01, enable_dyn=1, enable_pos=1)
for nd in (1, 2, 8):
    e = pkg.Engine(cfg, devices=[0] * nd) if nd > 1 else pkg.Engine(cfg)
    e.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e.set_shape(shape)
    r = bench.native_boundary_ms(pkg, e, 40, T, cm, None)
    print(nd, "devices: isdf_eval", None if r is None else round(r[0] * 1e3, 1), "us", "pull" if nd > 1 and os.environ.get("ISDF_MULTI_NO_PULL") != "1" else "")
