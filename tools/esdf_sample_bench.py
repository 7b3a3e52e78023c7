#!/usr/bin/env python
"""Developer tool: isdf_esdf_sample_device (GridMap3D::getSDFValue / getSDFValueWithGrad at world points) - the one pure gather of
the path.  n random points on a 256^3 and a 512^3 float32 ESDF, device resident; algorithmic bytes per point = 24 (xyz) + 8 x 4
(corner values) + 8 (value) [+ 24 (gradient)].  usage: esdf_sample_bench.py [n_points]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
st = torch.cuda.current_stream().cuda_stream
for gridn, res in ((256, 0.2), (512, 0.1)):
    occ = synth.random_box_map((gridn,) * 3, res=res, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=0.17)
    eng = pkg.Engine(cfg); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.generate_esdf()
    ext = gridn * res
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    for name, pts in (("uniform random", torch.rand(n, 3, dtype=torch.float64, device="cuda", generator=g) * ext),
                      ("along 4096 straight lines (trajectory-like: consecutive points 2 cm apart)", None)):
        if pts is None:
            m = n // 4096
            a = torch.rand(4096, 1, 3, dtype=torch.float64, device="cuda", generator=g) * ext
            d = torch.nn.functional.normalize(torch.randn(4096, 1, 3, dtype=torch.float64, device="cuda", generator=g), dim=2)
            pts = (a + d * (torch.arange(m, dtype=torch.float64, device="cuda").view(1, m, 1) * 0.02)).reshape(-1, 3).contiguous()
        npt = pts.shape[0]
        dv = torch.empty(npt, dtype=torch.float64, device="cuda"); dg = torch.empty(npt, 3, dtype=torch.float64, device="cuda")
        for grad, scat in ((False, False), (True, False), (True, True)):
            for _ in range(2):
                eng.esdf_sample_device(pts.data_ptr(), npt, dv.data_ptr(), dg.data_ptr() if grad else 0, st, scattered=scat)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                eng.esdf_sample_device(pts.data_ptr(), npt, dv.data_ptr(), dg.data_ptr() if grad else 0, st, scattered=scat)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            B = 24 + 32 + 8 + (24 if grad else 0)
            print(f"{gridn}^3 {name}{' [BRICKED copy]' if scat else ''}: {npt:.3g} points, {'value + gradient' if grad else 'value only      '}: {ms:8.3f} ms = {npt / ms / 1e6:8.2f} G points/s, "
                  f"algorithmic {B} B/point = {npt * B / ms / 1e6:7.1f} GB/s = {npt * B / ms / 1e6 / 8000:.3f} of the 8 TB/s HBM peak", flush=True)
        del pts, dv, dg
    del eng
