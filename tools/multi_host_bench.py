#!/usr/bin/env python
"""Developer tool: isdf_eval through the compiled binding on a ctx over 1 / 2 / 8 devices (this GPU listed repeatedly): what the
one-process multi-device step costs the calling thread.  ISDF_MULTI_NO_PULL=1: the peer-copy form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    # the device-resident step: what the calling thread spends QUEUEING a step (no synchronisation), the steps' throughput, one step's latency
    import time
    dT = torch.from_numpy(T).cuda(); dC = torch.from_numpy(cm).cuda(); dO = torch.zeros(e.out_stride(40), dtype=torch.float64, device="cuda")
    ts_ = torch.cuda.Stream(); torch.cuda.synchronize()
    st = ts_.cuda_stream                 # (not the null stream: its implicit synchronisation with every other stream is not what a caller would time)
    for _ in range(50): e.eval_device(1, 40, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, st)
    torch.cuda.synchronize()
    n = 60
    qs_, th_ = [], []
    for _ in range(7):                   # bursts of 60 steps, the queue drained in between; medians
        t0 = time.perf_counter()
        for _ in range(n): e.eval_device(1, 40, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, st)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        qs_.append(t1 - t0); th_.append(t2 - t0)
    t0 = 0.0; t1 = float(np.median(qs_)); t2 = float(np.median(th_))
    lat = []
    for _ in range(100):
        a = time.perf_counter(); e.eval_device(1, 40, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, st); torch.cuda.synchronize(); lat.append(time.perf_counter() - a)
    print(f"   device-resident: queueing {1e6 * (t1 - t0) / n:.1f} us/step, throughput {1e6 * (t2 - t0) / n:.1f} us/step, one step + synchronise {1e6 * np.median(lat):.1f} us")
