#!/usr/bin/env python
"""Developer tool: what a reference caller sees through the drop-in boundary at C2 (1 x 40 pieces, 256^3, k = 21, box robot):
isdf_eval (host arrays in, accumulate out) and isdf_cost_function (x -> cost, g), host-direct against the copy path
(ISDF_NO_HOST_DIRECT=1 in a second process), next to the device-resident step."""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for env in ({}, {"ISDF_NO_HOST_DIRECT": "1"}):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env})
    sys.exit(0)
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
N = 40
for shape_name, shape in (("box", synth.bench_box_shape()), ("rounded cone", synth.bench_rounded_cone_shape())):
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
    for _ in range(20):
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500):
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); dev_us = (time.perf_counter() - t0) / 500 * 1e6
    ref = d_out.cpu().numpy().copy()
    c, gT, gC = eng.eval_single(T, cm)
    err = np.linalg.norm(np.concatenate([[c], gT, gC]) - ref) / np.linalg.norm(ref)
    # the call a reference caller makes is C++: timed natively through the compiled binding (tests/native/binding_stub.cpp)
    import ctypes as C
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_binding import build_stub
    import tempfile
    stub = build_stub(pkg, tempfile.mkdtemp())
    dpt = C.POINTER(C.c_double)
    lat = np.zeros(2000)
    assert stub.stub_time_eval(eng.h, N, T.ctypes.data_as(dpt), cm.ctypes.data_as(dpt), lat.size, lat.ctypes.data_as(dpt)) == 0
    # the callback
    c6 = cm.reshape(3, 6 * N).T.reshape(N, 6, 3)
    head = np.stack([c6[0, 0], c6[0, 1], 2.0 * c6[0, 2]], axis=1)
    tl = T[-1]
    pw = lambda k: sum(np.prod([p - q for q in range(k)]) * c6[N - 1, p] * tl ** (p - k) for p in range(k, 6))
    tail = np.stack([pw(0), pw(1), pw(2)], axis=1)
    eng.set_trajectory(N, head, tail, 1000.0)
    x0 = eng.pack_variables(T, c6[1:, 0, :])
    cb = np.zeros(2000); cst = np.zeros(1)
    stub.stub_time_callback(eng.h, x0.size, x0.ctypes.data_as(dpt), cb.size, cb.ctypes.data_as(dpt), cst.ctypes.data_as(dpt))
    mode = "copy path (ISDF_NO_HOST_DIRECT=1)" if os.environ.get("ISDF_NO_HOST_DIRECT") == "1" else "host-direct"
    print(f"{shape_name:12s} {mode:34s}: device-resident step {dev_us:6.1f} us | isdf_eval median {np.median(lat):6.1f} us (p10 {np.percentile(lat,10):.1f}, p90 {np.percentile(lat,90):.1f}) "
          f"= {2600/np.median(lat)*1e6:.3g} evals/s | isdf_cost_function median {np.median(cb):6.1f} us | eval vs device-resident rel diff {err:.1e}", flush=True)
