#!/usr/bin/env python
"""Developer tool: per-WORKGROUP view of the sweep kernel's phase timing (ISDF_DEBUG_TIMING=1): which workgroups set the
kernel's makespan, and with what (pose / rows / exact).
Needs the statistics build: `make -C implicit-sdf-planner_amd/csrc EXTRA=-DISDF_SWEEP_STATS=1` after touching the kernel sources."""
import os, sys, ctypes as C
os.environ["ISDF_DEBUG_TIMING"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
shape = synth.bench_box_shape()
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
N = 40; dev = torch.device("cuda", 0)
d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
torch.cuda.synchronize()
lib = eng.lib
lib.isdf_debug_timing.restype = C.c_longlong; lib.isdf_debug_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
n = lib.isdf_debug_timing(eng.h, None, 0); buf = np.zeros(n, dtype=np.uint64); lib.isdf_debug_timing(eng.h, buf.ctypes.data_as(C.c_void_p), n)
ns = N * 65
sc = buf[:ns * 8].reshape(ns, 8).astype(np.int64)
G = ns // 4                                   # workgroups; wave w of WG b has sample b + w * G
t0 = sc[:, 0].min()
end = (sc[:, 6] - t0) / 100.0
start = (sc[:, 0] - t0) / 100.0
pose = sc[:, 2] - sc[:, 1]; rows = np.where(sc[:, 4] > 0, sc[:, 4] - sc[:, 2], 0); drain = np.where(sc[:, 4] > 0, sc[:, 5] - sc[:, 4], sc[:, 5] - sc[:, 2])
wg_end = end.reshape(4, G).max(axis=0); wg_start = start.reshape(4, G).min(axis=0)
wg_rows_max = rows.reshape(4, G).max(axis=0); wg_rows_mean = rows.reshape(4, G).mean(axis=0)
wg_rounds = sc[:, 7].reshape(4, G).sum(axis=0)
order = np.argsort(-wg_end)
print("WG end us: p50 %.2f p90 %.2f p99 %.2f max %.2f | start max %.2f" % (np.percentile(wg_end, 50), np.percentile(wg_end, 90), np.percentile(wg_end, 99), wg_end.max(), wg_start.max()))
print("corr(end, rows_max) %.2f  corr(end, rounds) %.2f  corr(end, start) %.2f" % (np.corrcoef(wg_end, wg_rows_max)[0, 1], np.corrcoef(wg_end, wg_rounds)[0, 1], np.corrcoef(wg_end, wg_start)[0, 1]))
print("slowest 10 WGs: end us, rows_max cyc, rows_mean cyc, exact rounds (sum of 4 waves), drain cyc of wave 0")
for b in order[:10]:
    print("  %6.2f  %6d %6d  %2d  %6d" % (wg_end[b], wg_rows_max[b], wg_rows_mean[b], wg_rounds[b], drain.reshape(4, G)[0, b]))
print("fastest 5:")
for b in order[-5:]:
    print("  %6.2f  %6d %6d  %2d  %6d" % (wg_end[b], wg_rows_max[b], wg_rows_mean[b], wg_rounds[b], drain.reshape(4, G)[0, b]))
print("rounds histogram (sum per WG):", np.bincount(wg_rounds))
for r in sorted(set(wg_rounds)):
    m = wg_rounds == r
    print("  rounds %2d: n %4d  mean end %.2f us  mean rows_max %.0f" % (r, m.sum(), wg_end[m].mean(), wg_rows_max[m].mean()))
# cycles per exact round estimate: regress (drain of the slowest wave) on rounds
