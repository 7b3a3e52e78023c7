#!/usr/bin/env python
"""Developer tool (CPU): distribution of the V1 query's work per obstacle point on the v1_bench workload - intervals per
point, sign-descent passes per interval and how many of them are full-step marches (oracle probe orc_swept_trace)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
orc = graft.load_oracle()
from bench import build_workload
res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
shape = synth.bench_rounded_cone_shape() if len(sys.argv) < 2 or sys.argv[1] != "box" else synth.bench_box_shape()
cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
way = cm.reshape(3, -1).T.reshape(40, 6, 3)[1:, 0, :]
pts = orc.gather_points(occ, (0, 0, 0), np.array(occ.shape) * res, res, way, (1.4, 1.4, 1.4))
o = orc.Oracle(cfg1); o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape)
dp = C.POINTER(C.c_double)
M = len(pts)
per_point = []; all_pass = []; all_march = []
out = np.zeros(3 * 64, dtype=np.int32)
for k in range(M):
    p = np.ascontiguousarray(pts[k], dtype=np.float64)
    n = o.lib.orc_swept_trace(o.h, 40, T.ctypes.data_as(dp), cm.ctypes.data_as(dp), p.ctypes.data_as(dp), 64, out.ctypes.data_as(C.POINTER(C.c_int)))
    r = out[:3 * n].reshape(n, 3)
    per_point.append((n, int(r[:, 0].sum()), int(r[:, 2].sum())))
    all_pass += list(r[:, 0]); all_march += list(r[:, 2])
pp = np.array(per_point); ap = np.array(all_pass); am = np.array(all_march)
print(f"M={M} intervals/point mean {pp[:,0].mean():.2f} max {pp[:,0].max()}")
print(f"passes/interval mean {ap.mean():.1f} p50 {np.percentile(ap,50):.0f} p90 {np.percentile(ap,90):.0f} p99 {np.percentile(ap,99):.0f} max {ap.max()}")
print(f"march share of passes: {am.sum()/ap.sum():.2f}")
print(f"passes/point mean {pp[:,1].mean():.1f} p90 {np.percentile(pp[:,1],90):.0f} p99 {np.percentile(pp[:,1],99):.0f} max {pp[:,1].max()} (march part of max point {pp[pp[:,1].argmax(),2]})")
# speculative marching, 6 steps per pass: passes after folding runs of march steps
