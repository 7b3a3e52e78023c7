#!/usr/bin/env python
"""Developer tool: per-wavefront timestamps of one V1 step (ISDF_DEBUG_TIMING=1; wall_clock64 = 100 MHz): where the scan and the
descent launches spend their time.  usage: v1_timing.py [cone|box]
Needs the statistics build: `make -C implicit-sdf-planner_amd/csrc EXTRA=-DISDF_SWEEP_STATS=1` after touching the kernel sources."""
import os, sys, ctypes as C
os.environ["ISDF_DEBUG_TIMING"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload


def pct(a):
    a = np.asarray(a, dtype=float)
    return "(none)" if a.size == 0 else "min %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (a.min(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max())


which = sys.argv[1] if len(sys.argv) > 1 else "cone"
NP = 40
if which in ("c5blob", "c5lthick"):          # the C5-shaped bench entries: 512^3 occupancy @ 0.1 m, 20 pieces, the 720-face blob / the reference's Lthick.obj
    res = 0.1; NP = 20
    occ = synth.random_box_map((512,) * 3, res=res, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
    T, Cf = synth.random_trajectory(np.array(occ.shape) * res, NP, seed=780, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    cm = synth.colmajor(Cf)
else:
    res = 0.2
    occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
if which == "box":
    shape = synth.bench_box_shape()
elif which == "mesh20":
    shape = synth.make_mesh_shape(*synth.l_prism_mesh())
elif which == "c5lthick":
    from benchlib.meshes import reference_mesh
    shape = synth.make_mesh_shape(*reference_mesh("Lthick", 0.83))
elif which in ("mesh720", "c5blob"):
    shape = synth.make_mesh_shape(*synth.blob_mesh(24, 16, scale=0.9))
else:
    shape = synth.bench_rounded_cone_shape()
cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
eng = pkg.Engine(cfg1); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape)
way = cm.reshape(3, -1).T.reshape(NP, 6, 3)[1:, 0, :]
M = eng.gather_points(way, 0.9 if which.startswith('c5') else 1.4)
if os.environ.get('V1_TIMING_SHARD'):      # 'rank/world': one rank's contiguous share of the points
    r_, w_ = os.environ['V1_TIMING_SHARD'].split('/'); eng.set_shard(int(r_), int(w_))
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
d_out = torch.zeros(eng.out_stride(NP), dtype=torch.float64, device=dev)
for _ in range(3 if (which.startswith('mesh') or which.startswith('c5')) else 6):
    eng.eval_device(1, NP, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
torch.cuda.synchronize()
lib = eng.lib
lib.isdf_debug_timing.restype = C.c_longlong
lib.isdf_debug_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
n = lib.isdf_debug_timing(eng.h, None, 0)
buf = np.zeros(n, dtype=np.uint64)
lib.isdf_debug_timing(eng.h, buf.ctypes.data_as(C.c_void_p), n)
N = NP
pt = buf[:M * 4].reshape(M, 4)
t0 = pt[:, 0].astype(np.int64); t1 = pt[:, 1].astype(np.int64); t2 = pt[:, 2].astype(np.int64)
t3 = (pt[:, 3] >> np.uint64(16)).astype(np.int64); npass = (pt[:, 3] & np.uint64(0xFFF)).astype(int); nrng = ((pt[:, 3] >> np.uint64(12)) & np.uint64(0xF)).astype(int)
base = t0.min()
us = lambda x: x / 100.0
print(f"V1 scan kernel, M = {M}: kernel span {us(t3.max() - base):.1f} us")
print("  workgroup start (us after the first)  :", pct(us(t0 - base)))
print("  staging the trajectory (block -> wave) :", pct(us(t1 - t0)))
print("  coarse scan per wavefront              :", pct(us(t2 - t1)))
print("  fine scans per wavefront               :", pct(us(t3 - t2)), " passes:", pct(npass), " intervals:", pct(nrng))
for k in range(0, 8):
    m = npass == k
    if m.any():
        print(f"    {k} fine passes: {m.sum():5d} wavefronts, fine time {pct(us((t3 - t2)[m]))}")
print("  wavefront end (us after kernel start)  :", pct(us(t3 - base)))
off = M * 4 + (N + 1) * 8
tk = buf[off:off + M * 3].reshape(M, 3)
ok = tk[:, 0] > 0
ts, te, tp = tk[ok, 0].astype(np.int64), tk[ok, 1].astype(np.int64), (tk[ok, 2] & np.uint64(0xFFFFFFFF)).astype(int)
b2 = ts.min()
print(f"V1 descent kernel: {ok.sum()} tasks recorded, span {us(te.max() - b2):.1f} us; gap scan end -> descent start {us(b2 - t3.max()):.1f} us")
print("  task start :", pct(us(ts - b2))); print("  task time  :", pct(us(te - ts))); print("  passes     :", pct(tp))
ti = ((tk[ok, 2] >> np.uint64(40)) & np.uint64(0xFFFF)).astype(int)
dur = us(te - ts)
lg = np.argsort(-dur)[:12]
print("  the 12 longest tasks (us, passes, sign-descent iterations):", ", ".join(f"{dur[i]:.0f}/{tp[i]}/{ti[i]}" for i in lg))
print("  iterations per pass:", pct(ti / np.maximum(tp, 1)))
print("  us per pass:", pct(us(te - ts) / np.maximum(tp, 1)))
print("  task end   :", pct(us(te - b2)))
idx = np.nonzero(ok)[0]
for lo, hi in ((0, 4096), (4096, 1 << 30)):      # the tasks that start with the launch / behind them (4 096 wavefronts are resident)
    m = (idx >= lo) & (idx < hi)
    if m.any():
        print(f"  tasks [{lo}, {min(hi, M)}): {m.sum()}  start {pct(us(ts[m] - b2))} | time {pct(dur[m])} | us per pass {pct(dur[m] / np.maximum(tp[m], 1))}")
