#!/usr/bin/env python
"""Developer tool (GPU): per-wavefront phase timestamps of swept_sweep_kernel (ISDF_DEBUG_TIMING=1; wall_clock64 ticks of
100 MHz): staging, coarse scan, scan/descent passes - where the time of the V1 sweep goes."""
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
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
shape = synth.bench_rounded_cone_shape()
cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
eng = pkg.Engine(cfg1); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape)
way = cm.reshape(3, -1).T.reshape(40, 6, 3)[1:, 0, :]
M = eng.gather_points(way, 1.4)
d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
d_out = torch.zeros(eng.out_stride(40), dtype=torch.float64, device=dev)
for _ in range(3):
    eng.eval_device(1, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
torch.cuda.synchronize()
buf = np.zeros(M * 7 + 41 * 8, dtype=np.uint64)
eng.lib.isdf_debug_timing.restype = C.c_longlong
n = eng.lib.isdf_debug_timing(eng.h, buf.ctypes.data_as(C.c_void_p), C.c_longlong(buf.size))
d = buf[:M * 4].reshape(M, 4)
rd = buf[M * 4:M * 4 + 41 * 8].reshape(41, 8).astype(np.int64)
td = buf[M * 4 + 41 * 8:].reshape(M, 3)
t0 = d[:, 0].astype(np.int64); t1 = d[:, 1].astype(np.int64); t2 = d[:, 2].astype(np.int64)
t3 = (d[:, 3] >> np.uint64(16)).astype(np.int64); npass = (d[:, 3] & np.uint64(0xFFF)).astype(np.int64); nrng = ((d[:, 3] >> np.uint64(12)) & np.uint64(0xF)).astype(np.int64)
base = t0.min()
tick = 10.0  # ns per wall_clock64 tick (100 MHz)
us = lambda a: a * tick / 1e3
print(f"M={M}; kernel span {us(t3.max() - base):.1f} us; wave life mean {us((t3 - t0).mean()):.1f} us, max {us((t3 - t0).max()):.1f}")
print(f"staging mean {us((t1 - t0).mean()):.2f} us; coarse scan mean {us((t2 - t1).mean()):.2f} us; rest mean {us((t3 - t2).mean()):.2f} us")
has = npass > 0
print(f"points with passes: {has.sum()}; passes mean {npass[has].mean():.1f} max {npass.max()}")
per = us((t3 - t2)[has]) / npass[has]
print(f"us per pass: mean {per.mean():.2f} p50 {np.percentile(per, 50):.2f} p90 {np.percentile(per, 90):.2f}")
k = np.argmax(t3 - t0)
print(f"longest wave: point {k}, {npass[k]} passes, staging {us(t1[k]-t0[k]):.1f} coarse {us(t2[k]-t1[k]):.1f} rest {us(t3[k]-t2[k]):.1f} us, starts at {us(t0[k]-base):.1f}")
print(f"start times: p50 {us(np.percentile(t0 - base, 50)):.1f} p90 {us(np.percentile(t0 - base, 90)):.1f} max {us((t0 - base).max()):.1f} us")
print("passes histogram (points with >= n passes):", {n: int((npass >= n).sum()) for n in (1, 4, 8, 12, 16, 20, 24, 28)})
print("by intervals: ", {int(r): (int((nrng == r).sum()), round(float(npass[nrng == r].mean()), 1), int(npass[nrng == r].max())) for r in np.unique(nrng)})
order = np.argsort(-(t3 - t0))[:8]
for k in order:
    print(f"  point {k}: intervals {nrng[k]}, passes {npass[k]}, life {us(t3[k]-t0[k]):.1f} us, start {us(t0[k]-base):.1f}")

rb = rd[:, 0].min()
print("reduce kernel, per block (us from first block start): start / loop / butterfly / stores / ticket / end")
for i in (0, 1, 20, 39, 40):
    print(f"  block {i}: " + " ".join(f"{us(rd[i, k] - rb):7.2f}" for k in range(6)))
print(f"  all blocks: latest end {us(rd[:, 5].max() - rb):.2f} us; loop mean {us((rd[:,1]-rd[:,0]).mean()):.2f}; butterfly mean {us((rd[:,2]-rd[:,1]).mean()):.2f}; ticket mean {us((rd[:,4]-rd[:,3]).mean()):.2f}")

tt = td[td[:, 1] > 0]
ts, te, tp = tt[:, 0].astype(np.int64), tt[:, 1].astype(np.int64), (tt[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
heavy = (tt[:, 2] >> np.uint64(32)).astype(np.int64)
b0 = ts.min()
print(f"descent kernel: {len(tt)} tasks ({heavy.sum()} dispatched first), span {us(te.max() - b0):.1f} us; task life mean {us((te - ts).mean()):.1f} max {us((te - ts).max()):.1f}; passes mean {tp.mean():.1f} max {tp.max()}")
print(f"  start times: p50 {us(np.percentile(ts - b0, 50)):.1f} p90 {us(np.percentile(ts - b0, 90)):.1f} max {us((ts - b0).max()):.1f} us;  us per pass: p50 {np.percentile(us(te - ts) / np.maximum(tp, 1), 50):.2f} p90 {np.percentile(us(te - ts) / np.maximum(tp, 1), 90):.2f}")
for k in np.argsort(-(te - b0))[:8]:
    print(f"  task {k}: heavy {heavy[k]} passes {tp[k]} start {us(ts[k] - b0):.1f} end {us(te[k] - b0):.1f} ({us(te[k]-ts[k])/max(tp[k],1):.2f} us/pass)")
