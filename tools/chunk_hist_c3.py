#!/usr/bin/env python
"""Developer tool: exact-pass iterations of the batch step C3 (128 trajectories), with the workgroups the dispatch order forms
(four neighbours of the per-trajectory class-sorted order), under the packing policies of the exact phase."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
res = 0.2
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
st = torch.cuda.current_stream().cuda_stream
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(synth.bench_box_shape())
ext = np.array(occ.shape) * res
NT = 128
Ts, Cs = [], []
for b in range(NT):
    Tb, Cf = synth.random_trajectory(ext, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    Ts.append(Tb); Cs.append(synth.colmajor(Cf))
Ts = np.concatenate(Ts); Cs = np.concatenate(Cs)
d_T = torch.from_numpy(Ts).to(dev); d_C = torch.from_numpy(Cs).to(dev)
d_out = torch.zeros(NT * eng.out_stride(40), dtype=torch.float64, device=dev)
for _ in range(3):
    eng.eval_device(NT, 40, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
torch.cuda.synchronize()
lib = eng.lib
lib.isdf_debug_sample_info.restype = C.c_longlong
lib.isdf_debug_sample_info.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
n = lib.isdf_debug_sample_info(eng.h, None, 0)
buf = np.zeros(2 * n, dtype=np.int32)
lib.isdf_debug_sample_info(eng.h, buf.ctypes.data_as(C.c_void_p), 2 * n)
pairs = buf[0::2].astype(int)
print(f"C3: {n} samples, exact pairs/sample mean {pairs.mean():.1f} p50 {np.percentile(pairs, 50):.0f} p90 {np.percentile(pairs, 90):.0f} max {pairs.max()};  zero: {(pairs == 0).mean():.3f}")
# NB: 'pairs' counts exact pairs (inside the cube); candidates per sample are what the chunks hold - same thing here (the list IS the cube test's input)
per = n // NT
cls = np.minimum((pairs + 31) >> 5, 7)
order = np.concatenate([b * per + np.argsort(-cls[b * per:(b + 1) * per], kind="stable") for b in range(NT)])
def items(p4, policy):
    full = sum(p // 64 for p in p4)
    rem = [p % 64 for p in p4 if p % 64]
    q = sum(1 for r in rem if r <= 16); h = sum(1 for r in rem if 16 < r <= 32); w = sum(1 for r in rem if r > 32)
    if policy == "none": return full + len(rem)
    if policy == "same": return full + w + -(-h // 2) + (q > 0)
    if policy == "mixed": return full + w + -(-(h + -(-q // 2)) // 2)
for name, grp in (("plain order", np.arange(n)), ("dispatch order", order)):
    g = pairs[grp][: (n // 4) * 4].reshape(-1, 4)
    for pol in ("none", "same", "mixed"):
        it = np.array([items(list(x), pol) for x in g])
        print(f"   {name:15s} {pol:6s}: iterations {it.sum():8d}  per workgroup mean {it.mean():.2f}  rounds (ceil/4) mean {np.ceil(it / 4).mean():.3f}")
