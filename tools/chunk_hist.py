#!/usr/bin/env python
"""Developer tool: candidates (exact pairs) per sample of the C2 step and what they cost in 64-lane chunks - per sample as the
exact pass deals them now, and packed across the four samples of a workgroup."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
res = 0.2
occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
for name, shape in (("box", synth.bench_box_shape()), ("rounded cone", synth.bench_rounded_cone_shape())):
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    eng.eval_single(T, cm)
    lib = eng.lib
    lib.isdf_debug_sample_info.restype = C.c_longlong
    lib.isdf_debug_sample_info.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    n = lib.isdf_debug_sample_info(eng.h, None, 0)
    buf = np.zeros(2 * n, dtype=np.int32)
    lib.isdf_debug_sample_info(eng.h, buf.ctypes.data_as(C.c_void_p), 2 * n)
    pairs = buf[0::2].astype(int)
    ch = np.ceil(pairs / 64).astype(int)
    print(f"{name}: {n} samples, pairs/sample mean {pairs.mean():.1f} p50 {np.percentile(pairs, 50):.0f} p90 {np.percentile(pairs, 90):.0f} max {pairs.max()}")
    print("   chunks per sample now:", np.bincount(ch), "mean", ch.mean(), " lane use", pairs.sum() / max(1, 64 * ch.sum()))
    # packed over groups of four neighbouring samples (plain order)
    g = pairs[: (n // 4) * 4].reshape(-1, 4).sum(axis=1)
    pk = np.ceil(g / 64)
    print("   packed per workgroup of 4: chunks", pk.sum(), "vs", ch[: (n // 4) * 4].sum(), f"({pk.sum() / ch[: (n // 4) * 4].sum():.2f}x)")
    # halves/quarters packing of the LAST block only (canonical trees kept): leftover sizes
    left = pairs % 64
    print("   leftover block sizes: <=16:", (left[left > 0] <= 16).mean(), " <=32:", (left[left > 0] <= 32).mean())
    # aligned power-of-two packing of the last (partial) block: quarter (<= 16), half (<= 32) or whole wavefront; blocks of the
    # same class share a wavefront iteration (canonical summation trees kept: the skipped butterfly levels only add zeros)
    def packed_iters(p4, mixed):
        full = sum(p // 64 for p in p4)
        rem = [p % 64 for p in p4 if p % 64]
        q = sum(1 for r in rem if r <= 16); h = sum(1 for r in rem if 16 < r <= 32); w = sum(1 for r in rem if r > 32)
        if mixed:
            return full + w + -(-(2 * h + q) // 4)
        return full + w + -(-h // 2) + -(-q // 4)
    m = (n // 4) * 4
    now = ch[:m].sum()
    for mixed in (False, True):
        it = sum(packed_iters(pairs[i:i + 4], mixed) for i in range(0, m, 4))
        print(f"   aligned packing ({'mixed' if mixed else 'same-class'} bins): {it} iterations vs {now} ({it / now:.3f}x)")
