#!/usr/bin/env python
"""Developer tool: per-wave phase timing of one optimizer step (ISDF_DEBUG_TIMING=1), bench workload.
wall_clock64 = 100 MHz; clock64 = shader cycles.  The per-wave statistics (row rounds, passes, listed voxels, candidates,
exact iterations) are compiled in only with `make -C implicit-sdf-planner_amd/csrc EXTRA=-DISDF_SWEEP_STATS=1` - zeros otherwise."""
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
    if a.size == 0:
        return "(none)"
    return "min %.2f p50 %.2f p90 %.2f max %.2f" % (a.min(), np.percentile(a, 50), np.percentile(a, 90), a.max())


def main():
    npieces = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    shape_name = sys.argv[2] if len(sys.argv) > 2 else "box"
    res = 0.2
    occ, esdf, T, cm = build_workload(pkg, npieces, 256, res)
    shape = synth.bench_box_shape() if shape_name == "box" else synth.bench_rounded_cone_shape()
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res,
                               weight_p=4000.0, smoothing_eps=0.01)
    eng = pkg.Engine(cfg)
    eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    N = T.size
    dev = torch.device("cuda", 0)
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    lib = eng.lib
    lib.isdf_debug_timing.restype = C.c_longlong
    lib.isdf_debug_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    n = lib.isdf_debug_timing(eng.h, None, 0)
    buf = np.zeros(n, dtype=np.uint64)
    lib.isdf_debug_timing(eng.h, buf.ctypes.data_as(C.c_void_p), n)
    ns = N * 65
    sc = buf[:ns * 8].reshape(ns, 8).astype(np.int64)
    sc = sc[sc[:, 0] > 0]
    tl = buf[ns * 8: ns * 8 + N * 4].reshape(N, 4).astype(np.int64)
    fn = buf[ns * 8 + N * 4:].astype(np.int64)
    t0 = sc[:, 0].min()
    us = lambda w: (w - t0) / 100.0
    print(f"== {npieces} pieces, {shape_name}: {ns} samples")
    print("scan  start us:", pct(us(sc[:, 0])))
    print("scan  end   us:", pct(us(sc[:, 6])))
    print("scan  wave duration us:", pct((sc[:, 6] - sc[:, 0]) / 100.0))
    cyc = sc[:, 5] - sc[:, 1]
    print("scan  wave cycles:", pct(cyc), " => MHz ~", np.median(cyc / np.maximum(1, (sc[:, 6] - sc[:, 0]) / 100.0)))
    print("  pose      cycles:", pct(sc[:, 2] - sc[:, 1]))
    ok = sc[:, 4] > 0
    print("  rows      cycles:", pct(sc[ok, 4] - sc[ok, 2]))
    print("  drain+sum cycles:", pct(sc[ok, 5] - sc[ok, 4]))
    c3 = sc[:, 3]
    print("  row rounds:", pct(c3 & 255), " compaction passes:", pct((c3 >> 8) & 255), " filter passes:", pct((c3 >> 16) & 255))
    print("  listed voxels:", pct((c3 >> 24) & 0xFFFFF), " candidates:", pct((c3 >> 44) & 0xFFFFF))
    print("  exact rounds per wave:", pct(sc[:, 7]), " waves:", sc.shape[0])
    print("tail  start us:", pct(us(tl[:, 0])))
    print("tail  end   us:", pct(us(tl[:, 1])))
    mk = tl[:, 2]
    f = lambda k: ((mk >> (16 * k)) & 0xFFFF) * 16
    print("tail  thread 0 cycles since start: derivatives", pct(f(0)), "| forward+dynamics", pct(f(1)), "| collision terms", pct(f(2)), "| backward", pct(f(3)), "| partials+sums", pct(tl[:, 3] * 16))
    print("tail  block duration us:", pct((tl[:, 1] - tl[:, 0]) / 100.0))



if __name__ == "__main__":
    main()
