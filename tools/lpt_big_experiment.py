#!/usr/bin/env python
"""Developer experiment: dispatch order of a tile sweep launch that is NOT resident as a whole (400 pieces = 6 500 workgroups,
or a shard of it).  Orders built on the host from the previous step's per-sample candidate counts."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
from lpt_experiment import step_us


def main():
    n_pieces = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    res = 0.2
    occ, esdf, T, cm = build_workload(pkg, n_pieces, 256, res)
    shape = synth.bench_box_shape()
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
    N = T.size
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    os.environ["ISDF_DEBUG_TIMING"] = "1"
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
    for _ in range(2):
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
    cand = (sc[:, 3] >> 44) & 0xFFFFF
    ref = d_out.cpu().numpy().copy()
    del eng
    os.environ["ISDF_DEBUG_TIMING"] = "0"
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    lib = eng.lib
    lib.isdf_debug_set_sample_map.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    nb = (ns + 3) // 4
    print(f"{n_pieces} pieces, {nb} workgroups: plain order {step_us(eng, N, d_T, d_C, d_out, st, n=100):8.2f} us/step")
    key = np.minimum((cand + 31) // 32, 7)
    order = np.argsort(-key, kind="stable")

    def run(label, m):
        mm = np.full(4 * nb, -1, dtype=np.int32); mm[:m.size] = m
        assert np.array_equal(np.sort(mm[mm >= 0]), np.arange(ns))
        lib.isdf_debug_set_sample_map(eng.h, mm.ctypes.data_as(C.c_void_p), mm.size)
        t = step_us(eng, N, d_T, d_C, d_out, st, n=100)
        print(f"  {label:70s} {t:8.2f} us/step  bitwise same: {np.array_equal(d_out.cpu().numpy(), ref)}")

    # rank r = workgroup r % nb, wave r // nb
    m = np.full(4 * nb, -1, dtype=np.int64)
    pos = np.arange(ns)
    m[(pos % 4) * nb + pos // 4] = order                      # workgroup b gets positions 4b .. 4b+3: heaviest workgroups first
    run("heaviest first, 4 neighbours of the sorted order per workgroup", m)
    m = np.full(4 * nb, -1, dtype=np.int64)
    m[:ns] = order                                            # rank = position: workgroup b gets one sample of each quarter
    run("sorted order dealt round-robin (one of each quarter per workgroup)", m)
    m = np.full(4 * nb, -1, dtype=np.int64)
    lo = np.sort(order[ns // 2:]); hi = np.sort(order[:ns // 2])
    m[:ns] = np.concatenate([hi[: hi.size // 2], lo[: lo.size // 2], hi[hi.size // 2:], lo[lo.size // 2:]])
    run("heavier half / lighter half, trajectory order kept, 2 + 2 per workgroup", m)
    m = np.full(4 * nb, -1, dtype=np.int64)
    m[(pos % 4) * nb + pos // 4] = order[::-1]
    run("LIGHTEST first (control)", m)


if __name__ == "__main__":
    main()
