#!/usr/bin/env python
"""Developer tool: the objective callback (isdf_cost_function, x -> cost, g) with its MINCO half on the device
(csrc/minco_dev.hip) against MINCO on the host (isdf_set_minco_mode(ctx, 1)) - timed natively through the compiled
reference-side binding (tests/native/binding_stub.cpp), at C2 (40 pieces), C4 (400 pieces) and on a swept-volume ctx; the
L-BFGS driver on both; the batched optimiser (configs[2] end to end) on both.  usage: callback_bench.py [quick]"""
import ctypes as C
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
import torch  # noqa: F401  (torch first: see tests/conftest.py)
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
from test_gpu_binding import build_stub

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
res = 0.2
stub = build_stub(pkg, tempfile.mkdtemp())
dpt = C.POINTER(C.c_double)


def ends_of(T, cm, N):
    c6 = cm.reshape(3, 6 * N).T.reshape(N, 6, 3)
    head = np.stack([c6[0, 0], c6[0, 1], 2.0 * c6[0, 2]], axis=1)
    tl = T[-1]
    pw = lambda k: sum(np.prod([p - q for q in range(k)]) * c6[N - 1, p] * tl ** (p - k) for p in range(k, 6))
    return head, np.stack([pw(0), pw(1), pw(2)], axis=1), c6[1:, 0, :]


def time_cb(eng, x0, iters):
    cb = np.zeros(iters); cst = np.zeros(1)
    stub.stub_time_callback(eng.h, x0.size, x0.ctypes.data_as(dpt), cb.size, cb.ctypes.data_as(dpt), cst.ctypes.data_as(dpt))
    return float(np.median(cb)), float(np.percentile(cb, 10)), float(np.percentile(cb, 90)), float(cst[0])


occ = esdf = None
for name, N, variant, piece_T in (("C2: 40 pieces, V3 tile sweep, box", 40, capi.V3_ESDF_TILE, 1.0), ("C4: 400 pieces, V3 tile sweep, box", 400, capi.V3_ESDF_TILE, 0.2),
                                  ("V1 ctx: 40 pieces, swept-volume + integral sweeps, rounded cone", 40, capi.V1_SWEPT, 1.0)):
    if occ is None:
        occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
    ext = np.array(occ.shape) * res
    Tn, Cf = synth.random_trajectory(ext, N, seed=777, piece_T=piece_T, jitter=0.5, margin=4.0, occ=occ, res=res)
    cmn = synth.colmajor(Cf)
    cfg = synth.default_config(variant, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    head, tail, way = ends_of(Tn, cmn, N)
    if variant == capi.V1_SWEPT:
        eng.set_shape(synth.bench_rounded_cone_shape())
        pts = synth.constraint_points(occ, (0, 0, 0), res, np.vstack([head[:, 0], way, tail[:, 0]]), half=3.0)
        eng.set_points(pts)
        name += f", M = {pts.shape[0]}"
    else:
        eng.set_shape(synth.bench_box_shape())
    eng.set_trajectory(N, head, tail, 1000.0)
    x0 = eng.pack_variables(Tn, way)
    out = {}
    for mode, label in ((capi.MINCO_DEVICE, "device"), (capi.MINCO_HOST, "host")):
        eng.set_minco_mode(mode)
        c, g = eng.cost_function(x0)
        med, p10, p90, cst = time_cb(eng, x0, 300 if quick else 1500)
        out[label] = (med, p10, p90, c, g, eng.minco_path(), eng.host_path())
    eng.set_minco_mode(capi.MINCO_AUTO)
    eng.cost_function(x0)
    auto = "device" if eng.minco_path() == 1 else "host"
    d, h = out["device"], out["host"]
    relc = abs(d[3] - h[3]) / abs(h[3]); relg = np.abs(d[4] - h[4]).max() / np.abs(h[4]).max()
    print(f"{name}\n    isdf_cost_function median us: device MINCO {d[0]:7.1f} (p10 {d[1]:.1f}, p90 {d[2]:.1f}; path {d[5]}/{d[6]})  |  host MINCO {h[0]:7.1f} (p10 {h[1]:.1f}, p90 {h[2]:.1f}; path {h[5]}/{h[6]})"
          f"  |  device vs host: cost {relc:.1e}, g {relg:.1e}  |  left to itself the library takes: {auto}", flush=True)
    if variant == capi.V1_SWEPT:
        # isdf_eval itself through the compiled binding: host-direct (BAR in, publish kernel + flag out) - and, in a second ctx, the copy path
        lat = np.zeros(600)
        stub.stub_time_eval.argtypes = [C.c_void_p, C.c_int, dpt, dpt, C.c_int, dpt]
        stub.stub_time_eval(eng.h, N, Tn.ctypes.data_as(dpt), cmn.ctypes.data_as(dpt), lat.size, lat.ctypes.data_as(dpt))
        p_direct = eng.host_path()
        os.environ["ISDF_NO_HOST_DIRECT"] = "1"
        e2 = pkg.Engine(cfg); e2.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e2.set_shape(synth.bench_rounded_cone_shape()); e2.set_points(pts)
        del os.environ["ISDF_NO_HOST_DIRECT"]
        lat2 = np.zeros(600)
        stub.stub_time_eval(e2.h, N, Tn.ctypes.data_as(dpt), cmn.ctypes.data_as(dpt), lat2.size, lat2.ctypes.data_as(dpt))
        print(f"    isdf_eval (host arrays, accumulate) median us: host-direct {np.median(lat):7.1f} (path {p_direct})  |  copy path {np.median(lat2):7.1f} (path {e2.host_path()})", flush=True)
        del e2
    if variant != capi.V1_SWEPT and not quick:
        for mode, label in ((capi.MINCO_DEVICE, "device"), (capi.MINCO_HOST, "host")):
            eng.set_minco_mode(mode)
            eng.optimize_lbfgs(x0, max_iterations=5, g_epsilon=0.0, past=0)
            _, r = eng.optimize_lbfgs(x0, max_iterations=40, g_epsilon=0.0, past=0)
            print(f"    L-BFGS, 40 iterations, {label:6s} MINCO: {r['wall_ms'] / max(1, r['iterations']):.4f} ms per iteration, {r['wall_ms'] / max(1, r['evaluations']):.4f} ms per evaluation "
                  f"({r['evaluations']} evaluations), cost {r['f']:.9g}", flush=True)
    del eng

# configs[2] end to end
if not quick:
    N = 40; n_traj = 128
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(synth.bench_box_shape())
    heads, tails, x0s = [], [], []
    for b in range(n_traj):
        Tb, Cf = synth.random_trajectory(ext, N, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
        head, tail, way = ends_of(Tb, synth.colmajor(Cf), N)
        eng.set_trajectory(N, head, tail, 1000.0)
        x0s.append(eng.pack_variables(Tb, way)); heads.append(head); tails.append(tail)
    x0s = np.array(x0s)
    for groups, slots in (("3", "2"), ("3", "3"), ("4", "3"), ("5", "3"), ("6", "3"), ("4", "4"), ("2", "2"), ("0", "2")):
        os.environ["ISDF_BATCH_GROUPS"] = groups; os.environ["ISDF_BATCH_SLOTS"] = slots
        for mode, label in ((capi.MINCO_DEVICE, "device"), (capi.MINCO_HOST, "host")):
            eng.set_minco_mode(mode)
            walls = []
            for rep in range(3):
                xs, results, wall = eng.optimize_lbfgs_batch(N, np.array(heads), np.array(tails), 1000.0, x0s, max_iterations=30, g_epsilon=0.0, past=0)
                walls.append(wall)
            ev = sum(r["evaluations"] for r in results)
            print(f"C3 end to end, 128 x 40 pieces, 30 iterations, {label:6s} MINCO, {'greedy rounds' if groups == '0' else groups + ' group(s), full rounds'}, {slots} rounds in flight: wall ms {', '.join(f'{w:.1f}' for w in walls)}; {results[0]['rounds']} rounds, {ev} evaluations "
                  f"-> {1e3 * ev * N * 65 / min(walls):.4g} point-evals/s (best), mean f {np.mean([r['f'] for r in results]):.9g}", flush=True)
    os.environ.pop("ISDF_BATCH_GROUPS", None); os.environ.pop("ISDF_BATCH_SLOTS", None)
