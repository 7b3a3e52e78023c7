#!/usr/bin/env python
"""Developer tool: BASELINE.json configs[2] end to end - 128 trajectories (40 pieces each, different start/goal) optimised
concurrently on the shared 256^3 map with the batched L-BFGS driver; compared with optimising them one after another."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload

res = 0.2; N = 40; n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 128; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
occ, esdf, T, cm = build_workload(pkg, N, 256, res)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(synth.bench_box_shape())
ext = np.array(occ.shape) * res
heads, tails, x0s = [], [], []
for b in range(n_traj):
    Tb, Cf = synth.random_trajectory(ext, N, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    c6 = Cf.reshape(N, 6, 3)
    head = np.stack([c6[0, 0], c6[0, 1], 2.0 * c6[0, 2]], axis=1)
    tl = Tb[-1]
    pw = lambda k: sum(np.prod([p - q for q in range(k)]) * c6[N - 1, p] * tl ** (p - k) for p in range(k, 6))
    tail = np.stack([pw(0), pw(1), pw(2)], axis=1)
    eng.set_trajectory(N, head, tail, 1000.0)
    x0s.append(eng.pack_variables(Tb, c6[1:, 0, :])); heads.append(head); tails.append(tail)
x0s = np.array(x0s)
c_first = []
for b in range(min(n_traj, 8)):
    eng.set_trajectory(N, heads[b], tails[b], 1000.0); c_first.append(eng.cost_function(x0s[b])[0])
eng.set_minco_mode(capi.MINCO_DEVICE)          # (the batch runs MINCO on the device by default; the one-at-a-time runs below are held to the same form: bitwise equal iterates)
eng.optimize_lbfgs_batch(N, np.array(heads), np.array(tails), 1000.0, x0s, max_iterations=iters, g_epsilon=0.0, past=0)      # warm-up: buffers' first touch
xs, results, wall = eng.optimize_lbfgs_batch(N, np.array(heads), np.array(tails), 1000.0, x0s, max_iterations=iters, g_epsilon=0.0, past=0)
it = sum(r["iterations"] for r in results); ev = sum(r["evaluations"] for r in results)
print(f"batch of {n_traj} x {N} pieces, {iters} L-BFGS iterations each: {wall:9.1f} ms wall, {results[0]['rounds']} device rounds, "
      f"{it} iterations / {ev} evaluations in total -> {wall / max(1, results[0]['rounds']):.3f} ms per round, "
      f"{1e3 * it / wall:.0f} trajectory-iterations/s, {1e3 * ev * N * 65 / wall:.4g} point-evals/s end to end", flush=True)
print("  cost first -> last (first 8):", ", ".join(f"{a:.3g}->{results[i]['f']:.3g}" for i, a in enumerate(c_first)), flush=True)
t0 = time.perf_counter()
nseq = min(n_traj, 16)
for b in range(nseq):
    eng.set_trajectory(N, heads[b], tails[b], 1000.0)
    x1, r1 = eng.optimize_lbfgs(x0s[b], max_iterations=iters, g_epsilon=0.0, past=0)
    assert np.array_equal(x1, xs[b])
seq = (time.perf_counter() - t0) * 1e3 / nseq
print(f"one at a time: {seq:9.1f} ms per trajectory -> {seq * n_traj:9.1f} ms for {n_traj}  (batched: x{seq * n_traj / wall:.1f})", flush=True)
