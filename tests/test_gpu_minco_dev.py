"""MINCO on the device (csrc/minco_dev.hip): the two kernels against the reference's OWN MINCO_S3NU
(oracle/_ref/libref_minco.so = utils/minco.hpp:43-198,433-513,530-582,584-654 compiled here) and against the product's host
MINCO bit for bit; and the batch optimizer FORCED onto them (ISDF_BATCH_DEVICE_MINCO=1): iterates, costs and evaluation counts
bitwise equal to the host-MINCO batch and to optimising every trajectory alone."""
import os

import numpy as np
import pytest

from common import assert_close, make_pair, small_world, tau_to_T
from test_gpu_callback import _engines, _setup

pytestmark = pytest.mark.gpu


def _batch(pkg, orc, n_traj, N, seed0):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=16)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg, eng, o = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
    heads, tails, x0s, ways, T0s = [], [], [], [], []
    for b in range(n_traj):
        head, tail, way, T0 = _setup(pkg, occ, res, N, seed0 + b)
        head[:, 1] = np.random.default_rng(seed0 + b).normal(0, 0.3, 3)        # non-zero boundary velocity / acceleration too
        tail[:, 2] = np.random.default_rng(seed0 + 100 + b).normal(0, 0.2, 3)
        eng.set_trajectory(N, head, tail, 1.0)
        x0s.append(eng.pack_variables(T0, way))
        heads.append(head); tails.append(tail); ways.append(way); T0s.append(T0)
    return eng, o, np.array(heads), np.array(tails), np.array(x0s), ways, T0s


@pytest.mark.parametrize("N", [1, 2, 5, 12, 40])
def test_device_minco_vs_reference_minco(pkg, orc, product_lib, N):
    """forward: T = forwardT(tau), coefficients of setParameters; backward: energy + partials + given sweep sums ->
    propogateGrad -> chain rule, per trajectory of a batch, against minco.hpp itself; and == the host MINCO bit for bit."""
    if not os.path.exists(orc.REF_MINCO):
        pytest.skip("oracle/_ref/libref_minco.so not built")
    n_traj = 5
    eng, o, heads, tails, xs, ways, _ = _batch(pkg, orc, n_traj, N, 300 + N)
    rng = np.random.default_rng(N)
    xs = xs.copy()
    xs[:, :N] += rng.normal(0, 0.6, (n_traj, N))              # both branches of the tau -> T map
    if N > 1:
        assert (xs[:, :N] > 0).any() and (xs[:, :N] < 0).any()
    rho = 3.5
    sweep = rng.normal(0, 1.0, (n_traj, 1 + 19 * N))
    T, Cc, cost, g = eng.minco_device(N, heads, tails, rho, xs, sweep_out=sweep)
    for t in range(n_traj):
        tau, xi = xs[t, :N], xs[t, N:]
        Tr = tau_to_T(tau)
        way = xi.reshape(N - 1, 3).T if N > 1 else np.zeros((3, 0))
        cm_r, e_r, gC_e, gT_e = orc.ref_minco(heads[t], tails[t], way, Tr)
        assert np.array_equal(T[t], Tr) or np.abs(T[t] - Tr).max() <= 1e-15 * np.abs(Tr).max()
        assert_close(Cc[t], cm_r, f"N={N} traj {t}: coefficients vs minco.hpp", tol=1e-11)
        gC = gC_e + sweep[t, 1 + N:]; gT = gT_e + sweep[t, 1:1 + N]
        gP, gTt = orc.ref_minco_propagate(heads[t], tails[t], way, Tr, gC, gT)
        total = e_r + sweep[t, 0] + rho * Tr.sum()
        gTt = gTt + rho
        den = (0.5 * tau - 1.0) * tau + 1.0
        g_tau = np.where(tau > 0, gTt * (tau + 1.0), gTt * (1.0 - tau) / (den * den))
        g_r = np.concatenate([g_tau, gP.T.reshape(-1)])
        assert abs(cost[t] - total) <= 1e-10 * max(1.0, abs(total)), (N, t, cost[t], total)
        assert_close(g[t], g_r, f"N={N} traj {t}: (cost, g) vs minco.hpp pieces", tol=1e-9)
        # the host MINCO of isdf_unpack_variables (setParameters): the same bits
        eng.set_trajectory(N, heads[t], tails[t], rho)
        Th, Ch = eng.unpack_variables(xs[t])
        assert np.array_equal(Th, T[t]) and np.array_equal(Ch, Cc[t]), (N, t, np.abs(Ch - Cc[t]).max())


def test_device_minco_callback_equals_host_callback_bitwise(pkg, orc, product_lib):
    """(cost, g) of isdf_cost_function (host MINCO) == device MINCO fed the same sweep sums, bit for bit: the sweep's packed
    output of the host callback's trajectory goes into the device backward pass."""
    capi = pkg.capi
    N, n_traj = 6, 4
    eng, o, heads, tails, xs, _, _ = _batch(pkg, orc, n_traj, N, 500)
    rho = 2.0
    sweeps = []
    host = []
    for t in range(n_traj):
        eng.set_trajectory(N, heads[t], tails[t], rho)
        c, g = eng.cost_function(xs[t])
        host.append((c, g))
        T, cm = eng.unpack_variables(xs[t])
        cs, gT, gC = eng.eval_single(T, cm)
        sweeps.append(np.concatenate([[cs], gT, gC]))
    assert any(s[0] > 0 for s in sweeps)
    _, _, cost, g = eng.minco_device(N, heads, tails, rho, xs, sweep_out=np.array(sweeps))
    for t in range(n_traj):
        assert cost[t] == host[t][0] and np.array_equal(g[t], host[t][1]), (t, cost[t] - host[t][0], np.abs(g[t] - host[t][1]).max())


@pytest.mark.parametrize("n_traj", [6, 16])
def test_batch_optimizer_forced_onto_device_minco(pkg, orc, product_lib, monkeypatch, n_traj):
    """ISDF_BATCH_DEVICE_MINCO=1: only x goes up and (cost, g) comes down; iterates / f / status / evaluation counts are BITWISE
    those of the host-MINCO batch (ISDF_BATCH_DEVICE_MINCO=0) and of isdf_optimize_lbfgs on every trajectory alone."""
    N = 4
    eng, o, heads, tails, x0s, _, _ = _batch(pkg, orc, n_traj, N, 700)
    rho = 5.0
    kw = dict(max_iterations=7, g_epsilon=0.0, past=0)
    monkeypatch.setenv("ISDF_BATCH_DEVICE_MINCO", "1")
    xs_d, res_d, _ = eng.optimize_lbfgs_batch(N, heads, tails, rho, x0s, **kw)
    monkeypatch.setenv("ISDF_BATCH_DEVICE_MINCO", "0")
    xs_h, res_h, _ = eng.optimize_lbfgs_batch(N, heads, tails, rho, x0s, **kw)
    monkeypatch.delenv("ISDF_BATCH_DEVICE_MINCO")
    assert res_d[0]["rounds"] >= 8
    for b in range(n_traj):
        for k in ("f", "status", "iterations", "evaluations"):
            assert res_d[b][k] == res_h[b][k], (b, k, res_d[b], res_h[b])
        assert np.array_equal(xs_d[b], xs_h[b]), (b, np.abs(xs_d[b] - xs_h[b]).max())
    for b in range(0, n_traj, 3):
        eng.set_trajectory(N, heads[b], tails[b], rho)
        x1, r1 = eng.optimize_lbfgs(x0s[b], **kw)
        assert r1["status"] == res_d[b]["status"] and r1["evaluations"] == res_d[b]["evaluations"] and r1["f"] == res_d[b]["f"]
        assert np.array_equal(x1, xs_d[b])
        assert res_d[b]["f"] < eng.cost_function(x0s[b])[0]
