"""Optimizer driver (csrc/lbfgs_host.hpp via isdf_lbfgs_minimize; no device needed) vs the oracle's restatement of the
reference's lbfgs_optimize, iterate by iterate, and on textbook problems."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    spec = importlib.util.spec_from_file_location("lbfgs_ref", os.path.join(ROOT, "oracle", "lbfgs_ref.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def rosen(x):
    x = np.asarray(x, dtype=np.float64)
    f = float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
    g = np.zeros_like(x)
    g[:-1] += -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
    return f, g


def quad(x):
    x = np.asarray(x, dtype=np.float64)
    d = np.arange(1, x.size + 1, dtype=np.float64)
    return float(0.5 * np.sum(d * x * x) + np.sum(x)), d * x + 1.0


@pytest.mark.parametrize("fun,n,iters", [(rosen, 6, 25), (quad, 12, 15)])
def test_driver_follows_reference_algorithm(pkg, product_lib, fun, n, iters):
    ref = _ref()
    rng = np.random.default_rng(n)
    x0 = rng.normal(0, 0.02, n) if fun is quad else np.full(n, -0.01) + rng.normal(0, 0.002, n)
    xr, fr, sr, kr, er = ref.lbfgs_optimize(lambda x: fun(x), list(x0), max_iterations=iters, g_epsilon=0.0, past=0)
    x, res = pkg.lbfgs_minimize(fun, x0, lib=product_lib, max_iterations=iters, g_epsilon=0.0, past=0)
    assert res["status"] == sr and res["iterations"] == kr and res["evaluations"] == er
    assert np.allclose(x, xr, rtol=1e-10, atol=1e-12), np.abs(x - np.array(xr)).max()
    assert abs(res["f"] - fr) <= 1e-10 * max(1.0, abs(fr))


def test_plain_lbfgs_converges(pkg, product_lib):
    """reference_patches = 0, weak_wolfe = 1: the unpatched Lewis-Overton L-BFGS solves the textbook problems."""
    x, res = pkg.lbfgs_minimize(rosen, np.full(8, -1.2), lib=product_lib, reference_patches=0, weak_wolfe=1, g_epsilon=1e-8, past=0,
                                max_iterations=2000)
    assert res["status"] == 0, res
    assert np.allclose(x, 1.0, atol=1e-5)
    x, res = pkg.lbfgs_minimize(quad, np.ones(20), lib=product_lib, reference_patches=0, weak_wolfe=1, g_epsilon=1e-7, past=0)
    assert res["status"] == 0 and np.allclose(x, -1.0 / np.arange(1, 21), atol=1e-6)


def test_status_codes(pkg, product_lib):
    ref = _ref()
    # invalid parameter -> the reference's code
    _, res = pkg.lbfgs_minimize(quad, np.ones(3), lib=product_lib, mem_size=0)
    assert res["status"] == ref.ERR_MEM
    # non-finite objective
    _, res = pkg.lbfgs_minimize(lambda x: (float("nan"), np.ones_like(x)), np.ones(3), lib=product_lib, g_epsilon=0.0)
    assert res["status"] == ref.ERR_FUNCVAL
    # stationary start
    _, res = pkg.lbfgs_minimize(lambda x: (0.0, np.zeros_like(x)), np.ones(3), lib=product_lib)
    assert res["status"] == ref.CONVERGENCE and res["evaluations"] == 1
    # delta-based stop
    _, res = pkg.lbfgs_minimize(quad, np.ones(4) * 1e-3, lib=product_lib, g_epsilon=0.0, past=2, delta=0.5)
    assert res["status"] == ref.STOP


def test_progress_hook_observes_and_cancels(pkg, product_lib):
    """isdf_lbfgs_minimize_progress: the reference's lbfgs_progress_t (lbfgs.hpp:256-262) - called once per iteration with the iterate,
    its gradient, the cost, the step, k and the line search's evaluation count; a non-zero return cancels (LBFGS_CANCELED = 2) and
    leaves the iterate the hook was shown in x.  Without a cancel the run is the run without a hook, iterate for iterate."""
    ref = _ref()
    x0 = np.full(6, -0.01)
    seen = []

    def watch(x, g, fx, step, k, ls):
        f, gg = rosen(x)
        assert fx == f and np.array_equal(g, gg) and step > 0 and ls >= 1
        seen.append((k, x.copy(), fx))
        return 0
    xa, ra = pkg.lbfgs_minimize(rosen, x0, lib=product_lib, max_iterations=12, g_epsilon=0.0, past=0)
    xb, rb = pkg.lbfgs_minimize(rosen, x0, lib=product_lib, progress=watch, max_iterations=12, g_epsilon=0.0, past=0)
    assert np.array_equal(xa, xb) and ra["f"] == rb["f"] and ra["evaluations"] == rb["evaluations"] and ra["status"] == rb["status"]
    assert [s[0] for s in seen] == list(range(1, rb["iterations"] + 1))
    assert all(seen[i + 1][2] <= seen[i][2] for i in range(len(seen) - 1))
    # cancel at iteration 5: status 2, five iterations, x = the fifth iterate
    xc, rc = pkg.lbfgs_minimize(rosen, x0, lib=product_lib, progress=lambda x, g, fx, step, k, ls: k >= 5, max_iterations=12, g_epsilon=0.0, past=0)
    assert rc["status"] == ref.CANCELED == 2 and rc["iterations"] == 5
    assert np.array_equal(xc, seen[4][1]) and rc["f"] == seen[4][2]
