"""Full objective callback through the C-ABI (isdf_cost_function == TrajOptimizer::costFunctionLmbm,
back_end_optimizer.hpp:358-430) vs the same callback composed from oracle pieces.  Tolerance 1e-5 relative."""
import ctypes as C

import numpy as np
import pytest

from common import REL_TOL, assert_close, make_pair, small_world, oracle_cost_function, T_to_tau

pytestmark = pytest.mark.gpu


def _setup(pkg, occ, res, N, seed):
    """Boundary states + a zig-zag of inner waypoints through free space, durations around 1.2 s."""
    rng = np.random.default_rng(seed)
    ext = np.array(occ.shape) * res
    T0, Cf = pkg.synth.random_trajectory(ext, N, seed=seed, piece_T=1.2, margin=4.0, occ=occ, res=res)
    c = Cf.reshape(N, 6, 3)
    head = np.zeros((3, 3)); tail = np.zeros((3, 3))
    head[:, 0] = c[0, 0]
    T0 = T0 * rng.uniform(0.8, 1.3, N)
    # end position of the last piece of the synthetic trajectory
    tl = T0[-1] * 0 + 1.2
    tail[:, 0] = sum(c[N - 1, p] * tl ** p for p in range(6))
    way = c[1:, 0, :].copy()                                   # piece start points = inner waypoints
    return head, tail, way, T0


def _engines(pkg, orc, variant, occ, esdf, res, shape, points=None):
    capi, synth = pkg.capi, pkg.synth
    cfg = synth.default_config(variant, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, occ=occ, esdf=esdf, res=res, points=points)
    return cfg, eng, o


@pytest.mark.parametrize("variant", ["V3", "V2"])
def test_callback_integral(pkg, orc, product_lib, variant):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    N = 6
    head, tail, way, T0 = _setup(pkg, occ, res, N, 21)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    v = capi.V3_ESDF_TILE if variant == "V3" else capi.V2_OCC_TILE
    cfg, eng, o = _engines(pkg, orc, v, occ, esdf, res, shape)
    rho = 3.5
    eng.set_trajectory(N, head, tail, rho)
    assert eng.num_variables() == N + 3 * (N - 1)
    x = eng.pack_variables(T0, way)
    assert_close(x[:N], T_to_tau(T0), "backwardT", tol=1e-12)
    T1, cm1 = eng.unpack_variables(x)
    assert_close(T1, T0, "forwardT(backwardT(T))", tol=1e-9)
    cost, g = eng.cost_function(x)
    c0, g0 = oracle_cost_function(orc, [o], head, tail, rho, x, N)
    assert abs(cost - c0) <= REL_TOL * abs(c0), (cost, c0)
    assert_close(g, g0, "g")
    parts = eng.cost_parts()
    assert parts["integral"] > 0 and parts["swept"] == 0.0
    assert abs(parts["energy"] + parts["integral"] + parts["time"] - cost) <= 1e-9 * abs(cost)
    # lmbm_evaluate_t entry point: same numbers
    gl = np.zeros_like(g)
    dp = C.POINTER(C.c_double)
    cl = eng.lib.isdf_cost_function_lmbm(eng.h, x.ctypes.data_as(dp), gl.ctypes.data_as(dp), x.size)
    assert cl == cost and np.array_equal(gl, g)


def test_callback_swept(pkg, orc, product_lib):
    """The reference's live configuration: swept-volume sweep + integral sweep without collision term."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=4)
    N = 5
    head, tail, way, T0 = _setup(pkg, occ, res, N, 33)
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=4 * res * 1.5)
    assert pts.shape[0] > 0
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg, eng, o1 = _engines(pkg, orc, capi.V1_SWEPT, occ, esdf, res, shape, points=pts)
    cfg2 = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5, enable_pos=0)
    o2 = orc.Oracle(cfg2, threads=4)
    rho = 1.0
    eng.set_trajectory(N, head, tail, rho)
    x = eng.pack_variables(T0 * 0.5, way)          # fast => dynamics penalties active as well
    cost, g = eng.cost_function(x)
    c0, g0 = oracle_cost_function(orc, [o1, o2], head, tail, rho, x, N)
    assert abs(cost - c0) <= REL_TOL * abs(c0), (cost, c0)
    assert_close(g, g0, "g")
    parts = eng.cost_parts()
    assert parts["swept"] > 0 and parts["integral"] > 0


def test_callback_gradient_is_the_derivative(pkg, orc, product_lib):
    """Directional central difference of the device callback itself (V3): g.d == dcost/deps."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=6)
    N = 4
    head, tail, way, T0 = _setup(pkg, occ, res, N, 44)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg, eng, o = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
    eng.set_trajectory(N, head, tail, 2.0)
    x = eng.pack_variables(T0, way)
    cost, g = eng.cost_function(x)
    rng = np.random.default_rng(1)
    for _ in range(3):
        d = rng.normal(0, 1, x.size); d /= np.linalg.norm(d)
        eps = 1e-6
        fd = (eng.cost_function(x + eps * d)[0] - eng.cost_function(x - eps * d)[0]) / (2 * eps)
        assert abs(fd - g @ d) <= 2e-4 * max(1.0, abs(fd)), (fd, g @ d)


def test_callback_errors(pkg, product_lib):
    capi, synth = pkg.capi, pkg.synth
    cfg = synth.default_config(capi.V3_ESDF_TILE, enable_pos=0)
    eng = pkg.Engine(cfg)
    x = np.zeros(7)
    with pytest.raises(pkg.engine.IsdfError) as e:
        eng._N = 2
        eng.cost_function(x)
    assert e.value.code == capi.ISDF_ERR_STATE
    eng.set_trajectory(2, np.zeros((3, 3)), np.eye(3), 1.0)
    with pytest.raises(pkg.engine.IsdfError) as e:
        eng.cost_function(np.zeros(9))
    assert e.value.code == capi.ISDF_ERR_INVALID_ARG
    assert eng.lib.isdf_cost_function_lmbm(eng.h, None, None, 5) == float("inf")


def test_optimize_lbfgs_follows_oracle(pkg, orc, product_lib):
    """The L-BFGS driver on the device callback vs the oracle's restatement of lbfgs_optimize on the oracle callback:
    same iterates for the first iterations (both sides see costs equal to ~1e-14)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("lbfgs_ref", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "lbfgs_ref.py"))
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=8)
    N = 4
    head, tail, way, T0 = _setup(pkg, occ, res, N, 55)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg, eng, o = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
    rho = 10.0
    eng.set_trajectory(N, head, tail, rho)
    x0 = eng.pack_variables(T0, way)
    iters = 6
    fun = lambda x: oracle_cost_function(orc, [o], head, tail, rho, np.asarray(x), N)
    xr, fr, sr, kr, er = ref.lbfgs_optimize(fun, list(x0), max_iterations=iters, g_epsilon=0.0, past=0)
    x, r = eng.optimize_lbfgs(x0, max_iterations=iters, g_epsilon=0.0, past=0)
    c_start = eng.cost_function(x0)[0]
    assert r["status"] == sr and r["iterations"] == kr and r["evaluations"] == er, (r, sr, kr, er)
    assert r["f"] < c_start
    assert_close(x, np.array(xr), "x after %d L-BFGS iterations" % iters, tol=1e-6)
    assert abs(r["f"] - fr) <= 1e-6 * abs(fr)


def test_callback_launch_finish_shards(pkg, orc, product_lib):
    """Multi-GPU form of the callback on one GPU: two sharded contexts, partial sums added by hand where the all-reduce
    would run, both finish with the unsharded (cost, g)."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=10)
    N = 5
    head, tail, way, T0 = _setup(pkg, occ, res, N, 66)
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=4 * res * 1.5)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    for variant, points in ((capi.V3_ESDF_TILE, None), (capi.V1_SWEPT, pts)):
        cfg, full, _ = _engines(pkg, orc, variant, occ, esdf, res, shape, points=points)
        full.set_trajectory(N, head, tail, 2.0)
        x = full.pack_variables(T0 * 0.7, way)
        c_full, g_full = full.cost_function(x)
        engs = []
        for r in range(2):
            _, e, _ = _engines(pkg, orc, variant, occ, esdf, res, shape, points=points)
            e.set_trajectory(N, head, tail, 2.0)
            e.set_shard(r, 2)
            engs.append(e)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipDeviceSynchronize.argtypes = []
        bufs = []
        for e in engs:
            ptr, cnt = e.cost_function_launch(x, 0)
            bufs.append((ptr, cnt))
        assert hip.hipDeviceSynchronize() == 0
        parts = []
        for ptr, cnt in bufs:
            h = np.zeros(cnt)
            assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), cnt * 8, 2) == 0     # device -> host
            parts.append(h)
        assert np.abs(parts[0]).sum() > 0 and np.abs(parts[1]).sum() > 0          # both shards contribute
        total = parts[0] + parts[1]                                               # where the all-reduce runs
        for ptr, cnt in bufs:
            assert hip.hipMemcpy(C.c_void_p(ptr), total.ctypes.data_as(C.c_void_p), cnt * 8, 1) == 0  # host -> device
        for e in engs:
            c, g = e.cost_function_finish(0)
            assert abs(c - c_full) <= 1e-12 * abs(c_full)
            assert_close(g, g_full, "sharded callback g", tol=1e-11)
    # finish without launch is a state error
    with pytest.raises(pkg.engine.IsdfError) as ei:
        engs[0].cost_function_finish(0)
    assert ei.value.code == capi.ISDF_ERR_STATE


@pytest.mark.parametrize("minco", ["device", "host"])
def test_optimize_lbfgs_batch_equals_individual(pkg, orc, product_lib, minco):
    """configs[2] at test size: 6 trajectories with different boundary states optimised concurrently (one batched device
    sweep per round) end on exactly the iterates they reach when optimised one at a time - with the callback's MINCO half on
    the device (one workgroup per trajectory of the round) and with it on the trajectories' host threads."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=16)
    N = 4
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg, eng, _ = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
    eng.set_minco_mode(capi.MINCO_DEVICE if minco == "device" else capi.MINCO_HOST)
    heads, tails, x0s = [], [], []
    rho = 5.0
    for b in range(6):
        head, tail, way, T0 = _setup(pkg, occ, res, N, 70 + b)
        eng.set_trajectory(N, head, tail, rho)
        x0s.append(eng.pack_variables(T0, way))
        heads.append(head); tails.append(tail)
    iters = 5 + np.arange(6) % 3        # not used per trajectory: the batch shares one parameter set
    xs, results, wall = eng.optimize_lbfgs_batch(N, np.array(heads), np.array(tails), rho, np.array(x0s), max_iterations=7,
                                                 g_epsilon=0.0, past=0)
    assert results[0]["rounds"] >= 8
    assert eng.minco_path() == (1 if minco == "device" else 0)
    for b in range(6):
        eng.set_trajectory(N, heads[b], tails[b], rho)
        x1, r1 = eng.optimize_lbfgs(x0s[b], max_iterations=7, g_epsilon=0.0, past=0)
        assert r1["status"] == results[b]["status"] and r1["evaluations"] == results[b]["evaluations"]
        assert r1["f"] == results[b]["f"] and np.array_equal(x1, xs[b]), (b, np.abs(x1 - xs[b]).max())
        assert results[b]["f"] < eng.cost_function(x0s[b])[0]
    # V1 contexts are refused
    cfg1, eng1, _ = _engines(pkg, orc, capi.V1_SWEPT, occ, esdf, res, shape)
    with pytest.raises(pkg.engine.IsdfError) as ei:
        eng1.optimize_lbfgs_batch(N, np.array(heads), np.array(tails), rho, np.array(x0s))
    assert ei.value.code == capi.ISDF_ERR_UNSUPPORTED


def test_progress_hook_on_the_ctx_drivers(pkg, orc, product_lib):
    """isdf_set_progress: the reference's callers watch and stop an optimisation through lbfgs_progress_t / lmbm_progress_t
    (back_end_optimizer.hpp:888-960).  On isdf_optimize_lbfgs the hook sees every iterate and can cancel; in the batch driver ONE
    trajectory is cancelled at its third iteration (status 2, x = that iterate) while the others run to the end on exactly the
    iterates they reach without a hook."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=16)
    N = 4
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg, eng, _ = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
    heads, tails, x0s = [], [], []
    rho = 5.0
    for b in range(5):
        head, tail, way, T0 = _setup(pkg, occ, res, N, 70 + b)
        eng.set_trajectory(N, head, tail, rho)
        x0s.append(eng.pack_variables(T0, way)); heads.append(head); tails.append(tail)
    kw = dict(max_iterations=7, g_epsilon=0.0, past=0)
    xs0, res0, _ = eng.optimize_lbfgs_batch(N, np.array(heads), np.array(tails), rho, np.array(x0s), **kw)
    import threading
    seen, lock = {}, threading.Lock()

    def hook(t, x, g, fx, step, k, ls):
        with lock:
            seen.setdefault(t, []).append((k, x.copy(), fx))
        return t == 2 and k >= 3
    eng.set_progress(hook, n_traj=5)
    xs, res1, _ = eng.optimize_lbfgs_batch(N, np.array(heads), np.array(tails), rho, np.array(x0s), **kw)
    assert sorted(seen) == [0, 1, 2, 3, 4]
    for t in range(5):
        ks = [s[0] for s in seen[t]]
        assert ks == list(range(1, len(ks) + 1))
        if t == 2:
            assert res1[t]["status"] == 2 and res1[t]["iterations"] == 3 and np.array_equal(xs[t], seen[t][2][1]) and res1[t]["f"] == seen[t][2][2]
        else:
            assert res1[t]["status"] == res0[t]["status"] and res1[t]["f"] == res0[t]["f"] and np.array_equal(xs[t], xs0[t])
            assert len(ks) == res1[t]["iterations"]
    # the single-trajectory driver: same hook (instance = trajectory 0's), cancel at k = 2; then the hook removed
    eng.set_trajectory(N, heads[0], tails[0], rho)
    seen.clear()
    eng.set_progress(lambda t, x, g, fx, step, k, ls: k >= 2)
    x1, r1 = eng.optimize_lbfgs(x0s[0], **kw)
    assert r1["status"] == 2 and r1["iterations"] == 2
    eng.set_progress(None)
    x2, r2 = eng.optimize_lbfgs(x0s[0], **kw)
    assert r2["status"] == res0[0]["status"] and r2["iterations"] > 2


# ---- the callback's MINCO half on the device (csrc/minco_dev.hip) vs the host's band LU (csrc/minco_host.hpp) -----------------
def _rel_inf(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("variant,N,piece_scale", [("V3", 1, 1.0), ("V3", 2, 1.0), ("V3", 6, 1.0), ("V3", 40, 1.0), ("V3", 40, 0.2),
                                                   ("V3", 65, 0.5), ("V2", 6, 1.0), ("V1", 6, 1.0), ("V1", 17, 1.0)])
def test_device_minco_callback_equals_host_minco_callback(pkg, orc, product_lib, variant, N, piece_scale, monkeypatch):
    """Same x through both forms of the callback: MINCO in log depth on the device (junction states + parallel cyclic reduction)
    and the reference's band LU on the host.  Different elimination orders, so equal to rounding, not bitwise: 1e-9 relative
    on (cost, g) - and both against the oracle at the north-star tolerance."""
    capi, synth = pkg.capi, pkg.synth
    for k in ("ISDF_HOST_MINCO", "ISDF_DEVICE_MINCO"):           # (the library's own choice is part of what is tested)
        monkeypatch.delenv(k, raising=False)
    occ, esdf, res = small_world(pkg)
    head, tail, way, T0 = _setup(pkg, occ, res, N, 300 + N)
    T0 = T0 * piece_scale
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    v = {"V3": capi.V3_ESDF_TILE, "V2": capi.V2_OCC_TILE, "V1": capi.V1_SWEPT}[variant]
    points = None
    if variant == "V1":
        pts_way = np.vstack([head[:, 0], way, tail[:, 0]])
        points = synth.constraint_points(occ, (0, 0, 0), res, pts_way, half=2.0)
    cfg, eng, o = _engines(pkg, orc, v, occ, esdf, res, shape, points=points)
    rho = 2.5
    eng.set_trajectory(N, head, tail, rho)
    x = eng.pack_variables(T0, way)
    # left to itself the library keeps MINCO on the host for small single-trajectory tile-sweep steps (the faster form there)
    eng.cost_function(x)
    assert eng.minco_path() == (1 if variant == "V1" or N > 64 else 0)
    eng.set_minco_mode(capi.MINCO_DEVICE)
    cost_d, g_d = eng.cost_function(x)
    assert eng.minco_path() == 1 and eng.host_path() == capi.HOST_PATH_DEVICE_CALLBACK
    parts_d = eng.cost_parts()
    cost_d2, g_d2 = eng.cost_function(x)                       # deterministic: the same bits again
    assert cost_d2 == cost_d and np.array_equal(g_d2, g_d)
    eng.set_minco_mode(capi.MINCO_HOST)
    cost_h, g_h = eng.cost_function(x)
    assert eng.minco_path() == 0
    parts_h = eng.cost_parts()
    # (V1: the argmin over t is a chain of accept / reject comparisons with a 1e-5 s stopping rule - coefficients that differ in
    # the 13th digit may stop it one halving apart)
    tol = 1e-6 if variant == "V1" else 1e-9
    assert abs(cost_d - cost_h) <= tol * abs(cost_h), (cost_d, cost_h)
    assert _rel_inf(g_d, g_h) <= tol, _rel_inf(g_d, g_h)
    for k in parts_h:
        assert abs(parts_d[k] - parts_h[k]) <= tol * max(abs(parts_h[k]), 1e-9 * abs(cost_h)), (k, parts_d[k], parts_h[k])
    if variant != "V1":
        c0, g0 = oracle_cost_function(orc, [o], head, tail, rho, x, N)
        assert abs(cost_d - c0) <= REL_TOL * abs(c0)
        assert_close(g_d, g0, "g (device MINCO) vs oracle")
    eng.set_minco_mode(capi.MINCO_DEVICE)
    cost_d3, g_d3 = eng.cost_function(x)
    assert cost_d3 == cost_d and np.array_equal(g_d3, g_d) and eng.minco_path() == 1


def test_device_minco_callback_largest_trajectory_and_fallback(pkg, orc, product_lib):
    """N = 400 (BASELINE configs[3]) is the largest trajectory the device form takes (19 N doubles of LDS); N = 401 falls back to
    the host's band LU without being asked."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    for N, want in ((400, 1), (401, 0)):
        head, tail, way, T0 = _setup(pkg, occ, res, N, 900 + N)
        T0 = T0 * 0.1
        cfg, eng, o = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
        eng.set_trajectory(N, head, tail, 1.0)
        x = eng.pack_variables(T0, way)
        cost, g = eng.cost_function(x)
        assert eng.minco_path() == want
        eng.set_minco_mode(capi.MINCO_HOST)
        cost_h, g_h = eng.cost_function(x)
        assert abs(cost - cost_h) <= 1e-9 * abs(cost_h) and _rel_inf(g, g_h) <= 1e-9, (N, cost, cost_h, _rel_inf(g, g_h))


def test_device_minco_gradient_is_the_derivative(pkg, orc, product_lib):
    """Directional central difference of the device-resident callback (x -> cost, g entirely on the device)."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=5)
    N = 5
    head, tail, way, T0 = _setup(pkg, occ, res, N, 31)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg, eng, o = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
    eng.set_trajectory(N, head, tail, 4.0)
    eng.set_minco_mode(capi.MINCO_DEVICE)
    x = eng.pack_variables(T0, way)
    cost, g = eng.cost_function(x)
    assert eng.minco_path() == 1
    rng = np.random.default_rng(0)
    for _ in range(3):
        d = rng.normal(0, 1, x.size); d /= np.linalg.norm(d)
        eps = 1e-6
        fd = (eng.cost_function(x + eps * d)[0] - eng.cost_function(x - eps * d)[0]) / (2 * eps)
        assert abs(fd - g @ d) <= 2e-4 * max(1.0, abs(fd)), (fd, g @ d)


def test_first_host_mapped_step_of_fresh_ctxs_is_complete(pkg, orc, product_lib):
    """The hand-overs that end in a flag in host-mapped memory (host-direct step, device callback): the FIRST step of a fresh ctx must
    return what every later step returns - the flag can reach the CPU before the result rows (seen on MI355X: a wrong first gradient
    about once in twenty fresh processes; csrc/isdf_host.hip host_rows_wait).  isdf_host_info counts the hand-overs."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    N = 6
    head, tail, way, T0 = _setup(pkg, occ, res, N, 21)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    late = 0
    for mode in (capi.MINCO_HOST, capi.MINCO_DEVICE) * 4:
        cfg, eng, o = _engines(pkg, orc, capi.V3_ESDF_TILE, occ, esdf, res, shape)
        eng.set_trajectory(N, head, tail, 3.5)
        eng.set_minco_mode(mode)
        x = eng.pack_variables(T0, way)
        c1, g1 = eng.cost_function(x)
        assert eng.host_path() in (capi.HOST_PATH_DIRECT_BAR, capi.HOST_PATH_DIRECT_MAPPED, capi.HOST_PATH_DEVICE_CALLBACK)
        info = eng.host_info()
        assert info["handovers"] == 1 and info["late"] in (0, 1)
        c2, g2 = eng.cost_function(x)
        assert c2 == c1 and np.array_equal(g2, g1)
        assert eng.host_info()["handovers"] == 2
        late += eng.host_info()["late"]
    print(f"host-mapped hand-overs of 8 fresh ctxs: {late} had results outstanding when their flag arrived")
