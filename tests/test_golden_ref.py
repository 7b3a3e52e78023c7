"""Golden fixtures produced by THE REFERENCE'S OWN CODE (tests/golden/ref_*.npz, tests/golden/make_golden_ref.py: the reference's
sweeps / ESDF / front end compiled from /root/reference and run on seeded inputs).  They travel with the repository, so these
checks need neither /root/reference nor oracle/_ref at run time: the oracle restatement on the CPU, the HIP path on the GPU."""
import os
import sys

import numpy as np
import pytest

from common import REL_TOL, assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import make_golden_ref as mg  # noqa: E402


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _targets(pkg, orc, gpu):
    return (lambda cfg: pkg.Engine(cfg)) if gpu else (lambda cfg: orc.Oracle(cfg, threads=4))


def _eval(t, gpu, T, cm, tstar=None):
    if gpu:
        return t.eval_single(T, cm, tstar=tstar)
    c, gT, gC, _ = t.eval(T, cm, tstar=tstar)
    return c, gT, gC


def _check_v1(pkg, orc, gpu, name):
    capi, synth = pkg.capi, pkg.synth
    w = _load(f"ref_v1_{name}")
    cfg = synth.default_config(capi.V1_SWEPT, kernel_size=9, integral_intervs=16, safety_hor=float(w["cfg_safety_hor"]))
    t = _targets(pkg, orc, gpu)(cfg)
    t.set_grid(w["occ"], (0, 0, 0), float(w["res"]), capi.GRID_OCCUPANCY)
    t.set_shape(synth.make_shape(name, poly_params=mg.POLY))
    t.set_points(w["points"])
    ts = np.zeros(len(w["points"]))
    c, gT, gC = _eval(t, gpu, w["T"], w["coeffs"], tstar=ts)
    c_r = float(w["cost"])
    tol = REL_TOL if gpu else 1e-7
    assert abs(c - c_r) <= tol * max(abs(c_r), 1e-9)
    assert_close(gT, w["gradT"], f"{name} gradT vs reference fixture", tol=tol)
    assert_close(gC, w["gradC"], f"{name} gradC vs reference fixture", tol=tol)
    assert np.allclose(ts, w["tstar"], rtol=0, atol=1e-7)


def _check_integral(pkg, orc, gpu):
    capi, synth = pkg.capi, pkg.synth
    w = _load("ref_integral")
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5, enable_pos=int(w["cfg_enable_pos"]),
                               vmax=float(w["cfg_vmax"]), omgmax=float(w["cfg_omgmax"]), thetamax=float(w["cfg_thetamax"]))
    t = _targets(pkg, orc, gpu)(cfg)
    t.set_grid(w["occ"], (0, 0, 0), float(w["res"]), capi.GRID_OCCUPANCY)
    t.set_shape(synth.make_shape("RoundedCone", poly_params=mg.POLY))
    c, gT, gC = _eval(t, gpu, w["T"], w["coeffs"])
    c_r = float(w["cost"])
    tol = REL_TOL if gpu else 1e-10
    assert c_r > 0 and abs(c - c_r) <= tol * abs(c_r)
    assert_close(gT, w["gradT"], "integral gradT vs reference fixture", tol=tol)
    assert_close(gC, w["gradC"], "integral gradC vs reference fixture", tol=tol)


def _check_frontend(pkg, orc, gpu, name):
    capi, synth = pkg.capi, pkg.synth
    w = _load(f"ref_frontend_{name}")
    kw = dict(params=tuple(w["box_params"])) if name == "Box" else {}
    shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25), **kw)
    t = _targets(pkg, orc, gpu)(synth.default_config(capi.V1_SWEPT))
    t.set_grid(w["occ"], (0, 0, 0), float(w["res"]), capi.GRID_OCCUPANCY)
    t.set_shape(shape)
    t.frontend_build(capi.frontend_config(kernel_size=int(w["kernel_size"])))
    assert np.array_equal(t.frontend_shape_kernels(), w["shape_kernels"])        # byte for byte
    assert np.array_equal(t.frontend_map_kernel(), w["map_kernel"])
    ok, cr, cp, _ = t.frontend_check(w["index"], w["father_roll"], w["father_pitch"])
    assert np.array_equal(ok, w["ok"])
    m = w["ok"] == 1
    assert np.array_equal(cr[m], w["child_roll"][m]) and np.array_equal(cp[m], w["child_pitch"][m])


def _check_astar(pkg, orc, gpu, name):
    """The reference's own AstarPathSearch / getPath / attitude chain (tests/golden/ref_astar_*.npz): the same cells, the same
    (roll, pitch) at every node, the same number of neighbour tests, the same failures."""
    capi, synth = pkg.capi, pkg.synth
    w = _load(f"ref_astar_{name}")
    kw = dict(params=tuple(w["box_params"])) if name == "Box" else {}
    shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25), **kw)
    t = _targets(pkg, orc, gpu)(synth.default_config(capi.V1_SWEPT))
    t.set_grid(w["occ"], (0, 0, 0), float(w["res"]), capi.GRID_OCCUPANCY)
    t.set_shape(shape)
    t.frontend_build(capi.frontend_config(kernel_size=int(w["kernel_size"])))
    at = 0
    assert (w["n"] > 1).sum() >= 8 and (w["n"] == -1).sum() >= 2 and (w["n"] == 1).sum() == 1
    for q, (s, g, n, checks) in enumerate(zip(w["start"], w["goal"], w["n"], w["checks"])):
        out = t.frontend_astar(s, g)
        xyz, rp, st = out[0], out[1], out[-1]
        got_checks = st.checks if gpu else st["checks"]
        assert got_checks == checks, (name, q, got_checks, checks)
        if n < 0:
            assert xyz is None, (name, q)
            continue
        assert xyz is not None and len(xyz) == n, (name, q, None if xyz is None else len(xyz), n)
        assert np.array_equal(xyz, w["xyz"][at:at + n]), (name, q)
        assert np.array_equal(rp, w["rp"][at:at + n]), (name, q)
        if gpu:                       # SE3State::rot of getastarSE3Path = Rx(roll) * Ry(pitch)
            for i in (0, n // 2, n - 1):
                a, b = np.deg2rad(rp[i])
                Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
                Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
                assert np.allclose(out[2][i], Rx @ Ry, rtol=0, atol=1e-15)
        at += n
    assert at == len(w["xyz"])


def _check_callback(pkg, orc, gpu):
    from common import oracle_cost_function
    capi, synth = pkg.capi, pkg.synth
    w = _load("ref_callback_v1")
    N = int(w["N"]); res = float(w["res"])
    kw = dict(kernel_size=9, integral_intervs=16, safety_hor=0.5, vmax=float(w["cfg_vmax"]), omgmax=float(w["cfg_omgmax"]), thetamax=float(w["cfg_thetamax"]))
    shape = synth.make_shape("RoundedCone", poly_params=mg.POLY)
    if gpu:
        eng = pkg.Engine(synth.default_config(capi.V1_SWEPT, **kw))
        eng.set_grid(w["occ"], (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape); eng.set_points(w["points"])
        eng.set_trajectory(N, w["head"], w["tail"], float(w["rho"]))
        cost, g = eng.cost_function(w["x"])
        tol = REL_TOL
    else:
        sw = orc.Oracle(synth.default_config(capi.V1_SWEPT, **kw), threads=4)
        integ = orc.Oracle(synth.default_config(capi.V3_ESDF_TILE, enable_pos=0, **kw), threads=4)
        for o in (sw, integ):
            o.set_grid(w["occ"], (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape)
        sw.set_points(w["points"])
        cost, g = oracle_cost_function(orc, [sw, integ], w["head"], w["tail"], float(w["rho"]), w["x"], N)
        tol = 1e-7
    c_r = float(w["cost"])
    assert abs(cost - c_r) <= tol * abs(c_r), (cost, c_r)
    assert_close(g, w["g"], "callback gradient vs reference fixture", tol=tol)


# ---- CPU: the oracle restatement against the reference's outputs
def test_oracle_callback_vs_reference_fixture(pkg, orc):
    _check_callback(pkg, orc, False)



@pytest.mark.parametrize("name", mg.V1_SHAPES)
def test_oracle_swept_sweep_vs_reference_fixture(pkg, orc, name):
    _check_v1(pkg, orc, False, name)


def test_oracle_integral_sweep_vs_reference_fixture(pkg, orc):
    _check_integral(pkg, orc, False)


def test_oracle_esdf_vs_reference_fixture(pkg, orc):
    w = _load("ref_esdf")
    assert np.array_equal(orc.build_esdf(w["occ"], float(w["res"])).astype(np.float32), w["esdf"])      # generateESDF3d restatement, bit-exact


@pytest.mark.parametrize("name", ["Box", "Torus_big"])
def test_oracle_frontend_vs_reference_fixture(pkg, orc, name):
    _check_frontend(pkg, orc, False, name)


@pytest.mark.parametrize("name", ["Box", "Torus_big"])
def test_oracle_astar_vs_reference_fixture(pkg, orc, name):
    _check_astar(pkg, orc, False, name)


# ---- GPU: the HIP path against the reference's outputs
@pytest.mark.gpu
@pytest.mark.parametrize("name", mg.V1_SHAPES)
def test_hip_swept_sweep_vs_reference_fixture(pkg, orc, product_lib, name):
    _check_v1(pkg, orc, True, name)


@pytest.mark.gpu
def test_hip_integral_sweep_vs_reference_fixture(pkg, orc, product_lib):
    _check_integral(pkg, orc, True)


@pytest.mark.gpu
def test_hip_esdf_vs_reference_fixture(pkg, product_lib):
    capi = pkg.capi
    w = _load("ref_esdf")
    eng = pkg.Engine(pkg.synth.default_config(capi.V3_ESDF_TILE))
    eng.set_grid(w["occ"], (0, 0, 0), float(w["res"]), capi.GRID_OCCUPANCY)
    eng.generate_esdf()
    e, _, _ = eng.get_grid(capi.GRID_ESDF)
    assert np.array_equal(e, w["esdf"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Box", "Torus_big"])
def test_hip_frontend_vs_reference_fixture(pkg, orc, product_lib, name):
    _check_frontend(pkg, orc, True, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Box", "Torus_big"])
def test_hip_astar_vs_reference_fixture(pkg, orc, product_lib, name):
    _check_astar(pkg, orc, True, name)


@pytest.mark.gpu
def test_hip_callback_vs_reference_fixture(pkg, orc, product_lib):
    """isdf_cost_function against costFunctionLmbm assembled from the reference's own MINCO + sweeps + propogateGrad."""
    _check_callback(pkg, orc, True)
