"""The device path through the C ABI against the reference's OWN code (oracle/_ref/*.so: line ranges of the reference
compiled from /root/reference against a mini-Eigen stand-in; the prebuilt libraries travel to the GPU box).
No oracle restatement in between: swept-volume sweep, integral sweep, ESDF, and the full costFunctionLmbm assembled from
the reference's MINCO_S3NU + its two sweeps + its propogateGrad."""
import os

import numpy as np
import pytest

from common import REL_TOL, assert_close, small_world, traj, tau_to_T, T_to_tau

pytestmark = pytest.mark.gpu


def _need(orc, *libs):
    for l in libs:
        if not os.path.exists(l):
            pytest.skip(f"{os.path.basename(l)} not built (needs /root/reference at build time)")


def _world(pkg, orc, variant, shape_name, seed, **cfgkw):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=seed)
    kw = dict(kernel_size=9, integral_intervs=16, safety_hor=0.5)
    kw.update(cfgkw)
    cfg = synth.default_config(variant, **kw)
    pp = (0.1, -0.05, 0.08, 20, -10, 40)
    shape = synth.make_shape(shape_name, poly_params=pp)
    ref = orc.RefWorld(cfg, shape_name, shape, res, threads=8)
    ref.set_occupancy(occ, (0, 0, 0), res)
    eng = pkg.Engine(cfg)
    eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    eng.set_shape(shape)
    return cfg, occ, res, ref, eng


@pytest.mark.parametrize("shape_name", ["RoundedCone", "Torus_big", "Box", "SmoothIntersection_big"])
def test_device_swept_sweep_vs_reference(pkg, orc, product_lib, shape_name):
    _need(orc, orc.REF_SWEEPS)
    capi, synth = pkg.capi, pkg.synth
    cfg, occ, res, ref, eng = _world(pkg, orc, capi.V1_SWEPT, shape_name, 3)
    N = 6
    T, cm = traj(pkg, occ, res, N=N, seed=43)
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    M = eng.gather_points(way, 3.0)
    pts = eng.get_points()
    assert M > 50
    ref.set_points(pts)
    c_r, gT_r, gC_r, ts_r = ref.sweep_swept(T, cm)
    ts = np.zeros(M)
    c, gT, gC = eng.eval_single(T, cm, tstar=ts)
    if shape_name in ("RoundedCone", "Box"):
        assert c_r > 0                              # these two are large enough to touch the obstacles of this scenario
    assert abs(c - c_r) <= REL_TOL * max(abs(c_r), 1e-9)
    assert_close(gT, gT_r, "gradT vs reference"); assert_close(gC, gC_r, "gradC vs reference")
    assert np.allclose(ts, ts_r, rtol=0, atol=1e-7)


def test_device_integral_sweep_vs_reference(pkg, orc, product_lib):
    _need(orc, orc.REF_SWEEPS)
    capi = pkg.capi
    cfg, occ, res, ref, eng = _world(pkg, orc, capi.V3_ESDF_TILE, "RoundedCone", 3, enable_pos=0, vmax=1.5, omgmax=0.6, thetamax=0.25)
    T, cm = traj(pkg, occ, res, N=6, seed=11, piece_T=0.45)
    c_r, gT_r, gC_r = ref.sweep_integral(T, cm)
    c, gT, gC = eng.eval_single(T, cm)
    assert c_r > 0 and abs(c - c_r) <= REL_TOL * abs(c_r)
    assert_close(gT, gT_r, "gradT vs reference"); assert_close(gC, gC_r, "gradC vs reference")


def test_device_esdf_vs_reference(pkg, orc, product_lib):
    _need(orc, orc.REF_SWEEPS)
    capi = pkg.capi
    cfg, occ, res, ref, eng = _world(pkg, orc, capi.V3_ESDF_TILE, "RoundedCone", 9)
    eng.generate_esdf()
    e, _, _ = eng.get_grid(capi.GRID_ESDF)
    e_r = ref.generate_esdf()                       # GridMap3D::generateESDF3d, the reference's doubles
    assert np.array_equal(e, e_r.astype(np.float32))


@pytest.mark.parametrize("variant", ["V1", "dyn"])
def test_device_callback_vs_reference_pieces(pkg, orc, product_lib, variant):
    """costFunctionLmbm assembled from the reference's own MINCO_S3NU (setParameters, getEnergy*, propogateGrad) and its
    own sweeps vs isdf_cost_function."""
    _need(orc, orc.REF_SWEEPS, orc.REF_MINCO)
    capi, synth = pkg.capi, pkg.synth
    v = capi.V1_SWEPT if variant == "V1" else capi.V3_ESDF_TILE
    kw = {} if variant == "V1" else dict(enable_pos=0)
    cfg, occ, res, ref, eng = _world(pkg, orc, v, "RoundedCone", 4, vmax=1.5, omgmax=0.6, thetamax=0.25, **kw)
    N = 5
    T0, cm0 = traj(pkg, occ, res, N=N, seed=33, piece_T=0.6)
    c6 = cm0.reshape(3, 6 * N).T.reshape(N, 6, 3)
    head = np.zeros((3, 3)); tail = np.zeros((3, 3))
    head[:, 0] = c6[0, 0]; tail[:, 0] = sum(c6[N - 1, p] * T0[-1] ** p for p in range(6))
    way = c6[1:, 0, :]
    rho = 2.5
    if variant == "V1":
        M = eng.gather_points(way, 3.0)
        ref.set_points(eng.get_points())
    eng.set_trajectory(N, head, tail, rho)
    x = eng.pack_variables(T0, way)
    cost, g = eng.cost_function(x)
    # the reference side
    tau, xi = x[:N], x[N:]
    T = tau_to_T(tau)
    cm_r, e_r, gC, gT = orc.ref_minco(head, tail, xi.reshape(N - 1, 3).T, T)
    total = e_r
    if variant == "V1":
        c1, gT1, gC1, _ = ref.sweep_swept(T, cm_r)
        total += c1; gT = gT + gT1; gC = gC + gC1
    c2, gT2, gC2 = ref.sweep_integral(T, cm_r)
    total += c2; gT = gT + gT2; gC = gC + gC2
    gP, gTt = orc.ref_minco_propagate(head, tail, xi.reshape(N - 1, 3).T, T, gC, gT)
    total += rho * T.sum()
    gTt = gTt + rho
    den = (0.5 * tau - 1.0) * tau + 1.0
    g_tau = np.where(tau > 0, gTt * (tau + 1.0), gTt * (1.0 - tau) / (den * den))
    g_r = np.concatenate([g_tau, gP.T.reshape(-1)])
    assert abs(cost - total) <= REL_TOL * abs(total), (cost, total)
    assert_close(g, g_r, "g vs reference pieces")


@pytest.mark.parametrize("case", ["zero offset", "offset", "at the map's edge"])
def test_device_gather_points_vs_reference(pkg, orc, product_lib, case):
    """isdf_gather_points against the reference's OWN getPointsInAABBOutOfLastOne (PCSmap_manager.h:182-216 in libref_sweeps.so, driven like
    plan_manager.cpp:232-251): the same set of voxel centres, bit for bit; the device orders them by (ix, iy, iz), the reference's
    unordered_map by nothing in particular - compared after sorting both by the reference's voxel id.  Includes the offset quirk (the
    offset moves the current waypoint's box only) and corners projected into the map."""
    _need(orc, orc.REF_SWEEPS)
    capi = pkg.capi
    cfg, occ, res, ref, eng = _world(pkg, orc, capi.V1_SWEPT, "RoundedCone", 12)
    dims = np.array(occ.shape); ext = dims * res
    rng = np.random.default_rng(77)
    occ_ijk = np.argwhere(occ)
    start = (occ_ijk[rng.integers(len(occ_ijk))] + 0.5) * res
    way = np.cumsum(np.vstack([start[None, :], rng.normal(0, 1.0, (8, 3))]), axis=0).clip(0.05 * ext, 0.95 * ext)
    half, offset = np.array([1.5, 1.2, 1.0]), None
    if case == "offset":
        offset = np.array([0.9, -0.7, 0.6])
    elif case == "at the map's edge":
        way[0] = [0.1, 0.2, ext[2] - 0.05]; way[-1] = ext - 0.01; way[4] = [-0.7, ext[1] + 2.0, 0.3]
        offset = np.array([-0.25, 0.6, 0.0])
    M = eng.gather_points(way, half, offset)
    pts = eng.get_points()
    ids_r, xyz_r = ref.gather_points(way, half, offset if offset is not None else (0, 0, 0))
    assert M == ids_r.size and M > 5
    ijk = np.floor(pts / res).astype(np.int64)
    ids_d = ijk[:, 0] + dims[0] * ijk[:, 1] + dims[0] * dims[1] * ijk[:, 2]
    order = np.argsort(ids_d, kind="stable")
    assert np.array_equal(ids_d[order], ids_r.astype(np.int64))
    assert np.array_equal(pts[order], xyz_r)


@pytest.mark.parametrize("name,occupancy", [("Box", 0.24), ("Table", 0.10), ("CSG", 0.12)])
def test_device_astar_vs_reference(pkg, orc, product_lib, name, occupancy):
    """isdf_frontend_astar_search / _path against the reference's OWN AstarPathSearch / getPath / attitude chain
    (front_end_Astar.hpp:160-376 in libref_sweeps.so, driven like PlannerManager::generatePath): the same path, cell for cell, the
    same (roll, pitch) at every node, the same number of neighbour tests, the same failures."""
    _need(orc, orc.REF_SWEEPS)
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occ = synth.random_box_map((44, 40, 14), res=res, occupancy=occupancy, seed=51, edge=(0.5, 2.5))
    kw = dict(params=(0.9, 0.35, 0.25)) if name == "Box" else {}
    shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25), **kw)
    cfg = synth.default_config(capi.V1_SWEPT)
    fe = capi.frontend_config(kernel_size=13 if name == "Box" else 15)
    eng = pkg.Engine(cfg)
    eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape); eng.frontend_build(fe)
    w = orc.RefWorld(cfg, name, shape, res)
    w.set_occupancy(occ, (0, 0, 0), res)
    w.frontend_build(fe, name)
    table, _ = eng.frontend_cspace()
    good = np.argwhere((table != 0).any(axis=-1))
    rng = np.random.default_rng(5)
    found = 0
    for q in range(10):
        a, b = good[rng.choice(len(good), 2, replace=False)]
        s, g = (a + rng.random(3)) * res, (b + rng.random(3)) * res
        xyz, rp, rot, r = eng.frontend_astar(s, g)
        x_r, rp_r, checks_r = w.astar(s, g)
        assert r.checks == checks_r, (name, q, r.checks, checks_r)
        assert (xyz is None) == (x_r is None), (name, q)
        if x_r is not None:
            found += 1
            assert np.array_equal(xyz, x_r) and np.array_equal(rp, rp_r), (name, q)
    assert found >= 5, found
