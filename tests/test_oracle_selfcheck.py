"""Self-consistency of the CPU oracle (the reference ships no tests for this path): finite-difference
gradient checks of the whole cost, analytic known answers, ESDF == exact EDT, MINCO adjoint vs finite differences."""
import numpy as np
import pytest

from benchlib.meshes import reference_mesh

from common import small_world, traj


def _fd_check(o, T, cm, gT, gC, n=10, seed=1, tol=2e-6):
    rng = np.random.default_rng(seed)
    idx = rng.choice(cm.size, n, replace=False)
    scale = max(np.abs(gC).max(), 1.0)
    for k in idx:
        h = 1e-6
        cp = cm.copy(); cp[k] += h
        cn = cm.copy(); cn[k] -= h
        fd = (o.eval(T, cp)[0] - o.eval(T, cn)[0]) / (2 * h)
        assert abs(fd - gC[k]) <= tol * scale + 1e-6 * abs(fd), (k, fd, gC[k])
    for k in range(T.size):
        h = 1e-6
        Tp = T.copy(); Tp[k] += h
        Tn = T.copy(); Tn[k] -= h
        fd = (o.eval(Tp, cm)[0] - o.eval(Tn, cm)[0]) / (2 * h)
        assert abs(fd - gT[k]) <= tol * max(np.abs(gT).max(), 1.0) + 1e-5 * abs(fd), (k, fd, gT[k])


@pytest.mark.parametrize("variant", ["V3", "V2"])
def test_integral_sweep_gradient_matches_finite_differences(pkg, orc, variant):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res)
    v = capi.V3_ESDF_TILE if variant == "V3" else capi.V2_OCC_TILE
    cfg = synth.default_config(v, kernel_size=9, integral_intervs=16, safety_hor=0.5, vmax=2.5, omgmax=1.5)
    o = orc.Oracle(cfg, threads=4)
    o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    o.set_shape(synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6)))
    c0, gT, gC, st = o.eval(T, cm)
    assert c0 > 0 and st[3] > 0
    _fd_check(o, T, cm, gT, gC)


def test_v2_equals_v3_and_cull_is_exact(pkg, orc):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, occupancy=0.05, seed=7)
    T, cm = traj(pkg, occ, res, seed=13)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6), bound_radius=1.9)
    outs = []
    for v, cull in ((capi.V3_ESDF_TILE, 0), (capi.V2_OCC_TILE, 0), (capi.V3_ESDF_TILE, 1)):
        cfg = synth.default_config(v, kernel_size=9, integral_intervs=16, safety_hor=0.5, enable_cull=cull)
        o = orc.Oracle(cfg)
        o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
        o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
        o.set_shape(shape)
        outs.append(o.eval(T, cm))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][2], outs[1][2])
    assert outs[2][3][1] > 0                                   # something was culled ...
    assert outs[2][0] == outs[0][0] and np.array_equal(outs[2][2], outs[0][2])   # ... without changing the result


def test_esdf_is_exact_edt(pkg, orc):
    from scipy import ndimage
    occ = pkg.synth.random_box_map((30, 24, 20), res=0.25, occupancy=0.1, seed=5, edge=(0.5, 1.5))
    e = orc.build_esdf(occ, 0.25)
    assert np.array_equal(e, 0.25 * ndimage.distance_transform_edt(occ == 0))
    assert np.array_equal(pkg.synth.esdf_from_occupancy(occ, 0.25), e.astype(np.float32))
    assert np.all((e == 0) == (occ != 0))                      # unsigned: zero exactly on occupied voxels


def test_trilinear_sampler(pkg, orc):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    o = orc.Oracle(synth.default_config())
    o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    # at a voxel centre the sampler returns the voxel value
    d, g = o.esdf_trilinear((10.25, 7.75, 5.25))
    assert abs(d - float(esdf[20, 15, 10])) < 1e-12
    # analytic gradient of the interpolant vs finite differences inside a cell
    p = np.array([10.4, 7.9, 5.35])
    d0, g0 = o.esdf_trilinear(p)
    for a in range(3):
        e = np.zeros(3); e[a] = 1e-6
        fd = (o.esdf_trilinear(p + e)[0] - o.esdf_trilinear(p - e)[0]) / 2e-6
        assert abs(fd - g0[a]) < 1e-6


def test_smoothed_l1_and_quat(orc):
    assert orc.smoothed_l1(-1.0, 0.01) == (0.0, 0.0)
    f, df = orc.smoothed_l1(0.5, 0.01)
    assert f == 0.5 - 0.005 and df == 1.0
    f, df = orc.smoothed_l1(0.01, 0.01)                        # C1 junction: f = mu/2, f' = 1
    assert abs(f - 0.005) < 1e-15 and abs(df - 1.0) < 1e-12
    q = np.array([0.9, 0.1, -0.3, 0.0]); q /= np.linalg.norm(q)
    R = orc.quat_to_rot(q)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14


def test_shape_known_answers(pkg, orc):
    synth, capi = pkg.synth, pkg.capi
    o = orc.Oracle(synth.default_config())
    rng = np.random.default_rng(0)
    P = rng.uniform(-3, 3, (500, 3))
    # Ball: |p| - r, gradient p/|p|
    o.set_shape(synth.make_shape("Ball", params=(0.7,)))
    s, g = o.shape_eval(P)
    assert np.allclose(s, np.linalg.norm(P, axis=1) - 0.7, atol=1e-15)
    assert np.allclose(g, P / np.linalg.norm(P, axis=1, keepdims=True), atol=1e-15)
    # Box (half extents) against the closed form; central-difference gradient is a unit vector
    o.set_shape(synth.make_shape("Box", params=(1.0, 0.5, 0.25), grad_mode=capi.GRAD_CENTRAL))
    s, g = o.shape_eval(P)
    q = np.abs(P) - np.array([1.0, 0.5, 0.25])
    ref = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
    assert np.allclose(s, ref, atol=1e-15)
    assert np.allclose(np.linalg.norm(g, axis=1), 1.0, atol=1e-9)
    # Box default gradient is the reference's forward difference (dx = 0.01, not normalised)
    o.set_shape(synth.make_shape("Box", params=(1.0, 0.5, 0.25)))
    s2, g2 = o.shape_eval(P[:5])
    for k in range(5):
        for a in range(3):
            e = np.zeros(3); e[a] = 0.01
            qq = np.abs(P[k] + e) - np.array([1.0, 0.5, 0.25])
            fwd = np.linalg.norm(np.maximum(qq, 0)) + min(qq.max(), 0)
            assert abs(g2[k, a] - (fwd - s2[k]) / 0.01) < 1e-12
    # Torus: distance to the circle of radius R in the xz-plane
    o.set_shape(synth.make_shape("Torus"))
    s, _ = o.shape_eval(P)
    assert np.allclose(s, np.hypot(np.hypot(P[:, 0], P[:, 2]) - 2.5, P[:, 1]) - 0.3, atol=1e-14)
    # body offset: (p - trans) * Rotate, yaw*pitch*roll
    sh = synth.make_shape("Torus", poly_params=(0.3, -0.2, 0.1, 20, -35, 120))
    o.set_shape(sh)
    s_off, _ = o.shape_eval(P)
    Rm = synth.poly_rotation(20, -35, 120)
    Pb = (P - np.array([0.3, -0.2, 0.1])) @ Rm
    assert np.allclose(s_off, np.hypot(np.hypot(Pb[:, 0], Pb[:, 2]) - 2.5, Pb[:, 1]) - 0.3, atol=1e-13)


@pytest.mark.parametrize("name", ["Torus", "Cappedtorus", "RoundedCone", "WireframeBox", "BendLinear", "TwistBox", "BendBox",
                                  "Table", "Trefoil", "SmoothDifference", "SmoothIntersection", "CSG", "CappedCone"])
def test_registered_shapes_are_finite_and_1_lipschitz_ish(pkg, orc, name):
    o = orc.Oracle(pkg.synth.default_config())
    o.set_shape(pkg.synth.make_shape(name))
    rng = np.random.default_rng(3)
    P = rng.uniform(-5, 5, (400, 3))
    s, g = o.shape_eval(P)
    assert np.all(np.isfinite(s))
    ok = np.isfinite(g).all(axis=1)
    assert ok.mean() > 0.95
    assert np.allclose(np.linalg.norm(g[ok], axis=1), 1.0, atol=1e-6)   # DEFINE_USEFUL_FUNCTION normalises


def test_minco_continuity_energy_and_adjoint(pkg, orc):
    rng = np.random.default_rng(2)
    N = 5
    T = rng.uniform(0.6, 1.8, N)
    head = np.zeros((3, 3)); head[:, 0] = [1, 2, 3]; head[:, 1] = [0.3, -0.1, 0.2]
    tail = np.zeros((3, 3)); tail[:, 0] = [8, 5, 4]
    P = head[:, :1] + (tail[:, :1] - head[:, :1]) * np.linspace(0, 1, N + 1)[1:-1] + rng.normal(0, 0.5, (3, N - 1))
    C = orc.minco_coeffs(head, tail, P, T)
    assert np.allclose(C, pkg.synth.minco_coeffs(head, tail, P, T), rtol=1e-9, atol=1e-9)   # banded LU vs dense solve
    # boundary + waypoint interpolation + C3 continuity at the junctions (through the trajectory evaluator)
    cm = pkg.synth.colmajor(C)
    st, piece, tl = orc.traj_eval(T, cm, 0.0)
    assert np.allclose(st[0], head[:, 0]) and np.allclose(st[1], head[:, 1])
    tc = np.cumsum(T)
    for i in range(N - 1):
        a, pa, _ = orc.traj_eval(T, cm, tc[i])            # junction time belongs to the EARLIER piece
        b, pb, _ = orc.traj_eval(T, cm, tc[i] + 1e-9)
        assert pa == i and pb == i + 1
        assert np.allclose(a[0], P[:, i], atol=1e-9)
        assert np.allclose(a, b, atol=1e-5)
    # energy partials vs finite differences
    e, gC, gT = orc.minco_energy(cm, T)
    k = 4 * 6 + 5
    h = 1e-6
    cp = cm.copy(); cp[k] += h; cn = cm.copy(); cn[k] -= h
    assert abs((orc.minco_energy(cp, T)[0] - orc.minco_energy(cn, T)[0]) / (2 * h) - gC[k]) < 1e-5 * max(1, abs(gC[k]))
    # adjoint: d energy(q, T) / d(waypoints, durations)
    gP, gTt = orc.minco_propagate(head, tail, P, T, gC, gT)
    def energy(Pp, Tt):
        return orc.minco_energy(pkg.synth.colmajor(orc.minco_coeffs(head, tail, Pp, Tt)), Tt)[0]
    for (r, c) in ((0, 1), (2, 3)):
        Pp = P.copy(); Pp[r, c] += 1e-6; Pn = P.copy(); Pn[r, c] -= 1e-6
        fd = (energy(Pp, T) - energy(Pn, T)) / 2e-6
        assert abs(fd - gP[r, c]) < 1e-5 * max(1.0, abs(fd))
    for i in (0, 3):
        Tp = T.copy(); Tp[i] += 1e-6; Tn = T.copy(); Tn[i] -= 1e-6
        fd = (energy(P, Tp) - energy(P, Tn)) / 2e-6
        assert abs(fd - gTt[i]) < 1e-5 * max(1.0, abs(fd))


def test_swept_volume_query_properties(pkg, orc):
    """V1 internals: t* is a local minimiser in its interval, min-sdf <= coarse samples, no interval -> 10.0."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    o = orc.Oracle(cfg)
    o.set_shape(synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6)))
    st, _, _ = orc.traj_eval(T, cm, 0.37 * T.sum())
    p_near = st[0] + np.array([0.3, -0.2, 0.4])
    s, ts, g, nr = o.swept_sdf(T, cm, p_near)
    assert nr >= 1 and s < 10.0 and 0 <= ts <= T.sum()
    assert abs(np.linalg.norm(g) - 1.0) < 1e-6
    s_far, ts_far, _, nr_far = o.swept_sdf(T, cm, st[0] + np.array([0, 0, 40.0]), tstar0=1.25)
    assert nr_far == 0 and s_far == 10.0 and ts_far == 1.25   # outputs untouched


def test_map_product_restatements(pkg, orc):
    """oracle.pointcloud_to_occupancy / gather_points (numpy restatements of rcvGlobalMapHandler and
    getPointsInAABBOutOfLastOne) against first principles."""
    synth = pkg.synth
    rng = np.random.default_rng(2)
    P = rng.uniform(0, 4, (5000, 3)).astype(np.float32)
    occ, b0, b1 = orc.pointcloud_to_occupancy(P, 0.5, 1)
    assert occ.shape == tuple(np.ceil((b1 - b0) / 0.5).astype(int)) and occ.all()      # 5000 points in 512 voxels
    occ3, _, _ = orc.pointcloud_to_occupancy(P, 0.5, 30)
    assert 0 <= occ3.sum() < occ.sum()
    # brute force count for one voxel
    idx = np.minimum(np.floor((P.astype(np.float64) - b0) / 0.5).astype(int), np.array(occ.shape) - 1)
    cnt = np.sum(np.all(idx == np.array([2, 3, 1]), axis=1))
    assert occ3[2, 3, 1] == (1 if cnt >= 30 else 0)
    # with zero offset the "out of the last box" rule only avoids duplicates: same set as the plain union
    occm = synth.random_box_map((30, 28, 20), res=0.5, occupancy=0.15, seed=4, edge=(1.0, 2.0))
    way = rng.uniform(3, 11, (5, 3)) * np.array([1.0, 1.0, 0.7])
    a = orc.gather_points(occm, np.zeros(3), np.array(occm.shape) * 0.5, 0.5, way, 1.6)
    b = synth.constraint_points(occm, (0, 0, 0), 0.5, way, 1.6)
    assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("which", ["blob 720", "Trefoil.obj 12 000", "blob with doubled faces"])
def test_mesh_closest_point_tree_equals_exhaustive_search(pkg, orc, which, monkeypatch):
    """The oracle's mesh distance goes through a bounding-box tree for meshes of more than 64 faces (test infrastructure: the
    exhaustive loop over the reference's 12 000-face Trefoil.obj made full-size parity runs impractical).  It must return what
    the exhaustive search returns, bit for bit - value, gradient and therefore the closest point, ties to the lowest face."""
    synth = pkg.synth
    if which == "blob 720":
        V, F = synth.blob_mesh(24, 16, scale=0.9)
    elif which.startswith("Trefoil"):
        V, F = reference_mesh("Trefoil", 0.83)
    else:
        V, F = synth.blob_mesh(12, 8)
        F = np.concatenate([F, F[:40]]).astype(np.int32)          # exact ties between a face and its copy
    V = np.asarray(V, dtype=np.float64)
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config()
    rng = np.random.default_rng(17)
    ext = np.abs(V).max()
    Q = np.concatenate([rng.uniform(-2.5 * ext, 2.5 * ext, (500, 3)),
                        V[rng.integers(0, V.shape[0], 500)] * (1.0 + rng.normal(0, 0.1, (500, 1))) + rng.normal(0, 0.02 * ext, (500, 3))])
    o_tree = orc.Oracle(cfg, threads=8); o_tree.set_shape(shape)
    monkeypatch.setenv("ORACLE_MESH_EXHAUSTIVE", "1")
    o_all = orc.Oracle(cfg, threads=8); o_all.set_shape(shape)
    s1, g1 = o_tree.shape_eval(Q)
    s0, g0 = o_all.shape_eval(Q)
    assert np.array_equal(s1, s0) and np.array_equal(g1, g0)
