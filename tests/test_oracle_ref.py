"""Pins the oracle against what can be built of the REAL reference (oracle/_ref, compiled from /root/reference by
oracle/Makefile — see oracle/ref_build).  CPU only.  Skipped when the prebuilt _ref libraries are absent."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FP = [0.61, 9.8, 0.1, 0.1, 0.01, 1e-4]


def test_flatness_bit_exact_vs_reference_header(orc):
    ref = orc.ref_flat_lib()
    if ref is None:
        pytest.skip("oracle/_ref/libref_flatness.so not built")
    rng = np.random.default_rng(0)
    for _ in range(3000):
        v, a, j = rng.normal(0, 3, 3), rng.normal(0, 4, 3), rng.normal(0, 6, 3)
        q, w = orc.flat_forward(FP, v, a, j)
        q2, w2 = orc.flat_forward(FP, v, a, j, lib=ref, prefix="ref")
        assert np.array_equal(q, q2) and np.array_equal(w, w2)
        pg, vg, og, qg = rng.normal(0, 1, 3), rng.normal(0, 1, 3), rng.normal(0, 1, 3), rng.normal(0, 1, 4)
        b1 = orc.flat_backward(FP, v, a, j, pg, vg, qg, og)
        b2 = orc.flat_backward(FP, v, a, j, pg, vg, qg, og, lib=ref, prefix="ref")
        assert np.array_equal(b1, b2)


def test_lmbm_flang_build_solves_nonsmooth_problems(orc):
    """The reference's live optimizer (Fortran 77 LMBM + the C++ shim lmbm.cpp, built by AMD flang into oracle/_ref/liblmbm.so,
    default lmbm_parameter_t like optimize_traj_lmbm): a smooth and a nonsmooth known-answer problem (MAXQ is one of the test
    problems of the upstream lmbm_test.f)."""
    if not os.path.exists(orc.REF_LMBM):
        pytest.skip("oracle/_ref/liblmbm.so not built (no flang)")

    def rosen(x):
        f = np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1]); g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return f, g

    def maxq(x):
        i = int(np.argmax(x * x)); g = np.zeros_like(x); g[i] = 2 * x[i]
        return x[i] ** 2, g
    r = orc.ref_lmbm_optimize(rosen, np.full(10, -1.2))
    assert r["status"] >= 0 and r["f"] < 1e-8 and np.abs(r["x"] - 1.0).max() < 1e-3 and r["evaluations"] >= r["iterations"] > 10
    r = orc.ref_lmbm_optimize(maxq, np.arange(1, 21, dtype=float) * np.where(np.arange(20) < 10, 1, -1))
    assert r["status"] >= 0 and r["f"] < 1e-4


def _cube():
    V = np.array([[x, y, z] for x in (-0.5, 0.5) for y in (-0.4, 0.4) for z in (-0.3, 0.3)], dtype=np.float64)
    F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6],
                  [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int32)
    return V, F


def test_mesh_winding_number_is_the_vendored_fwn(pkg, orc):
    """The oracle's mesh SDF takes w from the vendored HDK fast winding number itself (order 2, accuracy 2.0, float;
    oracle/_ref/libref_fwn.so, loaded by oracle/mesh_sdf.hpp): sdf == (1 - 2 w_hdk) * exact distance.  libigl approximates
    clusters farther than 2x their radius by an order-2 Taylor expansion, so w deviates from the exact winding number
    (0 / 1 for a closed mesh) by up to a few 1e-3 - measured here; that is why the exact solid-angle sum is NOT the
    reference's function and why the product walks the same hierarchy (csrc/fwn_host.hpp, tests/test_fwn_host.py)."""
    if not os.path.exists(orc.REF_FWN):
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    V, F = _cube()
    fwn = orc.RefFwn(V, F)
    cfg = pkg.synth.default_config()
    o = orc.Oracle(cfg)
    o.set_shape(pkg.synth.make_mesh_shape(V, F))
    assert o.mesh_backend() == 1
    rng = np.random.default_rng(1)
    Q = rng.uniform(-1.2, 1.2, (4000, 3))
    w_ref = fwn.eval(Q)
    sdf, _ = o.shape_eval(Q)
    inside = (np.abs(Q) < np.array([0.5, 0.4, 0.3])).all(axis=1)
    assert np.all((w_ref > 0.5) == inside)
    assert np.all((sdf < 0) == inside)
    w_exact = np.where(inside, 1.0, 0.0)                      # closed mesh: the exact winding number is 0 / 1
    err = np.abs(w_ref - w_exact)
    assert 1e-5 < err.max() < 5e-3                            # size of libigl's far-field approximation error
    half = np.array([0.5, 0.4, 0.3])
    d_out = np.linalg.norm(np.maximum(np.abs(Q) - half, 0), axis=1)
    d_in = np.min(half - np.abs(Q), axis=1)
    d = np.where(inside, d_in, d_out)
    assert np.allclose(sdf, (1.0 - 2.0 * w_ref) * d, rtol=1e-12, atol=1e-14)


def test_lbfgs_vs_reference_header(pkg, orc, product_lib, capfd):
    """The reference's own lbfgs_optimize (utils/lbfgs.hpp compiled from /root/reference against the Eigen stand-in) vs
    the oracle's restatement (oracle/lbfgs_ref.py) AND the product driver (csrc/lbfgs_host.hpp): same status, same number
    of callback evaluations, same iterate after a fixed number of iterations."""
    if not os.path.exists(orc.REF_LBFGS):
        pytest.skip("oracle/_ref/libref_lbfgs.so not built")
    import importlib.util
    spec = importlib.util.spec_from_file_location("lbfgs_ref", os.path.join(ROOT, "oracle", "lbfgs_ref.py"))
    ref_py = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_py)

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
    for fun, x0, iters in ((rosen, np.full(6, -0.01) + 0.002 * np.arange(6), 25), (quad, 0.02 * np.cos(np.arange(12.0)), 15),
                           (rosen, np.array([-1.2, 1.0, -0.5, 0.8]), 40)):
        xr, fr, sr, er = orc.ref_lbfgs_optimize(fun, x0, max_iterations=iters, g_epsilon=0.0, past=0)
        capfd.readouterr()        # the reference prints its progress
        xo, fo, so, ko, eo = ref_py.lbfgs_optimize(lambda x: fun(x), list(x0), max_iterations=iters, g_epsilon=0.0, past=0)
        xp, rp = pkg.lbfgs_minimize(fun, x0, lib=product_lib, max_iterations=iters, g_epsilon=0.0, past=0)
        assert sr == so == rp["status"], (sr, so, rp)
        assert er == eo == rp["evaluations"], (er, eo, rp)
        assert np.allclose(xo, xr, rtol=1e-9, atol=1e-11) and np.allclose(xp, xr, rtol=1e-9, atol=1e-11)
        assert abs(fo - fr) <= 1e-9 * max(1.0, abs(fr)) and abs(rp["f"] - fr) <= 1e-9 * max(1.0, abs(fr))
    # convergence / stop codes of the real thing match too
    xr, fr, sr, er = orc.ref_lbfgs_optimize(quad, np.ones(4) * 1e-3, g_epsilon=0.0, past=2, delta=0.5)
    capfd.readouterr()
    _, rp = pkg.lbfgs_minimize(quad, np.ones(4) * 1e-3, lib=product_lib, g_epsilon=0.0, past=2, delta=0.5)
    assert sr == rp["status"] == 1 and er == rp["evaluations"]


@pytest.mark.parametrize("N", [2, 5, 40])
def test_minco_and_trajectory_vs_reference_headers(pkg, orc, N):
    """The reference's own MINCO_S3NU + Trajectory<5> (utils/minco.hpp, trajectory.hpp, root_finder.hpp compiled from
    /root/reference against the Eigen stand-in) vs the oracle's restatements - which tests/test_minco_host.py shows
    bitwise equal to the product's host MINCO."""
    if not os.path.exists(orc.REF_MINCO):
        pytest.skip("oracle/_ref/libref_minco.so not built")
    rng = np.random.default_rng(50 + N)
    head = np.zeros((3, 3)); tail = np.zeros((3, 3))
    head[:, 0] = rng.uniform(0, 5, 3); head[:, 1] = rng.normal(0, 1, 3); head[:, 2] = rng.normal(0, 0.5, 3)
    tail[:, 0] = rng.uniform(15, 20, 3); tail[:, 1] = rng.normal(0, 1, 3)
    way = (np.linspace(head[:, 0], tail[:, 0], N + 1)[1:-1] + rng.normal(0, 0.6, (N - 1, 3))).T          # 3 x (N-1)
    T = rng.uniform(0.4, 2.5, N)
    cm_r, e_r, gC_r, gT_r = orc.ref_minco(head, tail, way, T)
    C6 = orc.minco_coeffs(head, tail, way, T)
    cm_o = np.ascontiguousarray(C6.T).reshape(-1)
    assert np.allclose(cm_o, cm_r, rtol=1e-12, atol=1e-12 * np.abs(cm_r).max())
    e_o, gC_o, gT_o = orc.minco_energy(cm_o, T)
    assert abs(e_o - e_r) <= 1e-12 * abs(e_r)
    assert np.allclose(gC_o, gC_r, rtol=1e-11, atol=1e-12 * np.abs(gC_r).max()) and np.allclose(gT_o, gT_r, rtol=1e-11)
    gc = rng.normal(0, 1, 18 * N); gt = rng.normal(0, 1, N)
    gP_r, gTt_r = orc.ref_minco_propagate(head, tail, way, T, gc, gt)
    gP_o, gTt_o = orc.minco_propagate(head, tail, way, T, gc, gt)
    sc = max(np.abs(gP_r).max(), np.abs(gTt_r).max())
    assert np.allclose(gP_o, gP_r, rtol=1e-10, atol=1e-12 * sc) and np.allclose(gTt_o, gTt_r, rtol=1e-10, atol=1e-12 * sc)
    # trajectory: piece lookup (junction time belongs to the earlier piece; overflow clamps) + state evaluation
    total = float(T.sum())
    times = list(rng.uniform(0, total, 12)) + [0.0, float(T[0]), float(np.cumsum(T)[N // 2]), total, total + 0.7]
    for t in times:
        st_r, dur = orc.ref_traj_eval(head, tail, way, T, t)
        st_o, piece, tl = orc.traj_eval(T, cm_o, t)
        assert abs(dur - total) <= 1e-12 * total
        assert np.allclose(st_o, st_r, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(st_r).max())), (t, st_o, st_r)


@pytest.mark.parametrize("name", ["Torus", "Torus_big", "Cappedtorus", "CappedCone", "RoundedCone", "WireframeBox", "BendLinear",
                                  "BendLinear_big", "TwistBox", "BendBox", "Table", "Trefoil", "SmoothDifference",
                                  "SmoothIntersection", "SmoothIntersection_big", "CSG", "Box"])
def test_analytic_shapes_vs_reference_classes(pkg, orc, name, capfd):
    """Every analytic robot shape of the registry (sw_manager.hpp:74-123) + Box: the reference's OWN class (Shape.hpp
    :824-2391, extracted at build time and compiled against the Eigen stand-in) vs the oracle's restatement - SDF and
    getonlyGrad1 (DEFINE_USEFUL_FUNCTION central difference / Box's forward difference), with a rotated body offset."""
    if not os.path.exists(orc.REF_SHAPES):
        pytest.skip("oracle/_ref/libref_shapes.so not built")
    synth = pkg.synth
    shape = synth.make_shape(name, poly_params=(0.2, -0.1, 0.15, 30, 10, 120))
    trans = np.array([shape.trans[i] for i in range(3)])
    R = np.array([shape.rotate[i] for i in range(9)]).reshape(3, 3)
    box = [shape.params[i] for i in range(3)] if name == "Box" else None
    ref = orc.RefShape(name, trans, R, box)
    capfd.readouterr()
    o = orc.Oracle(synth.default_config())
    o.set_shape(shape)
    rng = np.random.default_rng(len(name))
    Q = rng.uniform(-4.5, 4.5, (3000, 3))
    s_r, g_r = ref.eval(Q)
    s_o, g_o = o.shape_eval(Q)
    assert np.all(np.isfinite(s_r))
    assert np.allclose(s_o, s_r, rtol=1e-12, atol=1e-12), np.abs(s_o - s_r).max()
    # gradients are difference quotients with dx = 5e-6: rounding differences of the SDF are amplified by 1/dx
    ok = np.isfinite(g_r).all(axis=1)
    assert ok.mean() > 0.99
    assert np.allclose(g_o[ok], g_r[ok], rtol=0, atol=2e-7), np.abs(g_o[ok] - g_r[ok]).max()


# ---------------------------------------------------------------------------------------------------------------------
# The hot loop itself: the oracle vs the reference's OWN code (oracle/_ref/libref_sweeps.so)
# ---------------------------------------------------------------------------------------------------------------------
def _ref_world(pkg, orc, variant, shape_name="RoundedCone", seed=3, threads=4, **cfgkw):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import small_world, traj
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("oracle/_ref/libref_sweeps.so not built")
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=seed)
    kw = dict(kernel_size=9, integral_intervs=16, safety_hor=0.5)
    kw.update(cfgkw)
    cfg = synth.default_config(variant, **kw)
    shape = synth.make_shape(shape_name, poly_params=(0.1, -0.05, 0.08, 20, -10, 40)) if shape_name != "RoundedCone" else \
        synth.make_shape("RoundedCone", params=(1.5, 0.6, 4.5), poly_params=(0.1, -0.05, 0.08, 20, -10, 40))
    ref = orc.RefWorld(cfg, shape_name, shape, res, threads=threads)
    ref.set_occupancy(occ, (0, 0, 0), res)
    o = orc.Oracle(cfg, threads=threads)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    o.set_shape(shape)
    return cfg, occ, res, shape, ref, o


def _mesh_world(pkg, orc, V, F, seed=14, N=5, traj_seed=23, threads=8):
    """(cfg, T, cm, pts, reference world with the mesh Generalshape, oracle) on the small world of the GPU mesh tests."""
    from common import small_world, traj
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("oracle/_ref/libref_sweeps.so not built")
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=seed)
    T, cm = traj(pkg, occ, res, N=N, seed=traj_seed)
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    shape = synth.make_mesh_shape(V, F)
    ref = orc.RefWorld(cfg, "Box", synth.make_shape("Box"), res, threads=threads)      # any class: replaced by the mesh below
    ref.set_occupancy(occ, (0, 0, 0), res)
    ref.set_mesh(V, F)
    ref.set_points(pts)
    o = orc.Oracle(cfg, threads=threads)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.set_points(pts)
    return cfg, T, cm, pts, ref, o


@pytest.mark.parametrize("mesh", ["L prism", "blob 720"])
def test_mesh_generalshape_vs_reference_code(pkg, orc, mesh):
    """The reference's OWN mesh path - Generalshape::getonlySDF / getonlyGrad1 (Shape.cpp:105-151, extracted) over the vendored
    winding-number core, and the swept-volume sweep on top of it - vs the oracle's restatement (oracle/mesh_sdf.hpp): the
    shape by itself, then cost, t*, gradients of addSaftyPenaOnSweptVolumeParallel."""
    synth = pkg.synth
    V, F = synth.l_prism_mesh() if mesh == "L prism" else synth.blob_mesh(24, 16, scale=0.9)
    cfg, T, cm, pts, ref, o = _mesh_world(pkg, orc, V, F)
    if o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    ref.mesh_perturb(0)
    rng = np.random.default_rng(3)
    ext = np.abs(V).max()
    Q = np.concatenate([rng.uniform(-3 * ext, 3 * ext, (600, 3)), V[rng.integers(0, len(V), 600)] * (1 + rng.normal(0, 0.15, (600, 1)))])
    s_r, g_r = ref.shape_eval(Q)
    s_o, g_o = o.shape_eval(Q)
    assert np.array_equal(s_o, s_r) or np.abs(s_o - s_r).max() <= 1e-12 * max(1.0, np.abs(s_r).max())
    assert np.abs(g_o - g_r).max() <= 1e-9
    ts_r = np.zeros(len(pts)); ts_o = np.zeros(len(pts))
    c_r, gT_r, gC_r, _ = ref.sweep_swept(T, cm, tstar=ts_r)
    c_o, gT_o, gC_o, _ = o.eval(T, cm, tstar=ts_o)
    assert c_r > 0 and abs(c_o - c_r) <= 1e-9 * abs(c_r)
    assert np.abs(ts_o - ts_r).max() <= 1e-9
    assert np.linalg.norm(gC_o - gC_r) <= 1e-7 * np.linalg.norm(gC_r) and np.linalg.norm(gT_o - gT_r) <= 1e-7 * np.linalg.norm(gT_r)


def mesh_v1_sensitivity_on(pkg, orc, cfg, res, T, cm, pts, V, F, ulps=1, threads=8):
    """The reference's mesh x swept-volume sweep against ITSELF with the float solid angle moved by `ulps` x 2^-23 per query
    (one ulp of ONE order-one atan2f term - less than two libm implementations disagree by over a whole mesh; the device's
    w differs from the host library's by up to 4e-7, i.e. ~40 such units): the spread of t*, cost and gradients on the
    given scenario (the sweep reads the obstacle points, not the grid)."""
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("oracle/_ref/libref_sweeps.so not built")
    ref = orc.RefWorld(cfg, "Box", pkg.synth.make_shape("Box"), res, threads=threads)      # any class: replaced by the mesh
    ref.set_mesh(V, F)
    ref.set_points(pts)
    ref.mesh_perturb(0)
    ts0 = np.zeros(len(pts)); c0, gT0, gC0, _ = ref.sweep_swept(T, cm, tstar=ts0)
    ref.mesh_perturb(ulps)
    try:
        ts1 = np.zeros(len(pts)); c1, gT1, gC1, _ = ref.sweep_swept(T, cm, tstar=ts1)
    finally:
        ref.mesh_perturb(0)
    dt = np.abs(ts1 - ts0)
    return {"points": len(pts), "dt_max": float(dt.max()), "dt_frac_gt_1e-6": float((dt > 1e-6).mean()),
            "cost_rel": abs(c1 - c0) / abs(c0), "gradC_rel": float(np.linalg.norm(gC1 - gC0) / np.linalg.norm(gC0)),
            "gradT_rel": float(np.linalg.norm(gT1 - gT0) / np.linalg.norm(gT0))}


def mesh_v1_sensitivity(pkg, orc, V, F, ulps=1, **kw):
    cfg, T, cm, pts, ref, o = _mesh_world(pkg, orc, V, F, **kw)
    return mesh_v1_sensitivity_on(pkg, orc, cfg, 0.5, T, cm, pts, V, F, ulps=ulps)


@pytest.mark.parametrize("mesh", ["L prism", "blob 720"])
def test_mesh_swept_volume_argmin_is_sensitive_to_one_ulp_of_w(pkg, orc, mesh):
    """Measures what the GPU tests' mesh x V1 tolerances rest on: the REFERENCE's own code, run twice with the winding
    number differing by one float ulp, does not reproduce its own t* / gradients to 1e-5 - the argmin over t is a chain of
    `fc - fx < 0` accepts refined down to 1e-4 s steps, where the SDF changes by less than the float noise of w.  The cost
    (t* is a minimiser) is insensitive.  The spread found here is what tests/test_gpu_mesh.py allows the device."""
    synth = pkg.synth
    V, F = synth.l_prism_mesh() if mesh == "L prism" else synth.blob_mesh(24, 16, scale=0.9)
    r = mesh_v1_sensitivity(pkg, orc, V, F)
    print(mesh, r)
    assert r["cost_rel"] <= 1e-6
    assert r["dt_max"] > 1e-6                     # some point does move ...
    assert r["dt_max"] <= 5e-3                    # ... by refinement steps, not to another basin


@pytest.mark.parametrize("shape_name", ["RoundedCone", "Torus", "Box"])
def test_swept_sweep_vs_reference_code(pkg, orc, shape_name):
    """addSaftyPenaOnSweptVolumeParallel + getSDFofSweptVolume (choiceTInit, gradientDescent, getSDF_DOT...) of the
    reference itself vs the oracle's restatement: cost, gradT, gradC, t* per point."""
    from common import traj
    capi, synth = pkg.capi, pkg.synth
    cfg, occ, res, shape, ref, o = _ref_world(pkg, orc, capi.V1_SWEPT, shape_name, seed=3)
    N = 6
    T, cm = traj(pkg, occ, res, N=N, seed=43)
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    assert pts.shape[0] > 50
    ref.set_points(pts); o.set_points(pts)
    c_r, gT_r, gC_r, ts_r = ref.sweep_swept(T, cm)
    ts_o = np.zeros(pts.shape[0])
    c_o, gT_o, gC_o, st = o.eval(T, cm, tstar=ts_o)
    assert c_r > 0
    assert abs(c_o - c_r) <= 1e-9 * abs(c_r), (c_o, c_r)
    assert np.allclose(ts_o, ts_r, rtol=0, atol=1e-9)
    sc = np.abs(gC_r).max()
    assert np.allclose(gC_o, gC_r, rtol=1e-7, atol=1e-9 * sc) and np.allclose(gT_o, gT_r, rtol=1e-7, atol=1e-9 * np.abs(gT_r).max())
    # single queries incl. points with no qualifying interval (returns 10, t* untouched)
    for p in (pts[0], pts[len(pts) // 2], np.array([1.0, 1.0, 14.0])):
        s_r, t_r, g_r = ref.swept_sdf(T, cm, p, tstar0=0.3)
        s_o, t_o, g_o, nr = o.swept_sdf(T, cm, p, tstar0=0.3)
        assert abs(s_o - s_r) <= 1e-10 * max(1.0, abs(s_r)) and abs(t_o - t_r) <= 1e-9


def test_integral_sweep_vs_reference_code(pkg, orc):
    """addTimeIntPenaltyParallel of the reference itself (velocity / body-rate / tilt penalties, flatness back-prop,
    trapezoid weights) vs the oracle with the collision term off."""
    from common import traj
    capi, synth = pkg.capi, pkg.synth
    cfg, occ, res, shape, ref, o = _ref_world(pkg, orc, capi.V3_ESDF_TILE, seed=3, enable_pos=0, vmax=1.5, omgmax=0.6, thetamax=0.25)
    T, cm = traj(pkg, occ, res, N=6, seed=11, piece_T=0.45)
    c_r, gT_r, gC_r = ref.sweep_integral(T, cm)
    c_o, gT_o, gC_o, _ = o.eval(T, cm)
    assert c_r > 0
    assert abs(c_o - c_r) <= 1e-11 * abs(c_r)
    assert np.allclose(gC_o, gC_r, rtol=1e-9, atol=1e-11 * np.abs(gC_r).max()) and np.allclose(gT_o, gT_r, rtol=1e-9, atol=1e-11 * np.abs(gT_r).max())


def test_grad_cost_p_and_grid_vs_reference_code(pkg, orc):
    """grad_cost_p (per-pose tile term), getPointsInAABB, generateESDF3d and the trilinear ESDF sampler of the reference
    itself vs the oracle."""
    capi, synth = pkg.capi, pkg.synth
    cfg, occ, res, shape, ref, o = _ref_world(pkg, orc, capi.V2_OCC_TILE, seed=5)
    # ESDF: bit-exact
    e_r = ref.generate_esdf()
    e_o = orc.build_esdf(occ, res)
    assert np.array_equal(e_o, e_r)
    o2 = orc.Oracle(cfg); o2.set_grid(e_o, (0, 0, 0), res, capi.GRID_ESDF)
    rng = np.random.default_rng(8)
    ext = np.array(occ.shape) * res
    for _ in range(200):
        p = rng.uniform(-0.5, 1.0, 3) * 0 + rng.uniform(0.3, 0.97, 3) * ext
        v_r, g_r = ref.esdf_sample(p)
        v_o, g_o = o2.esdf_trilinear(p)
        # the oracle (like the device) stores the ESDF as float32 - the only place where it departs from the reference's doubles
        assert abs(v_o - v_r) <= 2e-7 * max(1.0, abs(v_r)) and np.allclose(g_o, g_r, rtol=1e-5, atol=2e-6)
    # the tile of a pose + the per-pose collision term
    n_active = 0
    for k in range(60):
        pos = rng.uniform(0.15, 0.85, 3) * ext
        lo, hi = o.aabb_index_box(pos, cfg.kernel_size * res / 2)
        pts_r = ref.points_in_aabb(pos, cfg.kernel_size * res / 2)
        sub = occ[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]
        assert pts_r.shape[0] == int(sub.sum())
        tilt = rng.normal(0, 0.25, 2)
        q = np.array([1.0, tilt[0], tilt[1], 0.0]); q /= np.linalg.norm(q)
        c_r, gp_r, gq_r, a_r = ref.grad_cost_p(pos, q)
        c_o, gp_o, gq_o, a_o = o.grad_cost_p(pos, q)
        assert a_r == a_o
        assert abs(c_o - c_r) <= 1e-11 * max(1.0, abs(c_r))
        assert np.allclose(gp_o, gp_r, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(gp_r).max()))
        assert np.allclose(gq_o, gq_r, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(gq_r).max()))
        n_active += int(a_r)
    assert n_active > 10


# ---- front end (kernel convolution): oracle/frontend.hpp vs the reference's own code ----------------------
def _frontend_world(pkg, seed=5, dims=(40, 36, 20), res=0.5):
    occ = pkg.synth.random_box_map(dims, res=res, occupancy=0.10, seed=seed, edge=(0.5, 2.0))
    return occ, res


@pytest.mark.parametrize("name", ["Box", "RoundedCone", "Torus_big", "BendLinear_big", "SmoothIntersection_big", "Trefoil", "CSG", "Table"])
def test_frontend_kernels_and_checks_vs_reference_code(pkg, orc, name):
    """Robot attitude kernels (BasicShape::initShape kernel part, Shape.hpp:406-462), the inflated bit-packed map
    (generateMapKernel, PCSmap_manager.h:46-78) and checkKernelValue (sw_manager.hpp:775-942: kernelConv + breadth-first
    attitude search) - the oracle's restatement against the reference's own code, byte for byte and query for query."""
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("oracle/_ref/libref_sweeps.so not built")
    capi, synth = pkg.capi, pkg.synth
    occ, res = _frontend_world(pkg)
    kw = dict(params=(0.9, 0.35, 0.25)) if name == "Box" else {}
    shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25), **kw)
    cfg = synth.default_config(capi.V1_SWEPT)
    k = 13 if name in ("Box", "RoundedCone") else 15
    fe = capi.frontend_config(kernel_size=k, max_roll=45.0, max_pitch=45.0, ang_res=9.0, safeh=0.0)
    o = orc.Oracle(cfg, threads=8)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape)
    o.frontend_build(fe)
    w = orc.RefWorld(cfg, name, shape, res)
    w.set_occupancy(occ, (0, 0, 0), res)
    w.frontend_build(fe, name)
    assert o.fe_dims == w.fe_dims and o.fe_dims[:2] == [11, 11]
    sk, sk_ref = o.frontend_shape_kernels(), w.frontend_shape_kernels()
    assert sk_ref.any()
    assert np.array_equal(sk, sk_ref), (name, np.argwhere(sk != sk_ref)[:5])
    assert np.array_equal(o.frontend_map_kernel(), w.frontend_map_kernel())
    # queries: free voxels all over the map incl. the borders, random parent attitudes on the attitude grid and off it
    rng = np.random.default_rng(3)
    free = np.argwhere(occ == 0)
    idx = free[rng.choice(len(free), 600, replace=False)].astype(np.int32)
    fr = rng.choice(np.arange(-45.0, 45.1, 9.0), 600) + rng.choice([0.0, 0.0, 2.5], 600) * (rng.random(600) < 0.3)
    fp = rng.choice(np.arange(-45.0, 45.1, 9.0), 600)
    fr = np.clip(fr, -45.0, 45.0)
    ok, cr, cp, ki = o.frontend_check(idx, fr, fp)
    n_ok = 0
    for q in range(600):
        r_ok, r_cr, r_cp = w.frontend_check(idx[q], fr[q], fp[q])
        assert r_ok == ok[q], (name, q)
        if r_ok:
            n_ok += 1
            assert r_cr == cr[q] and r_cp == cp[q], (name, q, r_cr, cr[q], r_cp, cp[q])
    assert 0 < n_ok < 600, n_ok        # the scenario exercises both outcomes


@pytest.mark.parametrize("name,occupancy,seed", [("RoundedCone", 0.10, 21), ("Table", 0.10, 22), ("Box", 0.24, 23)])
def test_astar_vs_reference_code(pkg, orc, name, occupancy, seed):
    """The SE(3) A* (AstarPathSearch / AstarGetSucc / getHeu / getPath and the nodes' attitude chain, front_end_Astar.hpp:160-376):
    the oracle's restatement against the reference's own code on cluttered worlds - the same cells, the same (roll, pitch) at
    every node and the same number of neighbour tests, which pins the open set's order (a multimap keyed by the fScore AT
    INSERTION), the re-opening of closed nodes and the overwriting of a node's attitude by later expansions."""
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("oracle/_ref/libref_sweeps.so not built")
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occ = synth.random_box_map((44, 40, 14), res=res, occupancy=occupancy, seed=seed, edge=(0.5, 2.5))
    kw = dict(params=(0.9, 0.35, 0.25)) if name == "Box" else {}
    shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25), **kw)
    cfg = synth.default_config(capi.V1_SWEPT)
    fe = capi.frontend_config(kernel_size=13 if name in ("Box", "RoundedCone") else 15)
    o = orc.Oracle(cfg, threads=8)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.frontend_build(fe)
    w = orc.RefWorld(cfg, name, shape, res)
    w.set_occupancy(occ, (0, 0, 0), res)
    w.frontend_build(fe, name)
    good = np.argwhere((o.frontend_cspace(occ.shape) != 0).any(axis=-1))
    rng = np.random.default_rng(seed)
    found = tilted = 0
    for q in range(10):
        a, b = good[rng.choice(len(good), 2, replace=False)]
        s, g = (a + rng.random(3)) * res, (b + rng.random(3)) * res
        x_r, rp_r, checks_r = w.astar(s, g)
        x_o, rp_o, st = o.frontend_astar(s, g)
        assert st["checks"] == checks_r, (name, q, st, checks_r)
        assert (x_r is None) == (x_o is None), (name, q)
        if x_r is not None:
            found += 1
            assert np.array_equal(x_o, x_r) and np.array_equal(rp_o, rp_r), (name, q)
            tilted += int((rp_r != 0).any())
    assert found >= 6 and tilted >= 3, (found, tilted)       # real searches through clutter, with attitudes other than level


def test_sampled_grid_shape_vs_reference_code(pkg, orc):
    """ISDF_SHAPE_GRID (a user-defined host shape tabulated on a body-frame lattice): the oracle's restatement of getonlySDFNum /
    getonlyGrad1Num / getSDFwithGrad1Num (Shape.hpp:481-600) against the reference's OWN functions on the reference's OWN lattice
    (BasicShape::initShape, :361-404, run over its RoundedCone class) - values bit for bit, incl. the 1e20 / untouched-gradient
    behaviour outside the lattice; and the lattice the product's host fill produces from the same class == the reference's."""
    if not os.path.exists(orc.REF_SHAPES):
        pytest.skip("oracle/_ref/libref_shapes.so not built")
    capi, synth = pkg.capi, pkg.synth
    rs = orc.RefShape("RoundedCone")
    nd, nres = (4.3, 4.0, 7.25), 0.25
    rg = rs.num_grid(nd, nres)
    cells, gmin = rg.cells()
    assert cells.shape[:3] == (int(np.ceil(nd[0] / nres)), int(np.ceil(nd[1] / nres)), int(np.ceil(nd[2] / nres)))
    assert np.allclose(gmin, [-nd[0] / 2, -nd[1] / 2, -nd[2] / 2], rtol=0, atol=0)
    rng = np.random.default_rng(5)
    hi = gmin + (np.array(cells.shape[:3]) - 1) * nres
    Q = np.concatenate([rng.uniform(gmin - 0.3, hi + 0.3, (4000, 3)),                 # inside and just outside
                        gmin + rng.integers(0, 8, (200, 3)) * nres,                   # exactly ON lattice nodes / cell faces
                        [gmin, hi, hi - 1e-12, gmin + 1e-12]])
    sa, ga, sb, gb = rg.eval(Q)
    o = orc.Oracle(synth.default_config(), threads=1)
    o.set_shape_grid(cells, gmin, nres)
    s0, g0 = o.shape_eval(Q)
    inside = sa < 1e19
    assert inside.sum() > 2500 and (~inside).sum() > 200
    assert np.array_equal(s0, sa) and np.array_equal(sb, sa)                      # value: same bits, 1e20 outside
    assert np.array_equal(g0, ga)                                                 # gradient: same bits, zero outside
    assert np.array_equal(gb[inside], ga[inside]) and np.all(gb[~inside] == 7.0)  # the combined sampler leaves grad alone outside
    # the sampled function approximates the class it came from: O(nres^2 x curvature) on the distance
    st, gt = rs.eval(Q[inside])
    assert np.abs(sa[inside] - st).max() < 0.5 * nres
    print(f"\\nsampled RoundedCone, nres {nres}: max |sdf_grid - sdf| = {np.abs(sa[inside] - st).max():.3e} m, "
          f"max |grad_grid - grad| = {np.linalg.norm(ga[inside] - gt, axis=1).max():.3e}")
    # isdf_set_shape_sampled's host fill (no device needed up to the upload: checked through the oracle-side twin below on the GPU)


@pytest.mark.parametrize("case", ["zero offset", "offset", "at the map's edge", "flat boxes"])
def test_gather_points_vs_reference_code(pkg, orc, case):
    """The obstacle-point set of the swept-volume sweep: the oracle's restatement (oracle.gather_points) against the reference's OWN
    getPointsInAABBOutOfLastOne (PCSmap_manager.h:182-216, extracted into libref_sweeps.so) driven like plan_manager.cpp:232-251 - the same
    SET of voxels (the reference keeps them in an unordered_map: its order is unspecified), the same centres bit for bit, the reference's
    voxel id = i + X j + X Y k.  The offset quirk is part of it: `offset` moves the CURRENT waypoint's box only, never the previous one's."""
    capi = pkg.capi
    cfg, occ, res, shape, ref, o = _ref_world(pkg, orc, capi.V1_SWEPT, seed=12)
    dims = np.array(occ.shape); ext = dims * res
    rng = np.random.default_rng(31)
    n_way = 7
    # waypoints next to obstacles, a metre or so apart: consecutive boxes overlap
    occ_ijk = np.argwhere(occ)
    start = (occ_ijk[rng.integers(len(occ_ijk))] + 0.5) * res
    way = np.cumsum(np.vstack([start[None, :], rng.normal(0, 1.0, (n_way - 1, 3))]), axis=0).clip(0.05 * ext, 0.95 * ext)
    half, offset = np.array([1.5, 1.2, 1.0]), np.zeros(3)
    if case == "offset":
        offset = np.array([0.9, -0.7, 0.6])
    elif case == "at the map's edge":
        way[0] = [0.1, 0.2, ext[2] - 0.05]; way[-1] = ext - 0.01; way[3] = [-0.7, ext[1] + 2.0, 0.3]      # corners projected into the map
        offset = np.array([-0.25, 0.6, 0.0])
    elif case == "flat boxes":
        half = np.array([2.1, 0.3, 0.6])
    ids_r, xyz_r = ref.gather_points(way, half, offset)
    want = orc.gather_points(occ, np.zeros(3), ext, res, way, half, offset)
    assert ids_r.size > 5
    ijk = np.floor(want / res).astype(np.int64)
    ids_o = ijk[:, 0] + dims[0] * ijk[:, 1] + dims[0] * dims[1] * ijk[:, 2]
    order = np.argsort(ids_o, kind="stable")
    assert np.array_equal(ids_o[order], ids_r.astype(np.int64))
    assert np.array_equal(want[order], xyz_r)
    if case == "offset":
        # the quirk matters on this input: with the offset applied to the previous box as well the set is another one
        def shifted(prev_also):
            ids = set()
            last = np.array([999.0, 999.0, 999.0])
            for w in way:
                box = orc.gather_points(occ, np.zeros(3), ext, res, w[None, :], half, offset)
                lo = np.floor((np.clip(last - half + (offset if prev_also else 0), 0, ext)) / res).astype(int).clip(0, dims - 1)
                hi = np.floor((np.clip(last + half + (offset if prev_also else 0), 0, ext)) / res).astype(int).clip(0, dims - 1)
                for p in box:
                    i = np.floor(p / res).astype(int)
                    if np.any(i > hi) or np.any(i < lo):
                        ids.add(tuple(i))
                last = w
            return ids
        assert shifted(False) == set(map(tuple, ijk)) and shifted(True) != shifted(False)
