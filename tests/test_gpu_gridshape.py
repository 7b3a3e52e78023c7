"""ISDF_SHAPE_GRID: a robot shape the library has no formula for, handed over as the lattice BasicShape::initShape tabulates
(Shape.hpp:361-404) and sampled on the device like getonlySDFNum / getonlyGrad1Num / getSDFwithGrad1Num (:481-600).  The
reference's plugin API stays open: any host class's getSDFwithGrad1 can be sampled (isdf_set_shape_sampled)."""
import os

import numpy as np
import pytest

from common import REL_TOL, assert_close, small_world, traj

pytestmark = pytest.mark.gpu


def _lattice(orc, nd, nres, params=(0.8, 0.3, 1.6)):
    """node values of a RoundedCone(r1, r2, h) through the oracle's analytic class: (cells [X,Y,Z,4], min)"""
    import __graft_entry__ as g
    pkg = g.load_package()
    X, Y, Z = (int(np.ceil(n / nres)) for n in nd)
    mn = np.array([-nd[0] / 2, -nd[1] / 2, -nd[2] / 2])
    I, J, K = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    P = np.stack([mn[0] + I * nres, mn[1] + J * nres, mn[2] + K * nres], axis=-1).reshape(-1, 3)
    o = orc.Oracle(pkg.synth.default_config(), threads=8)
    o.set_shape(pkg.synth.make_shape("RoundedCone", params=params))
    s, gr = o.shape_eval(P)
    return np.concatenate([gr, s[:, None]], axis=1).reshape(X, Y, Z, 4), mn


def test_grid_shape_plugin_and_sweeps(pkg, orc, product_lib):
    capi, synth = pkg.capi, pkg.synth
    nd, nres = (4.4, 4.4, 6.4), 0.1          # covers the cone (z -0.8 .. 1.9) plus the band the swept-volume query looks at
    cells, mn = _lattice(orc, nd, nres)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res)
    eng = pkg.Engine(cfg); o = orc.Oracle(cfg, threads=8)
    for t in (eng, o):
        t.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
        t.set_shape_grid(cells, mn, nres)
    # the plugin by itself: device == oracle restatement (same operations: bit for bit), inside, outside and on the faces
    rng = np.random.default_rng(2)
    hi = mn + (np.array(cells.shape[:3]) - 1) * nres
    Q = np.concatenate([rng.uniform(mn - 0.2, hi + 0.2, (5000, 3)), mn + rng.integers(0, 10, (100, 3)) * nres])
    s, g = eng.shape_eval(Q)
    s0, g0 = o.shape_eval(Q)
    assert np.array_equal(s, s0) and np.abs(g - g0).max() <= 1e-15
    assert (s0 > 1e19).sum() > 100 and (s0 < 1e19).sum() > 3000
    # V3 sweep
    c, gT, gC = eng.eval_single(T, cm)
    c0, gT0, gC0, st0 = o.eval(T, cm)
    assert c0 > 0 and st0[3] > 0
    assert abs(c - c0) <= REL_TOL * abs(c0); assert_close(gT, gT0, "grid V3 gradT"); assert_close(gC, gC0, "grid V3 gradC")
    st = eng.stats()
    assert st["grad_pairs"] == st0[3]
    # ... and it approximates the analytic class it was sampled from (documented approximation, not parity)
    ea = pkg.Engine(cfg); ea.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); ea.set_shape(synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6)))
    ca, _, gCa = ea.eval_single(T, cm)
    print(f"\nsampled (nres {nres}) vs analytic rounded cone, V3 sweep: cost {c:.6g} vs {ca:.6g} (rel {abs(c - ca) / ca:.2e}), "
          f"gradC rel {np.linalg.norm(gC - gCa) / np.linalg.norm(gCa):.2e}")
    assert abs(c - ca) <= 0.05 * ca and np.linalg.norm(gC - gCa) <= 0.1 * np.linalg.norm(gCa)
    # V1 sweep (the live path): own gradient mode, like Ball / the mesh kind
    N = T.size
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    e1 = pkg.Engine(cfg1); o1 = orc.Oracle(cfg1, threads=8)
    for t in (e1, o1):
        t.set_shape_grid(cells, mn, nres); t.set_points(pts)
    tg = np.zeros(len(pts)); to = np.zeros(len(pts))
    c, gT, gC = e1.eval_single(T, cm, tstar=tg)
    c0, gT0, gC0, _ = o1.eval(T, cm, tstar=to)
    assert c0 > 0 and abs(c - c0) <= REL_TOL * abs(c0)
    assert np.abs(tg - to).max() <= 1e-9
    assert_close(gT, gT0, "grid V1 gradT"); assert_close(gC, gC0, "grid V1 gradC")


def test_set_shape_sampled_fills_the_lattice_like_initshape(pkg, orc, product_lib):
    """isdf_set_shape_sampled calls a HOST callback (any BasicShape subclass's getSDFwithGrad1) once per node, in initShape's
    order and layout: the installed lattice behaves exactly like one filled by hand from the same function - and, where the
    reference's build is at hand, like the reference's own initShape over its own class."""
    capi, synth = pkg.capi, pkg.synth
    nd, nres = (1.7, 1.5, 2.3), 0.2
    oa = orc.Oracle(synth.default_config(), threads=1)
    oa.set_shape(synth.make_shape("RoundedCone", params=(0.5, 0.2, 1.0)))
    calls = []

    def user_shape(p):                       # a "user-defined" shape: here the oracle's rounded cone, one point per call
        s, g = oa.shape_eval(p.reshape(1, 3))
        calls.append(p.copy())
        return s[0], g[0]
    eng = pkg.Engine(synth.default_config())
    eng.set_shape_sampled(user_shape, nd, nres)
    X, Y, Z = (int(np.ceil(n / nres)) for n in nd)
    assert len(calls) == X * Y * Z
    assert np.allclose(calls[0], [-nd[0] / 2, -nd[1] / 2, -nd[2] / 2]) and np.allclose(calls[1] - calls[0], [0, 0, nres])     # z fastest
    cells, mn = _lattice(orc, nd, nres, params=(0.5, 0.2, 1.0))
    e2 = pkg.Engine(synth.default_config()); e2.set_shape_grid(cells, mn, nres)
    rng = np.random.default_rng(3)
    Q = rng.uniform(mn - 0.1, -mn + 0.1, (3000, 3))
    s, g = eng.shape_eval(Q); s2, g2 = e2.shape_eval(Q)
    assert np.array_equal(s, s2) and np.array_equal(g, g2)
    if os.path.exists(orc.REF_SHAPES):       # the reference's own initShape over its own RoundedCone (default constants 1.5 / 0.6 / 4.5)
        rs = orc.RefShape("RoundedCone")
        nd2 = (4.3, 4.0, 7.25)
        rg = rs.num_grid(nd2, 0.25)
        cells_r, mn_r = rg.cells()
        ob = orc.Oracle(synth.default_config(), threads=1)
        ob.set_shape(synth.make_shape("RoundedCone"))

        def ref_like(p):
            s, g = ob.shape_eval(p.reshape(1, 3))
            return s[0], g[0]
        e3 = pkg.Engine(synth.default_config()); e3.set_shape_sampled(ref_like, nd2, 0.25)
        Q = np.random.default_rng(4).uniform(mn_r, -mn_r, (3000, 3))
        s3, g3 = e3.shape_eval(Q)
        sa, ga, _, _ = rg.eval(Q)
        # (the oracle's analytic SDF equals the reference class's to 1e-12, its gradient to 2e-7: the lattices agree to that)
        ins = sa < 1e19
        assert np.array_equal(ins, s3 < 1e19)
        assert np.abs(s3[ins] - sa[ins]).max() <= 1e-10 and np.abs(g3[ins] - ga[ins]).max() <= 1e-5
