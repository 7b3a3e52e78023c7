"""MESH robot kind on the device against the reference's own winding-number library (through the oracle, whose w comes
from oracle/_ref/libref_fwn.so = the vendored HDK FastWindingNumberForSoups.h): the shape plugin by itself
(isdf_shape_eval) on meshes from 1 to ~5000 faces, then the sweeps with a 720-face robot."""
import numpy as np
import pytest

from benchlib.meshes import reference_mesh

from common import REL_TOL, assert_close, assert_mesh_v1_free_running, make_pair, small_world, traj

pytestmark = pytest.mark.gpu


def _need_hdk(o):
    if o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built: the oracle would use the exact winding number, which is NOT the reference's")


def _queries(V, rng, n):
    ext = np.abs(V).max()
    far = rng.uniform(-3.0 * ext, 3.0 * ext, (n, 3))
    # near the surface, both sides: vertices pushed in / out along their direction from the centroid
    idx = rng.integers(0, V.shape[0], n)
    near = V[idx] * (1.0 + rng.normal(0, 0.15, (n, 1))) + rng.normal(0, 0.02 * ext, (n, 3))
    return np.concatenate([far, near])


@pytest.mark.parametrize("name", ["one triangle", "tetrahedron", "L prism", "blob 80", "blob 720", "blob 4992"])
def test_mesh_shape_plugin_vs_reference_library(pkg, orc, product_lib, name):
    synth = pkg.synth
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    Ft = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]], dtype=np.int32)
    V, F = {"one triangle": lambda: (tet, Ft[:1]), "tetrahedron": lambda: (tet, Ft), "L prism": synth.l_prism_mesh,
            "blob 80": lambda: synth.blob_mesh(8, 6), "blob 720": lambda: synth.blob_mesh(24, 16),
            "blob 4992": lambda: synth.blob_mesh(64, 40)}[name]()
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config()
    eng = pkg.Engine(cfg); o = orc.Oracle(cfg, threads=8)
    eng.set_shape(shape); o.set_shape(shape)
    _need_hdk(o)
    rng = np.random.default_rng(11)
    Q = _queries(np.asarray(V, dtype=np.float64), rng, 1500)
    s, g = eng.shape_eval(Q)
    s0, g0 = o.shape_eval(Q)
    # w is the library's bit for bit (float atan2 = glibc's, the HDK traversal's tree-shaped sum); what is left is the rounding of
    # the fp64 closest-point distance (branch-and-bound walk here, exhaustive search in the oracle)
    assert np.all(np.abs(s - s0) <= 1e-13 * np.abs(s0) + 1e-14), (name, np.abs(s - s0).max())
    from common import _spread_log
    _spread_log(f"shape plugin, {name}: device sdf == oracle sdf bit for bit on {np.mean(s == s0):.2%} of {len(s)} queries, max |d sdf| {np.abs(s - s0).max():.2e}")
    # gradient = normalize((1 - 2w) (p - c)): a unit vector; identical closest point => identical to rounding
    ok = np.linalg.norm(g - g0, axis=1) <= 1e-6
    # a query equidistant (to rounding) from two triangles may pick either closest point: must be (very) rare
    assert ok.mean() > 0.999, (name, (~ok).sum())


def test_mesh_sweeps_with_a_720_face_robot(pkg, orc, product_lib):
    """V3 integral sweep and V1 swept-volume sweep, 720-face concave-ish robot (hierarchy 3 levels deep, far-field clusters
    active inside the tile), device vs the oracle with the reference's winding-number library."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=5, seed=23)
    V, F = synth.blob_mesh(24, 16, scale=0.9)
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=12, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res, threads=16)
    _need_hdk(o)
    c, gT, gC = eng.eval_single(T, cm)
    c0, gT0, gC0, st0 = o.eval(T, cm)
    assert c0 > 0 and st0[3] > 0
    assert abs(c - c0) <= REL_TOL * abs(c0)
    assert_close(gT, gT0, "mesh-720 V3 gradT"); assert_close(gC, gC0, "mesh-720 V3 gradC")
    way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    eng1, o1 = make_pair(pkg, orc, cfg1, shape, occ=occ, res=res, points=pts, threads=16)
    tg = np.zeros(len(pts)); to = np.zeros(len(pts))
    c, gT, gC = eng1.eval_single(T, cm, tstar=tg)
    c0, gT0, gC0, st0 = o1.eval(T, cm, tstar=to)
    assert c0 > 0
    assert abs(c - c0) <= REL_TOL * abs(c0)
    # free-running argmin: since the device's winding number is the library's bit for bit, t* has to be identical
    assert_mesh_v1_free_running("mesh-720 V1 (test_gpu_mesh)", tg, to, gT, gT0, gC, gC0)


@pytest.mark.parametrize("name", ["L prism (20)", "blob 720", "blob 4992"])
def test_mesh_v1_arithmetic_at_the_reference_minimisers(pkg, orc, product_lib, name):
    """Mesh robot x swept-volume sweep with the argmin taken out of the comparison: the oracle (reference winding-number
    library) finds t* per obstacle point, the device evaluates the SAME back-prop (SDF and unit gradient at t*, penalty, pose
    Jacobians, flatness reverse mode, per-piece sums) at those t* (isdf_eval_swept_at_tstar) - cost, gradC and gradT must
    agree at the north-star's 1e-5.  What remains of the free-running difference is then the argmin's own sensitivity."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=5, seed=23)
    V, F = {"L prism (20)": synth.l_prism_mesh, "blob 720": lambda: synth.blob_mesh(24, 16, scale=0.9),
            "blob 4992": lambda: synth.blob_mesh(64, 40, scale=0.9)}[name]()
    shape = synth.make_mesh_shape(V, F)
    way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    if name == "blob 4992":
        pts = np.ascontiguousarray(pts[::3])               # the oracle's exhaustive closest-point search is the slow part
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    eng1, o1 = make_pair(pkg, orc, cfg1, shape, occ=occ, res=res, points=pts, threads=30)
    _need_hdk(o1)
    to = -np.ones(len(pts))                                # -1 survives exactly where no time interval qualified
    c0, gT0, gC0, st0 = o1.eval(T, cm, tstar=to)
    assert c0 > 0 and (to >= 0).sum() > 50
    c, gT, gC = eng1.eval_swept_at_tstar_host(T, cm, to)
    assert abs(c - c0) <= REL_TOL * abs(c0), (c, c0)
    assert_close(gC, gC0, f"{name} gradC at the reference's t*")
    assert_close(gT, gT0, f"{name} gradT at the reference's t*")


@pytest.mark.parametrize("stem", ["Trefoil", "Lthick", "mybox", "icosahedron", "drone", "kuang", "box"])
def test_reference_meshes_through_the_mesh_kind(pkg, orc, product_lib, stem):
    """"C5 mesh Lthick.obj (concave, 20 tris) and Trefoil.obj (12 k tris) scaled to the same bound, through the FWN
    path" - the reference's own files (tests/golden/ref_meshes.npz; isdf_read_obj's view of src/plan_manager/shapes/*.obj), as
    Generalshape would load them (sw_manager.hpp:255-275 falls through to the mesh class for a stem the registry does not
    know; Shape.cpp:27-151): shape plugin, then the swept-volume sweep with both sides searching their own minimisers."""
    capi, synth = pkg.capi, pkg.synth
    V, F = reference_mesh(stem, bound_radius=1.66)
    # (round 6: + the reference's small robot meshes - mybox / icosahedron take the FLAT evaluation like Lthick, drone / kuang / box the quad walks)
    assert len(F) == {"Trefoil": 12000, "Lthick": 20, "mybox": 12, "icosahedron": 18, "drone": 52, "kuang": 60, "box": 96}[stem]
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config()
    eng = pkg.Engine(cfg); o = orc.Oracle(cfg, threads=30)
    eng.set_shape(shape); o.set_shape(shape)
    _need_hdk(o)
    rng = np.random.default_rng(5)
    Q = _queries(V, rng, 600 if stem == "Trefoil" else 1500)
    s, g = eng.shape_eval(Q)
    s0, g0 = o.shape_eval(Q)
    assert np.all(np.abs(s - s0) <= 1e-13 * np.abs(s0) + 1e-14), (stem, np.abs(s - s0).max())
    assert (np.linalg.norm(g - g0, axis=1) <= 1e-6).mean() > 0.995
    # V1 sweep
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=4, seed=23, piece_T=1.0)
    way = cm.reshape(3, -1).T.reshape(4, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=2.5)
    if stem == "Trefoil":
        pts = np.ascontiguousarray(pts[::4])               # (the oracle's exhaustive closest-point search over 12 000 faces)
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    eng1, o1 = make_pair(pkg, orc, cfg1, shape, occ=occ, res=res, points=pts, threads=30)
    tg = -np.ones(len(pts)); to = -np.ones(len(pts))
    c, gT, gC = eng1.eval_single(T, cm, tstar=tg)
    c0, gT0, gC0, _ = o1.eval(T, cm, tstar=to)
    assert c0 > 0 and (to >= 0).sum() > 20
    assert abs(c - c0) <= REL_TOL * abs(c0), (c, c0)
    from common import assert_mesh_v1_free_running
    assert_mesh_v1_free_running(f"{stem}.obj V1", tg, to, gT, gT0, gC, gC0)


def test_mesh_with_degenerate_and_duplicate_triangles(pkg, orc, product_lib):
    """Zero-area triangles, duplicated triangles and an open surface: the winding-number hierarchy (zero-area items get zero
    coefficients, HDK :6700) and the closest-point walk must still agree with the reference library / the exact search."""
    synth = pkg.synth
    V, F = synth.blob_mesh(8, 6)
    V = np.concatenate([V, V[:3] + 0.0])                        # duplicate vertices
    deg = np.array([[0, 0, 1], [2, 2, 2], [len(V) - 3, len(V) - 2, len(V) - 1]], dtype=np.int32)     # degenerate + a copy of face material
    F2 = np.concatenate([F[:-5], deg, F[:4]]).astype(np.int32)                                       # 5 faces removed (open), 4 doubled
    shape = synth.make_mesh_shape(V, F2)
    cfg = synth.default_config()
    eng = pkg.Engine(cfg); o = orc.Oracle(cfg, threads=8)
    eng.set_shape(shape); o.set_shape(shape)
    _need_hdk(o)
    rng = np.random.default_rng(5)
    Q = _queries(np.asarray(V, dtype=np.float64), rng, 1500)
    s, g = eng.shape_eval(Q)
    s0, g0 = o.shape_eval(Q)
    assert np.all(np.isfinite(s)) and np.all(np.isfinite(s0))
    assert np.all(np.abs(s - s0) <= 5e-6 * np.abs(s0) + 4e-6 * np.abs(Q).max())
    assert (np.linalg.norm(g - g0, axis=1) <= 1e-6).mean() > 0.995     # doubled faces tie exactly: either copy's closest point


def test_mesh_v1_does_not_depend_on_the_dispatch_orders(pkg, orc, product_lib, monkeypatch):
    """Mesh robots, V1: the scan's points and the descents are dispatched longest-first by LAST step's measured durations (the
    prepare kernel's counting sort, the task list's front / back) - scheduling only: repeated steps on one ctx (whose orders
    change from step to step), a fresh ctx and ISDF_NO_LPT=1 return the same bits, t* included."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=5, seed=23)
    V, F = synth.blob_mesh(24, 16, scale=0.9)
    way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)

    def mk():
        e = pkg.Engine(cfg1); e.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e.set_shape(synth.make_mesh_shape(V, F)); e.set_points(pts)
        return e

    def step(e, T_, cm_):
        ts = -np.ones(len(pts))
        c, gT, gC = e.eval_single(T_, cm_, tstar=ts)
        return c, gT, gC, ts

    e = mk()
    ref = step(e, T, cm)
    assert ref[0] > 0 and (ref[3] >= 0).any()
    T2, cm2 = traj(pkg, occ, res, N=5, seed=29)
    other = step(e, T2, cm2)                               # another trajectory in between: the durations it leaves order the next step
    for _ in range(3):
        again = step(e, T, cm)
        assert again[0] == ref[0] and all(np.array_equal(x, y) for x, y in zip(again[1:], ref[1:]))
    monkeypatch.setenv("ISDF_NO_LPT", "1")
    plain = mk()
    monkeypatch.delenv("ISDF_NO_LPT")
    for want, (T_, cm_) in ((ref, (T, cm)), (other, (T2, cm2))):
        got = step(plain, T_, cm_)
        assert got[0] == want[0] and all(np.array_equal(x, y) for x, y in zip(got[1:], want[1:]))


def test_mesh_walk_forms_agree(pkg, orc, product_lib, monkeypatch):
    """The mesh kind has two forms of each sweep: V1 with one task per wavefront (wave-cooperative walks, small meshes) or per
    workgroup (quad-cooperative walks; ISDF_MESH_WG forces either), and the tile sweep with its exact pass inside the sweep kernel
    or as its own launch over a queue of 16-voxel blocks (ISDF_MESH_QUEUE=0 forces the former).  Every query sees the same
    arithmetic either way: V1 must agree BITWISE (t*, cost, gradients - the reduction over the points is the same), the tile
    sweep to rounding (its per-sample sums are partitioned into blocks of 16 / of up to 64 voxels)."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=5, seed=23)
    V, F = synth.blob_mesh(24, 16, scale=0.9)
    way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    out = {}
    for wg in ("0", "1"):
        monkeypatch.setenv("ISDF_MESH_WG", wg)
        cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
        e1 = pkg.Engine(cfg1); e1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e1.set_shape(synth.make_mesh_shape(V, F)); e1.set_points(pts)
        ts = -np.ones(len(pts))
        c, gT, gC = e1.eval_single(T, cm, tstar=ts)
        out[wg] = (c, gT, gC, ts, e1.stats())
    monkeypatch.delenv("ISDF_MESH_WG")
    a, b = out["0"], out["1"]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert a[4]["pairs"] == b[4]["pairs"] and a[4]["grad_pairs"] == b[4]["grad_pairs"] > 0
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=12, safety_hor=0.5)
    res3 = {}
    for q in ("1", "0"):
        monkeypatch.setenv("ISDF_MESH_QUEUE", q)
        e = pkg.Engine(cfg); e.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e.set_shape(synth.make_mesh_shape(V, F))
        res3[q] = e.eval_single(T, cm) + (e.stats(),)
    monkeypatch.delenv("ISDF_MESH_QUEUE")
    a, b = res3["1"], res3["0"]
    assert a[0] > 0 and abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
    assert_close(a[1], b[1], "queued vs in-kernel exact pass gradT", tol=1e-11); assert_close(a[2], b[2], "queued vs in-kernel exact pass gradC", tol=1e-11)
    assert a[3]["pairs"] == b[3]["pairs"] and a[3]["grad_pairs"] == b[3]["grad_pairs"] and a[3]["overflow"] == 0
    # a batch through the queue == each trajectory alone, bitwise (a sample's blocks depend on its own candidates only)
    Ts, Cs = zip(*[traj(pkg, occ, res, N=5, seed=40 + k) for k in range(4)])
    e = pkg.Engine(cfg); e.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e.set_shape(synth.make_mesh_shape(V, F))
    cost, gTs, gCs = e.eval(list(Ts), list(Cs))
    for k in range(4):
        c1, gT1, gC1 = e.eval_single(Ts[k], Cs[k])
        assert c1 == cost[k] and np.array_equal(gT1, gTs[k]) and np.array_equal(gC1, gCs[k])


def test_deep_unbalanced_mesh_is_refused_by_the_tile_sweep(pkg, orc, product_lib):
    """A strongly unbalanced mesh (triangles in geometric progression: the hierarchy degenerates to 14 levels) exceeds the 12
    levels of walk frames the tile sweep keeps in LDS: V2 / V3 refuse it with ISDF_ERR_UNSUPPORTED instead of walking past their
    frames; the shape plugin itself and the swept-volume sweep still take it (per-lane / wave walks)."""
    capi, synth = pkg.capi, pkg.synth
    n = 60
    V, F = [], []
    for i in range(n):
        x = 2.0 ** i * 1e-9; s = 0.1 * x
        V += [[x, 0, 0], [x + s, 0, 0], [x, s, 0.1 * s]]
        F.append([3 * i, 3 * i + 1, 3 * i + 2])
    V = np.array(V); F = np.array(F, dtype=np.int32)
    V *= 1.5 / np.abs(V).max()
    shape = synth.make_mesh_shape(V, F)
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res, N=4)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    eng.set_shape(shape)
    with pytest.raises(pkg.engine.IsdfError) as ei:
        eng.eval_single(T, cm)
    assert ei.value.code == capi.ISDF_ERR_UNSUPPORTED
    # the plugin itself still evaluates it
    o = orc.Oracle(cfg, threads=4); o.set_shape(shape)
    _need_hdk(o)
    Q = _queries(V, np.random.default_rng(2), 400)
    s, g = eng.shape_eval(Q)
    s0, g0 = o.shape_eval(Q)
    assert np.all(np.isfinite(s)) and np.all(np.abs(s - s0) <= 5e-6 * np.abs(s0) + 4e-6 * np.abs(Q).max())


def test_mesh_distance_lattice_filter_rejects_only_zero_penalties(pkg, orc, product_lib, monkeypatch):
    """The tile sweep's pre-filter for mesh robots (a lattice of node-to-surface distances built by isdf_set_shape) may only drop
    voxels without a penalty: with it and without it (ISDF_NO_F32_FILTER=1) the same active pairs, cost and gradients to rounding
    (a sample's candidates are cut into blocks of 16 at other places), fewer exact pairs.  An OPEN mesh gets no lattice - next to
    it the winding number is a fraction and (1 - 2 w) * distance says nothing about the distance."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=5, seed=23)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=12, safety_hor=0.5)
    V, F = synth.blob_mesh(24, 16, scale=0.9)
    F_open = np.ascontiguousarray(F[: len(F) - 40])                  # the same surface with a hole
    F_flip = np.ascontiguousarray(F[:, ::-1])                        # closed, but inside out: the winding number is -1 inside
    for faces, closed in ((F, True), (F_open, False), (F_flip, False)):
        out = {}
        for off in ("0", "1"):
            monkeypatch.setenv("ISDF_NO_F32_FILTER", off)
            e = pkg.Engine(cfg); e.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e.set_shape(synth.make_mesh_shape(V, faces))
            out[off] = e.eval_single(T, cm) + (e.stats(),)
        monkeypatch.delenv("ISDF_NO_F32_FILTER")
        a, b = out["0"], out["1"]
        assert a[0] > 0 and abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
        assert_close(a[1], b[1], "lattice filter on / off gradT", tol=1e-11); assert_close(a[2], b[2], "lattice filter on / off gradC", tol=1e-11)
        assert a[3]["grad_pairs"] == b[3]["grad_pairs"] > 0 and a[3]["overflow"] == 0
        if closed: assert a[3]["pairs"] < b[3]["pairs"]
        else: assert a[3]["pairs"] == b[3]["pairs"]


@pytest.mark.parametrize("case", ["blob 720", "blob 168", "Trefoil.obj"])
def test_mesh_v1_distance_lattice_changes_no_result(pkg, orc, product_lib, monkeypatch, case):
    """The swept-volume sweep uses the same lattice to leave queries out whose answer cannot matter (a capped query the lattice
    already answers; fine samples that cannot hold their pass's minimum; coarse samples inside the band behind another such
    sample).  What is left out is never looked at: with and without the lattice (ISDF_NO_F32_FILTER=1 builds none) t*, cost and
    gradients are the same BITS, in both forms of the walks."""
    capi, synth = pkg.capi, pkg.synth
    seed_w, seed_t, safety = {"blob 720": (14, 23, 0.5), "blob 168": (14, 29, 0.35), "Trefoil.obj": (16, 31, 0.3)}[case]
    occ, esdf, res = small_world(pkg, seed=seed_w)
    T, cm = traj(pkg, occ, res, N=5, seed=seed_t)
    if case == "Trefoil.obj": V, F = reference_mesh("Trefoil", 0.83)
    elif case == "blob 720": V, F = synth.blob_mesh(24, 16, scale=0.9)
    else: V, F = synth.blob_mesh(12, 8, scale=0.9)
    way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    for wg in ("1", "0"):
        monkeypatch.setenv("ISDF_MESH_WG", wg)
        out = {}
        for off in ("0", "1"):
            monkeypatch.setenv("ISDF_NO_F32_FILTER", off)
            cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=safety)
            e1 = pkg.Engine(cfg1); e1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e1.set_shape(synth.make_mesh_shape(V, F)); e1.set_points(pts)
            ts = -np.ones(len(pts))
            steps = []
            for _ in range(3):                       # (lastTstar and the dispatch orders carry over from step to step)
                c, gT, gC = e1.eval_single(T, cm, tstar=ts)
                steps.append((c, gT.copy(), gC.copy(), ts.copy()))
            out[off] = (steps, e1.stats())
        for a, b in zip(out["0"][0], out["1"][0]):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert out["0"][1]["grad_pairs"] == out["1"][1]["grad_pairs"] > 0
    monkeypatch.delenv("ISDF_MESH_WG"); monkeypatch.delenv("ISDF_NO_F32_FILTER")


ADVERSARIAL = ["thin shell 1 mm", "shell in shell", "shell in shell, 2 mm apart", "two sheets a cell apart", "overlapping boxes", "inverted box", "open box",
               "mybox.obj", "Lthick.obj x 0.1", "Lthick.obj x 10", "box.obj", "kuang.obj", "drone.obj", "icosahedron.obj"]


@pytest.mark.parametrize("name", ADVERSARIAL)
def test_distance_lattice_on_adversarial_meshes(pkg, orc, product_lib, monkeypatch, name):
    """The distance lattice rests on two facts about a mesh, both CHECKED by isdf_set_shape before one is built (isdf_mesh_info): the
    surface bounds a solid (exact winding number 0 / 1 either side of every face) and the reference's approximate winding number stays
    within 4 % of that at the lattice's nodes, cell centres and edge midpoints.  Meshes that break the first - a closed sheet inside
    another of the same orientation (also a hair inside it), overlapping bodies (a doubly covered slab thinner than a lattice cell),
    an inverted body, an open one - get NO lattice; a thin-walled hollow shell, two bodies a lattice cell apart, the reference's own
    small robot meshes (triangle soups: no shared vertices) and Lthick.obj at a tenth and ten times the bench's scale do.  Either way
    the swept-volume sweep's t*, cost and gradients are the same BITS with and without it, in every walk form the mesh can take."""
    from benchlib.meshes import adversarial_mesh
    capi, synth = pkg.capi, pkg.synth
    V, F, expect_solid = adversarial_mesh(name)
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=5, seed=23)
    way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    if name == "Lthick.obj x 10":
        pts = pts[::6]                       # (an 8 m robot covers every point of this map: a sixth of them is plenty)
    infos = {}
    across = []
    # every form of the evaluation the mesh can take: flat (small meshes; ISDF_MESH_FLAT=0 switches it off), quad walks, wave walks
    for form, flat, wg in (("flat", "1", "1"), ("quad", "0", "1"), ("wave", "0", "0")):
        monkeypatch.setenv("ISDF_MESH_FLAT", flat); monkeypatch.setenv("ISDF_MESH_WG", wg)
        out = {}
        for off in ("0", "1"):
            monkeypatch.setenv("ISDF_NO_F32_FILTER", off)
            cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.3)
            e1 = pkg.Engine(cfg1); e1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e1.set_shape(synth.make_mesh_shape(V, F)); e1.set_points(pts)
            infos[(wg if form in ("quad", "wave") else form, off)] = e1.mesh_info()
            ts = -np.ones(len(pts))
            steps = []
            for _ in range(2):
                c, gT, gC = e1.eval_single(T, cm, tstar=ts)
                steps.append((c, gT.copy(), gC.copy(), ts.copy()))
            out[off] = (steps, e1.stats())
        for a, b in zip(out["0"][0], out["1"][0]):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), form
        assert out["0"][1] == out["1"][1]
        across.append((form, out["0"]))
    for form, o in across[1:]:             # ... and the forms agree with each other, bit for bit
        for a, b in zip(across[0][1][0], o[0]):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), (across[0][0], form)
        assert across[0][1][1] == o[1]
    monkeypatch.delenv("ISDF_MESH_WG"); monkeypatch.delenv("ISDF_NO_F32_FILTER"); monkeypatch.delenv("ISDF_MESH_FLAT")
    small = infos[("flat", "0")]["flat_slots"] > 0
    assert small == (name in ("mybox.obj", "Lthick.obj x 0.1", "Lthick.obj x 10", "icosahedron.obj")), infos[("flat", "0")]
    on, off = infos[("1", "0")], infos[("1", "1")]
    assert off["lattice"] == (0, 0, 0) and off["solid"] == -1
    if expect_solid is not None:
        assert on["solid"] == expect_solid, on
    if on["solid"] == 1:
        assert min(on["lattice"]) >= 2 and 0.96 <= on["s_range"][0] <= on["s_range"][1] <= 1.04, on
    else:
        assert on["lattice"] == (0, 0, 0), on
    print(f"\nadversarial mesh {name!r}: {on}")
