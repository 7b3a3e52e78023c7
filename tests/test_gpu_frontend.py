"""Front end on the device (pose feasibility by kernel convolution) against the oracle, whose restatement is
pinned byte for byte against the reference's own code (tests/test_oracle_ref.py): integer / bit work => bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(pkg, orc, shape, occ, res, fe):
    capi, synth = pkg.capi, pkg.synth
    cfg = synth.default_config(capi.V1_SWEPT)
    eng = pkg.Engine(cfg); o = orc.Oracle(cfg, threads=16)
    for t in (eng, o):
        t.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
        t.set_shape(shape)
        t.frontend_build(fe)
    return eng, o


def _queries(occ, rng, n, xk=11):
    X, Y, Z = occ.shape
    idx = np.stack([rng.integers(-1, X + 1, n), rng.integers(-1, Y + 1, n), rng.integers(-1, Z + 1, n)], axis=1).astype(np.int32)   # incl. invalid and occupied voxels
    fr = rng.choice(np.arange(-45.0, 45.1, 9.0), n)
    fp = rng.choice(np.arange(-45.0, 45.1, 9.0), n)
    off = rng.random(n) < 0.3                       # parents whose attitude is not on the grid (the reference truncates)
    fr = np.clip(fr + off * rng.uniform(0.0, 8.9, n), -45.0, 45.0)
    return idx, fr, fp


@pytest.mark.parametrize("name", ["Box", "RoundedCone", "Torus_big", "BendLinear_big", "SmoothIntersection_big", "Trefoil", "CSG", "Table",
                                  "Cappedtorus", "WireframeBox", "TwistBox", "Ball", "mesh"])
def test_frontend_matches_oracle_bit_for_bit(pkg, orc, product_lib, name):
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occ = synth.random_box_map((40, 36, 20), res=res, occupancy=0.10, seed=5, edge=(0.5, 2.0))
    if name == "mesh":
        V, F = synth.l_prism_mesh()
        shape = synth.make_mesh_shape(V * 1.5, F)
    elif name == "Box":
        shape = synth.make_shape("Box", params=(0.9, 0.35, 0.25), poly_params=(0.05, -0.1, 0.02, 12, -8, 25))
    elif name == "Ball":
        shape = synth.make_shape("Ball", params=(0.8,))
    else:
        shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25))
    k = 13 if name in ("Box", "RoundedCone", "Ball", "mesh") else 15
    fe = capi.frontend_config(kernel_size=k, max_roll=45.0, max_pitch=45.0, ang_res=9.0, safeh=0.0)
    eng, o = _pair(pkg, orc, shape, occ, res, fe)
    if name == "mesh" and o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    sk, sk0 = eng.frontend_shape_kernels(), o.frontend_shape_kernels()
    assert sk0.any() and sk.shape == sk0.shape == (121, k * k * ((k + 7) // 8))
    nbad = int((np.unpackbits(sk ^ sk0)).sum())
    # transcendental shapes: device and host sin/cos/atan2 may differ in the last ulp, which can only matter for a voxel whose
    # SDF is within 1e-15 of the margin - none observed; the polynomial / sqrt-only shapes must be exact by construction
    assert nbad == 0, (name, nbad)
    assert np.array_equal(eng.frontend_map_kernel(), o.frontend_map_kernel())
    rng = np.random.default_rng(9)
    idx, fr, fp = _queries(occ, rng, 20000)
    ok, cr, cp, ki = eng.frontend_check(idx, fr, fp)
    ok0, cr0, cp0, ki0 = o.frontend_check(idx, fr, fp)
    assert np.array_equal(ok, ok0) and np.array_equal(ki, ki0)
    m = ok0 == 1
    assert 0 < m.sum() < len(m)
    assert np.array_equal(cr[m], cr0[m]) and np.array_equal(cp[m], cp0[m])
    # not every accepted query is the level attitude: the breadth-first search is exercised
    assert name == "Ball" or (ki0[m] != 60).any()      # (a ball looks the same at every attitude)


def test_frontend_axis_aligned_box_on_the_margin(pkg, orc, product_lib):
    """A box whose faces sit exactly half a voxel from voxel centres at the level attitude (sdf == margin up to rounding):
    the comparison sdf <= margin must round like the reference's - the reason frontend.hip is built without FMA contraction."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.2
    occ = synth.random_box_map((32, 32, 32), res=res, occupancy=0.12, seed=8, edge=(0.4, 1.6))
    shape = synth.make_shape("Box", params=(0.9, 0.3, 0.3))            # identity body offset; 0.3 + res / 2 = 0.4 = 2 voxels
    fe = capi.frontend_config(kernel_size=21, max_roll=45.0, max_pitch=45.0, ang_res=9.0, safeh=0.0)
    eng, o = _pair(pkg, orc, shape, occ, res, fe)
    assert np.array_equal(eng.frontend_shape_kernels(), o.frontend_shape_kernels())
    rng = np.random.default_rng(1)
    idx, fr, fp = _queries(occ, rng, 5000)
    a, b = eng.frontend_check(idx, fr, fp), o.frontend_check(idx, fr, fp)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])


def test_frontend_needs_build_and_tracks_state(pkg, product_lib):
    capi, synth = pkg.capi, pkg.synth
    eng = pkg.Engine(synth.default_config(capi.V1_SWEPT))
    with pytest.raises(Exception):
        eng.frontend_check(np.zeros((1, 3), dtype=np.int32), 0.0, 0.0)
    occ = synth.random_box_map((16, 16, 16), res=0.5, occupancy=0.1, seed=2, edge=(0.5, 1.0))
    eng.set_grid(occ, (0, 0, 0), 0.5, capi.GRID_OCCUPANCY)
    eng.set_shape(synth.make_shape("Box", params=(0.5, 0.3, 0.2)))
    with pytest.raises(Exception):
        eng.frontend_build(capi.frontend_config(kernel_size=12))       # even
    eng.frontend_build(capi.frontend_config(kernel_size=9))
    eng.frontend_check(np.array([[8, 8, 8]], dtype=np.int32), 0.0, 0.0)
    eng.set_shape(synth.make_shape("Box", params=(0.6, 0.3, 0.2)))      # tables are stale now
    with pytest.raises(Exception):
        eng.frontend_check(np.array([[8, 8, 8]], dtype=np.int32), 0.0, 0.0)


@pytest.mark.parametrize("name", ["Box", "Torus_big", "mesh"])
def test_whole_configuration_space_matches_kernel_conv(pkg, orc, product_lib, name):
    """isdf_frontend_cspace: every voxel x every attitude in one pass == the oracle's kernelConv per (voxel, attitude), and the
    batched checkKernelValue answers are reproduced from the table with the breadth-first order."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occ = synth.random_box_map((40, 36, 70), res=res, occupancy=0.10, seed=6, edge=(0.5, 2.0))      # Z > 64: two z-blocks per column
    if name == "mesh":
        V, F = synth.l_prism_mesh()
        shape = synth.make_mesh_shape(V * 1.5, F)
    elif name == "Box":
        shape = synth.make_shape("Box", params=(0.9, 0.35, 0.25), poly_params=(0.05, -0.1, 0.02, 12, -8, 25))
    else:
        shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25))
    fe = capi.frontend_config(kernel_size=13 if name != "Torus_big" else 15)
    eng, o = _pair(pkg, orc, shape, occ, res, fe)
    if name == "mesh" and o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    table, ms = eng.frontend_cspace()
    ref = o.frontend_cspace(occ.shape)
    assert table.shape == ref.shape and np.array_equal(table, ref)
    assert ref.any() and (ref[occ == 1] == 0).all()
    # checkKernelValue from the table: first attitude of the breadth-first order whose bit is set
    rng = np.random.default_rng(4)
    idx, fr, fp = _queries(occ, rng, 3000)
    ok, cr, cp, ki = eng.frontend_check(idx, fr, fp)
    X, Y, Z = occ.shape
    inside = (idx >= 0).all(axis=1) & (idx[:, 0] < X) & (idx[:, 1] < Y) & (idx[:, 2] < Z)
    for q in np.flatnonzero(inside)[:1500]:
        m = table[idx[q, 0], idx[q, 1], idx[q, 2]]
        bit = lambda a: (int(m[a >> 5]) >> (a & 31)) & 1
        assert (ok[q] == 1) == any(bit(a) for a in range(121))      # some attitude fits <=> the search succeeds (it visits all 121)
        if ok[q]:
            assert bit(int(ki[q]))


def test_frontend_edge_configurations(pkg, orc, product_lib):
    """Degenerate attitude grids and kernels: a single attitude (ang_res > 2 * max), kernel_size 1 (the robot is one voxel),
    rectangular grids (roll range != pitch range), parents outside the attitude range (rejected, never indexed)."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occ = synth.random_box_map((24, 20, 16), res=res, occupancy=0.15, seed=3, edge=(0.5, 1.5))
    shape = synth.make_shape("Box", params=(0.7, 0.3, 0.2))
    rng = np.random.default_rng(0)
    X, Y, Z = occ.shape
    idx = np.stack([rng.integers(0, X, 4000), rng.integers(0, Y, 4000), rng.integers(0, Z, 4000)], axis=1).astype(np.int32)
    for fe, rolls, pitches in (
            (capi.frontend_config(kernel_size=9, max_roll=10.0, max_pitch=10.0, ang_res=30.0), [-10.0], [-10.0]),          # 1 x 1
            (capi.frontend_config(kernel_size=1, max_roll=45.0, max_pitch=45.0, ang_res=9.0), np.arange(-45.0, 45.1, 9.0), np.arange(-45.0, 45.1, 9.0)),
            (capi.frontend_config(kernel_size=7, max_roll=20.0, max_pitch=40.0, ang_res=10.0), np.arange(-20.0, 20.1, 10.0), np.arange(-40.0, 40.1, 10.0))):
        eng, o = _pair(pkg, orc, shape, occ, res, fe)
        assert np.array_equal(eng.frontend_shape_kernels(), o.frontend_shape_kernels())
        assert np.array_equal(eng.frontend_map_kernel(), o.frontend_map_kernel())
        fr = rng.choice(rolls, 4000); fp = rng.choice(pitches, 4000)
        a, b = eng.frontend_check(idx, fr, fp), o.frontend_check(idx, fr, fp)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
        m = b[0] == 1
        assert np.array_equal(a[1][m], b[1][m]) and np.array_equal(a[2][m], b[2][m])
    # a parent attitude outside the grid: the reference would index visited[][] out of bounds; here the query is rejected
    ok, _, _, ki = eng.frontend_check(idx[:8], 500.0, 0.0)
    assert not ok.any() and (ki == -1).all()


def _astar_pairs(table, res, rng, n):
    good = np.argwhere((table != 0).any(axis=-1))
    out = []
    for _ in range(n):
        a, b = good[rng.choice(len(good), 2, replace=False)]
        out.append(((a + rng.random(3)) * res, (b + rng.random(3)) * res))
    return out


@pytest.mark.parametrize("name", ["Box", "RoundedCone", "Torus_big", "mesh"])
def test_astar_matches_oracle_node_for_node(pkg, orc, product_lib, name):
    """isdf_frontend_astar_search / _path (the SE(3) A* over the device-built configuration space) against the oracle's restatement
    of AstarPathSearcher (front_end_Astar.hpp:172-403; pinned against the reference's own code in tests/test_oracle_ref.py and
    tests/golden/ref_astar_*.npz): the same cells, the same (roll, pitch) at every node, the same number of neighbour tests and
    expansions, the same failures."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occupancy = {"Box": 0.24, "RoundedCone": 0.10, "Torus_big": 0.14, "mesh": 0.20}[name]
    occ = synth.random_box_map((44, 40, 14), res=res, occupancy=occupancy, seed=31, edge=(0.5, 2.5))
    if name == "mesh":
        V, F = synth.l_prism_mesh()
        shape = synth.make_mesh_shape(V * 1.5, F)
    elif name == "Box":
        shape = synth.make_shape("Box", params=(0.9, 0.35, 0.25), poly_params=(0.05, -0.1, 0.02, 12, -8, 25))
    else:
        shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25))
    fe = capi.frontend_config(kernel_size=15 if name == "Torus_big" else 13)
    eng, o = _pair(pkg, orc, shape, occ, res, fe)
    if name == "mesh" and o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    table, _ = eng.frontend_cspace()
    rng = np.random.default_rng(12)
    found = tilted = 0
    first = True
    for s, g in _astar_pairs(table, res, rng, 10) + [(np.array([1.0, 1.0, -0.1]), np.array([3.0, 3.0, 3.0]))]:
        xyz, rp, rot, r = eng.frontend_astar(s, g)
        x0, rp0, st = o.frontend_astar(s, g)
        assert (r.checks, r.expansions) == (st["checks"], st["expansions"]), (name, r.checks, st)
        assert (first and r.table_ms > 0 and r.cspace_ms > 0) or (not first and r.table_ms == 0)        # the table is computed once
        first = False
        assert (xyz is None) == (x0 is None)
        if x0 is None:
            assert r.success == 0 and r.n_path == 0
            continue
        found += 1
        assert r.success == 1 and r.n_path == len(x0)
        assert np.array_equal(xyz, x0) and np.array_equal(rp, rp0)
        tilted += int((rp0 != 0).any())
        # every node of the path is a cell where its attitude fits; consecutive cells are 26-neighbours
        cells = np.floor(xyz / res).astype(int)
        assert (np.abs(np.diff(cells, axis=0)).max(axis=1) == 1).all() or len(cells) == 1
    assert found >= 5 and (tilted >= 2 or name == "RoundedCone"), (name, found, tilted)
    # the path read-back: capacity shorter than the path, NULL outputs
    C = __import__("ctypes")
    n = eng.lib.isdf_frontend_astar_path(eng.h, 0, None, None, None)
    assert n == (0 if xyz is None else len(xyz))


def test_astar_follows_the_table_after_a_rebuild(pkg, orc, product_lib):
    """A new map / robot + isdf_frontend_build drops the cached configuration space: the next search recomputes it."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    shape = synth.make_shape("Box", params=(0.9, 0.35, 0.25))
    fe = capi.frontend_config(kernel_size=13)
    occ_a = synth.random_box_map((32, 32, 12), res=res, occupancy=0.15, seed=41, edge=(0.5, 2.0))
    occ_b = synth.random_box_map((32, 32, 12), res=res, occupancy=0.15, seed=42, edge=(0.5, 2.0))
    eng, o = _pair(pkg, orc, shape, occ_a, res, fe)
    rng = np.random.default_rng(2)
    for occ in (occ_a, occ_b):
        if occ is occ_b:
            for t in (eng, o):
                t.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
            with pytest.raises(Exception):
                eng.frontend_astar((1.0, 1.0, 1.0), (5.0, 5.0, 3.0))         # stale tables are refused like every front-end call
            for t in (eng, o):
                t.frontend_build(fe)
        table, _ = eng.frontend_cspace()
        n_table = 0
        for s, g in _astar_pairs(table, res, rng, 4):
            xyz, rp, rot, r = eng.frontend_astar(s, g)
            n_table += int(r.table_ms > 0)
            x0, rp0, st = o.frontend_astar(s, g)
            assert r.checks == st["checks"] and (xyz is None) == (x0 is None)
            if x0 is not None:
                assert np.array_equal(xyz, x0) and np.array_equal(rp, rp0)
        assert n_table == 1


def test_astar_other_grids_and_origins(pkg, orc, product_lib):
    """The A* with a map origin off zero, a 0.2 m grid, a rectangular attitude grid whose spacing is not an integer (roll +-30 in
    steps of 7.5 -> 9, pitch +-15 -> 5 attitudes), kernel_size 9 - against the oracle, node for node; start == goal cell; a goal in an
    occupied voxel (the search exhausts what it can reach and fails)."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.2
    origin = (-3.7, 12.25, 0.4)
    occ = synth.random_box_map((60, 52, 24), res=res, occupancy=0.12, seed=77, edge=(0.4, 1.4))
    shape = synth.make_shape("Box", params=(0.55, 0.16, 0.12), poly_params=(0.02, -0.03, 0.01, 5, -8, 10))
    fe = capi.frontend_config(kernel_size=9, max_roll=30.0, max_pitch=15.0, ang_res=7.5, safeh=0.0)
    cfg = synth.default_config(capi.V1_SWEPT)
    eng = pkg.Engine(cfg); o = orc.Oracle(cfg, threads=16)
    for t in (eng, o):
        t.set_grid(occ, origin, res, capi.GRID_OCCUPANCY); t.set_shape(shape); t.frontend_build(fe)
    table, _ = eng.frontend_cspace()
    good = np.argwhere((table != 0).any(axis=-1))
    rng = np.random.default_rng(3)
    cases = []
    for _ in range(8):
        a, b = good[rng.choice(len(good), 2, replace=False)]
        cases.append(((a + rng.random(3)) * res + origin, (b + rng.random(3)) * res + origin))
    cases.append((cases[0][0], cases[0][0] + 1e-3))                                         # one cell
    occv = np.argwhere(occ == 1)
    cases.append((cases[1][0], (occv[rng.choice(len(occv))] + 0.5) * res + origin))        # goal inside an obstacle
    cases.append((np.array(origin) - 0.01, cases[2][1]))                                    # start outside the map
    found = tilted = 0
    for q, (s, g) in enumerate(cases):
        xyz, rp, rot, r = eng.frontend_astar(s, g)
        x0, rp0, st = o.frontend_astar(s, g)
        assert (r.checks, r.expansions) == (st["checks"], st["expansions"]), (q, r.checks, st)
        assert (xyz is None) == (x0 is None), q
        if x0 is not None:
            found += 1
            assert np.array_equal(xyz, x0) and np.array_equal(rp, rp0), q
            tilted += int((rp0 != 0).any())
            assert set(np.unique(rp0[:, 0])) <= set(np.arange(-30.0, 30.1, 7.5)) and set(np.unique(rp0[:, 1])) <= set(np.arange(-15.0, 15.1, 7.5))
    assert cases and found >= 6 and tilted >= 1, (found, tilted)
    assert eng.frontend_astar(*cases[-1])[3].success == 0 and eng.frontend_astar(*cases[-2])[3].success == 0
    x1 = eng.frontend_astar(*cases[-3])[0]
    assert x1 is not None and len(x1) == 1


def test_more_than_128_attitudes(pkg, orc, product_lib):
    """kernel_ang_res 5 degrees: 19 x 19 = 361 attitudes -> 12 dwords per voxel in the configuration-space table (three 128-attitude
    groups, the last one partly filled); table == the oracle's kernelConv per (voxel, attitude), A* == the oracle's, node for node."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occ = synth.random_box_map((30, 28, 12), res=res, occupancy=0.2, seed=91, edge=(0.5, 2.0))
    shape = synth.make_shape("Box", params=(0.9, 0.35, 0.25), poly_params=(0.05, -0.1, 0.02, 12, -8, 25))
    fe = capi.frontend_config(kernel_size=11, max_roll=45.0, max_pitch=45.0, ang_res=5.0, safeh=0.0)
    eng, o = _pair(pkg, orc, shape, occ, res, fe)
    assert o.fe_dims[:2] == [19, 19]
    table, _ = eng.frontend_cspace()
    ref = o.frontend_cspace(occ.shape)
    assert table.shape == ref.shape == occ.shape + (12,) and np.array_equal(table, ref)
    assert ref[..., 4:8].any() and ref[..., 8:].any() and (ref[..., 11] >> 9 == 0).all()          # bits 361 .. 383 stay clear
    rng = np.random.default_rng(8)
    found = high = 0
    for s, g in _astar_pairs(table, res, rng, 8):
        xyz, rp, rot, r = eng.frontend_astar(s, g)
        x0, rp0, st = o.frontend_astar(s, g)
        assert (r.checks, r.expansions) == (st["checks"], st["expansions"]) and (xyz is None) == (x0 is None)
        if x0 is not None:
            found += 1
            assert np.array_equal(xyz, x0) and np.array_equal(rp, rp0)
            high += int((np.abs(rp0) > 0).any())
    assert found >= 4 and high >= 1, (found, high)
    # the batched per-neighbour test agrees with the table beyond attitude 127 as well
    idx, fr, fp = _queries(occ, rng, 4000, xk=19)
    fr = rng.choice(np.arange(-45.0, 45.1, 5.0), len(fr)); fp = rng.choice(np.arange(-45.0, 45.1, 5.0), len(fp))
    ok, cr, cp, ki = eng.frontend_check(idx, fr, fp)
    ok0, cr0, cp0, ki0 = o.frontend_check(idx, fr, fp)
    assert np.array_equal(ok, ok0) and np.array_equal(ki, ki0) and (ki0 > 127).any()
