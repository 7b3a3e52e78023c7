"""GPU parity proper: the HIP path through the C-ABI vs the CPU oracle on identical seeded inputs.
Tolerance: 1e-5 relative (norm-wise on gradients), the figure BASELINE.json's north_star states."""
import numpy as np
import pytest

from common import REL_TOL, assert_close, make_pair, small_world, traj

pytestmark = pytest.mark.gpu


def _check(eng, o, T, cm, what, tstar=False):
    cost, gT, gC = eng.eval_single(T, cm)
    c0, gT0, gC0, st0 = o.eval(T, cm)
    assert abs(cost - c0) <= REL_TOL * max(abs(c0), 1e-9), (what, cost, c0)
    assert_close(gT, gT0, what + " gradT")
    assert_close(gC, gC0, what + " gradC")
    return cost, st0


@pytest.mark.parametrize("variant", ["V3", "V2"])
@pytest.mark.parametrize("shape_name", ["RoundedCone", "Box"])
def test_tile_sweep_matches_oracle(pkg, orc, product_lib, variant, shape_name):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res)
    v = capi.V3_ESDF_TILE if variant == "V3" else capi.V2_OCC_TILE
    cfg = synth.default_config(v, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    if shape_name == "Box":
        shape = synth.make_shape("Box", params=(1.2, 0.4, 0.3), grad_mode=capi.GRAD_CENTRAL)
    else:
        shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    eng, o = make_pair(pkg, orc, cfg, shape, occ=occ, esdf=esdf, res=res)
    cost, st0 = _check(eng, o, T, cm, f"{variant}/{shape_name}")
    assert cost > 0, "scenario must exercise the collision term"
    st = eng.stats()
    # units and active-penalty pairs are exact counts; "pairs" on the GPU counts only the voxels that survive the
    # conservative fp32 pre-filter (every active pair must be among them)
    assert st["units"] == st0[0] and st["grad_pairs"] == st0[3] and st0[3] <= st["pairs"] <= st0[2], (st, st0)


@pytest.mark.parametrize("K", [127, 128, 150, 300])
def test_more_samples_per_piece_than_one_tail_pass(pkg, orc, product_lib, K):
    """integral_intervs beyond what one pass of the tail's 128 threads covers: K + 1 <= 128 is a condition of the fused single launch
    (its tail makes ONE pass); from K = 128 on the step takes the two-launch path, whose tail walks a piece's samples in passes
    of 128 - and the result is still the oracle's (also with the dynamics penalties, which the tail owns)."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=4)
    T, cm = traj(pkg, occ, res, seed=21)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=K, safety_hor=0.5)
    shape = synth.make_shape("Box", params=(1.2, 0.4, 0.3), grad_mode=capi.GRAD_CENTRAL)
    eng, o = make_pair(pkg, orc, cfg, shape, occ=occ, esdf=esdf, res=res)
    cost, st0 = _check(eng, o, T, cm, f"K={K}")
    assert cost > 0
    st = eng.stats()
    assert st["units"] == st0[0] and st["grad_pairs"] == st0[3], (st, st0)


def test_v3_equals_v2_at_zero_threshold(pkg, orc, product_lib):
    """V3 with occ_thresh = 0 and culling off is V2 up to summation order."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=5)
    T, cm = traj(pkg, occ, res, seed=12)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    out = []
    for v in (capi.V3_ESDF_TILE, capi.V2_OCC_TILE):
        cfg = synth.default_config(v, kernel_size=9, integral_intervs=16, safety_hor=0.5)
        eng = pkg.Engine(cfg)
        eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
        eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
        eng.set_shape(shape)
        out.append(eng.eval_single(T, cm))
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_cull_preserves_result(pkg, orc, product_lib):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=7, occupancy=0.04)
    T, cm = traj(pkg, occ, res, seed=13)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6), bound_radius=1.9)
    res_ = {}
    for cull in (0, 1):
        cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5, enable_cull=cull)
        eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
        _check(eng, o, T, cm, f"cull={cull}")
        res_[cull] = (eng.eval_single(T, cm), eng.stats())
    assert res_[1][1]["culled"] > 0, "scenario must cull something"
    assert res_[1][1]["pairs"] <= res_[0][1]["pairs"]
    assert_close(res_[1][0][2], res_[0][0][2], "cull on vs off gradC", tol=1e-12)
    assert abs(res_[1][0][0] - res_[0][0][0]) <= 1e-12 * max(1.0, abs(res_[0][0][0]))


def test_dynamics_only_sweep(pkg, orc, product_lib):
    """enable_pos = 0 is exactly the reference's live addTimeIntPenaltyParallel (velocity / body-rate / tilt)."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res, piece_T=0.35)     # fast => velocity / omega limits violated
    cfg = synth.default_config(capi.V3_ESDF_TILE, integral_intervs=16, enable_pos=0, vmax=2.0, omgmax=1.0, thetamax=0.3)
    eng, o = make_pair(pkg, orc, cfg, None)
    cost, _ = _check(eng, o, T, cm, "dyn-only")
    assert cost > 0


def test_accumulate_semantics_and_batch(pkg, orc, product_lib):
    """Outputs are += like the reference; a ragged batch equals per-trajectory calls."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=8, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    trajs = [traj(pkg, occ, res, N=n, seed=20 + i) for i, n in enumerate([4, 4, 7, 3])]
    Ts = [t for t, _ in trajs]
    Cs = [c for _, c in trajs]
    cost = np.full(4, 2.5)
    gTs = [np.full(t.size, -1.0) for t in Ts]
    gCs = [np.full(18 * t.size, 3.0) for t in Ts]
    eng.eval(Ts, Cs, accumulate_into=(cost, gTs, gCs))
    for b in range(4):
        c0, gT0, gC0, _ = o.eval(Ts[b], Cs[b])
        assert abs((cost[b] - 2.5) - c0) <= REL_TOL * max(abs(c0), 1.0)
        assert_close(gTs[b] + 1.0, gT0, f"batch gradT[{b}]")
        assert_close(gCs[b] - 3.0, gC0, f"batch gradC[{b}]")


def test_shards_sum_to_full(pkg, orc, product_lib):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res, N=7)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=8, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    full = eng.eval_single(T, cm)
    for world in (2, 3, 8):
        acc = [0.0, np.zeros(7), np.zeros(18 * 7)]
        for r in range(world):
            eng.set_shard(r, world)
            c, gT, gC = eng.eval_single(T, cm)
            acc[0] += c; acc[1] += gT; acc[2] += gC
        eng.set_shard(0, 1)
        assert abs(acc[0] - full[0]) <= 1e-12 * max(1.0, abs(full[0]))
        assert_close(acc[1], full[1], f"world={world} gradT", tol=1e-12)
        assert_close(acc[2], full[2], f"world={world} gradC", tol=1e-12)


def test_cost_slots_survive_changing_launch_geometry(pkg, orc, product_lib):
    """The trajectory cost is collected from per-piece slots that every step must leave empty again: one context,
    alternating piece counts (incl. > 64 pieces: more than one slot per collecting lane), batches and shards, repeated."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=8, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    cases = {n: traj(pkg, occ, res, N=n, seed=40 + n) for n in (3, 70, 9)}
    want = {n: o.eval(*cases[n])[0] for n in cases}
    for rep in range(3):
        for n in (70, 3, 9, 70):
            T, cm = cases[n]
            c = eng.eval_single(T, cm)[0]
            assert abs(c - want[n]) <= REL_TOL * max(abs(want[n]), 1e-9), (rep, n, c, want[n])
            parts = 0.0
            for r in range(3):
                eng.set_shard(r, 3)
                parts += eng.eval_single(T, cm)[0]
            eng.set_shard(0, 1)
            assert abs(parts - c) <= 1e-12 * max(1.0, abs(c)), (rep, n, parts, c)
        # 5 trajectories: still one fused launch; 40: sweep and tail as two launches (the slots change hands differently)
        for nb in (5, 40):
            Ts = [cases[9][0]] * nb
            Cs = [cases[9][1]] * nb
            cost = np.zeros(nb)
            eng.eval(Ts, Cs, accumulate_into=(cost, [np.zeros(9) for _ in range(nb)], [np.zeros(18 * 9) for _ in range(nb)]))
            assert np.all(cost == cost[0]) and abs(cost[0] - want[9]) <= REL_TOL * max(abs(want[9]), 1e-9)


def test_edge_cases(pkg, orc, product_lib):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    # single piece, K = 1
    T, cm = traj(pkg, occ, res, N=1, seed=31)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=1, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    _check(eng, o, T, cm, "N=1,K=1")
    # empty map: no collision term at all
    cfg = synth.default_config(capi.V2_OCC_TILE, kernel_size=9, integral_intervs=8, safety_hor=0.5, enable_dyn=0)
    eng, o = make_pair(pkg, orc, cfg, shape, occ=np.zeros_like(occ), res=res)
    T, cm = traj(pkg, occ, res, N=3, seed=32)
    c, gT, gC = eng.eval_single(T, cm)
    assert c == 0.0 and not gT.any() and not gC.any()
    # fully occupied map + trajectory hugging the map boundary (tiles clipped by projInMap), non-trivial bmax
    full = np.ones_like(occ)
    bmax = np.array(occ.shape) * res - 0.3
    cfg = synth.default_config(capi.V2_OCC_TILE, kernel_size=7, integral_intervs=8, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, occ=full, res=res, bmax=bmax)
    T, cm = traj(pkg, np.zeros_like(occ), res, N=3, seed=33, margin=0.6)
    _check(eng, o, T, cm, "boundary tiles")
    # kernel_size larger than the map
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=101, integral_intervs=4, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    T, cm = traj(pkg, occ, res, N=2, seed=34)
    _check(eng, o, T, cm, "kernel larger than map")


@pytest.mark.parametrize("name", ["Torus", "Cappedtorus", "CappedCone", "WireframeBox", "BendLinear", "TwistBox", "BendBox",
                                  "Table", "Trefoil", "SmoothDifference", "SmoothIntersection", "CSG", "Box", "Ball",
                                  "Torus_big", "BendLinear_big", "SmoothIntersection_big"])
def test_every_registered_shape(pkg, orc, product_lib, name):
    """Each analytic kind of the registry (sw_manager.hpp:74-123) + Box/Ball with reference constants, rotated body offset."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, shape=(40, 40, 40), occupancy=0.10, seed=9)
    T, cm = traj(pkg, occ, res, N=3, seed=41, margin=6.0)
    shape = synth.make_shape(name, poly_params=(0.2, -0.1, 0.15, 30, 10, 120))
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=17, integral_intervs=6, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    cost, st = _check(eng, o, T, cm, name)
    assert st[2] > 0


_ALL_KINDS = ["Torus", "Cappedtorus", "CappedCone", "RoundedCone", "WireframeBox", "BendLinear", "TwistBox", "BendBox", "Table", "Trefoil",
              "SmoothDifference", "SmoothIntersection", "CSG", "Box", "Ball"]


@pytest.mark.parametrize("name", _ALL_KINDS)
def test_every_shape_with_identity_body_offset(pkg, orc, product_lib, name):
    """poly_params all zero (every shipped config): the sweep kernel's identity-offset instantiation of every shape class."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, shape=(40, 40, 40), occupancy=0.10, seed=9)
    T, cm = traj(pkg, occ, res, N=3, seed=41, margin=6.0)
    shape = synth.make_shape(name)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=17, integral_intervs=6, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    cost, st = _check(eng, o, T, cm, name + " (identity offset)")
    assert st[2] > 0


@pytest.mark.parametrize("name", _ALL_KINDS)
def test_swept_sweep_every_shape_class(pkg, orc, product_lib, name):
    """The V1 kernel's instantiation of every shape class (with and without a body offset) against the oracle."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res, T, cm, pts = _swept_world(pkg, seed=5, N=5)
    pts = pts[:: max(1, len(pts) // 120)]
    for pp in ((0, 0, 0, 0, 0, 0), (0.2, -0.1, 0.15, 30, 10, 120)):
        shape = synth.make_shape(name, poly_params=pp)
        cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
        eng, o = make_pair(pkg, orc, cfg, shape, occ=occ, res=res, points=pts)
        tg = np.zeros(len(pts)); to = np.zeros(len(pts))
        c, gT, gC = eng.eval_single(T, cm, tstar=tg)
        c0, gT0, gC0, _ = o.eval(T, cm, tstar=to)
        assert abs(c - c0) <= REL_TOL * max(abs(c0), 1e-9), (name, pp)
        assert_close(gT, gT0, f"{name} V1 gradT"); assert_close(gC, gC0, f"{name} V1 gradC")
        assert np.max(np.abs(tg - to)) <= 2e-5


# ------------------------------------------------------------------------------------------------------------------
# V1: the reference's live swept-volume sweep (addSaftyPenaOnSweptVolumeParallel)
# ------------------------------------------------------------------------------------------------------------------
def _swept_world(pkg, seed=3, N=6, piece_T=1.5):
    synth = pkg.synth
    occ, esdf, res = small_world(pkg, seed=seed)
    ext = np.array(occ.shape) * res
    T, Cf = synth.random_trajectory(ext, N, seed=seed + 40, piece_T=piece_T, margin=4.0, occ=occ, res=res)
    cm = synth.colmajor(Cf)
    # waypoints = piece junctions (positions at the start of pieces 1..N-1): c0 coefficients
    way = Cf.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=4 * res * 1.5)
    return occ, esdf, res, T, cm, pts


@pytest.mark.parametrize("shape_name", ["RoundedCone", "Torus", "Box"])
def test_swept_sweep_matches_oracle(pkg, orc, product_lib, shape_name):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res, T, cm, pts = _swept_world(pkg)
    assert pts.shape[0] > 50
    params = {"RoundedCone": (0.8, 0.3, 1.6), "Torus": (1.2, 0.25), "Box": (1.2, 0.4, 0.3)}[shape_name]
    shape = synth.make_shape(shape_name, params=params)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, points=pts)
    ts_g = np.zeros(pts.shape[0]); ts_o = np.zeros(pts.shape[0])
    cost, gT, gC = eng.eval_single(T, cm, tstar=ts_g)
    c0, gT0, gC0, st0 = o.eval(T, cm, tstar=ts_o)
    assert c0 > 0 and st0[3] > 0, "scenario must have active points"
    assert abs(cost - c0) <= REL_TOL * abs(c0), (cost, c0)
    assert_close(gT, gT0, "V1 gradT")
    assert_close(gC, gC0, "V1 gradC")
    # t* is the argmin of a flat-bottomed function: compare where the oracle found an interval, to descent resolution
    assert np.max(np.abs(ts_g - ts_o)) <= 2e-5, np.max(np.abs(ts_g - ts_o))
    st = eng.stats()
    assert st["units"] == pts.shape[0] and st["grad_pairs"] == st0[3] and st["pairs"] == st0[2], (st, st0)
    # second call: lastTstar persists for points without an interval (back_end_optimizer.hpp:578)
    T2 = T * 1.1
    cost2, gT2, gC2 = eng.eval_single(T2, cm, tstar=ts_g)
    c2, gT20, gC20, _ = o.eval(T2, cm, tstar=ts_o)
    assert abs(cost2 - c2) <= REL_TOL * max(abs(c2), 1e-9)
    assert_close(gC2, gC20, "V1 gradC second call")
    assert_close(gT2, gT20, "V1 gradT second call")


def test_swept_host_paths_agree_bitwise(pkg, orc, product_lib, monkeypatch):
    """isdf_eval on a swept-volume ctx: host-direct (inputs through the PCIe BAR, a publish kernel behind the five launches, a flag
    in host-mapped memory) against the copy path (ISDF_NO_HOST_DIRECT=1, read at isdf_create) - cost, gradients, lastTstar and the
    statistics are the same bits; with and without a lastTstar array; accumulate semantics kept."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res, T, cm, pts = _swept_world(pkg)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    got = {}
    for name in ("direct", "copy"):
        if name == "copy":
            monkeypatch.setenv("ISDF_NO_HOST_DIRECT", "1")
        eng = pkg.Engine(cfg); eng.set_shape(shape); eng.set_points(pts)
        ts = np.full(pts.shape[0], 0.25)
        r1 = eng.eval_single(T, cm, tstar=ts)
        st = eng.stats()
        path = eng.host_path()
        r2 = eng.eval_single(T * 1.05, cm, tstar=ts)            # lastTstar carried over
        r3 = eng.eval_single(T, cm)                            # no lastTstar array
        # a FRESH lastTstar array after steps in which the device wrote its own: the values the caller passes are the ones used
        # (host-direct: they travel through a staging region only the CPU writes, never straight into the array the GPU wrote)
        ts2 = np.full(pts.shape[0], 0.6); ts2[::3] = 0.05
        r4 = eng.eval_single(T, cm, tstar=ts2)
        got[name] = (r1, r2, r3, ts.copy(), st, path, r4, ts2.copy())
    d, c = got["direct"], got["copy"]
    assert c[5] == capi.HOST_PATH_COPY and d[5] in (capi.HOST_PATH_DIRECT_BAR, capi.HOST_PATH_COPY)
    assert d[6][0] == c[6][0] and np.array_equal(d[6][1], c[6][1]) and np.array_equal(d[6][2], c[6][2]) and np.array_equal(d[7], c[7])
    untouched = d[7] == np.where(np.arange(pts.shape[0]) % 3 == 0, 0.05, 0.6)      # points without an interval keep the caller's value
    assert untouched.any() and not untouched.all()
    for k in range(3):
        assert d[k][0] == c[k][0] and np.array_equal(d[k][1], c[k][1]) and np.array_equal(d[k][2], c[k][2])
    assert np.array_equal(d[3], c[3]) and d[4] == c[4]
    assert d[0][0] > 0


def test_swept_shards_and_empty(pkg, orc, product_lib):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res, T, cm, pts = _swept_world(pkg, seed=5)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, points=pts)
    full = eng.eval_single(T, cm)
    acc = [0.0, np.zeros(T.size), np.zeros(18 * T.size)]
    for r in range(3):
        eng.set_shard(r, 3)
        c, gT, gC = eng.eval_single(T, cm)
        acc[0] += c; acc[1] += gT; acc[2] += gC
    eng.set_shard(0, 1)
    assert abs(acc[0] - full[0]) <= 1e-12 * max(1.0, abs(full[0]))
    assert_close(acc[1], full[1], "V1 shards gradT", tol=1e-12)
    assert_close(acc[2], full[2], "V1 shards gradC", tol=1e-12)
    # no obstacle points: nothing is added
    eng.set_points(np.zeros((0, 3)))
    c, gT, gC = eng.eval_single(T, cm)
    assert c == 0 and not gT.any() and not gC.any()
    # a trajectory longer than 300 s keeps the stale duration (sw_manager.hpp:287-296): oracle and GPU agree
    eng.set_points(pts); o.set_points(pts)
    eng.eval_single(T, cm); o.eval(T, cm)
    Tl = np.full(T.size, 60.0)
    c, gT, gC = eng.eval_single(Tl, cm)
    c0, gT0, gC0, _ = o.eval(Tl, cm)
    assert abs(c - c0) <= REL_TOL * max(abs(c0), 1e-9)
    assert_close(gC, gC0, "V1 stale-duration gradC")


@pytest.mark.parametrize("shape_name", ["Box", "RoundedCone"])
def test_crowded_tiles_fill_the_candidate_ring(pkg, orc, product_lib, shape_name):
    """A map that is 55 % occupied with a wide penalty band: samples gather far more than the 512 candidates their ring holds,
    so the scan has to stop, let the workgroup drain the ring and resume (both the filtered path and Box's direct path)."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=31, shape=(56, 56, 48), res=0.2, occupancy=0.55)
    T, cm = traj(pkg, occ, res, N=4, seed=5, piece_T=0.8)
    if shape_name == "Box":
        shape = synth.make_shape("Box", params=(1.3, 1.0, 0.8), grad_mode=capi.GRAD_CENTRAL, bbox=((0, 0, 0), (1.3, 1.0, 0.8)))
    else:
        shape = synth.make_shape("RoundedCone", params=(1.0, 0.6, 1.3), bbox=((0, 0, 0.45), (1.0, 1.0, 1.45)))
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=10, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res, threads=16)
    c, gT, gC = eng.eval_single(T, cm)
    c0, gT0, gC0, st0 = o.eval(T, cm)
    st = eng.stats()
    assert st0[3] / st0[0] > 520                     # more active voxels per sample (on average!) than a ring holds
    assert st["grad_pairs"] == st0[3]
    assert abs(c - c0) <= REL_TOL * abs(c0)
    assert_close(gT, gT0, "crowded gradT"); assert_close(gC, gC0, "crowded gradC")


@pytest.mark.parametrize("shape_name,N,K", [("RoundedCone", 70, 16), ("Box", 70, 16), ("Box", 40, 64), ("RoundedCone", 24, 99)])
def test_longest_first_order_is_scheduling_only(pkg, orc, product_lib, shape_name, N, K):
    """A fused single-launch step with more workgroups than CUs dispatches its samples longest-first from the third step on
    (tile_sweep.hip, plan_piece).  The order is a permutation built on the device from the previous steps' work; results must
    be bit for bit those of the plain order - same trajectory repeated, trajectory changed under a running plan, batch of two."""
    import ctypes as C, os
    if os.environ.get("ISDF_NO_LPT") == "1":
        pytest.skip("dispatch order switched off (ISDF_NO_LPT=1)")
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=K, safety_hor=0.5)
    if shape_name == "Box":
        shape = synth.make_shape("Box", params=(1.2, 0.4, 0.3), grad_mode=capi.GRAD_CENTRAL)
    else:
        shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    K1 = K + 1                       # samples per piece: one wavefront covers up to 128 of them, lanes l and l + 64
    ns, nb = N * K1, (N * K1 + 3) // 4
    Ta, ca = traj(pkg, occ, res, N=N, seed=5)
    Tb, cb = traj(pkg, occ, res, N=N, seed=6)
    fresh = {}
    for name, (T, cm) in (("a", (Ta, ca)), ("b", (Tb, cb))):
        e0, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
        fresh[name] = e0.eval_single(T, cm)               # first step of a context: plain order
        c0, gT0, gC0, _ = o.eval(T, cm)
        assert abs(fresh[name][0] - c0) <= REL_TOL * max(abs(c0), 1e-9)
        assert_close(fresh[name][2], gC0, "gradC vs oracle")
    eng = pkg.Engine(cfg)
    eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    lib = eng.lib
    lib.isdf_debug_plan_map.restype = C.c_int
    lib.isdf_debug_plan_map.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    for step in range(6):
        got = eng.eval_single(Ta, ca)
        assert got[0] == fresh["a"][0] and np.array_equal(got[1], fresh["a"][1]) and np.array_equal(got[2], fresh["a"][2]), step
    m = np.full(4096, -7, dtype=np.int32)
    k = lib.isdf_debug_plan_map(eng.h, m.ctypes.data_as(C.c_void_p), m.size)
    assert k >= 3, "the plan must be running (fused launch with more workgroups than CUs)"
    m = m[m != -7]                     # 4 x the sweep workgroups of the launch (it may run more of them than the samples need)
    assert m.size % 4 == 0 and m.size >= 4 * nb
    assert np.array_equal(np.sort(m[m >= 0]), np.arange(ns)) and np.count_nonzero(m == -1) == m.size - ns, "not a permutation"
    assert not np.array_equal(m[:ns], np.arange(ns)), "the order should differ from the plain one in a map with obstacles"
    for step in range(4):                                   # the plan keeps running across a changed trajectory
        got = eng.eval_single(Tb, cb)
        assert got[0] == fresh["b"][0] and np.array_equal(got[2], fresh["b"][2]), step
    # a different launch geometry resets it, and coming back starts it again
    T3, c3 = traj(pkg, occ, res, N=9, seed=8)
    eng.eval_single(T3, c3)
    for step in range(4):
        got = eng.eval_single(Ta, ca)
        assert got[0] == fresh["a"][0] and np.array_equal(got[2], fresh["a"][2]), step
    # two trajectories in one fused launch
    T2, c2 = traj(pkg, occ, res, N=N // 2, seed=9)
    e1 = pkg.Engine(cfg); e1.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e1.set_shape(shape)
    one = e1.eval_single(T2, c2)
    for step in range(5):
        cost, gTs, gCs = eng.eval([T2, T2], [c2, c2])
        assert cost[0] == one[0] and cost[1] == one[0] and np.array_equal(gCs[1], one[2]), step


@pytest.mark.parametrize("long_axis", [0, 1, 2])
@pytest.mark.parametrize("variant", ["V3", "V2"])
def test_rows_along_every_axis_and_clipped_tiles(pkg, orc, product_lib, long_axis, variant):
    """The tile rows run along the world axis the robot is longest on, read from the bit-grid packed along that axis
    (tile_sweep.hip).  Robots long in body x, y and z, a grid whose extents differ and are no multiples of 32 (41 x 70 x 36:
    2, 3 and 2 dwords per row), trajectories hugging the map boundary (tiles clipped on every side, non-trivial bmax), a dense
    map - against the oracle; pair counts must agree exactly (no voxel lost or visited twice)."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    occ = synth.random_box_map((41, 70, 36), res=res, occupancy=0.3, seed=17 + long_axis, edge=(0.5, 2.0))
    esdf = synth.esdf_from_occupancy(occ, res)
    half = [0.35, 0.35, 0.35]
    half[long_axis] = 1.7
    shape = synth.make_shape("Box", params=tuple(half), grad_mode=capi.GRAD_CENTRAL)
    v = capi.V3_ESDF_TILE if variant == "V3" else capi.V2_OCC_TILE
    cfg = synth.default_config(v, kernel_size=11, integral_intervs=12, safety_hor=0.5)
    bmax = np.array(occ.shape) * res - 0.2
    kw = dict(esdf=esdf) if variant == "V3" else dict(occ=occ)
    eng, o = make_pair(pkg, orc, cfg, shape, res=res, bmax=bmax, **kw)
    for seed, margin in ((3, 0.6), (4, 0.6), (5, 3.0)):
        T, cm = traj(pkg, np.zeros_like(occ), res, N=5, seed=seed + 10 * long_axis, margin=margin)
        cost, st0 = _check(eng, o, T, cm, f"axis {long_axis} {variant} seed {seed}")
        st = eng.stats()
        assert cost > 0 and st["units"] == st0[0] and st["grad_pairs"] == st0[3] and st["pairs"] == st0[2], (st, st0)


def _plan_map(eng, cap):
    import ctypes as C
    lib = eng.lib
    lib.isdf_debug_plan_map.restype = C.c_int
    lib.isdf_debug_plan_map.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    m = np.full(cap, -7, dtype=np.int32)
    k = lib.isdf_debug_plan_map(eng.h, m.ctypes.data_as(C.c_void_p), m.size)
    return k, m[m != -7]


@pytest.mark.parametrize("case", ["long trajectory", "body offset", "batch", "shard"])
def test_dispatch_order_of_two_launch_steps_is_scheduling_only(pkg, orc, product_lib, case):
    """Steps that are sweep launch + tail launch (too many workgroups to be resident, a shape that cannot be fused, a batch, a
    shard) are dispatched, from the fourth step on, in workgroups of like weight, heaviest first (tile_sweep.hip, plan_wave;
    a batch sorts every trajectory by itself).  Results must be bit for bit those of the plain order, and the order the
    device built must be a permutation of the launch's samples."""
    import os
    if os.environ.get("ISDF_NO_LPT") == "1":
        pytest.skip("dispatch order switched off (ISDF_NO_LPT=1)")
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    K1 = 17
    shape = synth.make_shape("Box", params=(1.2, 0.4, 0.3), grad_mode=capi.GRAD_CENTRAL)
    n_traj, N, shard = 1, 200, None
    if case == "body offset":
        N = 70
        shape = synth.make_shape("Box", params=(1.2, 0.4, 0.3), grad_mode=capi.GRAD_CENTRAL, poly_params=(0.1, -0.2, 0.05, 0.3, 0.2, -0.4))
    elif case == "batch":
        n_traj, N = 100, 40            # 68 000 samples: more than one sorting group holds -> every trajectory by itself
    elif case == "shard":
        N, shard = 600, (1, 3)         # 200 pieces per rank: sweep launch + tail launch
    trajs = [traj(pkg, occ, res, N=N, seed=60 + b) for b in range(min(n_traj, 3))]
    Ts = [trajs[b % len(trajs)][0] for b in range(n_traj)]
    Cs = [trajs[b % len(trajs)][1] for b in range(n_traj)]

    def engine():
        e = pkg.Engine(cfg); e.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e.set_shape(shape)
        if shard:
            e.set_shard(*shard)
        return e
    first = engine().eval(Ts, Cs)                     # first step of a context: plain order
    if not shard and n_traj == 1:
        o = orc.Oracle(cfg, threads=4); o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); o.set_shape(shape)
        c0, gT0, gC0, _ = o.eval(Ts[0], Cs[0])
        assert abs(first[0][0] - c0) <= REL_TOL * max(abs(c0), 1e-9)
        assert_close(first[2][0], gC0, "gradC vs oracle")
    eng = engine()
    for step in range(7):
        got = eng.eval(Ts, Cs)
        assert np.array_equal(got[0], first[0]), (case, step)
        for b in range(n_traj):
            assert np.array_equal(got[1][b], first[1][b]) and np.array_equal(got[2][b], first[2][b]), (case, step, b)
    k, m = _plan_map(eng, 4 * ((n_traj * N * K1 + 3) // 4) + 64)
    assert k >= 4, "the plan must be running"
    live = m[m >= 0]
    assert live.size > 0 and np.array_equal(np.sort(live), np.arange(live.size)) and np.count_nonzero(m == -1) == m.size - live.size, "not a permutation"
    if not shard:
        assert live.size == n_traj * N * K1
    if case == "batch":       # every trajectory's samples stay inside its own stretch of the order
        pos = np.empty(live.size, dtype=np.int64)
        nb = m.size // 4
        r = np.where(m >= 0)[0]
        pos[m[r]] = 4 * (r % nb) + r // nb               # rank = wave * nb + workgroup, position = 4 * workgroup + wave
        assert np.array_equal(pos // (N * K1), np.arange(live.size) // (N * K1))


# ---- V3 with a REAL ESDF threshold (Gridmap3D.cpp:361-414,494-533, GridMap3D.h:155-193) ----------------------------
def _thresh_shape(pkg, name):
    synth, capi = pkg.synth, pkg.capi
    if name == "Box":
        return synth.make_shape("Box", params=(1.2, 0.4, 0.3), grad_mode=capi.GRAD_CENTRAL, bound_radius=1.4,
                                bbox=((0, 0, 0), (1.2, 0.4, 0.3)))
    if name == "RoundedCone":
        return synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6), bound_radius=1.95, bbox=((0, 0, 0.55), (0.8, 0.8, 1.35)))
    V, F = synth.l_prism_mesh()
    return synth.make_mesh_shape(V, F)


@pytest.mark.parametrize("frac", [0.5, 1.0, 1.7, 2.3])
@pytest.mark.parametrize("shape_name", ["Box", "RoundedCone", "mesh"])
def test_v3_with_a_real_esdf_threshold(pkg, orc, product_lib, shape_name, frac):
    """occ_thresh > 0 is what distinguishes V3 from V2: the qualifier is `esdf(i,j,k) <= occ_thresh` on the ESDF grid, a shell of
    voxels around every obstacle.  Three checks per (robot, threshold): (a) device vs oracle at 1e-5 with the SAME set of
    active pairs, cull off and on; (b) the whole-tile cull changes nothing (it has to allow for the shell: cull threshold +
    occ_thresh); (c) V3(threshold) == V2 on the occupancy grid {esdf <= threshold} BIT FOR BIT - the V2 term is the reference's
    own grad_cost_p (pinned against its code, tests/test_oracle_ref.py), so this ties the thresholded qualifier to it."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=9, occupancy=0.05)
    T, cm = traj(pkg, occ, res, seed=13)
    thresh = frac * res
    shape = _thresh_shape(pkg, shape_name)
    got = {}
    for cull in (0, 1):
        cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5, enable_cull=cull)
        cfg.occ_thresh = thresh
        eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res, threads=8)
        if shape_name == "mesh" and o.mesh_backend() != 1:
            pytest.skip("oracle/_ref/libref_fwn.so not built")
        cost, st0 = _check(eng, o, T, cm, f"V3 thresh={frac} res/{shape_name}/cull={cull}")
        st = eng.stats()
        assert st["units"] == st0[0] and st["grad_pairs"] == st0[3] and st0[3] <= st["pairs"] <= st0[2], (st, st0)
        got[cull] = (eng.eval_single(T, cm), st, st0)
    assert got[0][0][0] > 0, "scenario must exercise the collision term"
    # the shell really is there: more active pairs than with the bare occupancy.  The ESDF of a voxel grid takes the values
    # res * sqrt(integer): a threshold below one voxel (0.5 res) qualifies nothing new and must reproduce threshold 0 exactly,
    # 1.0 res is the TIE case (the six face neighbours sit at exactly res), 1.7 / 2.3 res reach d^2 <= 2 / <= 5
    cfg0 = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    e0, o0 = make_pair(pkg, orc, cfg0, shape, esdf=esdf, res=res, threads=8)
    base_pairs = o0.eval(T, cm)[3][3]
    if frac < 1.0:
        assert got[0][2][3] == base_pairs
        c0_, gT0_, gC0_ = e0.eval_single(T, cm)
        assert c0_ == got[0][0][0] and np.array_equal(gC0_, got[0][0][2])
    else:
        assert got[0][2][3] > base_pairs, "threshold must qualify voxels the occupancy does not"
    # (b) cull on == cull off (the cull only skips poses whose sums are exact zeros)
    if shape.bound_radius > 0:
        assert got[1][1]["culled"] > 0, "scenario must cull something"
    assert got[1][0][0] == got[0][0][0] and np.array_equal(got[1][0][1], got[0][0][1]) and np.array_equal(got[1][0][2], got[0][0][2])
    # (c) the same voxel set handed over as an occupancy grid to the V2 term
    shell = (esdf.astype(np.float64) <= thresh).astype(np.uint8)
    cfg2 = synth.default_config(capi.V2_OCC_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    e2 = pkg.Engine(cfg2); e2.set_grid(shell, (0, 0, 0), res, capi.GRID_OCCUPANCY); e2.set_shape(shape)
    c2, gT2, gC2 = e2.eval_single(T, cm)
    assert c2 == got[0][0][0] and np.array_equal(gT2, got[0][0][1]) and np.array_equal(gC2, got[0][0][2])
