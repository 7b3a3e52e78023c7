"""BASELINE.json's configurations at FULL size on the GPU, through size-independent properties (and the oracle where it
finishes in seconds): C2 single trajectory, C3 batch of 128, C4 400 pieces in 8 shards, C5 swept-volume sweep with a
concave mesh robot on a 512^3 map.  Plus the mesh kind against the oracle at test size."""
import numpy as np
import pytest

from benchlib.meshes import reference_mesh

from common import REL_TOL, assert_close, assert_mesh_v1_free_running, make_pair, small_world, traj

pytestmark = pytest.mark.gpu

RES = 0.2


@pytest.fixture(scope="module")
def world256(pkg):
    synth = pkg.synth
    occ = synth.random_box_map((256,) * 3, res=RES, occupancy=0.15, seed=12345)
    esdf = synth.esdf_from_occupancy(occ, RES)
    return occ, esdf


def _traj(pkg, occ, n_pieces, seed=777):
    ext = np.array(occ.shape) * RES
    T, Cf = pkg.synth.random_trajectory(ext, n_pieces, seed=seed, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=RES)
    return T, pkg.synth.colmajor(Cf)


def _cfg(pkg, **kw):
    capi = pkg.capi
    return pkg.synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * RES,
                                    weight_p=4000.0, smoothing_eps=0.01, **kw)


def test_c2_full_size_vs_oracle(pkg, orc, product_lib, world256):
    """configs[1]: 40 pieces x 65 samples, 256^3 ESDF, k = 21, both bench robots."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf = world256
    T, cm = _traj(pkg, occ, 40)
    for shape in (synth.bench_box_shape(), synth.bench_rounded_cone_shape()):
        eng, o = make_pair(pkg, orc, _cfg(pkg), shape, esdf=esdf, res=RES, threads=30)
        c, gT, gC = eng.eval_single(T, cm)
        c0, gT0, gC0, st0 = o.eval(T, cm)
        assert c0 > 0 and abs(c - c0) <= REL_TOL * abs(c0)
        assert_close(gT, gT0, "C2 gradT"); assert_close(gC, gC0, "C2 gradC")
        st = eng.stats()
        assert st["units"] == 2600 == st0[0] and st["grad_pairs"] == st0[3]


def test_v3_on_a_512_cubed_map(pkg, orc, product_lib):
    """The tile sweep on the 512^3 map of configs[4] (0.1 m voxels: 8x the voxels of C2, three 16 MiB qualifier bit-grids, the
    ESDF built on the device): one 40-piece trajectory vs the oracle, then a batch of 16 == each one alone, bitwise - the bench's
    two 512^3 entries carry no other parity evidence."""
    capi, synth = pkg.capi, pkg.synth
    res5 = 0.1
    occ5 = synth.random_box_map((512,) * 3, res=res5, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
    cfg5 = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res5,
                                weight_p=4000.0, smoothing_eps=0.01, enable_dyn=1, enable_pos=1)
    shape5 = synth.make_shape("Box", params=(0.8, 0.15, 0.15), grad_mode=capi.GRAD_CENTRAL, bound_radius=0.83, bbox=((0, 0, 0), (0.8, 0.15, 0.15)))
    eng = pkg.Engine(cfg5)
    eng.set_grid(occ5, (0, 0, 0), res5, capi.GRID_OCCUPANCY)
    eng.generate_esdf()
    eng.set_shape(shape5)
    esdf5, _, _ = eng.get_grid(capi.GRID_ESDF)
    ext5 = np.array(occ5.shape) * res5
    Ts, Cs = [], []
    for b in range(16):
        Tb, Cf = synth.random_trajectory(ext5, 40, seed=900 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ5, res=res5)
        Ts.append(Tb); Cs.append(synth.colmajor(Cf))
    o = orc.Oracle(cfg5, threads=30)
    o.set_grid(esdf5, (0, 0, 0), res5, capi.GRID_ESDF); o.set_shape(shape5)
    c, gT, gC = eng.eval_single(Ts[0], Cs[0])
    c0, gT0, gC0, st0 = o.eval(Ts[0], Cs[0])
    assert c0 > 0 and st0[3] > 0 and abs(c - c0) <= REL_TOL * abs(c0)
    assert_close(gT, gT0, "512^3 V3 gradT"); assert_close(gC, gC0, "512^3 V3 gradC")
    assert eng.stats()["grad_pairs"] == st0[3]
    cost, gTs, gCs = eng.eval(Ts, Cs)
    for b in (0, 7, 15):
        c1, gT1, gC1 = eng.eval_single(Ts[b], Cs[b])
        assert c1 == cost[b] and np.array_equal(gT1, gTs[b]) and np.array_equal(gC1, gCs[b])


def test_c3_batch_is_independent(pkg, orc, product_lib, world256):
    """configs[2]: 128 trajectories evaluated concurrently == each one evaluated alone, bitwise; spot checks vs the oracle."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf = world256
    shape = synth.bench_box_shape()
    eng, o = make_pair(pkg, orc, _cfg(pkg), shape, esdf=esdf, res=RES, threads=30)
    Ts, Cs = zip(*[_traj(pkg, occ, 40, seed=777 + b) for b in range(128)])
    cost, gTs, gCs = eng.eval(list(Ts), list(Cs))
    assert eng.stats()["units"] == 128 * 2600
    for b in (0, 17, 127):
        c1, gT1, gC1 = eng.eval_single(Ts[b], Cs[b])
        assert c1 == cost[b] and np.array_equal(gT1, gTs[b]) and np.array_equal(gC1, gCs[b])
        c0, gT0, gC0, _ = o.eval(Ts[b], Cs[b])
        assert abs(cost[b] - c0) <= REL_TOL * abs(c0)
        assert_close(gCs[b], gC0, f"C3 traj {b} gradC")
    assert len(set(np.round(cost, 6))) > 100          # genuinely different trajectories


def test_c4_eight_shards_sum_to_full(pkg, orc, product_lib, world256):
    """configs[3]: one 400-piece trajectory, pieces sharded over 8 ranks; the partial outputs sum to the unsharded result
    (what the per-step all-reduce computes) and match the oracle."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf = world256
    T, cm = _traj(pkg, occ, 400)
    shape = synth.bench_box_shape()
    eng, o = make_pair(pkg, orc, _cfg(pkg), shape, esdf=esdf, res=RES, threads=30)
    full = eng.eval_single(T, cm)
    acc = [0.0, np.zeros(400), np.zeros(18 * 400)]
    units = 0
    for r in range(8):
        eng.set_shard(r, 8)
        c, gT, gC = eng.eval_single(T, cm)
        units += eng.stats()["units"]
        acc[0] += c; acc[1] += gT; acc[2] += gC
        lo, hi = pkg.parallel.shard_range(400, r, 8)
        mask = np.ones(400, bool); mask[lo:hi] = False
        assert np.all(gT[mask] == 0.0)                 # zeros outside the rank's pieces
    assert units == 400 * 65
    assert abs(acc[0] - full[0]) <= 1e-12 * abs(full[0])
    assert_close(acc[1], full[1], "C4 gradT", tol=1e-12); assert_close(acc[2], full[2], "C4 gradC", tol=1e-12)
    c0, gT0, gC0, _ = o.eval(T, cm)
    assert abs(full[0] - c0) <= REL_TOL * abs(c0)
    assert_close(full[2], gC0, "C4 gradC vs oracle")


def test_mesh_kind_matches_oracle(pkg, orc, product_lib):
    """MESH robot (libigl's fast winding number: the device walks the HDK hierarchy, the oracle calls the vendored
    library; exact closest point on both sides): V3 tile sweep and V1 sweep."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=14)
    T, cm = traj(pkg, occ, res, N=5, seed=23)
    V, F = synth.l_prism_mesh()
    assert F.shape == (20, 3)
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=12, safety_hor=0.5)
    eng, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res)
    c, gT, gC = eng.eval_single(T, cm)
    c0, gT0, gC0, st0 = o.eval(T, cm)
    assert c0 > 0 and st0[3] > 0
    assert abs(c - c0) <= REL_TOL * abs(c0)
    assert_close(gT, gT0, "mesh V3 gradT"); assert_close(gC, gC0, "mesh V3 gradC")
    way = cm.reshape(3, -1).T.reshape(5, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    eng1, o1 = make_pair(pkg, orc, cfg1, shape, occ=occ, res=res, points=pts)
    tg = np.zeros(len(pts)); to = np.zeros(len(pts))
    c, gT, gC = eng1.eval_single(T, cm, tstar=tg)
    c0, gT0, gC0, st0 = o1.eval(T, cm, tstar=to)
    assert c0 > 0
    assert abs(c - c0) <= REL_TOL * abs(c0)
    assert_mesh_v1_free_running("mesh-20 V1 (test_mesh_kind_matches_oracle)", tg, to, gT, gT0, gC, gC0)


def test_c5_swept_mesh_512_shards(pkg, orc, product_lib):
    """configs[4]: swept-volume sweep, concave mesh robot, 512^3 map (res 0.1 m): obstacle points gathered on the device,
    8 point-shards sum to the unsharded result; a sample of the points is checked against the oracle."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.1
    occ = synth.random_box_map((512,) * 3, res=res, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
    ext = np.array(occ.shape) * res
    T, Cf = synth.random_trajectory(ext, 20, seed=780, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    cm = synth.colmajor(Cf)
    V, F = synth.l_prism_mesh()
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
    eng = pkg.Engine(cfg)
    eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    eng.set_shape(shape)
    way = cm.reshape(3, -1).T.reshape(20, 6, 3)[1:, 0, :]
    M = eng.gather_points(way, 0.9)
    assert M > 2000
    pts = eng.get_points()
    full = eng.eval_single(T, cm)
    assert full[0] > 0
    acc = [0.0, np.zeros(20), np.zeros(18 * 20)]
    for r in range(8):
        eng.set_shard(r, 8)
        c, gT, gC = eng.eval_single(T, cm)
        acc[0] += c; acc[1] += gT; acc[2] += gC
    assert abs(acc[0] - full[0]) <= 1e-11 * abs(full[0])
    assert_close(acc[2], full[2], "C5 gradC shards", tol=1e-11)
    # the oracle on ALL points, both sides free-running: identical t*, cost and gradients at 1e-5; then once more with the argmin
    # taken out (the device evaluated AT the oracle's minimisers)
    eng.set_shard(0, 1)
    o = orc.Oracle(cfg, threads=30)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.set_points(pts)
    if o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    tg = -np.ones(M); to = -np.ones(M)
    c, gT, gC = eng.eval_single(T, cm, tstar=tg)
    c0, gT0, gC0, _ = o.eval(T, cm, tstar=to)
    assert c0 > 0 and abs(c - c0) <= REL_TOL * abs(c0)
    assert_mesh_v1_free_running("C5 512^3 mesh-20 V1", tg, to, gT, gT0, gC, gC0)
    c, gT, gC = eng.eval_swept_at_tstar_host(T, cm, to)
    assert abs(c - c0) <= REL_TOL * abs(c0)
    assert_close(gC, gC0, "C5 gradC at the oracle's t*"); assert_close(gT, gT0, "C5 gradT at the oracle's t*")


def test_c5_trefoil_512_every_point(pkg, orc, product_lib):
    """configs[4] with the reference's own 12 000-face Trefoil.obj, full size: EVERY obstacle point of the 512^3
    bench workload against the oracle, both sides searching their own minimisers (the oracle's closest point through its
    bounding-box tree, which tests/test_oracle_selfcheck.py holds to the exhaustive search bit for bit)."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.1
    occ = synth.random_box_map((512,) * 3, res=res, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
    ext = np.array(occ.shape) * res
    T, Cf = synth.random_trajectory(ext, 20, seed=780, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    cm = synth.colmajor(Cf)
    shape = synth.make_mesh_shape(*reference_mesh("Trefoil", 0.83))
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
    eng = pkg.Engine(cfg)
    eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    eng.set_shape(shape)
    way = cm.reshape(3, -1).T.reshape(20, 6, 3)[1:, 0, :]
    M = eng.gather_points(way, 0.9)
    assert M > 10000
    o = orc.Oracle(cfg, threads=30)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.set_points(eng.get_points())
    if o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    tg = -np.ones(M); to = -np.ones(M)
    c, gT, gC = eng.eval_single(T, cm, tstar=tg)
    c0, gT0, gC0, _ = o.eval(T, cm, tstar=to)
    assert c0 > 0 and abs(c - c0) <= REL_TOL * abs(c0)
    assert_mesh_v1_free_running("C5 512^3 Trefoil.obj V1, all %d points" % M, tg, to, gT, gT0, gC, gC0, allow_ties=2)


def test_c1_narrow_slit_mesh_robot(pkg, orc, product_lib):
    """configs[0] (demo1.launch): a narrow-slit map, a mesh robot (the concave L prism) and a ~20-piece trajectory through the
    slit on the reference's live path - the swept-volume sweep - device vs the oracle (reference winding-number library)."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.25
    occ = np.zeros((96, 64, 40), dtype=np.uint8)
    occ[46:50, :, :] = 1                     # a wall across the corridor ...
    occ[46:50, 28:36, 6:30] = 0              # ... with a 2 m x 6 m slit: the 1.6 x 1.0 x 0.5 m robot has to thread it
    occ[20:24, 10:30, :20] = 1; occ[70:74, 34:54, 10:] = 1      # two more obstacles off the straight line
    ext = np.array(occ.shape) * res
    N = 20
    p0 = np.array([3.0, 8.0, 4.5]); p1 = np.array([21.0, 8.0, 4.5])
    lam = np.linspace(0, 1, N + 1)[1:-1]
    rng = np.random.default_rng(7)
    way = p0[:, None] + (p1 - p0)[:, None] * lam[None, :] + rng.normal(0, 0.15, (3, N - 1))
    head = np.zeros((3, 3)); head[:, 0] = p0
    tail = np.zeros((3, 3)); tail[:, 0] = p1
    T = np.full(N, 0.8)
    cm = synth.colmajor(synth.minco_coeffs(head, tail, way, T))
    V, F = synth.l_prism_mesh()
    shape = synth.make_mesh_shape(V, F)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
    eng, o = make_pair(pkg, orc, cfg, shape, occ=occ, res=res, threads=30)
    if o.mesh_backend() != 1:
        pytest.skip("oracle/_ref/libref_fwn.so not built")
    M = eng.gather_points(way.T, 2.0)
    pts = eng.get_points()
    assert M > 300
    o.set_points(pts)
    tg = np.zeros(M); to = np.zeros(M)
    c, gT, gC = eng.eval_single(T, cm, tstar=tg)
    c0, gT0, gC0, _ = o.eval(T, cm, tstar=to)
    assert c0 > 0 and abs(c - c0) <= REL_TOL * abs(c0)
    assert_mesh_v1_free_running("C1 narrow slit mesh-20 V1", tg, to, gT, gT0, gC, gC0)
    # ... and with the argmin taken out (the device evaluated AT the oracle's minimisers): 1e-5
    to2 = -np.ones(M); c0, gT0, gC0, _ = o.eval(T, cm, tstar=to2)
    c, gT, gC = eng.eval_swept_at_tstar_host(T, cm, to2)
    assert abs(c - c0) <= REL_TOL * abs(c0)
    assert_close(gC, gC0, "C1 gradC at the oracle's t*"); assert_close(gT, gT0, "C1 gradT at the oracle's t*")
