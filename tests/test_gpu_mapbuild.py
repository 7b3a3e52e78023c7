"""Map products built on the device (isdf_set_pointcloud / isdf_generate_esdf / isdf_gather_points) vs the oracle's
restatements of rcvGlobalMapHandler, generateESDF3d and getPointsInAABBOutOfLastOne.  Integer / byte work: bit-exact."""
import numpy as np
import pytest

from common import REL_TOL, assert_close, small_world, traj

pytestmark = pytest.mark.gpu


def _cloud(seed, n=20000, ext=(12.0, 9.0, 6.0)):
    rng = np.random.default_rng(seed)
    # a few solid boxes sampled densely + sparse noise (exercises the count threshold)
    pts = [rng.uniform(0, 1, (n // 10, 3)) * np.array(ext)]
    for _ in range(6):
        c = rng.uniform(1, 1, 3) * rng.uniform(0.1, 0.9, 3) * np.array(ext)
        sz = rng.uniform(0.4, 1.5, 3)
        pts.append(c + (rng.uniform(-0.5, 0.5, (n // 6, 3)) * sz))
    P = np.concatenate(pts).astype(np.float32)
    return P


@pytest.mark.parametrize("thr,explicit", [(1, False), (3, False), (2, True)])
def test_pointcloud_to_occupancy(pkg, orc, product_lib, thr, explicit):
    capi, synth = pkg.capi, pkg.synth
    P = _cloud(5 + thr)
    res = 0.25
    eng = pkg.Engine(synth.default_config(capi.V2_OCC_TILE))
    kw = {}
    if explicit:    # a boundary that cuts the cloud: outside points pile up in voxel (0,0,0)
        kw = dict(bmin=np.array([1.0, 0.5, 0.25]), bmax=np.array([10.3, 8.1, 5.2]))
    dims = eng.set_pointcloud(P, res, thr, **kw)
    occ0, b0, b1 = orc.pointcloud_to_occupancy(P, res, thr, **kw)
    occ, o, bm = eng.get_grid(capi.GRID_OCCUPANCY)
    assert dims == occ0.shape == occ.shape
    assert np.array_equal(o, b0) and np.array_equal(bm, b1)
    assert np.array_equal(occ, occ0)
    assert 0 < occ.sum() < occ.size


@pytest.mark.parametrize("shape", [(48, 40, 33), (7, 70, 5), (1, 1, 40)])
def test_generate_esdf_bit_exact(pkg, orc, product_lib, shape):
    capi, synth = pkg.capi, pkg.synth
    res = 0.3
    occ = synth.random_box_map(shape, res=res, occupancy=0.08, seed=sum(shape), edge=(0.3, 1.5))
    if occ.sum() == 0:
        occ[shape[0] // 2, shape[1] // 2, shape[2] // 3] = 1
    eng = pkg.Engine(synth.default_config(capi.V3_ESDF_TILE))
    eng.set_grid(occ, (0.5, -1.0, 2.0), res, capi.GRID_OCCUPANCY)
    eng.generate_esdf()
    esdf, _, _ = eng.get_grid(capi.GRID_ESDF)
    want = orc.build_esdf(occ, res).astype(np.float32)           # the reference's double, stored as float32
    assert np.array_equal(esdf, want), np.abs(esdf.astype(np.float64) - want).max()


def test_generate_esdf_empty_map_and_use_in_sweep(pkg, orc, product_lib):
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    # no occupied voxel at all: res * sqrt(DBL_MAX) everywhere (+inf as float32)
    eng = pkg.Engine(synth.default_config(capi.V3_ESDF_TILE))
    eng.set_grid(np.zeros((6, 5, 4), dtype=np.uint8), (0, 0, 0), res, capi.GRID_OCCUPANCY)
    eng.generate_esdf()
    e, _, _ = eng.get_grid(capi.GRID_ESDF)
    with np.errstate(over="ignore"):
        want = orc.build_esdf(np.zeros((6, 5, 4), dtype=np.uint8), res).astype(np.float32)
    assert np.all(np.isinf(e)) and np.array_equal(e, want)
    # a device-built ESDF drives the V3 sweep exactly like an uploaded one
    occ, esdf, res = small_world(pkg)
    T, cm = traj(pkg, occ, res)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    a = pkg.Engine(cfg); a.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); a.set_shape(shape)
    b = pkg.Engine(cfg); b.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); b.generate_esdf(); b.set_shape(shape)
    ra, rb = a.eval_single(T, cm), b.eval_single(T, cm)
    assert ra[0] == rb[0] and np.array_equal(ra[2], rb[2])


@pytest.mark.parametrize("offset", [None, (0.4, -0.3, 0.2)])
def test_gather_points(pkg, orc, product_lib, offset):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=12)
    T, cm = traj(pkg, occ, res, N=7, seed=3)
    way = cm.reshape(3, -1).T.reshape(7, 6, 3)[1:, 0, :]
    half = np.array([1.5, 1.2, 1.0])
    eng = pkg.Engine(synth.default_config(capi.V1_SWEPT))
    eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    M = eng.gather_points(way, half, offset)
    pts = eng.get_points()
    bmax = np.array(occ.shape) * res
    want = orc.gather_points(occ, np.zeros(3), bmax, res, way, half, offset if offset is not None else (0, 0, 0))
    assert M == want.shape[0] and M > 0
    assert np.array_equal(pts, want)
    # ... and they are the V1 sweep's obstacle points
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    eng.set_shape(shape)
    c_dev = eng.eval_single(T, cm)
    eng2 = pkg.Engine(synth.default_config(capi.V1_SWEPT))
    eng2.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng2.set_shape(shape); eng2.set_points(want)
    c_host = eng2.eval_single(T, cm)
    assert c_dev[0] == c_host[0] and np.array_equal(c_dev[2], c_host[2])
    # no waypoints -> empty set
    assert eng.gather_points(np.zeros((0, 3)), half) == 0 and eng.get_points().shape == (0, 3)
