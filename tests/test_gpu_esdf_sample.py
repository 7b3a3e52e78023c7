"""isdf_esdf_sample on the device: GridMap3D::getSDFValue / getSDFValueWithGrad
(src/map_manager/include/map_manager/GridMap3D.h:114-146, :155-193) at world points - bit for bit the oracle's restatement
(oracle/grid.hpp esdf_trilinear, same float32 grid), and against the reference's own code (oracle/_ref/libref_sweeps.so, whose
grid_esdf holds doubles: the difference is the float rounding of the grid values)."""
import os

import numpy as np
import pytest

from common import small_world

pytestmark = pytest.mark.gpu


def _points(rng, ext, res, n):
    """inside, near and beyond every face of the map, on voxel centres and voxel corners"""
    p = rng.uniform(-1.5 * res, 1.0, (n, 3)) * np.array([1, 1, 1]) + rng.uniform(0, 1, (n, 3)) * (ext + 1.5 * res)
    k = n // 8
    idx = rng.integers(0, (ext / res).astype(int), (k, 3))
    p[:k] = (idx + 0.5) * res                                   # voxel centres
    p[k:2 * k] = rng.integers(0, (ext / res).astype(int) + 1, (k, 3)) * res          # voxel corners / the map's faces
    p[2 * k:3 * k] = rng.uniform(-3, 3, (k, 3)) + np.where(rng.uniform(size=(k, 3)) < 0.5, 0.0, ext)   # around the faces, both sides
    return p


def test_esdf_sample_matches_oracle_bit_for_bit(pkg, orc, product_lib):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=5)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    eng = pkg.Engine(cfg); eng.set_grid(esdf, (0.3, -1.0, 2.0), res, capi.GRID_ESDF)
    o = orc.Oracle(cfg); o.set_grid(esdf, (0.3, -1.0, 2.0), res, capi.GRID_ESDF)
    rng = np.random.default_rng(0)
    ext = np.array(occ.shape) * res
    pts = _points(rng, ext, res, 4000) + np.array([0.3, -1.0, 2.0])
    val, grad = eng.esdf_sample(pts)
    ref = np.array([np.concatenate([[d], g]) for d, g in (o.esdf_trilinear(p) for p in pts)])
    assert np.array_equal(val, ref[:, 0])
    assert np.array_equal(grad, ref[:, 1:])
    # value only / gradient only
    v2, _ = eng.esdf_sample(pts, want_grad=False)
    assert np.array_equal(v2, val)
    assert np.abs(val).max() > 1.0 and np.abs(grad).max() > 0.1
    # empty input, no grid
    v0, g0 = eng.esdf_sample(np.zeros((0, 3)))
    assert v0.size == 0
    e2 = pkg.Engine(cfg); e2.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    with pytest.raises(RuntimeError):
        e2.esdf_sample(pts[:4])


def test_esdf_sample_vs_reference_code(pkg, orc, product_lib):
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("libref_sweeps.so not built (needs /root/reference at build time)")
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=9)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    shape = synth.make_shape("RoundedCone")
    ref = orc.RefWorld(cfg, "RoundedCone", shape, res, threads=2)
    ref.set_occupancy(occ, (0, 0, 0), res)
    e_r = ref.generate_esdf()                                   # the reference's doubles (GridMap3D::generateESDF3d)
    eng = pkg.Engine(cfg); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.generate_esdf()
    rng = np.random.default_rng(1)
    pts = _points(rng, np.array(occ.shape) * res, res, 1500)
    val, grad = eng.esdf_sample(pts)
    r = np.array([np.concatenate([[d], g]) for d, g in (ref.esdf_sample(p) for p in pts)])
    scale = e_r.max()
    # float32 storage of the grid: 6e-8 relative per corner value, 1 / res in the gradient
    assert np.abs(val - r[:, 0]).max() <= 2e-7 * scale
    assert np.abs(grad - r[:, 1:]).max() <= 2e-7 * scale / res * 4


def test_esdf_sample_full_size_properties(pkg, product_lib):
    """256^3: the sampler returns the grid's own value at voxel centres, and its gradient is the finite difference of its value
    inside a cell (the interpolant is trilinear: the central difference is exact up to rounding)."""
    import torch
    capi, synth = pkg.capi, pkg.synth
    res = 0.2
    occ = synth.random_box_map((256,) * 3, res=res, occupancy=0.15, seed=12345)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=0.17)
    eng = pkg.Engine(cfg); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.generate_esdf()
    e, _, _ = eng.get_grid(capi.GRID_ESDF)
    rng = np.random.default_rng(2)
    idx = rng.integers(1, 255, (200000, 3))
    val, _ = eng.esdf_sample((idx + 0.5) * res, want_grad=False)
    assert np.abs(val - e[idx[:, 0], idx[:, 1], idx[:, 2]]).max() <= 1e-12 * e.max()
    # gradient vs central difference, well inside a cell of the DUAL grid (between voxel centres)
    base = rng.integers(1, 254, (50000, 3)) + 0.5
    p = (base + rng.uniform(0.2, 0.8, base.shape)) * res
    _, g = eng.esdf_sample(p)
    h = 0.02 * res
    for a in range(3):
        d = np.zeros(3); d[a] = h
        vp, _ = eng.esdf_sample(p + d, want_grad=False); vm, _ = eng.esdf_sample(p - d, want_grad=False)
        assert np.abs((vp - vm) / (2 * h) - g[:, a]).max() <= 1e-9 * e.max() / res
    # device-resident form on 2e6 points equals the host form
    n = 2_000_000
    q = rng.uniform(-1, 52, (n, 3))
    dq = torch.from_numpy(q).cuda(); dv = torch.zeros(n, dtype=torch.float64, device="cuda"); dg = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    eng.esdf_sample_device(dq.data_ptr(), n, dv.data_ptr(), dg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    vh, gh = eng.esdf_sample(q[:5000])
    assert np.array_equal(dv[:5000].cpu().numpy(), vh) and np.array_equal(dg[:5000].cpu().numpy(), gh)


@pytest.mark.parametrize("dims", [(48, 48, 32), (37, 21, 15), (2, 3, 1)])
def test_esdf_sample_scattered_equals_plain_bitwise(pkg, orc, product_lib, dims):
    """The bricked copy (2 x 2 x 2-cell bricks with apron, one 128-byte line each: isdf_esdf_sample_scattered) gives the z-fastest
    path's bits - even and odd grid sizes, points beyond every face, a grid smaller than a brick, and again after the ESDF
    changed (the copy is rebuilt)."""
    capi, synth = pkg.capi, pkg.synth
    res = 0.5
    rng = np.random.default_rng(sum(dims))
    occ = (rng.uniform(size=dims) < 0.1).astype(np.uint8)
    occ[0, 0, 0] = 1
    esdf = synth.esdf_from_occupancy(occ, res)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    org = np.array([0.3, -1.0, 2.0])
    eng = pkg.Engine(cfg); eng.set_grid(esdf, org, res, capi.GRID_ESDF)
    ext = np.array(dims) * res
    pts = _points(rng, ext, res, 6000) + org
    v0, g0 = eng.esdf_sample(pts)
    v1, g1 = eng.esdf_sample(pts, scattered=True)
    assert np.array_equal(v0, v1) and np.array_equal(g0, g1)
    o = orc.Oracle(cfg); o.set_grid(esdf, org, res, capi.GRID_ESDF)
    ref = np.array([np.concatenate([[d], g]) for d, g in (o.esdf_trilinear(p) for p in pts[:500])])
    assert np.array_equal(v1[:500], ref[:, 0]) and np.array_equal(g1[:500], ref[:, 1:])
    # another ESDF on the same ctx: the copy follows
    esdf2 = (esdf * 0.5 + 0.25).astype(np.float32)
    eng.set_grid(esdf2, org, res, capi.GRID_ESDF)
    v2, g2 = eng.esdf_sample(pts, scattered=True)
    v3, g3 = eng.esdf_sample(pts)
    assert np.array_equal(v2, v3) and np.array_equal(g2, g3) and not np.array_equal(v2, v1)
