"""Shared scenario builders for the parity tests (inputs only; expected values always come from the oracle)."""
import numpy as np

REL_TOL = 1e-5   # BASELINE.json north_star: cost/grad within 1e-5 relative of the CPU reference arithmetic


def small_world(pkg, seed=3, shape=(48, 48, 32), res=0.5, occupancy=0.12):
    synth = pkg.synth
    occ = synth.random_box_map(shape, res=res, occupancy=occupancy, seed=seed, edge=(1.0, 3.0))
    esdf = synth.esdf_from_occupancy(occ, res)
    return occ, esdf, res


def traj(pkg, occ, res, N=6, seed=11, piece_T=1.5, margin=4.0, jitter=0.5):
    ext = np.array(occ.shape) * res
    T, Cf = pkg.synth.random_trajectory(ext, N, seed=seed, piece_T=piece_T, margin=margin, occ=occ, res=res, jitter=jitter)
    return T, pkg.synth.colmajor(Cf)


def make_pair(pkg, orc, cfg, shape, occ=None, esdf=None, res=0.5, origin=(0, 0, 0), bmax=None, points=None, threads=4):
    """(Engine, Oracle) with identical once-per-plan state."""
    capi = pkg.capi
    eng = pkg.Engine(cfg)
    o = orc.Oracle(cfg, threads=threads)
    for tgt in (eng, o):
        if esdf is not None:
            tgt.set_grid(esdf, origin, res, capi.GRID_ESDF, bmax=bmax)
        if occ is not None:
            tgt.set_grid(occ, origin, res, capi.GRID_OCCUPANCY, bmax=bmax)
        if shape is not None:
            tgt.set_shape(shape)
        if points is not None:
            tgt.set_points(points)
    return eng, o


def assert_close(got, want, what, tol=REL_TOL):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.all(np.isfinite(got)) or not np.all(np.isfinite(want)), f"{what}: non-finite output"
    scale = max(np.linalg.norm(want.ravel()), 1e-12)
    err = np.linalg.norm((got - want).ravel())
    assert err <= tol * max(scale, 1e-9) + 1e-12, f"{what}: |diff|={err:.3e} vs |ref|={scale:.3e} (rel {err / scale:.3e} > {tol})"
