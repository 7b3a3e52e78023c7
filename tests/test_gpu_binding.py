"""The reference-side binding, COMPILED (tests/native/binding_stub.cpp): TrajOptimizer's two static sweeps with the
reference's exact signatures (back_end_optimizer.hpp:432-438, :557-562) over Eigen objects, bodies = isdf_eval.  Driven on
the GPU against the reference's own sweeps (oracle/_ref/libref_sweeps.so) fed the same Eigen::MatrixX3d contents."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import REL_TOL, assert_close, small_world, traj

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def build_stub(pkg, out_dir):
    out = os.path.join(str(out_dir), "libbinding_stub.so")
    libdir = os.path.dirname(pkg.capi.LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "oracle", "ref_build", "eigen_shim"),
                           os.path.join(ROOT, "tests", "native", "binding_stub.cpp"), "-o", out,
                           "-L", libdir, "-lisdf_accel", f"-Wl,-rpath,{libdir}"])
    lib = C.CDLL(out)
    lib.stub_call.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_int]
    lib.stub_time_eval.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int, dp]
    lib.stub_time_callback.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, dp, dp]
    return lib


@pytest.fixture(scope="module")
def stub(pkg, product_lib, tmp_path_factory):
    return build_stub(pkg, tmp_path_factory.mktemp("stub"))


def _p(a):
    return a.ctypes.data_as(dp)


def _call(stub, eng, which, T, cm, cost0, gT0, gC0_cm, tstar=None):
    """cm / gC0_cm: column-major 6N x 3 (the library's layout); the stub receives ROW-major copies and fills its Eigen
    matrices element by element, so the layout under test is Eigen's."""
    N = T.size
    to_rm = lambda v: np.ascontiguousarray(v.reshape(3, 6 * N).T)
    Crm = to_rm(cm); gCrm = to_rm(gC0_cm).copy()
    cost = np.array([cost0]); gT = gT0.copy()
    ts = None if tstar is None else tstar
    rc = stub.stub_call(eng.h, which, N, _p(np.ascontiguousarray(T)), _p(Crm), _p(cost), _p(gT), _p(gCrm),
                        None if ts is None else _p(ts), 0 if ts is None else ts.size)
    assert rc == 0, rc
    return float(cost[0]), gT, np.ascontiguousarray(gCrm.T).reshape(-1)


def test_stub_integral_sweep_vs_reference_code(pkg, orc, stub):
    """addTimeIntPenaltyParallel through the compiled binding == the reference's own function, accumulate semantics incl."""
    capi, synth = pkg.capi, pkg.synth
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("oracle/_ref/libref_sweeps.so not built")
    occ, esdf, res = small_world(pkg, seed=3)
    N = 6
    T, cm = traj(pkg, occ, res, N=N, seed=43, piece_T=0.6)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5, enable_pos=0, enable_dyn=1,
                               vmax=2.0, omgmax=1.0, thetamax=0.3)
    shape = synth.make_shape("RoundedCone")
    eng = pkg.Engine(cfg); eng.set_shape(shape)
    ref = orc.RefWorld(cfg, "RoundedCone", shape, res, threads=4)
    c_r, gT_r, gC_r = ref.sweep_integral(T, cm)
    assert c_r > 0
    rng = np.random.default_rng(1)
    cost0 = 3.25; gT0 = rng.normal(0, 1, N); gC0 = rng.normal(0, 1, 18 * N)
    c, gT, gC = _call(stub, eng, 0, T, cm, cost0, gT0, gC0)
    assert abs((c - cost0) - c_r) <= REL_TOL * abs(c_r)
    assert_close(gT - gT0, gT_r, "binding: addTimeIntPenaltyParallel gradT"); assert_close(gC - gC0, gC_r, "binding: addTimeIntPenaltyParallel gradC")


def test_stub_swept_sweep_vs_reference_code(pkg, orc, stub):
    """addSaftyPenaOnSweptVolumeParallel through the compiled binding == the reference's own function; lastTstar lives in the
    optimizer's std::vector<double> and comes back filled."""
    capi, synth = pkg.capi, pkg.synth
    if not os.path.exists(orc.REF_SWEEPS):
        pytest.skip("oracle/_ref/libref_sweeps.so not built")
    occ, esdf, res = small_world(pkg, seed=3)
    N = 6
    T, cm = traj(pkg, occ, res, N=N, seed=43)
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    shape = synth.make_shape("RoundedCone", params=(1.5, 0.6, 4.5), poly_params=(0.1, -0.05, 0.08, 20, -10, 40))
    eng = pkg.Engine(cfg); eng.set_shape(shape); eng.set_points(pts)
    ref = orc.RefWorld(cfg, "RoundedCone", shape, res, threads=4)
    ref.set_points(pts)
    c_r, gT_r, gC_r, ts_r = ref.sweep_swept(T, cm)
    assert c_r > 0
    rng = np.random.default_rng(2)
    cost0 = -1.5; gT0 = rng.normal(0, 1, N); gC0 = rng.normal(0, 1, 18 * N)
    ts = np.zeros(len(pts))
    c, gT, gC = _call(stub, eng, 1, T, cm, cost0, gT0, gC0, tstar=ts)
    assert abs((c - cost0) - c_r) <= REL_TOL * abs(c_r)
    assert np.abs(ts - ts_r).max() <= 1e-7
    assert_close(gT - gT0, gT_r, "binding: addSafty... gradT"); assert_close(gC - gC0, gC_r, "binding: addSafty... gradC")


def test_stub_every_boundary_path_gives_the_same_sums(pkg, orc, stub, monkeypatch):
    """The three ways a host-array step can cross PCIe (include/isdf_accel.h, isdf_host_path): the CPU writes the inputs through
    the PCIe BAR, the launch fetches them from host-mapped memory (ISDF_NO_BAR_WRITES=1: what a box without a large BAR gets), or
    the copy path (ISDF_NO_HOST_DIRECT=1).  All three through the compiled binding's addTimeIntPenaltyParallel with the collision
    term on, repeated (the BAR path overwrites device memory between back-to-back launches), against the oracle and against
    each other bit for bit - they run the same kernel arithmetic."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=5)
    N = 7
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    shape = synth.make_shape("Box", params=(1.2, 0.4, 0.3), bound_radius=1.4)
    o = orc.Oracle(cfg, threads=4)
    o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); o.set_shape(shape)
    trajs = [traj(pkg, occ, res, N=N, seed=60 + k) for k in range(4)]
    want = []
    for T, cm in trajs:
        c0, gT0, gC0, _ = o.eval(T, cm)
        want.append((c0, gT0, gC0))
    assert any(w[0] > 0 for w in want)
    results, paths = {}, {}
    for name, env in (("bar", {}), ("mapped", {"ISDF_NO_BAR_WRITES": "1"}), ("copy", {"ISDF_NO_HOST_DIRECT": "1"})):
        for k in ("ISDF_NO_BAR_WRITES", "ISDF_NO_HOST_DIRECT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = pkg.Engine(cfg)                      # the switches are read per ctx, in isdf_create
        eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
        got = []
        for rep in range(3):                       # back-to-back steps with CHANGING inputs
            for (T, cm), (c0, gT0, gC0) in zip(trajs, want):
                c, gT, gC = _call(stub, eng, 0, T, cm, 0.0, np.zeros(N), np.zeros(18 * N))
                assert abs(c - c0) <= REL_TOL * max(1.0, abs(c0)), (name, rep)
                assert_close(gT, gT0, f"{name}: gradT"); assert_close(gC, gC0, f"{name}: gradC")
                got.append((c, gT, gC))
        results[name] = got
        paths[name] = eng.host_path()
        eng.close()
    assert paths["copy"] == capi.HOST_PATH_COPY
    assert paths["mapped"] == capi.HOST_PATH_DIRECT_MAPPED
    assert paths["bar"] in (capi.HOST_PATH_DIRECT_BAR, capi.HOST_PATH_DIRECT_MAPPED)     # BAR only where the box has a large one
    print(f"\nboundary paths exercised: {paths}")
    for name in ("mapped", "copy"):
        for a, b in zip(results["bar"], results[name]):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), f"{name} differs from the BAR path"
