"""ONE host process, several devices, through the C ABI (isdf_create_multi; SURVEY 8(b) "Threading"): the reference's caller is a
single ROS process, so the multi-GPU step has to be reachable from one C++ thread.  The test box has one GPU: the device is listed
two / three / five times - every shard then runs on its own ctx and stream of that device and the exchange code (peer copies of
the inputs, per-device launches from the calling thread, the rank-ordered sum / the staged sum / RCCL's all-reduce) is the
code a real node runs.  Driven through the COMPILED reference-side binding (tests/native/binding_stub.cpp) where the reference has a
signature for it."""
import os

import numpy as np
import pytest

from common import REL_TOL, assert_close, make_pair, small_world, traj
from test_gpu_binding import _call, build_stub

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stub(pkg, product_lib, tmp_path_factory):
    return build_stub(pkg, tmp_path_factory.mktemp("stub_multi"))


def _n_gpus():
    import torch
    return torch.cuda.device_count()


def _devs(n):
    """n device ordinals: DISTINCT GPUs wherever the node has them (on an 8-GPU node these tests run real peer copies / peer reads
    over xGMI), the test GPU listed repeatedly otherwise."""
    k = max(1, _n_gpus())
    return [i % k for i in range(n)]


def _mk(pkg, cfg, devices, shape, esdf=None, occ=None, res=0.5, points=None):
    capi = pkg.capi
    eng = pkg.Engine(cfg, devices=devices)
    if esdf is not None:
        eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    if occ is not None:
        eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    if shape is not None:
        eng.set_shape(shape)
    if points is not None:
        eng.set_points(points)
    return eng


@pytest.mark.parametrize("collective", ["peer", "staged"])
@pytest.mark.parametrize("n_dev", [2, 3, 5])
def test_multi_integral_sweep_through_the_binding(pkg, orc, stub, monkeypatch, n_dev, collective):
    """addTimeIntPenaltyParallel (collision term on) on a ctx over n 'devices' == the single-device ctx == the oracle."""
    capi, synth = pkg.capi, pkg.synth
    monkeypatch.setenv("ISDF_MULTI_COLLECTIVE", collective)
    occ, esdf, res = small_world(pkg, seed=5)
    N = 7
    T, cm = traj(pkg, occ, res, N=N, seed=61)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5, vmax=2.0, omgmax=1.0, thetamax=0.3)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    single, o = make_pair(pkg, orc, cfg, shape, esdf=esdf, res=res, threads=4)
    c1, gT1, gC1 = single.eval_single(T, cm)
    c0, gT0, gC0, st0 = o.eval(T, cm)
    multi = _mk(pkg, cfg, _devs(n_dev), shape, esdf=esdf, res=res)
    nd, mode = multi.multi_info()
    assert nd == n_dev and mode == (capi.MULTI_PEER_SUM if collective == "peer" else capi.MULTI_STAGED)
    rng = np.random.default_rng(4)
    for rep in range(3):                                   # repeated: the part buffers and events are reused
        cost0 = 1.5 * rep; gTi = rng.normal(0, 1, N); gCi = rng.normal(0, 1, 18 * N)
        c, gT, gC = _call(stub, multi, 0, T, cm, cost0, gTi, gCi)
        assert abs((c - cost0) - c1) <= 1e-12 * abs(c1)
        assert_close(gT - gTi, gT1, "multi vs single gradT", tol=1e-11); assert_close(gC - gCi, gC1, "multi vs single gradC", tol=1e-11)
        assert abs((c - cost0) - c0) <= REL_TOL * abs(c0)
        assert_close(gC - gCi, gC0, "multi vs oracle gradC")
    st = multi.stats()
    assert st["units"] == st0[0] and st["grad_pairs"] == st0[3] and st["overflow"] == 0
    # a different trajectory length on the same ctx (buffers grow), then back
    T2, cm2 = traj(pkg, occ, res, N=11, seed=62)
    a = multi.eval_single(T2, cm2); b = single.eval_single(T2, cm2)
    assert abs(a[0] - b[0]) <= 1e-12 * max(1.0, abs(b[0])); assert_close(a[2], b[2], "N=11 gradC", tol=1e-11)
    a = multi.eval_single(T, cm)
    assert abs(a[0] - c1) <= 1e-12 * abs(c1)


def test_multi_host_written_outputs_equal_the_downloaded_ones(pkg, orc, product_lib, monkeypatch):
    """isdf_eval on a multi-device ctx: the sum kernel writes outputs, statistics words and a completion word into the pinned
    buffer (default) - bitwise what the download commands + stream synchronisation deliver (ISDF_MULTI_NO_HOST_OUT=1)."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=5)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5, vmax=2.0, omgmax=1.0, thetamax=0.3)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    a = _mk(pkg, cfg, _devs(3), shape, esdf=esdf, res=res)
    monkeypatch.setenv("ISDF_MULTI_NO_HOST_OUT", "1")
    b = _mk(pkg, cfg, _devs(3), shape, esdf=esdf, res=res)
    monkeypatch.delenv("ISDF_MULTI_NO_HOST_OUT")
    for N, seed in ((7, 61), (12, 62), (7, 63)):
        T, cm = traj(pkg, occ, res, N=N, seed=seed)
        ra = a.eval_single(T, cm); rb = b.eval_single(T, cm)
        assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])
        assert a.stats() == b.stats()
    # ragged batch (group by group, statistics added up per group)
    Ts, Cs = zip(*[traj(pkg, occ, res, N=n, seed=70 + n) for n in (5, 5, 9)])
    ra = a.eval(list(Ts), list(Cs)); rb = b.eval(list(Ts), list(Cs))
    assert np.array_equal(ra[0], rb[0])
    for x, y in zip(ra[2], rb[2]): assert np.array_equal(x, y)
    assert a.stats() == b.stats()


@pytest.mark.parametrize("n_dev", [2, 4])
def test_multi_swept_sweep_through_the_binding(pkg, orc, stub, n_dev):
    """addSaftyPenaOnSweptVolumeParallel: obstacle points sharded over the devices, lastTstar merged back in point order."""
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=3)
    N = 6
    T, cm = traj(pkg, occ, res, N=N, seed=43)
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    shape = synth.make_shape("RoundedCone", params=(1.5, 0.6, 4.5), poly_params=(0.1, -0.05, 0.08, 20, -10, 40))
    single, o = make_pair(pkg, orc, cfg, shape, points=pts, threads=4)
    ts1 = np.zeros(len(pts)); c1, gT1, gC1 = single.eval_single(T, cm, tstar=ts1)
    multi = _mk(pkg, cfg, _devs(n_dev), shape, points=pts)
    ts = np.full(len(pts), -7.0)                           # every entry has to be overwritten or kept per the reference's rule
    ts_single = np.full(len(pts), -7.0); single.eval_single(T, cm, tstar=ts_single)
    c, gT, gC = _call(stub, multi, 1, T, cm, 0.25, np.zeros(N), np.zeros(18 * N), tstar=ts)
    assert np.array_equal(ts, ts_single)
    assert abs((c - 0.25) - c1) <= 1e-12 * abs(c1)
    assert_close(gT, gT1, "multi V1 gradT", tol=1e-11); assert_close(gC, gC1, "multi V1 gradC", tol=1e-11)
    st = multi.stats(); st1 = single.stats()
    assert st["units"] == st1["units"] == len(pts) and st["grad_pairs"] == st1["grad_pairs"] and st["pairs"] == st1["pairs"]
    c0, gT0, gC0, _ = o.eval(T, cm)
    assert abs((c - 0.25) - c0) <= REL_TOL * abs(c0); assert_close(gC, gC0, "multi V1 vs oracle gradC")


def test_multi_callback_and_optimizer(pkg, orc, product_lib):
    """isdf_cost_function[_lmbm] and isdf_optimize_lbfgs on a 3-'device' ctx == the single-device ctx (live configuration: V1 +
    dynamics; and the V3 integral sweep), and the device-resident entry point with the result left on devices[0]."""
    import torch
    capi, synth = pkg.capi, pkg.synth
    from test_gpu_callback import _setup
    occ, esdf, res = small_world(pkg, seed=10)
    N = 5
    head, tail, way, T0 = _setup(pkg, occ, res, N, 66)
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=4 * res * 1.5)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    for variant, points in ((capi.V3_ESDF_TILE, None), (capi.V1_SWEPT, pts)):
        cfg = synth.default_config(variant, kernel_size=9, integral_intervs=16, safety_hor=0.5)
        single = _mk(pkg, cfg, None, shape, esdf=esdf, occ=occ, res=res, points=points)
        multi = _mk(pkg, cfg, [0, 0, 0], shape, esdf=esdf, occ=occ, res=res, points=points)
        for e in (single, multi):
            e.set_trajectory(N, head, tail, 2.0)
        x = single.pack_variables(T0 * 0.7, way)
        c1, g1 = single.cost_function(x)
        c, g = multi.cost_function(x)
        assert abs(c - c1) <= 1e-12 * abs(c1); assert_close(g, g1, "multi callback g", tol=1e-10)
        x1, r1 = single.optimize_lbfgs(x, max_iterations=4, g_epsilon=0.0, past=0)
        xm, rm = multi.optimize_lbfgs(x, max_iterations=4, g_epsilon=0.0, past=0)
        assert rm["f"] < c1 and abs(rm["f"] - r1["f"]) <= 1e-6 * abs(r1["f"])
    # device-resident step (V3): inputs and output on devices[0]
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    single = _mk(pkg, cfg, None, shape, esdf=esdf, res=res); multi = _mk(pkg, cfg, [0, 0], shape, esdf=esdf, res=res)
    T, cm = traj(pkg, occ, res, N=N, seed=77)
    dev = torch.device("cuda", 0)
    dT = torch.from_numpy(T).to(dev); dC = torch.from_numpy(cm).to(dev)
    outs = []
    for e in (single, multi):
        dO = torch.zeros(e.out_stride(N), dtype=torch.float64, device=dev)
        for _ in range(3):
            e.eval_device(1, N, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(dO.cpu().numpy())
    assert_close(outs[1], outs[0], "multi eval_device", tol=1e-12)
    # what such a ctx refuses
    with pytest.raises(pkg.engine.IsdfError) as ei:
        multi.set_shard(0, 2)
    assert ei.value.code == capi.ISDF_ERR_STATE


def test_multi_rccl_collective_world_of_one(pkg, orc, product_lib, monkeypatch):
    """ISDF_MULTI_COLLECTIVE=rccl: librccl.so by dlopen, ncclCommInitAll, one grouped ncclAllReduce(sum, double) per sweep.  RCCL
    wants distinct devices, so on this one-GPU box the world has ONE rank - the call sequence is the real one."""
    capi, synth = pkg.capi, pkg.synth
    monkeypatch.setenv("ISDF_MULTI_COLLECTIVE", "rccl")
    occ, esdf, res = small_world(pkg, seed=5)
    T, cm = traj(pkg, occ, res, N=6, seed=63)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    try:
        multi = _mk(pkg, cfg, [0], shape, esdf=esdf, res=res)
    except pkg.engine.IsdfError as e:
        if e.code == capi.ISDF_ERR_UNSUPPORTED:
            pytest.skip(f"RCCL not usable here: {e}")
        raise
    assert multi.multi_info() == (1, capi.MULTI_NONE) or multi.multi_info()[0] == 1
    monkeypatch.delenv("ISDF_MULTI_COLLECTIVE")
    single = _mk(pkg, cfg, None, shape, esdf=esdf, res=res)
    for _ in range(3):
        a = multi.eval_single(T, cm)
    b = single.eval_single(T, cm)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    # duplicates are refused with a message, not a crash
    monkeypatch.setenv("ISDF_MULTI_COLLECTIVE", "rccl")
    with pytest.raises(pkg.engine.IsdfError):
        _mk(pkg, cfg, [0, 0], shape, esdf=esdf, res=res)


def test_multi_refuses_the_fixed_minimiser_entry_points(pkg, orc, product_lib):
    """isdf_eval_swept_at_tstar[_host] on a multi-device ctx would return the lead's shard only: ISDF_ERR_UNSUPPORTED, as the header says."""
    import torch
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=3)
    N = 6
    T, cm = traj(pkg, occ, res, N=N, seed=43)
    way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    shape = synth.make_shape("RoundedCone", params=(1.5, 0.6, 4.5))
    multi = _mk(pkg, cfg, _devs(2), shape, points=pts)
    with pytest.raises(pkg.engine.IsdfError) as ei:
        multi.eval_swept_at_tstar_host(T, cm, np.zeros(len(pts)))
    assert ei.value.code == capi.ISDF_ERR_UNSUPPORTED
    dT = torch.from_numpy(T).cuda(); dC = torch.from_numpy(cm).cuda(); dts = torch.zeros(len(pts), dtype=torch.float64, device="cuda")
    dO = torch.zeros(multi.out_stride(N), dtype=torch.float64, device="cuda")
    with pytest.raises(pkg.engine.IsdfError) as ei:
        multi.eval_swept_at_tstar(N, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), dts.data_ptr(), 0)
    assert ei.value.code == capi.ISDF_ERR_UNSUPPORTED
    # the sweep itself still works on that ctx
    c, gT, gC = multi.eval_single(T, cm)
    assert np.isfinite(c)


@pytest.mark.parametrize("collective", ["rccl", "staged", "peer"])
def test_multi_on_distinct_gpus(pkg, orc, product_lib, monkeypatch, collective):
    """Upgrades itself on a node with >= 2 GPUs (the driver's 8-GPU box): every listed device is a DIFFERENT GPU - real
    hipMemcpyPeerAsync of the inputs, the lead reading the peers' parts over xGMI / staged copies / ONE grouped ncclAllReduce with
    more than one rank - against the single-device ctx, bitwise in the rank-ordered modes."""
    n = _n_gpus()
    if n < 2:
        pytest.skip("one GPU on this box: the same code runs with the device listed repeatedly (tests above)")
    capi, synth = pkg.capi, pkg.synth
    monkeypatch.setenv("ISDF_MULTI_COLLECTIVE", collective)
    occ, esdf, res = small_world(pkg, seed=5)
    N = 16
    T, cm = traj(pkg, occ, res, N=N, seed=61)
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=16, safety_hor=0.5)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    try:
        multi = _mk(pkg, cfg, list(range(min(n, 8))), shape, esdf=esdf, res=res)
    except pkg.engine.IsdfError as e:
        if e.code == capi.ISDF_ERR_UNSUPPORTED:
            pytest.skip(f"{collective} not usable between these devices: {e}")
        raise
    monkeypatch.delenv("ISDF_MULTI_COLLECTIVE")
    single = _mk(pkg, cfg, None, shape, esdf=esdf, res=res)
    b = single.eval_single(T, cm)
    for _ in range(4):
        a = multi.eval_single(T, cm)
        assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
        assert_close(a[1], b[1], f"{collective} gradT", tol=1e-11); assert_close(a[2], b[2], f"{collective} gradC", tol=1e-11)
    assert multi.stats()["units"] == single.stats()["units"]
