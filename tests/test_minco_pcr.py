"""The log-depth MINCO of the device-resident callback (csrc/minco_pcr.hpp: junction-state form, parallel cyclic reduction)
against the reference's own minco.hpp compiled here (oracle/_ref/libref_minco.so) and against the oracle's restatement of
its banded LU.  CPU only: the header's inline arithmetic is compiled into a test shim with g++ and run one loop per parallel
round - the identical functions minco_dev.hip runs with one thread per row.  The elimination order differs from the band
LU's, so the bar is 1e-10 relative, not bit equality."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("pcr") / "libminco_pcr_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "implicit-sdf-planner_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "minco_pcr_shim.cpp"), "-o", out])
    return C.CDLL(out)


def _problem(N, seed, t_lo, t_hi):
    rng = np.random.default_rng(seed)
    head = np.zeros((3, 3)); tail = np.zeros((3, 3))
    head[:, 0] = rng.uniform(0, 5, 3); head[:, 1] = rng.normal(0, 1, 3); head[:, 2] = rng.normal(0, 0.5, 3)
    tail[:, 0] = rng.uniform(15, 20, 3); tail[:, 1] = rng.normal(0, 1, 3); tail[:, 2] = rng.normal(0, 0.3, 3)
    way = np.linspace(head[:, 0], tail[:, 0], N + 1)[1:-1] + rng.normal(0, 0.6, (N - 1, 3))      # (N-1) x 3
    T = np.exp(rng.uniform(np.log(t_lo), np.log(t_hi), N))
    return head, tail, way, T


def _pcr(shim, head, tail, way, T, gc=None, gt=None):
    N = T.size
    h = np.ascontiguousarray(head.T).reshape(-1); t = np.ascontiguousarray(tail.T).reshape(-1)
    W = np.ascontiguousarray(way).reshape(-1) if N > 1 else np.zeros(3)
    cm = np.zeros(18 * N); gC = np.zeros(18 * N); gT = np.zeros(N); e = C.c_double(0)
    shim.pcr_minco(C.c_int(N), _p(h), _p(t), _p(W), _p(T), _p(cm), C.byref(e), _p(gC), _p(gT))
    out = [cm, e.value, gC, gT]
    if gc is not None:
        gP = np.zeros(3 * max(N - 1, 1)); gTo = np.zeros(N)
        shim.pcr_minco_propagate(C.c_int(N), _p(h), _p(t), _p(W), _p(T), _p(gc), _p(gt), _p(gP), _p(gTo))
        out += [gP[:3 * (N - 1)].reshape(N - 1, 3).T, gTo]
    return out


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


# short and long durations, mixed by a factor of 100 inside one trajectory
CASES = [(1, 0.4, 2.5), (2, 0.4, 2.5), (3, 0.4, 2.5), (5, 0.4, 2.5), (17, 0.4, 2.5), (40, 0.4, 2.5), (40, 0.05, 0.3), (40, 3.0, 12.0),
         (40, 0.1, 10.0), (64, 0.2, 2.0), (65, 0.2, 2.0), (129, 0.1, 1.0), (400, 0.15, 0.3), (400, 0.05, 5.0), (512, 0.2, 1.0)]


@pytest.mark.parametrize("N,t_lo,t_hi", CASES)
def test_pcr_minco_vs_band_lu(shim, orc, N, t_lo, t_hi):
    head, tail, way, T = _problem(N, 900 + N, t_lo, t_hi)
    rng = np.random.default_rng(3 * N + 1)
    gc = rng.normal(0, 1, 18 * N); gt = rng.normal(0, 1, N)
    cm, e, gC, gT, gP, gTo = _pcr(shim, head, tail, way, T, gc, gt)
    wayT = way.T.reshape(3, N - 1) if N > 1 else np.zeros((3, 0))
    C6 = orc.minco_coeffs(head, tail, wayT, T)                              # the oracle's band LU (bitwise the host MINCO)
    cm0 = np.ascontiguousarray(C6.T).reshape(-1)
    e0, gC0, gT0 = orc.minco_energy(cm0, T)
    gP0, gTo0 = orc.minco_propagate(head, tail, wayT, T, gc, gt)
    # per coefficient order: the trajectory's derivatives differ in scale by powers of 1 / T
    c = cm.reshape(3, N, 6); c0 = cm0.reshape(3, N, 6)
    for r in range(6):
        assert _rel(c[:, :, r], c0[:, :, r]) <= 1e-10, (r, _rel(c[:, :, r], c0[:, :, r]))
    assert abs(e - e0) <= 1e-10 * abs(e0)
    assert _rel(gC, gC0) <= 1e-10 and _rel(gT, gT0) <= 1e-10
    assert _rel(gTo, gTo0) <= 1e-10, _rel(gTo, gTo0)
    if N > 1:
        assert _rel(gP, gP0) <= 1e-10, _rel(gP, gP0)


@pytest.mark.parametrize("N,t_lo,t_hi", [(2, 0.4, 2.5), (5, 0.4, 2.5), (40, 0.4, 2.5), (40, 0.05, 0.3), (40, 0.1, 10.0)])
def test_pcr_minco_vs_reference_headers(shim, orc, N, t_lo, t_hi):
    """... and straight against the reference's MINCO_S3NU (utils/minco.hpp compiled from /root/reference)."""
    if not os.path.exists(orc.REF_MINCO):
        pytest.skip("oracle/_ref/libref_minco.so not built")
    head, tail, way, T = _problem(N, 70 + N, t_lo, t_hi)
    rng = np.random.default_rng(N)
    gc = rng.normal(0, 1, 18 * N); gt = rng.normal(0, 1, N)
    cm, e, gC, gT, gP, gTo = _pcr(shim, head, tail, way, T, gc, gt)
    cm_r, e_r, gC_r, gT_r = orc.ref_minco(head, tail, way.T, T)
    gP_r, gTo_r = orc.ref_minco_propagate(head, tail, way.T, T, gc, gt)
    c = cm.reshape(3, N, 6); c0 = np.asarray(cm_r).reshape(3, N, 6)
    for r in range(6):
        assert _rel(c[:, :, r], c0[:, :, r]) <= 1e-10
    assert abs(e - e_r) <= 1e-10 * abs(e_r) and _rel(gC, gC_r) <= 1e-10 and _rel(gT, gT_r) <= 1e-10
    assert _rel(gP, gP_r) <= 1e-10 and _rel(gTo, gTo_r) <= 1e-10


def test_pcr_adjoint_is_the_gradient(shim):
    """First principles: (gradP, gradT) is the derivative of sum(gc * coeffs(waypoints, T)) + gt . T (central differences)."""
    N = 7
    head, tail, way, T = _problem(N, 11, 0.5, 2.0)
    rng = np.random.default_rng(2)
    gc = rng.normal(0, 1, 18 * N); gt = rng.normal(0, 1, N)
    _, _, _, _, gP, gTo = _pcr(shim, head, tail, way, T, gc, gt)

    def f(w, t):
        return float(np.dot(gc, _pcr(shim, head, tail, w, t)[0]) + np.dot(gt, t))
    eps = 1e-6
    for i in range(N):
        tp = T.copy(); tm = T.copy(); tp[i] += eps; tm[i] -= eps
        fd = (f(way, tp) - f(way, tm)) / (2 * eps)
        assert abs(fd - gTo[i]) <= 1e-6 * max(1.0, abs(fd)), (i, fd, gTo[i])
    for j in range(N - 1):
        for d in range(3):
            wp = way.copy(); wm = way.copy(); wp[j, d] += eps; wm[j, d] -= eps
            fd = (f(wp, T) - f(wm, T)) / (2 * eps)
            assert abs(fd - gP[d, j]) <= 1e-6 * max(1.0, abs(fd)), (j, d, fd, gP[d, j])
