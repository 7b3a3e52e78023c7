"""Committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them (restatement regression anchor).  GPU: the HIP path reproduces them through the
C-ABI without touching the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from common import REL_TOL, assert_close  # noqa: E402
import make_golden  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(make_golden.CASES)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(pkg, orc, name):
    want = np.load(os.path.join(GOLD, name + ".npz"))
    got = make_golden.build_case(pkg, orc, name)
    assert np.array_equal(got["occ"], want["occ"]) and np.array_equal(got["coeffs"], want["coeffs"])   # generators are stable
    assert abs(got["cost"] - float(want["cost"])) <= 1e-12 * abs(float(want["cost"]))
    assert_close(got["gradC"], want["gradC"], name + " gradC", tol=1e-11)
    assert_close(got["gradT"], want["gradT"], name + " gradT", tol=1e-11)
    assert np.array_equal(got["stats"], want["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(pkg, product_lib, name):
    capi, synth = pkg.capi, pkg.synth
    want = np.load(os.path.join(GOLD, name + ".npz"))
    variant, sname, sparams, kw = make_golden.CASES[name]
    v = {"V1": capi.V1_SWEPT, "V2": capi.V2_OCC_TILE, "V3": capi.V3_ESDF_TILE}[variant]
    eng = pkg.Engine(synth.default_config(v, **kw))
    occ, res = want["occ"], float(want["res"])
    eng.set_grid(synth.esdf_from_occupancy(occ, res), (0, 0, 0), res, capi.GRID_ESDF)
    eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    if sname:
        eng.set_shape(synth.make_shape(sname, params=sparams))
    tstar = None
    if variant == "V1":
        eng.set_points(want["points"])
        tstar = np.zeros(want["points"].shape[0])
    cost, gT, gC = eng.eval_single(want["T"], want["coeffs"], tstar=tstar)
    assert abs(cost - float(want["cost"])) <= REL_TOL * abs(float(want["cost"]))
    assert_close(gC, want["gradC"], name + " gradC")
    assert_close(gT, want["gradT"], name + " gradT")
    if variant == "V1":
        assert np.max(np.abs(tstar - want["tstar"])) <= 2e-5
