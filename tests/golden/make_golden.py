#!/usr/bin/env python
"""Generates tests/golden/*.npz: seeded inputs + outputs of the CPU oracle for small cases of every sweep variant.

The reference ships no golden vectors for this path and cannot be built here (Eigen/ROS/PCL/libigl absent,
SURVEY.md §8c), so these fixtures pin the ORACLE (a regression anchor for the restatement) and give the GPU tests a
device-side target that does not need the oracle at run time.  Re-run after any deliberate oracle change:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
from common import small_world, traj  # noqa: E402

CASES = {
    # name: (variant, shape name, shape params, config overrides)
    "v3_roundedcone": ("V3", "RoundedCone", (0.8, 0.3, 1.6), dict(kernel_size=9, integral_intervs=12, safety_hor=0.5)),
    "v2_torus": ("V2", "Torus", (1.2, 0.25), dict(kernel_size=11, integral_intervs=8, safety_hor=0.4)),
    "v3_dyn_only": ("V3", None, None, dict(integral_intervs=12, enable_pos=0, vmax=2.0, omgmax=1.0, thetamax=0.3)),
    "v1_roundedcone": ("V1", "RoundedCone", (0.8, 0.3, 1.6), dict(safety_hor=0.5)),
}


def build_case(pkg, orc, name):
    capi, synth = pkg.capi, pkg.synth
    variant, sname, sparams, kw = CASES[name]
    occ, esdf, res = small_world(pkg, seed=21, shape=(40, 40, 28))
    N = 5
    T, cm = traj(pkg, occ, res, N=N, seed=77, piece_T=0.5 if name == "v3_dyn_only" else 1.4)
    v = {"V1": capi.V1_SWEPT, "V2": capi.V2_OCC_TILE, "V3": capi.V3_ESDF_TILE}[variant]
    cfg = synth.default_config(v, **kw)
    o = orc.Oracle(cfg)
    o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY)
    shape = None
    if sname:
        shape = synth.make_shape(sname, params=sparams)
        o.set_shape(shape)
    pts = np.zeros((0, 3))
    tstar = None
    if variant == "V1":
        way = np.asarray(cm).reshape(3, 6 * N).T.reshape(N, 6, 3)[1:, 0, :]
        pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
        o.set_points(pts)
        tstar = np.zeros(pts.shape[0])
    cost, gT, gC, st = o.eval(T, cm, tstar=tstar)
    return dict(occ=occ, res=res, T=T, coeffs=cm, points=pts, cost=cost, gradT=gT, gradC=gC, stats=np.array(st),
                tstar=np.zeros(0) if tstar is None else tstar)


def main():
    pkg = g.load_package(); orc = g.load_oracle()
    for name in CASES:
        d = build_case(pkg, orc, name)
        assert d["cost"] > 0, name
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "cost", d["cost"], "stats", d["stats"], "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
