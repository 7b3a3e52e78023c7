#!/usr/bin/env python
"""Generates tests/golden/ref_*.npz: outputs of THE REFERENCE'S OWN CODE (oracle/_ref/libref_sweeps.so - line ranges of the
reference compiled from /root/reference, see oracle/ref_build) on seeded small worlds.  The fixtures travel with the repository,
so the oracle (CPU tests) and the device (GPU tests) are checked against the reference even where neither /root/reference nor
oracle/_ref exists.  Needs /root/reference at generation time only:
    make -C oracle ref && python tests/golden/make_golden_ref.py
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

POLY = (0.1, -0.05, 0.08, 20, -10, 40)          # body offset of the robot (yaml poly_params)
V1_SHAPES = ["RoundedCone", "Box", "Torus_big"]


def world(pkg, orc, variant, shape_name, seed, **cfgkw):
    capi, synth = pkg.capi, pkg.synth
    occ, esdf, res = small_world(pkg, seed=seed)
    kw = dict(kernel_size=9, integral_intervs=16, safety_hor=0.5)
    kw.update(cfgkw)
    cfg = synth.default_config(variant, **kw)
    shape = synth.make_shape(shape_name, poly_params=POLY)
    return cfg, kw, occ, res, shape


def main():
    pkg = g.load_package(); orc = g.load_oracle(); capi, synth = pkg.capi, pkg.synth
    # ---- V1 swept-volume sweep (addSaftyPenaOnSweptVolumeParallel + the swept SDF query)
    for name in V1_SHAPES:
        cfg, kw, occ, res, shape = world(pkg, orc, capi.V1_SWEPT, name, 3)
        ref = orc.RefWorld(cfg, name, shape, res, threads=8); ref.set_occupancy(occ, (0, 0, 0), res)
        N = 6
        T, cm = traj(pkg, occ, res, N=N, seed=43)
        way = cm.reshape(3, -1).T.reshape(N, 6, 3)[1:, 0, :]
        pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
        ref.set_points(pts)
        c, gT, gC, ts = ref.sweep_swept(T, cm)
        np.savez_compressed(os.path.join(HERE, f"ref_v1_{name}.npz"), occ=occ, res=res, T=T, coeffs=cm, points=pts, cost=c, gradT=gT, gradC=gC, tstar=ts,
                            cfg_safety_hor=kw["safety_hor"])
        print(f"ref_v1_{name}: M={len(pts)} cost={c:.9g}")
    # ---- integral sweep, dynamics penalties (addTimeIntPenaltyParallel)
    dyn = dict(enable_pos=0, vmax=1.5, omgmax=0.6, thetamax=0.25)
    cfg, kw, occ, res, shape = world(pkg, orc, capi.V3_ESDF_TILE, "RoundedCone", 3, **dyn)
    ref = orc.RefWorld(cfg, "RoundedCone", shape, res, threads=8); ref.set_occupancy(occ, (0, 0, 0), res)
    T, cm = traj(pkg, occ, res, N=6, seed=11, piece_T=0.45)
    c, gT, gC = ref.sweep_integral(T, cm)
    np.savez_compressed(os.path.join(HERE, "ref_integral.npz"), occ=occ, res=res, T=T, coeffs=cm, cost=c, gradT=gT, gradC=gC, **{f"cfg_{k}": v for k, v in dyn.items()})
    print(f"ref_integral: cost={c:.9g}")
    # ---- the full objective callback: costFunctionLmbm assembled from the reference's own MINCO_S3NU (setParameters, getEnergy*,
    # propogateGrad) and its own two sweeps (back_end_optimizer.hpp:358-430)
    from common import tau_to_T  # noqa: E402
    cfg, kw, occ, res, shape = world(pkg, orc, capi.V1_SWEPT, "RoundedCone", 4, vmax=1.5, omgmax=0.6, thetamax=0.25)
    ref = orc.RefWorld(cfg, "RoundedCone", shape, res, threads=8); ref.set_occupancy(occ, (0, 0, 0), res)
    N = 5
    T0, cm0 = traj(pkg, occ, res, N=N, seed=33, piece_T=0.6)
    c6 = cm0.reshape(3, 6 * N).T.reshape(N, 6, 3)
    head = np.zeros((3, 3)); tail = np.zeros((3, 3))
    head[:, 0] = c6[0, 0]; tail[:, 0] = sum(c6[N - 1, p] * T0[-1] ** p for p in range(6))
    way = c6[1:, 0, :]
    rho = 2.5
    pts = synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0)
    ref.set_points(pts)
    # x = [tau | waypoints] exactly as isdf_pack_variables builds it
    from common import T_to_tau  # noqa: E402
    x = np.concatenate([T_to_tau(T0), way.reshape(-1)])
    tau, xi = x[:N], x[N:]
    T = tau_to_T(tau)
    cm_r, e_r, gC, gT = orc.ref_minco(head, tail, xi.reshape(N - 1, 3).T, T)
    total = e_r
    c1, gT1, gC1, _ = ref.sweep_swept(T, cm_r)
    total += c1; gT = gT + gT1; gC = gC + gC1
    c2, gT2, gC2 = ref.sweep_integral(T, cm_r)
    total += c2; gT = gT + gT2; gC = gC + gC2
    gP, gTt = orc.ref_minco_propagate(head, tail, xi.reshape(N - 1, 3).T, T, gC, gT)
    total += rho * T.sum()
    gTt = gTt + rho
    den = (0.5 * tau - 1.0) * tau + 1.0
    g_tau = np.where(tau > 0, gTt * (tau + 1.0), gTt * (1.0 - tau) / (den * den))
    grad_x = np.concatenate([g_tau, gP.T.reshape(-1)])
    np.savez_compressed(os.path.join(HERE, "ref_callback_v1.npz"), occ=occ, res=res, N=N, head=head, tail=tail, rho=rho, x=x, points=pts, cost=total, g=grad_x,
                        energy=e_r, cost_swept=c1, cost_integral=c2, cfg_vmax=1.5, cfg_omgmax=0.6, cfg_thetamax=0.25)
    print(f"ref_callback_v1: cost={total:.9g} (energy {e_r:.6g}, swept {c1:.6g}, integral {c2:.6g})")
    # ---- ESDF (GridMap3D::generateESDF3d)
    occ9, _, res9 = small_world(pkg, seed=9)
    ref.set_occupancy(occ9, (0, 0, 0), res9)
    e = ref.generate_esdf()
    np.savez_compressed(os.path.join(HERE, "ref_esdf.npz"), occ=occ9, res=res9, esdf=e.astype(np.float32))
    # ---- front end (attitude kernels, generateMapKernel, checkKernelValue)
    occf = synth.random_box_map((40, 36, 20), res=0.5, occupancy=0.10, seed=5, edge=(0.5, 2.0))
    for name, k in (("Box", 13), ("Torus_big", 15)):
        kwf = dict(params=(0.9, 0.35, 0.25)) if name == "Box" else {}
        shape = synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25), **kwf)
        cfg = synth.default_config(capi.V1_SWEPT)
        fe = capi.frontend_config(kernel_size=k)
        w = orc.RefWorld(cfg, name, shape, 0.5); w.set_occupancy(occf, (0, 0, 0), 0.5); w.frontend_build(fe, name)
        rng = np.random.default_rng(3)
        free = np.argwhere(occf == 0)
        idx = free[rng.choice(len(free), 400, replace=False)].astype(np.int32)
        fr = rng.choice(np.arange(-45.0, 45.1, 9.0), 400); fp = rng.choice(np.arange(-45.0, 45.1, 9.0), 400)
        res_q = np.array([w.frontend_check(idx[q], fr[q], fp[q]) for q in range(400)])
        np.savez_compressed(os.path.join(HERE, f"ref_frontend_{name}.npz"), occ=occf, res=0.5, kernel_size=k, shape_kernels=w.frontend_shape_kernels(),
                            map_kernel=w.frontend_map_kernel(), index=idx, father_roll=fr, father_pitch=fp, ok=res_q[:, 0].astype(np.uint8),
                            child_roll=res_q[:, 1], child_pitch=res_q[:, 2], box_params=np.array(kwf.get("params", (0, 0, 0))))
        print(f"ref_frontend_{name}: feasible {res_q[:, 0].mean():.2f}")


ASTAR_WORLDS = (("Box", 13, 0.22, 9), ("Torus_big", 15, 0.14, 7))      # robot class, kernel_size, occupancy, seed


def astar_world(pkg, name, occupancy, seed):
    synth = pkg.synth
    occ = synth.random_box_map((48, 40, 16), res=0.5, occupancy=occupancy, seed=seed, edge=(0.5, 2.5))
    kwf = dict(params=(0.9, 0.35, 0.25)) if name == "Box" else {}
    return occ, synth.make_shape(name, poly_params=(0.05, -0.1, 0.02, 12, -8, 25), **kwf)


def astar_fixtures():
    """ref_astar_<robot>.npz: the reference's own AstarPathSearch / getPath / attitude chain (front_end_Astar.hpp:160-376, compiled
    into oracle/_ref/libref_sweeps.so) on seeded cluttered worlds: start / goal pairs -> path cells, (roll, pitch) per node and
    the number of neighbour tests.  Searches that fail (goal where no attitude fits; start outside the map) are part of it."""
    pkg = g.load_package(); orc = g.load_oracle(); capi, synth = pkg.capi, pkg.synth
    for name, k, occupancy, seed in ASTAR_WORLDS:
        occ, shape = astar_world(pkg, name, occupancy, seed)
        cfg = synth.default_config(capi.V1_SWEPT)
        fe = capi.frontend_config(kernel_size=k)
        w = orc.RefWorld(cfg, name, shape, 0.5); w.set_occupancy(occ, (0, 0, 0), 0.5); w.frontend_build(fe, name)
        o = orc.Oracle(cfg, threads=8)              # only to pick start / goal cells where SOME attitude fits
        o.set_grid(occ, (0, 0, 0), 0.5, capi.GRID_OCCUPANCY); o.set_shape(shape); o.frontend_build(fe)
        fits = (o.frontend_cspace(occ.shape) != 0).any(axis=-1)
        good = np.argwhere(fits); bad = np.argwhere(~fits & (occ == 0))
        rng = np.random.default_rng(seed)
        starts, goals = [], []
        for _ in range(12):
            a, b = good[rng.choice(len(good), 2, replace=False)]
            starts.append((a + rng.random(3)) * 0.5); goals.append((b + rng.random(3)) * 0.5)
        starts.append(starts[0]); goals.append(starts[0] + 0.01)                                    # start cell == goal cell
        starts.append(np.array([-0.2, 3.0, 3.0])); goals.append(goals[1])                          # start outside the map
        starts.append((good[rng.choice(len(good))] + 0.5) * 0.5); goals.append((bad[rng.choice(len(bad))] + 0.5) * 0.5)      # free goal voxel where no attitude fits
        n, xyz, rp, checks = [], [], [], []
        for s_, g_ in zip(starts, goals):
            x, r, c = w.astar(s_, g_)
            n.append(-1 if x is None else len(x)); checks.append(c)
            if x is not None:
                xyz.append(x); rp.append(r)
        np.savez_compressed(os.path.join(HERE, f"ref_astar_{name}.npz"), occ=occ, res=0.5, kernel_size=k, start=np.array(starts), goal=np.array(goals),
                            n=np.array(n), checks=np.array(checks), xyz=np.concatenate(xyz), rp=np.concatenate(rp),
                            box_params=np.array((0.9, 0.35, 0.25) if name == "Box" else (0, 0, 0)))
        print(f"ref_astar_{name}: path lengths {n}, attitudes on paths {len(np.unique(np.concatenate(rp), axis=0))}, neighbour tests {checks}")


if __name__ == "__main__":
    if sys.argv[1:] == ["astar"]:
        astar_fixtures()
    else:
        main()
        astar_fixtures()
