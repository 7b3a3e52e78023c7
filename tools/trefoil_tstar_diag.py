"""Developer diagnostic: the points of the C5 / Trefoil.obj workload whose free-running t* differs between device and oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
from benchlib.meshes import reference_mesh
import torch  # noqa
pkg = graft.load_package(); orc = graft.load_oracle(); capi, synth = pkg.capi, pkg.synth
res = 0.1
occ = synth.random_box_map((512,) * 3, res=res, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
ext = np.array(occ.shape) * res
T, Cf = synth.random_trajectory(ext, 20, seed=780, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
cm = synth.colmajor(Cf)
shape = synth.make_mesh_shape(*reference_mesh("Trefoil", 0.83))
cfg = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
eng = pkg.Engine(cfg); eng.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); eng.set_shape(shape)
way = cm.reshape(3, -1).T.reshape(20, 6, 3)[1:, 0, :]
M = eng.gather_points(way, 0.9)
pts = eng.get_points()
o = orc.Oracle(cfg, threads=30); o.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o.set_shape(shape); o.set_points(pts)
tg = -np.ones(M); to = -np.ones(M)
eng.eval_single(T, cm, tstar=tg); o.eval(T, cm, tstar=to)
bad = np.nonzero(np.abs(tg - to) > 1e-9)[0]
print("M", M, "differing:", bad)
for k in bad:
    p = pts[k]
    print(f"point {k} {p}: device t* {tg[k]!r}  oracle t* {to[k]!r}")
    for seed in (tg[k], to[k]):
        r = o.swept_sdf(T, cm, p, tstar0=seed)
        print("   oracle swept_sdf(seeded at", seed, ") ->", r)
    # single-point device run
    e1 = pkg.Engine(cfg); e1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e1.set_shape(shape); e1.set_points(pts[k:k + 1])
    t1 = -np.ones(1); c1 = e1.eval_single(T, cm, tstar=t1)
    o1 = orc.Oracle(cfg, threads=1); o1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o1.set_shape(shape); o1.set_points(pts[k:k + 1])
    t2 = -np.ones(1); c2 = o1.eval(T, cm, tstar=t2)
    print("   single-point: device t*", t1[0], "cost", c1[0], "| oracle t*", t2[0], "cost", c2[0])
    fp = np.array([cfg.vehicle_mass, cfg.grav_acc, cfg.horiz_drag, cfg.vert_drag, cfg.paras_drag, cfg.speed_eps])
    for t in [tg[k]] + [tg[k] + 0.02 / 2 ** q for q in range(9)] + [tg[k] - 0.02 / 2 ** q for q in range(9)]:
        st, piece, tl = orc.traj_eval(T, cm, t)
        q, w = orc.flat_forward(fp, st[1], st[2], st[3])
        R = orc.quat_to_rot(q)
        pr = R.T @ (p - st[0])
        sd, gd = eng.shape_eval(pr[None, :]); so, go = o.shape_eval(pr[None, :])
        print(f"   t={t!r}: device sdf {sd[0]!r} grad {gd[0]} | oracle sdf {so[0]!r} grad {go[0]}  diff {sd[0]-so[0]:.3e}")
