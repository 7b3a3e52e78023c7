import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as graft
import numpy as np
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from common import small_world, traj
name = sys.argv[1] if len(sys.argv) > 1 else "Torus"
occ, esdf, res = small_world(pkg, shape=(40, 40, 40), occupancy=0.10, seed=9)
T, cm = traj(pkg, occ, res, N=3, seed=41, margin=6.0)
shape = synth.make_shape(name, poly_params=(0.2, -0.1, 0.15, 30, 10, 120))
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=17, integral_intervs=6, safety_hor=0.5)
eng = pkg.Engine(cfg)
eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
eng.set_shape(shape)
print("eval...", flush=True)
print(eng.eval_single(T, cm)[0], eng.stats(), flush=True)
