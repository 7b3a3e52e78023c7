#!/usr/bin/env python
"""Developer tool: isdf_mesh_info of the bench's mesh robots (what isdf_set_shape found and decided)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
from benchlib.meshes import reference_mesh
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
occ = synth.random_box_map((64,) * 3, res=0.2, occupancy=0.1, seed=1, edge=(0.4, 2.0))
for nm, mk in (("blob720", lambda: synth.blob_mesh(24, 16, scale=0.9)), ("blob4992", lambda: synth.blob_mesh(64, 40, scale=0.9)), ("trefoil", lambda: reference_mesh("Trefoil", 0.83)),
               ("lthick", lambda: reference_mesh("Lthick", 0.83)), ("mybox", lambda: reference_mesh("mybox", 0.83)), ("lprism", synth.l_prism_mesh)):
    V, F = mk()
    e = pkg.Engine(synth.default_config(capi.V1_SWEPT, safety_hor=0.0866, weight_p=4000.0)); e.set_grid(occ, (0, 0, 0), 0.2, capi.GRID_OCCUPANCY)
    e.set_shape(synth.make_mesh_shape(V, F))
    print(nm, e.mesh_info(), flush=True)
