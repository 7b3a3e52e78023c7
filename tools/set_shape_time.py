#!/usr/bin/env python
"""Developer tool: what isdf_set_shape costs for a mesh robot with and without the distance lattice (ISDF_NO_F32_FILTER=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
from benchlib.meshes import reference_mesh
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
import torch
occ = synth.random_box_map((64,) * 3, res=0.2, occupancy=0.1, seed=1, edge=(0.4, 2.0))
for nm, mk in (("blob720", lambda: synth.blob_mesh(24, 16, scale=0.9)), ("blob4992", lambda: synth.blob_mesh(64, 40, scale=0.9)), ("trefoil", lambda: reference_mesh("Trefoil", 0.83)), ("lthick", lambda: reference_mesh("Lthick", 0.83))):
    try:
        V, F = mk()
    except Exception as e:
        print(nm, "skip", e); continue
    for off in ("1", "0"):
        os.environ["ISDF_NO_F32_FILTER"] = off
        cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.0866, weight_p=4000.0)
        e = pkg.Engine(cfg); e.set_grid(occ, (0, 0, 0), 0.2, capi.GRID_OCCUPANCY)
        sh = synth.make_mesh_shape(V, F)
        e.set_shape(sh); torch.cuda.synchronize()
        t0 = time.perf_counter(); e.set_shape(sh); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{nm} ({len(F)} faces) set_shape {'without' if off == '1' else 'with   '} lattice: {dt * 1e3:.1f} ms", flush=True)
