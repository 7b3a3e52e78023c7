#!/usr/bin/env python
"""Developer tool: the C5-shaped V1 entries of bench.py alone (512^3 occupancy @ 0.1 m, 20-piece trajectory, mesh robots), under
ISDF_MESH_WG = 0 / 1 (wave-cooperative / quad-cooperative walks).  usage: c5_mesh_bench.py [--shards=N] [blob720|trefoil|lthick|lprism ...]
--shards=N [--rank=R]: the step of rank R (default 0) of N (its contiguous share of the obstacle points), as one rank of an N-GPU run executes it."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
from benchlib.meshes import reference_mesh
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
res5 = 0.1
occ5 = synth.random_box_map((512,) * 3, res=res5, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
ext5 = np.array(occ5.shape) * res5
T5, Cf5 = synth.random_trajectory(ext5, 20, seed=780, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ5, res=res5)
cm5 = synth.colmajor(Cf5)
way5 = cm5.reshape(3, -1).T.reshape(20, 6, 3)[1:, 0, :]
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
dT = torch.from_numpy(T5).to(dev); dC = torch.from_numpy(cm5).to(dev)
meshes = {"lprism": synth.l_prism_mesh, "blob720": lambda: synth.blob_mesh(24, 16, scale=0.9), "lthick": lambda: reference_mesh("Lthick", 0.83),
          "trefoil": lambda: reference_mesh("Trefoil", 0.83), "mybox": lambda: reference_mesh("mybox", 0.83), "drone": lambda: reference_mesh("drone", 0.83),
          "kuang": lambda: reference_mesh("kuang", 0.83), "box96": lambda: reference_mesh("box", 0.83), "icosa": lambda: reference_mesh("icosahedron", 0.83)}
shards = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--shards=")]
rank = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--rank=")] or [0])[0]
names = [a for a in sys.argv[1:] if not a.startswith("--")]
for nm in (names or ["blob720", "trefoil"]):
    V, F = meshes[nm]()
    for wg in ("0", "1"):
        os.environ["ISDF_MESH_WG"] = wg
        cfg1 = synth.default_config(capi.V1_SWEPT, safety_hor=(3 ** 0.5 / 2) * res5, weight_p=4000.0)
        e1 = pkg.Engine(cfg1); e1.set_grid(occ5, (0, 0, 0), res5, capi.GRID_OCCUPANCY); e1.set_shape(synth.make_mesh_shape(V, F))
        M = e1.gather_points(way5, 0.9)
        if shards: e1.set_shard(rank, shards[0])
        dO = torch.zeros(e1.out_stride(20), dtype=torch.float64, device=dev)
        for _ in range(2):
            e1.eval_device(1, 20, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, st)
        torch.cuda.synchronize(); t0 = time.perf_counter(); steps = 5
        for _ in range(steps):
            e1.eval_device(1, 20, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        e1.profile_enable(1, secondary=True)
        for _ in range(3):
            e1.eval_device(1, 20, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, st)
        torch.cuda.synchronize()
        _, k_search = e1.profile_read(); e1.profile_enable(0)
        print(f"{nm} ({len(F)} faces) ISDF_MESH_WG={wg}{' rank %d of %d' % (rank, shards[0]) if shards else ''}: M = {M}, {dt * 1e3:8.3f} ms/step (search launches {k_search:8.3f} ms) cost {float(dO[0].item()):.12g} stats {e1.stats()}", flush=True)
        del e1
