#!/usr/bin/env python
"""Developer tool: one rank's share of a long trajectory (BASELINE configs[3]: 400 pieces on 8 GPUs = 50 pieces per rank) as
ONE fused launch vs sweep + tail - run with ISDF_FUSE_MAX_BLOCKS=768 (two launches at this size) and =1024 (fused)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
res = 0.2
occ, esdf, T, cm = build_workload(pkg, N, 256, res)
cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01)
eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(synth.bench_box_shape())
dev = torch.device("cuda", 0)
d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rank in (0, world // 2, world - 1):
    eng.set_shard(rank, world)
    for _ in range(20): eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"N={N} world={world} rank={rank} limit={os.environ.get('ISDF_FUSE_MAX_BLOCKS', 'default')}: {dt * 1e6:8.2f} us/step", flush=True)
