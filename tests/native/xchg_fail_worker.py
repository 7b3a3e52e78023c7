"""Worker of tests/test_gpu_xchg.py::test_xchg_failure_is_agreed_in_the_kernel: two processes on GPU 0.  Rank 1 shows up six seconds
late for one exchange: rank 0's bounded wait expires, and through the kernel's verdict round BOTH ranks end that exchange with
the status word set and a NaN in slot 0 - without any host-side collective.  At callback level both ranks then redo the step's
sum on the group's all-reduce and return the right (cost, g)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as graft

pkg = graft.load_package(); capi, synth, par = pkg.capi, pkg.synth, pkg.parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
LATE = 6.0      # seconds rank 1 shows up late: several times the exchange's bounded wait (~0.5-2 s of polling, depending on the box)
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
occ = synth.random_box_map((48, 48, 32), res=0.5, occupancy=0.12, seed=3, edge=(1.0, 3.0))
esdf = synth.esdf_from_occupancy(occ, 0.5)
ext = np.array(occ.shape) * 0.5
T, Cf = synth.random_trajectory(ext, 6, seed=21, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=0.5)
cfg = synth.default_config(capi.V3_ESDF_TILE, device=0, kernel_size=9, integral_intervs=12, safety_hor=0.5)
shape = synth.make_shape("RoundedCone", params=(0.6, 0.25, 1.6))
eng = pkg.Engine(cfg); eng.set_grid(esdf, (0, 0, 0), 0.5, capi.GRID_ESDF); eng.set_shape(shape); eng.set_shard(rank, world)
full = pkg.Engine(cfg); full.set_grid(esdf, (0, 0, 0), 0.5, capi.GRID_ESDF); full.set_shape(shape)
Cf6 = Cf.reshape(6, 6, 3)
head = np.zeros((3, 3)); head[:, 0] = Cf6[0, 0, :]
tail = np.zeros((3, 3)); tail[:, 0] = sum(Cf6[5, k, :] * T[5] ** k for k in range(6))
for e in (eng, full):
    e.set_trajectory(6, head, tail, 2.0)
xv = full.pack_variables(T, Cf6[1:, 0, :])
c_ref, g_ref = full.cost_function(xv)
x = par.XgmiExchange(eng, eng.out_stride(6), dev)
assert x.ok, x.why
# healthy: the callback through the exchange, no host collective involved
c, g = par.cost_function_allreduced(eng, xv, x)
assert abs(c - c_ref) <= 1e-12 * abs(c_ref) and np.allclose(g, g_ref, rtol=0, atol=1e-11 * np.abs(g_ref).max()) and x.ok
# 1) the raw exchange with a late peer: both ranks must see the failure
v = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
dist.barrier()
if rank == 1:
    time.sleep(LATE)
t0 = time.perf_counter()
x.allreduce(v); torch.cuda.synchronize()
assert eng.xchg_status() == 1 and not np.isfinite(float(v[0].item())), (rank, eng.xchg_status(), float(v[0].item()))
assert time.perf_counter() - t0 < 30.0
dist.barrier()
# the next exchanges work again (sticky status notwithstanding): the mailboxes are back in step
for _ in range(4):
    v = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
    x.allreduce(v); torch.cuda.synchronize()
    assert float(v[0].item()) == 3.0, (rank, float(v[0].item()))
dist.barrier()
# 2) callback level: a late peer during a step -> both ranks redo the sum on the group's all-reduce and agree
if rank == 1:
    time.sleep(LATE)
c, g = par.cost_function_allreduced(eng, xv, x)
assert not x.ok, "the failed exchange must have been abandoned on every rank"
assert abs(c - c_ref) <= 1e-12 * abs(c_ref) and np.allclose(g, g_ref, rtol=0, atol=1e-11 * np.abs(g_ref).max()), (rank, c, c_ref)
c, g = par.cost_function_allreduced(eng, xv, x)          # and stays on the all-reduce
assert abs(c - c_ref) <= 1e-12 * abs(c_ref)
dist.barrier()
if rank == 0:
    print("XCHG FAILURE AGREED OK", flush=True)
dist.destroy_process_group()
