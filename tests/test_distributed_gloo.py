"""N>1 path on CPU (world_size 2 and 3, gloo): the shard partition + the single all-reduce of the packed
[cost | gradT | gradC] vector that bench.py performs over RCCL.  Per-rank compute is stood in by the oracle restricted
to the SAME shard rule as isdf_set_shard (the product kernels need a GPU); the GPU-side shard arithmetic itself is
covered by tests/test_gpu_parity.py::test_shards_sum_to_full."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, variant_name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    from common import small_world, traj
    pkg = g.load_package(); orc = g.load_oracle()
    capi, synth, par = pkg.capi, pkg.synth, pkg.parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    par.init_process_group("gloo")
    occ, esdf, res = small_world(pkg)
    N = 7
    T, cm = traj(pkg, occ, res, N=N)
    shape = synth.make_shape("RoundedCone", params=(0.8, 0.3, 1.6))
    if variant_name == "V1":
        cfg = synth.default_config(capi.V1_SWEPT, safety_hor=0.5)
    else:
        cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=9, integral_intervs=8, safety_hor=0.5)
    o = orc.Oracle(cfg)
    o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    o.set_shape(shape)
    if variant_name == "V1":
        way = np.asarray(cm).reshape(3, 6 * N).T.reshape(N, 6, 3)[1:, 0, :]
        o.set_points(synth.constraint_points(occ, (0, 0, 0), res, way, half=3.0))
    if variant_name == "callback":
        _callback_worker(pkg, orc, o, rank, world, occ, res, q)
        dist.barrier()
        dist.destroy_process_group()
        return
    o.set_shard(rank, world)
    c, gT, gC, _ = o.eval(T, cm)
    out = torch.from_numpy(np.concatenate([[c], gT, gC]))
    assert out.numel() == 1 + 19 * N
    par.allreduce_packed(out)                      # ONE collective per step
    cost, gradT, gradC = par.unpack_out(out.numpy(), N)
    o.set_shard(0, 1)
    c0, gT0, gC0, _ = o.eval(T, cm)
    ok = (abs(cost - c0) <= 1e-12 * abs(c0) and np.allclose(gradT, gT0, rtol=1e-11, atol=1e-9)
          and np.allclose(gradC, gC0, rtol=1e-11, atol=1e-9) and c0 > 0)
    # every rank holds the same reduced vector
    gathered = [torch.zeros_like(out) for _ in range(world)]
    dist.all_gather(gathered, out)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _callback_worker(pkg, orc, o, rank, world, occ, res, q):
    """The multi-GPU form of the full objective callback (isdf_cost_function_launch / all-reduce / _finish), with the
    oracle standing in for the device sweeps: energy and MINCO are replicated, only the sweeps' partial sums are reduced."""
    import torch
    from common import oracle_cost_function, T_to_tau
    par = pkg.parallel
    N = 5
    rng = np.random.default_rng(3)
    T, cm = __import__("common").traj(pkg, occ, res, N=N, seed=5)
    c6 = np.asarray(cm).reshape(3, 6 * N).T.reshape(N, 6, 3)
    head = np.zeros((3, 3)); tail = np.zeros((3, 3))
    head[:, 0] = c6[0, 0]; tail[:, 0] = sum(c6[N - 1, p] * T[-1] ** p for p in range(6))
    way = c6[1:, 0, :]
    x = np.concatenate([T_to_tau(T), way.reshape(-1)])
    rho = 2.0
    c_full, g_full = oracle_cost_function(orc, [o], head, tail, rho, x, N)

    class Sharded:      # an oracle "sweep" whose partial sums are all-reduced before they are accumulated
        def eval(self, T_, cm_, accumulate_into):
            o.set_shard(rank, world)
            c, gT, gC, _ = o.eval(T_, cm_)
            o.set_shard(0, 1)
            buf = torch.from_numpy(np.concatenate([[c], gT, gC]))
            par.allreduce_packed(buf)
            cost, gT0, gC0 = accumulate_into
            b = buf.numpy()
            cost[0] += b[0]; gT0 += b[1:1 + N]; gC0 += b[1 + N:]
    c_sh, g_sh = oracle_cost_function(orc, [Sharded()], head, tail, rho, x, N)
    ok = abs(c_sh - c_full) <= 1e-12 * abs(c_full) and np.allclose(g_sh, g_full, rtol=1e-10, atol=1e-9)
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,variant", [(2, "V3"), (3, "V3"), (2, "V1"), (2, "callback")])
def test_sharded_allreduce_matches_single_rank(world, variant):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, variant, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res), res


def test_shard_range_partition(pkg):
    sr = pkg.parallel.shard_range
    for total in (0, 1, 7, 40, 400, 1001):
        for world in (1, 2, 3, 8):
            spans = [sr(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
