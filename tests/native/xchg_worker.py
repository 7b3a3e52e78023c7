"""Worker of tests/test_gpu_xchg.py: one of several processes that all use GPU 0 (the exchange's IPC, flag and double-buffer
logic does not care which device a peer's mailbox lives on), rendezvous over gloo."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as graft

pkg = graft.load_package(); capi, synth, par = pkg.capi, pkg.synth, pkg.parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import traceback


def main():
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = pkg.Engine(synth.default_config(capi.V3_ESDF_TILE, device=0))
    N_MAX = 7601                                   # 1 + 19 * 400: the packed vector of the 400-piece configuration
    x = par.XgmiExchange(eng, N_MAX, dev)
    assert x.ok, f"rank {rank}: exchange not usable: {x.why}"
    gen = torch.Generator(device="cpu"); gen.manual_seed(99 + rank)
    for it in range(300):
        n = [1, 7, 761, 4096, N_MAX][it % 5]
        v = torch.randn(n, dtype=torch.float64, generator=gen)
        ref = v.clone(); dist.all_reduce(ref)
        d = v.to(dev)
        x.allreduce(d)
        if it % 7 == 0:
            torch.cuda.synchronize()               # most exchanges are queued back to back (ranks run ahead of each other)
        got = d.cpu()
        assert eng.xchg_status() == 0
        assert torch.allclose(got, ref, rtol=0, atol=1e-12 * float(ref.abs().max() + 1)), (rank, it, float((got - ref).abs().max()))
        # every rank forms the SAME sum bit for bit (rank-order addition)
        allv = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(allv, got)
        assert all(torch.equal(allv[0], a) for a in allv), (rank, it)
    # run-ahead: 2000 exchanges queued back to back with no host synchronisation in between (a fast rank is at most one
    # exchange ahead of a slow one: the double-buffered mailboxes must cope).  d <- allreduce(d) / world keeps d at the mean.
    d = torch.full((N_MAX,), float(rank + 1), dtype=torch.float64, device=dev)
    for it in range(2000):
        x.allreduce(d)
        d.mul_(1.0 / world)
        if rank == it % world and it % 97 == 0:
            torch.cuda.synchronize()               # perturb the ranks' relative timing
    torch.cuda.synchronize()
    assert eng.xchg_status() == 0
    assert torch.allclose(d.cpu(), torch.full((N_MAX,), (world + 1) / 2.0, dtype=torch.float64), rtol=1e-12, atol=0), (rank, float(d[0]))

    # a sharded engine step through the exchange == through the group's all-reduce
    occ = synth.random_box_map((48, 48, 32), res=0.5, occupancy=0.12, seed=3, edge=(1.0, 3.0))
    esdf = synth.esdf_from_occupancy(occ, 0.5)
    ext = np.array(occ.shape) * 0.5
    T, Cf = synth.random_trajectory(ext, 6, seed=21, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=0.5)
    cm = synth.colmajor(Cf)
    cfg = synth.default_config(capi.V3_ESDF_TILE, device=0, kernel_size=9, integral_intervs=12, safety_hor=0.5)
    e2 = pkg.Engine(cfg); e2.set_grid(esdf, (0, 0, 0), 0.5, capi.GRID_ESDF); e2.set_shape(synth.make_shape("RoundedCone", params=(0.6, 0.25, 1.6)))
    e2.set_shard(rank, world)
    x2 = par.XgmiExchange(e2, e2.out_stride(6), dev)
    assert x2.ok, x2.why
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    out = torch.zeros(e2.out_stride(6), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    e2.eval_device(1, 6, d_T.data_ptr(), d_C.data_ptr(), out.data_ptr(), 0, st)
    part = out.cpu().clone(); dist.all_reduce(part)
    x2.allreduce(out); torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), part, rtol=1e-13, atol=1e-13 * float(part.abs().max()))
    assert float(part[0]) > 0
    # the exchange INSIDE the step (isdf_xchg_fuse): every rank's launch returns the full result, bitwise the same on all ranks
    def _step():
        e2.eval_device(1, 6, d_T.data_ptr(), d_C.data_ptr(), out.data_ptr(), 0, st)
    assert x2.try_fused_step(_step, out), "in-kernel exchange did not reproduce step + all-reduce"
    for it in range(40):
        out.zero_(); _step()
        if it % 5 == rank % 5:
            torch.cuda.synchronize()               # perturb the ranks' relative timing
        if it % 8 == 0:
            got = out.cpu()
            assert torch.allclose(got, part, rtol=1e-12, atol=1e-12 * float(part.abs().max())), (rank, it, float((got - part).abs().max()))
            allv = [torch.zeros_like(got) for _ in range(world)]
            dist.all_gather(allv, got)
            assert all(torch.equal(allv[0], a) for a in allv), (rank, it)
    torch.cuda.synchronize()
    assert e2.stats()["overflow"] == 0
    e2.xchg_fuse(False)
    out.zero_(); _step(); torch.cuda.synchronize()     # back on the ordinary path: this rank's partial sums again
    p2 = out.cpu().clone(); dist.all_reduce(p2)
    assert torch.allclose(p2, part, rtol=1e-13, atol=1e-13 * float(part.abs().max()))
    # a batch of 3 trajectories (shard boundaries fall inside trajectories: the cost collector of each trajectory changes place)
    x3 = par.XgmiExchange(e2, 3 * e2.out_stride(6), dev)
    assert x3.ok, x3.why
    d_T3 = d_T.repeat(3); d_C3 = d_C.repeat(3)
    out3 = torch.zeros(3 * e2.out_stride(6), dtype=torch.float64, device=dev)
    def _step3():
        e2.eval_device(3, 6, d_T3.data_ptr(), d_C3.data_ptr(), out3.data_ptr(), 0, st)
    assert x3.try_fused_step(_step3, out3), "in-kernel exchange (batch) did not reproduce step + all-reduce"
    for it in range(6):
        out3.zero_(); _step3()
    torch.cuda.synchronize()
    got3 = out3.cpu().reshape(3, -1)
    for b in range(3):
        assert torch.allclose(got3[b], part, rtol=1e-12, atol=1e-12 * float(part.abs().max())), (rank, b, float((got3[b] - part).abs().max()))
    assert e2.stats()["overflow"] == 0
    e2.xchg_fuse(False)
    dist.barrier()
    # a step that is NOT one fused launch (here: a body offset; likewise shards beyond the residency limit, mesh robots) carries the
    # exchange in its tail launch (tail_kernel_xf)
    e5 = pkg.Engine(cfg); e5.set_grid(esdf, (0, 0, 0), 0.5, capi.GRID_ESDF)
    e5.set_shape(synth.make_shape("RoundedCone", params=(0.6, 0.25, 1.6), poly_params=(0.05, -0.02, 0.03, 10, -5, 20)))
    e5.set_shard(rank, world)
    x5 = par.XgmiExchange(e5, e5.out_stride(6), dev)
    assert x5.ok, x5.why
    out5 = torch.zeros(e5.out_stride(6), dtype=torch.float64, device=dev)
    def _step5():
        e5.eval_device(1, 6, d_T.data_ptr(), d_C.data_ptr(), out5.data_ptr(), 0, st)
    _step5(); torch.cuda.synchronize()
    part5 = out5.cpu().clone(); dist.all_reduce(part5)
    assert float(part5[0]) > 0
    assert x5.try_fused_step(_step5, out5), "exchange in the tail launch did not reproduce step + all-reduce"
    for it in range(20):
        out5.zero_(); _step5()
        if it % 5 == rank % 5:
            torch.cuda.synchronize()
        if it % 6 == 0:
            got = out5.cpu()
            assert torch.allclose(got, part5, rtol=1e-12, atol=1e-12 * float(part5.abs().max())), (rank, it)
            allv = [torch.zeros_like(got) for _ in range(world)]
            dist.all_gather(allv, got)
            assert all(torch.equal(allv[0], a) for a in allv), (rank, it)
    torch.cuda.synchronize()
    assert e5.stats()["overflow"] == 0
    e5.xchg_fuse(False)
    dist.barrier()
    # a peer that never shows up must not hang the stream: rank 0 runs one fused step ALONE - its polls run into their bound,
    # the step ends with the overflow flag and a NaN cost (last check: the boards are out of step afterwards)
    x4 = par.XgmiExchange(e2, e2.out_stride(6), dev)
    assert x4.ok, x4.why
    # callback level (MINCO + sweeps + adjoint): exchange kernel after the sweep vs exchange inside the sweep's launch
    Cf6 = Cf.reshape(6, 6, 3)
    head = np.zeros((3, 3)); head[:, 0] = Cf6[0, 0, :]
    tail = np.zeros((3, 3)); tail[:, 0] = sum(Cf6[5, k, :] * T[5] ** k for k in range(6))
    e2.set_trajectory(6, head, tail, 2.0)
    xv = e2.pack_variables(T, Cf6[1:, 0, :])
    c1, g1 = par.cost_function_allreduced(e2, xv, x4)
    assert x4.try_fused_step(_step, out) and x4.fused
    c2, g2 = par.cost_function_allreduced(e2, xv, x4)
    assert abs(c1 - c2) <= 1e-12 * abs(c1) and np.allclose(g1, g2, rtol=0, atol=1e-12 * np.abs(g1).max()), (rank, c1, c2)
    gg = [torch.zeros(g2.size, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gg, torch.from_numpy(np.ascontiguousarray(g2)))
    assert all(torch.equal(gg[0], a) for a in gg), rank
    dist.barrier()
    if rank == 0:
        import time
        t0 = time.perf_counter()
        out.zero_(); _step(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert e2.stats()["overflow"] != 0 and not np.isfinite(float(out[0].item())), (e2.stats(), float(out[0].item()))
        assert dt < 60.0, dt
    dist.barrier()
    if rank == 0:
        print("XCHG WORKERS OK", flush=True)
    dist.destroy_process_group()


try:
    main()
except BaseException:       # noqa: BLE001 - the text of the failure goes to stdout FIRST (a cut stderr tail once hid it), then the non-zero exit
    print(f"[xchg_worker] rank {rank} FAILED:\n" + traceback.format_exc(), flush=True)
    raise
