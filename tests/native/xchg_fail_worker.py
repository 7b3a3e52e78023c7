"""Worker of tests/test_gpu_xchg.py::test_xchg_failure_is_agreed_in_the_kernel: two processes on GPU 0.  Rank 1 shows up LATE = three
times the exchange's time-out (isdf_xchg_timeout_ms: a bound in milliseconds of the device wall clock, the same on every box) for one
exchange: rank 0's bounded wait expires after exactly that time-out, and through the kernel's verdict round BOTH ranks end that exchange
with the status word set and a NaN in slot 0 - without any host-side collective.  At callback level both ranks then redo the step's
sum on the group's all-reduce and return the right (cost, g).  Any failure is printed to STDOUT first (a cut stderr tail once hid it)."""
import os, sys, time
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
    print(f"[xchg_fail_worker] rank {rank}: first step of the unsharded ctx, host hand-over {full.host_info()} (path {full.lib.isdf_host_path(full.h)})", flush=True)
    assert eng.xchg_timeout_ms() == 2000.0                      # the default: generous (ranks are separate processes)
    x = par.XgmiExchange(eng, eng.out_stride(6), dev)
    assert x.ok, x.why
    # healthy: the callback through the exchange, no host collective involved (twice: the first call loads kernels)
    for it in range(2):
        c, g = par.cost_function_allreduced(eng, xv, x)
        if not (abs(c - c_ref) <= 1e-12 * abs(c_ref) and np.allclose(g, g_ref, rtol=0, atol=1e-11 * np.abs(g_ref).max()) and x.ok):
            c2, g2 = full.cost_function(xv)            # which side is wrong: the unsharded reference evaluated once more
            bad = np.flatnonzero(np.abs(g - g_ref) > 1e-9 * np.abs(g_ref).max())
            raise AssertionError(("healthy callback", it, "cost", c, c_ref, c2, "max |g - g_ref|", float(np.abs(g - g_ref).max()), "max |g - g_ref2|", float(np.abs(g - g2).max()),
                                  "max |g_ref2 - g_ref|", float(np.abs(g2 - g_ref).max()), "of", float(np.abs(g_ref).max()), "entries off", bad.tolist(), "x.ok", x.ok, "status", eng.xchg_status()))
    # from here on a SHORT bound, so that the failures below take a second and not ten; the ranks are aligned by barriers
    dist.barrier()
    eng.xchg_set_timeout_ms(400.0)
    TIMEOUT = eng.xchg_timeout_ms() * 1e-3
    assert abs(TIMEOUT - 0.4) < 1e-12
    LATE = 3.0 * TIMEOUT          # seconds rank 1 shows up late
    # 1) the raw exchange with a late peer: both ranks must see the failure
    v = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
    dist.barrier()
    if rank == 1:
        time.sleep(LATE)
    t0 = time.perf_counter()
    x.allreduce(v); torch.cuda.synchronize()
    took = time.perf_counter() - t0
    assert eng.xchg_status() == 1 and not np.isfinite(float(v[0].item())), (rank, eng.xchg_status(), float(v[0].item()))
    # the failed kernel lasts the time-out on the rank whose wait expired (it does not wait for verdicts it no longer needs) and
    # no time at all on the late rank (rank 0's "bad" verdict is already in its mailbox)
    if rank == 0:
        assert 0.9 * TIMEOUT <= took <= TIMEOUT + 1.0, ("failed exchange kernel, seconds", took, "time-out", TIMEOUT)      # (+ launch and wake-up latency of a busy host)
    else:
        assert took <= 1.0, ("late rank's exchange kernel, seconds", took)
    print(f"[xchg_fail_worker] rank {rank}: failed exchange took {took:.3f} s (time-out {TIMEOUT:.3f} s, peer {LATE:.1f} s late)", flush=True)
    dist.barrier()
    # the next exchanges work again (sticky status notwithstanding): the mailboxes are back in step (healthy exchanges: the default bound)
    eng.xchg_set_timeout_ms(2000.0)
    for _ in range(4):
        v = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
        x.allreduce(v); torch.cuda.synchronize()
        assert float(v[0].item()) == 3.0, (rank, float(v[0].item()))
    dist.barrier()
    # 2) callback level: a late peer during a step -> both ranks redo the sum on the group's all-reduce and agree
    eng.xchg_set_timeout_ms(400.0)
    dist.barrier()
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


try:
    main()
except BaseException:       # noqa: BLE001 - the text of the failure goes to stdout FIRST, then the non-zero exit
    print(f"[xchg_fail_worker] rank {rank} FAILED:\n" + traceback.format_exc(), flush=True)
    raise
