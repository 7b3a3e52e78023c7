#!/usr/bin/env python
"""bench.py — headline benchmark of the collision cost/gradient hot path on MI355X.

Metric (BASELINE.json): SDF cost+grad point-evals/sec on a 256^3 map.  A "step" is one pass of the V3 ESDF-tile sweep
(include/isdf_accel.h: isdf_eval_device) over every constraint sample of the trajectory; one "point-eval" is one sampled
pose (piece i, sample j) incl. its kernel_size^3 ESDF tile.  Inputs are resident in HBM before the timed
region; results stay on the device (the PCIe-inclusive rate through the drop-in boundary is reported separately as
`host_api_ms`, never as `value`).  Nothing but the steps is inside the timed region: kernel durations come from separate
passes.

Workload at 1 GPU = BASELINE.json configs[1]: one trajectory, 40 pieces x 64 intervals (65 samples / piece), 256^3 fp32 ESDF
at 0.2 m, kernel_size 21, convex box robot (half extents 1.6/0.3/0.3 m, central-difference gradient).
At N > 1 GPUs = BASELINE.json configs[3]: ONE 400-piece trajectory, its pieces sharded contiguously across the ranks
(50 per rank at 8), the one exchange of [cost | gradT | gradC] per step inside the step's launches (xGMI peer stores), as a
one-shot exchange kernel, or as RCCL's all-reduce - whichever verifies, in that order; the RCCL figure is reported beside it.
Total work is fixed (26 000 evals per step): strong scaling.  (`--weak`: 40 pieces per rank instead.)

Launch: python bench.py [--gpus N --steps K --warmup W]; for N>1 via torch.distributed.run (one rank per GPU).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

from benchlib.roofline import (FP64_VALU_PEAK_TFLOPS, HBM_PEAK_GBS, N_SIMD, PEAK_CLOCK_HZ, algorithmic_bytes_per_eval, compulsory_bytes,      # noqa: E402,F401
                               issue_from_counters, kernel_source_hash, v1_source_hash)
from benchlib.workloads import build_workload      # noqa: E402,F401  (the developer tools import these names from here)
from benchlib.timing import native_boundary_ms     # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pieces", type=int, default=0, help="pieces of the trajectory (default: 40 at 1 GPU = configs[1], 400 at N > 1 = configs[3])")
    ap.add_argument("--workload", choices=("auto", "c2", "c4"), default="auto",
                    help="headline workload: c2 = BASELINE configs[1] (1 x 40 pieces), c4 = configs[3] (1 x 400 pieces); auto = c2 on one GPU, c4 on several "
                         "(so `--gpus 1 --workload c4` gives the one-GPU point of the N > 1 curve)")
    ap.add_argument("--no-c5", action="store_true", help="N > 1: skip the configs[4] entry (V1, mesh robot, 512^3 map, sharded by obstacle points)")
    ap.add_argument("--weak", action="store_true", help="N > 1: 40 pieces per rank (weak scaling) instead of the fixed 400-piece trajectory")
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--kernel-size", type=int, default=21)
    ap.add_argument("--cull", type=int, default=0, help="1: enable the V3 whole-tile cull (reported separately, never as value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the C3 / C4 / 512^3 entries of `configs`")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-pointer boundary timing (its ~1 500 host-direct launches of the same kernel run ~2 us longer and would dilute a rocprofv3 average of the device-resident step)")
    ap.add_argument("--lbfgs-iters", type=int, default=40, help="iterations of the L-BFGS driver timed after the sweep (0: skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--steady-steps", type=int, default=3000, help="steps of the extra steady-state measurement `value_steady` (0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # developer switches (not used by the driver): run an N > 1 launch on a box with ONE GPU to exercise the multi-process
    # flow - every rank on cuda:0, gloo instead of RCCL (which refuses two ranks on one device)
    single_dev = os.environ.get("ISDF_BENCH_SINGLE_DEVICE") == "1"
    backend = os.environ.get("ISDF_BENCH_BACKEND", "nccl")
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pkg = graft.load_package()
    capi, synth, par = pkg.capi, pkg.synth, pkg.parallel
    if world > 1:
        par.init_process_group(backend, device=dev)     # "nccl" == RCCL on ROCm
    res = 0.2 * 256.0 / args.grid if args.grid != 256 else 0.2
    K = 64
    if args.pieces > 0:
        n_pieces = args.pieces
    elif args.workload != "auto":
        n_pieces = 40 if args.workload == "c2" else 400
    elif world == 1:
        n_pieces = 40                    # configs[1]
    else:
        n_pieces = 40 * world if args.weak else 400      # configs[3]: ONE 400-piece trajectory
    scaling = "weak" if (world > 1 and args.weak) else "strong"
    occ, esdf, T, cm = build_workload(pkg, n_pieces, args.grid, res)
    shape = synth.bench_box_shape()
    cfg = synth.default_config(capi.V3_ESDF_TILE, device=local_rank, kernel_size=args.kernel_size, integral_intervs=K,
                               safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01, enable_dyn=1,
                               enable_pos=1, enable_cull=args.cull)
    eng = pkg.Engine(cfg)
    eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
    eng.set_shape(shape)
    eng.set_shard(rank, world)

    N = n_pieces
    d_T = torch.from_numpy(T).to(dev)
    d_C = torch.from_numpy(cm).to(dev)
    d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    # The ONE exchange of the step: the sum of [cost | gradT | gradC] over the ranks.  The library's one-shot peer-to-peer form
    # over xGMI (csrc/xchg.hip) is used when it sets up and verifies against the group's all-reduce on this node; otherwise
    # (or with ISDF_NO_XGMI_EXCHANGE=1) RCCL's all_reduce.
    xchg = None
    if world > 1 and os.environ.get("ISDF_NO_XGMI_EXCHANGE") != "1":
        xchg = par.XgmiExchange(eng, eng.out_stride(N), dev)
        if not xchg.ok:
            if rank == 0:
                print(f"[bench] peer-to-peer exchange not used ({xchg.why}); RCCL all_reduce", file=sys.stderr, flush=True)
            xchg = None

    def eval_only():
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, stream)

    # Preferred: the exchange INSIDE the step's launches (isdf_xchg_fuse: the owning workgroups store their rows into the
    # peers' outputs) - used when it reproduces "step + the group's all-reduce" on every rank.
    in_kernel = bool(xchg is not None and os.environ.get("ISDF_NO_FUSED_EXCHANGE") != "1" and xchg.try_fused_step(eval_only, d_out))

    def make_step(mode):       # "fused": exchange inside the launches | "xchg": one-shot peer-to-peer kernel | "rccl"
        def step():
            eval_only()
            if world > 1 and mode != "fused":
                if mode == "xchg":
                    xchg.allreduce(d_out)
                else:
                    par.allreduce_packed(d_out)
        return step

    def timed(step, steps, with_events=False):
        """warm-up, then EXACTLY `steps` steps between barrier + synchronize on both sides; nothing else in the region.  The host
        SPINS on the stream (query) before the closing synchronise, which then returns at once instead of adding the scheduler's
        wake-up latency to a region that lasts 0.3 ms at 20 steps.  with_events (the long steady-state run only): the region's
        duration on the launch stream itself (HIP events: torch's current stream IS the stream handed to isdf_eval_device) / steps -
        for a one-launch step that bounds the kernel's duration from above; the two event records cost the host ~10 us, so the
        headline region carries none."""
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream()
        e0 = e1 = None
        if with_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if with_events:
            e0.record()
        for _ in range(steps):
            step()
        if with_events:
            e1.record()
        while not cur.query():
            pass
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        return t1 - t0, (e0.elapsed_time(e1) / steps if with_events else 0.0)

    def any_rank(flag):
        t = torch.tensor([int(flag)], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item()) != 0

    # ---- duration of the dominant kernel, in a SEPARATE pass (events attached to every dispatch cost the stream 1-3 us each),
    # BEFORE the timed region: the process has spent seconds building the map on the CPU, and the pass doubles as the first GPU
    # activity of the run (the timed region below is still exactly W warm-up + K timed steps of the plain step)
    eng.profile_enable(1)
    for _ in range(40):
        eval_only()
    torch.cuda.synchronize()
    n_launch, kern_ms_dispatch = eng.profile_read()
    eng.profile_enable(0)

    mode = "none" if world == 1 else ("fused" if in_kernel else ("xchg" if xchg else "rccl"))
    names = {"none": "none", "fused": "in-kernel peer-to-peer exchange over xGMI (rows stored into the peers' outputs by the owning workgroups of the step's launches), verified against RCCL at start-up",
             "xchg": "xgmi peer-to-peer one-shot sum (csrc/xchg.hip), verified against RCCL at start-up", "rccl": "RCCL all_reduce"}
    elapsed, region_ms = timed(make_step(mode), args.steps)
    # a bounded wait that timed out anywhere invalidates the run: measure again one level down
    if mode == "fused" and any_rank(eng.stats()["overflow"] != 0 or not np.isfinite(float(d_out[0].item()))):
        eng.xchg_fuse(False)
        mode = "xchg"
        elapsed, region_ms = timed(make_step(mode), args.steps)
        names["xchg"] += " (the in-kernel exchange timed out and was abandoned)"
    if mode == "xchg" and any_rank(eng.xchg_status()):
        mode = "rccl"
        names["rccl"] = "RCCL all_reduce (the peer-to-peer exchange timed out and was abandoned)"
        elapsed, region_ms = timed(make_step(mode), args.steps)
    collective = names[mode]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    evals_per_step = N * (K + 1)
    value = evals_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    stats = eng.stats()
    out_gpu = d_out.cpu().numpy().copy()
    # the driver's 20-step region lasts half a millisecond, right after start-up (clocks still ramping): the same step over a
    # region long enough to be in steady state, as an EXTRA key - `value` stays what the K steps measured
    value_steady = None
    if world == 1 and args.steady_steps > 0:
        el_s, region_steady = timed(make_step(mode), args.steady_steps, with_events=True)
        if region_steady > 0:
            region_ms = region_steady
        value_steady = {"value": round(evals_per_step * args.steady_steps / el_s, 1), "steps": args.steady_steps,
                        "ms_per_step": round(1e3 * el_s / args.steady_steps, 5)}

    # the same trajectory with the exchange done by RCCL, for comparison (shorter; after the headline run)
    rccl_ms = None
    if world > 1 and mode != "rccl":
        if mode == "fused":
            eng.xchg_fuse(False)
        el2, _ = timed(make_step("rccl"), max(20, args.steps // 4))
        t2 = torch.tensor([el2], dtype=torch.float64, device=dev)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        rccl_ms = 1e3 * float(t2.item()) / max(20, args.steps // 4)
        if mode == "fused":
            eng.xchg_fuse(True)

    # ---- N > 1: the group's facts and a scaling line that can be READ (benchlib/multi.py)
    rccl_seen = one_gpu = c3_sharded = c5_sharded = None
    if world > 1:
        from benchlib.multi import scaling_entries
        from types import SimpleNamespace
        rccl_seen, one_gpu, c3_sharded, c5_sharded = scaling_entries(SimpleNamespace(
            args=args, pkg=pkg, capi=capi, synth=synth, par=par, world=world, rank=rank, local_rank=local_rank, dev=dev, stream=stream, eng=eng, cfg=cfg,
            shape=shape, occ=occ, esdf=esdf, res=res, K=K, N=N, d_T=d_T, d_C=d_C, mode=mode))

    evals_this_rank = (par.shard_range(N, rank, world)[1] - par.shard_range(N, rank, world)[0]) * (K + 1)
    one_launch = evals_this_rank // 4 + (N // world) <= 768 and world == 1
    # two live bounds of the kernel's duration, both from above: the dispatch-attached events (an instrumented dispatch runs
    # 1-3 us longer than a plain one) and - when the step IS one launch - the timed region's own stream duration / steps
    kern_ms = (min(kern_ms_dispatch, region_ms) if region_ms > 0 else kern_ms_dispatch) if (one_launch and kern_ms_dispatch > 0) else (kern_ms_dispatch if kern_ms_dispatch > 0 else region_ms)
    B = algorithmic_bytes_per_eval(args.kernel_size)
    achieved = (evals_this_rank * B) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    # ---- the roofline of the dominant kernel.  What bounds this design is NOT HBM: the sweep reads a 1-bit qualifier grid that
    # lives in L2 (three 2 MiB arrays at 256^3), its HBM traffic is ~2 MB per launch (PMC) - the kernel is bound by VALU
    # instruction ISSUE (and, on the 2 600-wave C2 launch, by latency at ~2 waves per SIMD).  So `roofline` is the issue-slot
    # model: frac = the share of SIMD-cycles in which a VALU instruction issues, from the committed counter passes of THIS
    # command (profiles/pmc_counters.json, tools/profile_round.sh; used only while the kernel sources are the ones it was
    # measured on) over the kernel duration measured live with HIP events.  SURVEY 8(d)'s byte model (k^3 x 4 B per eval: bytes
    # this kernel never moves) is kept as `nominal_bytes` - a work-rate proxy that EXCEEDS the HBM peak on the batch configs.
    traffic = None
    hbm_meas = None
    issue = None
    geom_has_counters = world == 1 and N == 40 and args.grid == 256 and args.kernel_size == 21
    tpath = os.path.join(ROOT, "profiles", "pmc_counters.json")
    if geom_has_counters:
        issue = issue_from_counters(tpath, kernel_source_hash(), live_kernel_us=kern_ms * 1e3 if kern_ms > 0 else None)
        if issue is not None:
            try:
                with open(tpath) as f:
                    c = json.load(f)["counters"]
                traffic = int(round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))        # gfx950: FETCH_SIZE x 2 (guide, HBM)
                hbm_meas = traffic / (kern_ms * 1e-3) / 1e9
            except Exception:       # noqa: BLE001
                traffic = None
    # SURVEY.md 8(d) asks for the COMPULSORY lower bound next to the algorithmic figure: the unique voxels that the tiles of
    # this rank's launch touch (neighbouring samples' tiles overlap heavily) x 4 B, plus the per-sample inputs / outputs
    compulsory = None
    if world == 1:
        compulsory = compulsory_bytes(T, cm, N, K, args.kernel_size, args.grid, res)
    nominal = {"achieved_gbs": round(achieved, 2), "frac_of_hbm_peak": round(achieved / HBM_PEAK_GBS, 4),
               "algorithmic_bytes_per_eval": B, "algorithmic_bytes_per_launch": evals_this_rank * B, "compulsory_bytes_per_launch": compulsory,
               "is_a_bound": False,
               "definition": "SURVEY 8(d): (k^3 x 4 B tile + corners + coefficients + outputs) x evals per launch / kernel duration / 8 TB/s - a work-rate "
                             "proxy in the north-star's unit, NOT traffic and NOT a bound: the kernel reads a 1-bit qualifier grid, its HBM bytes are `traffic`",
               "exceeds_1_on": "filled below from `configs` (every entry whose nominal_bytes_frac > 1)"}
    # `roofline` follows the contract (SURVEY 8(d) / BASELINE.md): achieved = algorithmic bytes per launch / the kernel's duration,
    # peak = the HBM spec peak, frac = achieved / peak.  What actually limits the kernel is instruction issue: `issue_frac`,
    # `lane_utilisation`, the measured HBM traffic (`traffic`, `hbm_frac_measured`) stand beside it.
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "issue_frac": None if issue is None else issue["valu_issue_busy_frac"],
                "lane_utilisation": None if issue is None else issue["lane_utilisation"],
                "waves_per_simd_avg": None if issue is None else issue["waves_per_simd_avg"],
                "wait_frac_of_wave_cycles": None if issue is None else issue["wait_frac_of_wave_cycles"],
                "counters": None if issue is None else issue["source"],
                "limits": "VALU instruction issue / latency (C2 is ONE launch of 2 600 wavefronts for 1 024 SIMDs); the kernel reads a 1-bit qualifier grid from L2, "
                          "so the contract's byte figure is a work rate in the north-star's unit, not traffic"}
    roofline.update({"hbm_frac_measured": None if hbm_meas is None else round(hbm_meas / HBM_PEAK_GBS, 5),
                     "hbm_gbs_measured": None if hbm_meas is None else round(hbm_meas, 1),
                     "kernel": "sweep_kernel", "kernel_ms": round(kern_ms, 5), "launches": n_launch,
                     "kernel_ms_dispatch_events": round(kern_ms_dispatch, 5), "kernel_ms_region_events": round(region_ms, 5),
                     "nominal_bytes": nominal,
                     # secondary figure of SURVEY 8(d): robot-SDF evaluations per second inside the tiles (every pair costs one
                     # evaluation, every pair with a non-zero penalty six more for the central-difference gradient)
                     "pair_sdf_evals_per_s": round((stats["pairs"] + 6 * stats["grad_pairs"]) / (ms_per_step * 1e-3), 1)})

    result = None
    if rank == 0:
        # ---- the other single-GPU configurations SURVEY 7 names as the memory-side-heavy ones, same JSON (device resident)
        extra = None
        if world > 1 and (c3_sharded is not None or c5_sharded is not None):
            extra = [e_ for e_ in (c3_sharded, c5_sharded) if e_ is not None]
        if world == 1 and not args.no_extra_configs:
            from benchlib.configs import extra_configs
            from types import SimpleNamespace
            extra = extra_configs(SimpleNamespace(args=args, graft=graft, pkg=pkg, capi=capi, synth=synth, dev=dev, stream=stream, local_rank=local_rank, eng=eng, cfg=cfg,
                                                  shape=shape, occ=occ, esdf=esdf, res=res, T=T, cm=cm, K=K, roofline=roofline))

        # ---- through the drop-in boundary (PCIe inclusive, synchronous, accumulate) - informational, never `value`
        host_ms = None
        cb_native_host_minco_ms = None
        cb_native_device_minco_ms = None
        cb_minco_auto = None
        host_ms_multi = None
        host_ms_multi8 = None
        host_queue_multi8 = None
        host_timer = None
        cb_native_ms = None
        x0 = None
        c6 = cm.reshape(3, 6 * N).T.reshape(N, 6, 3)
        if world == 1 and not (args.no_host_api and args.lbfgs_iters <= 0):
            eng.set_shard(0, 1)
            head = np.stack([c6[0, 0], c6[0, 1], 2.0 * c6[0, 2]], axis=1)
            tl = T[-1]
            pw = lambda k: sum(np.prod([p - q for q in range(k)]) * c6[N - 1, p] * tl ** (p - k) for p in range(k, 6))
            tail = np.stack([pw(0), pw(1), pw(2)], axis=1)
            eng.set_trajectory(N, head, tail, 1000.0)
            x0 = eng.pack_variables(T, c6[1:, 0, :])
            nat = None if args.no_host_api else native_boundary_ms(pkg, eng, N, T, cm, x0)
            if args.no_host_api:
                pass
            elif nat is not None:
                host_ms, cb_native_ms = nat
                # the same callback with MINCO forced onto the host (the reference's band LU) / onto the device (minco_dev.hip) - the
                # A/B of N1; left to itself (above) the library takes the faster one for the configuration
                cb_minco_auto = "device" if eng.minco_path() == 1 else "host"
                try:
                    eng.set_minco_mode(capi.MINCO_HOST)
                    nat_h = native_boundary_ms(pkg, eng, N, T, cm, x0, eval_too=False)
                    cb_native_host_minco_ms = None if nat_h is None else nat_h[1]
                    eng.set_minco_mode(capi.MINCO_DEVICE)
                    nat_d = native_boundary_ms(pkg, eng, N, T, cm, x0, eval_too=False)
                    cb_native_device_minco_ms = None if nat_d is None else nat_d[1]
                finally:
                    eng.set_minco_mode(capi.MINCO_AUTO)
                host_timer = "native: median of 1000 isdf_eval calls inside the compiled reference-side binding (tests/native/binding_stub.cpp)"
                # the same call on a ctx over TWO 'devices' (this GPU listed twice: isdf_create_multi) - what the single-process
                # multi-GPU path adds on top of the copy path (input broadcast, two shard launches, the rank-ordered sum kernel)
                try:
                    em = pkg.Engine(cfg, devices=[local_rank, local_rank])
                    em.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); em.set_shape(shape)
                    natm = native_boundary_ms(pkg, em, N, T, cm, None)
                    host_ms_multi = None if natm is None else round(natm[0], 4)
                    del em
                    em8 = pkg.Engine(cfg, devices=[local_rank] * 8)
                    em8.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); em8.set_shape(shape)
                    natm8 = native_boundary_ms(pkg, em8, N, T, cm, None)
                    host_ms_multi8 = None if natm8 is None else round(natm8[0], 4)
                    # what the calling thread spends QUEUEING a device-resident step on those 8 shards (isdf_eval_device on a
                    # caller stream, no synchronisation; the figure above is the synchronous host-pointer call and contains the
                    # eight shards' kernels running on this ONE GPU)
                    qs = torch.cuda.Stream(); torch.cuda.synchronize()
                    dTq = torch.from_numpy(T).cuda(); dCq = torch.from_numpy(cm).cuda(); dOq = torch.zeros(em8.out_stride(N), dtype=torch.float64, device="cuda")
                    for _ in range(50): em8.eval_device(1, N, dTq.data_ptr(), dCq.data_ptr(), dOq.data_ptr(), 0, qs.cuda_stream)
                    torch.cuda.synchronize()
                    qt = []
                    for _ in range(7):           # 7 bursts of 60 steps, the queue drained in between
                        tq = time.perf_counter()
                        for _ in range(60): em8.eval_device(1, N, dTq.data_ptr(), dCq.data_ptr(), dOq.data_ptr(), 0, qs.cuda_stream)
                        qt.append((time.perf_counter() - tq) / 60)
                        torch.cuda.synchronize()
                    host_queue_multi8 = {"median": round(1e3 * float(np.median(qt)), 4), "min": round(1e3 * float(np.min(qt)), 4), "max": round(1e3 * float(np.max(qt)), 4),
                                         "bursts": len(qt), "steps_per_burst": 60}      # (a burst that runs into the runtime's queue limits waits for the GPU: the min is the calling thread's own cost)
                    del em8
                except Exception:       # noqa: BLE001
                    host_ms_multi = None
            else:
                eng.eval_single(T, cm)
                th0 = time.perf_counter()
                for _ in range(20):
                    eng.eval_single(T, cm)
                host_ms = 1e3 * (time.perf_counter() - th0) / 20
                host_timer = "python/ctypes (no C++ compiler at hand): includes ~15 us of marshalling per call"

        # second half of BASELINE.json's metric: "L-BFGS iter ms" - the optimizer driver (isdf_optimize_lbfgs) on the
        # full objective callback (MINCO where the library puts it - `lbfgs.minco` says - + the device sweeps), same map / robot / piece count.  Informational.
        lbfgs = None
        if world == 1 and args.lbfgs_iters > 0:
            c_first = eng.cost_function(x0)[0]
            xo, r = eng.optimize_lbfgs(x0, max_iterations=args.lbfgs_iters, g_epsilon=0.0, past=0)
            lbfgs = {"minco": "device (csrc/minco_dev.hip: x -> (cost, g) without leaving the device)" if eng.minco_path() == 1 else "host (the library's choice for a single trajectory of <= 64 pieces: the band LU's 10 us beat two more launches)",
                     "iterations": r["iterations"], "evaluations": r["evaluations"], "status": r["status"],
                     "ms_per_iteration": round(r["wall_ms"] / max(1, r["iterations"]), 4),
                     "ms_per_evaluation": round(r["wall_ms"] / max(1, r["evaluations"]), 4),
                     "callback_ms_native": None if cb_native_ms is None else round(cb_native_ms, 4),
                     "callback_ms_native_host_minco": None if cb_native_host_minco_ms is None else round(cb_native_host_minco_ms, 4),
                     "callback_ms_native_device_minco": None if cb_native_device_minco_ms is None else round(cb_native_device_minco_ms, 4),
                     "cost_first": c_first, "cost_last": r["f"]}
            try:        # the same driver with MINCO forced the other way
                other = capi.MINCO_HOST if eng.minco_path() == 1 else capi.MINCO_DEVICE
                eng.set_minco_mode(other)
                _, rh = eng.optimize_lbfgs(x0, max_iterations=args.lbfgs_iters, g_epsilon=0.0, past=0)
                lbfgs["host_minco" if other == capi.MINCO_HOST else "device_minco"] = {"iterations": rh["iterations"], "evaluations": rh["evaluations"],
                                       "ms_per_iteration": round(rh["wall_ms"] / max(1, rh["iterations"]), 4),
                                       "ms_per_evaluation": round(rh["wall_ms"] / max(1, rh["evaluations"]), 4), "cost_last": rh["f"]}
            finally:
                eng.set_minco_mode(capi.MINCO_AUTO)

        # measured HBM copy rate of this box next to the spec peak the roofline uses
        hbm_copy = None
        if world == 1:
            a = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB
            b_ = torch.empty_like(a)
            b_.copy_(a); torch.cuda.synchronize()
            tcp = time.perf_counter()
            for _ in range(10):
                b_.copy_(a)
            torch.cuda.synchronize()
            hbm_copy = round(10 * 2 * a.numel() * 4 / (time.perf_counter() - tcp) / 1e9, 1)
            del a, b_

        cpu = None
        parity = None
        if not args.no_cpu_baseline:
            # cpu_baseline leg: the oracle (CPU restatement of the reference's OpenMP sweep) on the SAME workload,
            # bounded to ~cpu-seconds of wall time.  This is the only place bench.py touches oracle/.
            orc = graft.load_oracle()
            ncpu = os.cpu_count() or 1
            o = orc.Oracle(cfg, threads=ncpu)
            o.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
            o.set_shape(shape)
            c0, gT0, gC0, _ = o.eval(T, cm)   # warm-up + parity reference
            # the box's logical cores, and the 30 threads of the shipped configs (threads_num);
            # the faster of the two is the reported baseline
            runs = []
            for threads in sorted({min(30, ncpu), ncpu}):
                o.set_threads(threads)
                o.eval(T, cm)
                n_cpu = 0
                tc0 = time.perf_counter()
                while True:
                    o.eval(T, cm)
                    n_cpu += 1
                    if time.perf_counter() - tc0 > args.cpu_seconds / 2:
                        break
                cpu_el = time.perf_counter() - tc0
                runs.append((evals_per_step * n_cpu / cpu_el, threads, n_cpu, cpu_el))
            best = max(runs)
            cpu = {"value": round(best[0], 1), "unit": "point-evals/s", "cores": best[1], "kind": "port",
                   "sample": f"{best[2]} full steps of the same workload ({evals_per_step} evals each) in {best[3]:.1f} s, "
                             f"oracle OpenMP schedule(dynamic) + critical-section merge like the reference; "
                             + "; ".join(f"{r[1]} threads: {r[0]:.0f} evals/s" for r in runs),
                   "ms_per_step": round(1e3 * best[3] / best[2], 2)}
            ref = np.concatenate([[c0], gT0, gC0])
            parity = float(np.linalg.norm(out_gpu - ref) / max(np.linalg.norm(ref), 1e-300))
        # the per-configuration issue fractions the documents quote, as top-level keys (the driver's record keeps those)
        def _frac(pred, key):
            for e_ in (extra or []):
                if isinstance(e_, dict) and pred(e_.get("name", "")) and isinstance(e_.get(key), dict):
                    return e_[key].get("valu_issue_busy_frac")
            return None
        issue_keys = {"c2_issue_frac": None if issue is None else issue["valu_issue_busy_frac"],
                      "c3_issue_frac": _frac(lambda n: n.startswith("C3 (configs[2])"), "roofline"),
                      "v3mesh_issue_frac": _frac(lambda n: "MESH robot (blob, 720" in n, "roofline"),
                      "v1_issue_frac": _frac(lambda n: n.startswith("V1 swept-volume sweep") and "rounded cone" in n, "roofline_issue"),
                      "v1mesh_issue_frac": _frac(lambda n: n.startswith("C5 (configs[4]) shape") and "720" in n, "roofline_issue")}
        result = {
            "metric": "SDF cost+grad point-evals/sec", "value": round(value, 1), "unit": "point-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"V3 ESDF-tile sweep: 1 trajectory x {N} pieces x {K + 1} samples/piece, {args.grid}^3 fp32 ESDF "
                                   f"@ {res:g} m, kernel_size {args.kernel_size}, box robot (1.6,0.3,0.3) central-diff grad, "
                                   f"occ_thresh = 0, cull {'on' if args.cull else 'off'}"
                                   + (" = BASELINE configs[1]" if (world == 1 and N == 40) else "")
                                   + ("" if world == 1 else (f" = BASELINE configs[3], {N // world}-{-(-N // world)} pieces per rank" if N == 400 else ""))
                                   + ("" if world == 1 else "; the --gpus 1 line is configs[1] (another workload): read `speedup`, not value(N)/value(1)"),
                       "pieces": N, "samples_per_piece": K + 1, "grid": args.grid, "kernel_size": args.kernel_size,
                       "sharding": ("none" if world == 1 else ("pieces contiguous by rank; each piece's 19 rows + cost stored by its owner into every peer's output inside the step's launches" if mode == "fused" else "pieces contiguous by rank + all_reduce(sum) of 1+19N doubles")),
                       "collective": collective,
                       "rccl_all_reduce_ms_per_step": None if rccl_ms is None else round(rccl_ms, 5),
                       "group": rccl_seen},
            "roofline": roofline, "roofline_issue": issue, "cpu_baseline": cpu, "configs": extra,
            "one_gpu_same_workload": one_gpu, "speedup": None if one_gpu is None else round(value / one_gpu["evals_per_s"], 3),
            "value_steady": value_steady, "parity_rel_err_vs_oracle": parity, "host_api_ms": None if host_ms is None else round(host_ms, 4), "host_api_timer": host_timer,
            "host_api_ms_two_shards_one_gpu": host_ms_multi, "host_api_ms_eight_shards_one_gpu": host_ms_multi8, "host_queue_ms_eight_shards_one_gpu": host_queue_multi8,
            "stats_last_step_rank0": stats, "lbfgs": lbfgs, "hbm_copy_gbs_measured": hbm_copy,
        }
        result.update(issue_keys)
        # the long form (every entry with its prose) goes to a file; the ONE printed line is the compact record (< 4 KB: the driver
        # keeps an 8 KB tail of stdout - round 5's 21 KB line could not be parsed)
        from benchlib.compact import MAX_LINE, compact_line
        detail_rel = os.path.join("gpurun_out", f"bench_detail_n{world}.json")
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, detail_rel), "w") as f:
                json.dump(result, f, indent=1)
            result["detail_file"] = detail_rel
        except OSError:
            result["detail_file"] = None
        line = json.dumps(compact_line(result))
        assert len(line) < MAX_LINE, len(line)
        print(line, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
