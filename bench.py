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

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6  # fp64 vector peak: half the 157.3 TFLOP/s fp32 vector rate of the same guide
N_SIMD = 1024                # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9        # max shader clock (same guide)


def kernel_source_hash():
    """sha256 over the sources the step's dominant kernel is compiled from: PMC counters measured on another version of them
    are not reported (profiles/pmc_counters.json carries the hash it was measured on)."""
    h = hashlib.sha256()
    for f in ("tile_sweep.hip", "dev_math.hpp", "dev_shapes.hpp", "dev_mesh.hpp", "isdf_internal.hpp"):
        with open(os.path.join(ROOT, "implicit-sdf-planner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def v1_source_hash():
    """the same for the swept-volume sweep's kernels (profiles/pmc_counters_v1.json)"""
    h = hashlib.sha256()
    for f in ("swept_sweep.hip", "dev_math.hpp", "dev_shapes.hpp", "dev_mesh.hpp", "isdf_internal.hpp"):
        with open(os.path.join(ROOT, "implicit-sdf-planner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def issue_from_counters(path, want_hash, live_kernel_us=None):
    """Issue-slot roofline block from a tools/pmc_collect.py JSON (per-dispatch means of the SQ counters, summed over the kernels it
    lists); None when the file is missing, unreadable or was measured on other kernel sources.  The counters are properties of
    the kernel + workload (instructions issued per launch); the DURATION they are divided by is `live_kernel_us` when given (the
    kernel's duration measured in this run with HIP events), the rocprofv3 average of the counter run otherwise."""
    try:
        with open(path) as f:
            pj = json.load(f)
        if pj.get("source_hash") != want_hash:
            return None
        c = pj["counters"]
        k_us = live_kernel_us if live_kernel_us else pj["avg_kernel_us"]
        cyc = k_us * 1e-6 * PEAK_CLOCK_HZ
        f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
        flops64 = (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]) * 64
        busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc)
        lane = pj.get("lane_utilisation_valu")
        if lane is None and "SQ_THREAD_CYCLES_VALU" in c:
            lane = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        share64 = f64 / c["SQ_INSTS_VALU"]
        return {"kernels": pj.get("kernel"), "valu_issue_busy_frac": round(busy, 4),
                "lane_utilisation": None if lane is None else round(lane, 4),
                "fp64_share_of_valu_insts": round(share64, 4),
                "useful_fp64_frac": None if lane is None else round(busy * share64 * lane, 4),
                "wait_frac_of_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                "waves_per_simd_avg": round(4.0 * c["SQ_WAVE_CYCLES"] / (N_SIMD * pj["avg_kernel_us"] * 1e-6 * PEAK_CLOCK_HZ), 3),
                "valu_insts_per_step": int(c["SQ_INSTS_VALU"]), "salu_insts_per_step": int(c["SQ_INSTS_SALU"]),
                "fp64_tflops": None if lane is None else round(lane * flops64 / (k_us * 1e-6) / 1e12, 3),
                "fp64_valu_peak_tflops": FP64_VALU_PEAK_TFLOPS,
                "fp64_frac_of_peak": None if lane is None else round(lane * flops64 / (k_us * 1e-6) / 1e12 / FP64_VALU_PEAK_TFLOPS, 4),
                "definition": "valu_issue_busy = SQ_ACTIVE_INST_VALU (quad-cycles per launch, PMC) x 4 / (1024 SIMDs x kernel duration x 2.4 GHz): 1.0 = a VALU instruction "
                              "issuing on every SIMD in every cycle; lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); useful_fp64_frac = busy x fp64 share "
                              "of the VALU instructions x lane utilisation; fp64 flops = (ADD + MUL + 2 FMA + TRANS wave-instructions) x 64 lanes x lane utilisation",
                "kernel_us": round(k_us, 3), "kernel_us_source": "HIP events, this run" if live_kernel_us else "rocprofv3 average of the counter run",
                "kernel_us_under_rocprof": round(pj["avg_kernel_us"], 3), "measured_on_commit": pj.get("commit"), "source": os.path.relpath(path, ROOT)}
    except Exception:       # noqa: BLE001
        return None


def compulsory_bytes(T, cm, N, K, k, grid, res):
    """Unique ESDF voxels under the tiles of all N*(K+1) samples x 4 B + per-sample coefficients / T / outputs (the
    compulsory traffic if every byte were read once): numpy restatement of the tile boxes (getPointsInAABB floor indexing)."""
    c6 = cm.reshape(3, 6 * N).T.reshape(N, 6, 3)                   # piece, power, axis
    touched = np.zeros((grid, grid, grid), dtype=bool)
    half = k * res / 2
    for i in range(N):
        t = np.arange(K + 1) * (T[i] / K)
        pw = np.stack([t ** p for p in range(6)], axis=1)          # (K+1, 6)
        pos = pw @ c6[i]                                           # (K+1, 3)
        lo = np.clip(np.floor((np.clip(pos - half, 0, grid * res)) / res).astype(int), 0, grid - 1)
        hi = np.clip(np.floor((np.clip(pos + half, 0, grid * res)) / res).astype(int), 0, grid - 1)
        for a, b in zip(lo, hi):
            touched[a[0]:b[0] + 1, a[1]:b[1] + 1, a[2]:b[2] + 1] = True
    return int(touched.sum()) * 4 + N * (K + 1) * (144 + 8 + 152)


def algorithmic_bytes_per_eval(k, s_vox=4):
    """tile staged once + trilinear corners + 6x3 coefficients + T_i + 19 partial outputs."""
    return k ** 3 * s_vox + 8 * s_vox + 144 + 8 + 152


def build_workload(pkg, n_pieces, grid_n, res, seed_map=12345, seed_traj=777):
    synth = pkg.synth
    occ = synth.random_box_map((grid_n,) * 3, res=res, occupancy=0.15, seed=seed_map)
    esdf = synth.esdf_from_occupancy(occ, res)
    ext = np.array(occ.shape) * res
    T, Cf = synth.random_trajectory(ext, n_pieces, seed=seed_traj, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    return occ, esdf, T, synth.colmajor(Cf)


def native_boundary_ms(pkg, eng, N, T, cm, x0, eval_too=True):
    """isdf_eval and the lmbm_evaluate_t callback as a C++ caller sees them: timed inside the compiled reference-side
    binding (tests/native/binding_stub.cpp, built here with g++) - ctypes marshalling costs as much as the call itself.
    Returns (eval_ms, callback_ms) medians, or None when no compiler is at hand."""
    import ctypes as C
    try:
        out = os.path.join(tempfile.mkdtemp(), "libbinding_stub.so")
        libdir = os.path.dirname(pkg.capi.LIB_PATH)
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "oracle", "ref_build", "eigen_shim"),
                               os.path.join(ROOT, "tests", "native", "binding_stub.cpp"), "-o", out,
                               "-L", libdir, "-lisdf_accel", f"-Wl,-rpath,{libdir}"], stderr=subprocess.DEVNULL)
        stub = C.CDLL(out)
    except Exception:
        return None
    dp = C.POINTER(C.c_double)
    stub.stub_time_eval.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int, dp]
    stub.stub_time_callback.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, dp, dp]
    lat = np.zeros(1000 if eval_too else 4)
    if stub.stub_time_eval(eng.h, N, T.ctypes.data_as(dp), cm.ctypes.data_as(dp), lat.size, lat.ctypes.data_as(dp)) != 0:
        return None
    cb = None
    if x0 is not None:
        cbl = np.zeros(500); cst = np.zeros(1)
        stub.stub_time_callback(eng.h, x0.size, x0.ctypes.data_as(dp), cbl.size, cbl.ctypes.data_as(dp), cst.ctypes.data_as(dp))
        cb = float(np.median(cbl)) * 1e-3
    return float(np.median(lat)) * 1e-3, cb


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

    # what the collective library saw, and what ONE all-reduce of the step's packed vector costs by itself on this group
    rccl_seen = None
    if world > 1:
        ar = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
        for _ in range(5):
            dist.all_reduce(ar)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(ar)
        torch.cuda.synchronize()
        ta = torch.tensor([(time.perf_counter() - ta) / 50], dtype=torch.float64, device=dev)
        dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        devs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([local_rank], dtype=torch.int64, device=dev))
        rccl_seen = {"backend": dist.get_backend(), "rccl_ranks_seen": dist.get_world_size(), "devices_by_rank": [int(d.item()) for d in devs],
                     "distinct_devices": len({int(d.item()) for d in devs}),
                     "all_reduce_alone_ms": round(1e3 * float(ta.item()), 5), "all_reduce_doubles": int(ar.numel())}
        del ar

    # ---- N > 1: a scaling line that can be READ.  (1) the SAME workload (the N-piece trajectory) unsharded on ONE GPU, measured on
    # rank 0 in this run -> `one_gpu_same_workload` + `speedup` (the N = 1 line of this bench is another workload: C2, 40 pieces);
    # (2) the workload of BASELINE's configs that CAN scale - configs[2], 128 trajectories - sharded by TRAJECTORY (no exchange:
    # every trajectory's outputs stay on its rank, as in the batch optimiser), with its own one-GPU figure from rank 0.
    one_gpu = None
    c3_sharded = None
    c5_sharded = None
    if world > 1 and not args.weak:
        def plain_timed(fn, steps, warm=5):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize(); dist.barrier()
            return time.perf_counter() - t0_

        fused_was_on = mode == "fused"
        if fused_was_on:
            eng.xchg_fuse(False)
        e1g = None
        if rank == 0:
            e1g = pkg.Engine(cfg); e1g.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e1g.set_shape(shape)
            d_o1 = torch.zeros(e1g.out_stride(N), dtype=torch.float64, device=dev)
        st1 = max(20, args.steps // 4)
        el1 = plain_timed((lambda: e1g.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_o1.data_ptr(), 0, stream)) if rank == 0 else (lambda: None), st1)
        if rank == 0:
            one_gpu = {"ms_per_step": round(1e3 * el1 / st1, 5), "evals_per_s": round(N * (K + 1) * st1 / el1, 1), "steps": st1,
                       "what": f"the same {N}-piece trajectory UNSHARDED on rank 0's GPU while the other ranks wait, same run"}
        # (2) configs[2] sharded by trajectory
        n_traj_all = 128
        tb, te = par.shard_range(n_traj_all, rank, world)
        ext_ = np.array(occ.shape) * res
        Tl, Cl = [], []
        for b in range(n_traj_all if rank == 0 else 0):     # rank 0 also needs all of them for the one-GPU figure
            Tb_, Cf_ = synth.random_trajectory(ext_, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
            Tl.append(Tb_); Cl.append(synth.colmajor(Cf_))
        if rank != 0:
            for b in range(tb, te):
                Tb_, Cf_ = synth.random_trajectory(ext_, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
                Tl.append(Tb_); Cl.append(synth.colmajor(Cf_))
            Tm, Cm_ = Tl, Cl
        else:
            Tm, Cm_ = Tl[tb:te], Cl[tb:te]
        e3 = pkg.Engine(cfg); e3.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); e3.set_shape(shape)
        nb_ = te - tb
        if nb_ > 0:
            d3T = torch.from_numpy(np.concatenate(Tm)).to(dev); d3C = torch.from_numpy(np.concatenate(Cm_)).to(dev)
            d3O = torch.zeros(nb_ * e3.out_stride(40), dtype=torch.float64, device=dev)
        st3 = 20
        el3 = plain_timed((lambda: e3.eval_device(nb_, 40, d3T.data_ptr(), d3C.data_ptr(), d3O.data_ptr(), 0, stream)) if nb_ > 0 else (lambda: None), st3)
        t3 = torch.tensor([el3], dtype=torch.float64, device=dev); dist.all_reduce(t3, op=dist.ReduceOp.MAX); el3 = float(t3.item())
        if rank == 0:
            dAT = torch.from_numpy(np.concatenate(Tl)).to(dev); dAC = torch.from_numpy(np.concatenate(Cl)).to(dev)
            dAO = torch.zeros(n_traj_all * e3.out_stride(40), dtype=torch.float64, device=dev)
        el3one = plain_timed((lambda: e3.eval_device(n_traj_all, 40, dAT.data_ptr(), dAC.data_ptr(), dAO.data_ptr(), 0, stream)) if rank == 0 else (lambda: None), st3)
        if rank == 0:
            ev3 = n_traj_all * 40 * (K + 1)
            # parity of the sharded launch: rank 0's first trajectory, sharded launch vs the 128-trajectory launch (bitwise: one trajectory's
            # result does not depend on its neighbours in the batch, tests/test_gpu_fullsize.py::test_c3_batch_is_independent)
            same = bool(torch.equal(d3O[:e3.out_stride(40)], dAO[:e3.out_stride(40)]))
            c3_sharded = {"name": f"C3 (configs[2]) sharded by TRAJECTORY over {world} GPUs: 128 trajectories x 40 pieces, {nb_} per rank, no exchange (outputs stay with their trajectory's rank)",
                          "evals_per_step": ev3, "ms_per_step": round(1e3 * el3 / st3, 5), "evals_per_s": round(ev3 * st3 / el3, 1), "scaling": "strong",
                          "one_gpu_same_workload": {"ms_per_step": round(1e3 * el3one / st3, 5), "evals_per_s": round(ev3 * st3 / el3one, 1),
                                                    "what": "all 128 trajectories in one batched launch on rank 0's GPU while the other ranks wait, same run"},
                          "speedup": round(el3one / el3, 3), "sharded_equals_unsharded_bitwise_traj0": same}
        del e3
        # (3) configs[4]: the swept-volume sweep (V1) with a mesh robot through the fast winding number on a 512^3 map - the one
        # north-star workload whose step lasts milliseconds.  Obstacle points sharded contiguously over the ranks (isdf_set_shard),
        # ONE all-reduce of [cost | gradT | gradC] per step over the group, lastTstar merged once at the end (every rank writes its
        # own points' minimisers; the optimizer never reads them: set_ts = false, sw_manager.hpp:710-747).
        if not args.no_c5:
            try:
                res5 = 0.1
                occ5 = synth.random_box_map((512,) * 3, res=res5, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
                ext5 = np.array(occ5.shape) * res5
                T5, Cf5 = synth.random_trajectory(ext5, 20, seed=780, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ5, res=res5)
                cm5 = synth.colmajor(Cf5)
                way5 = cm5.reshape(3, -1).T.reshape(20, 6, 3)[1:, 0, :]
                cfg5 = synth.default_config(capi.V1_SWEPT, device=local_rank, safety_hor=(3 ** 0.5 / 2) * res5, weight_p=4000.0)
                e5 = pkg.Engine(cfg5); e5.set_grid(occ5, (0, 0, 0), res5, capi.GRID_OCCUPANCY)
                e5.set_shape(synth.make_mesh_shape(*synth.blob_mesh(24, 16, scale=0.9)))
                M5 = e5.gather_points(way5, 0.9)
                e5.set_shard(rank, world)
                d5T = torch.from_numpy(T5).to(dev); d5C = torch.from_numpy(cm5).to(dev)
                d5O = torch.zeros(e5.out_stride(20), dtype=torch.float64, device=dev)
                d5ts = torch.full((M5,), -1.0, dtype=torch.float64, device=dev)

                def c5_step():
                    e5.eval_device(1, 20, d5T.data_ptr(), d5C.data_ptr(), d5O.data_ptr(), d5ts.data_ptr(), stream)
                    par.allreduce_packed(d5O)
                st5 = 5
                el5 = plain_timed(c5_step, st5, warm=2)
                t5 = torch.tensor([el5], dtype=torch.float64, device=dev); dist.all_reduce(t5, op=dist.ReduceOp.MAX); el5 = float(t5.item())
                merged = d5ts.clone(); dist.all_reduce(merged, op=dist.ReduceOp.MAX)      # a rank's untouched entries are -1, minimisers are >= 0
                sum5 = d5O.clone()
                if rank == 0:
                    e5.set_shard(0, 1)
                    d5O1 = torch.zeros_like(d5O); d5ts1 = torch.full((M5,), -1.0, dtype=torch.float64, device=dev)
                el5one = plain_timed((lambda: e5.eval_device(1, 20, d5T.data_ptr(), d5C.data_ptr(), d5O1.data_ptr(), d5ts1.data_ptr(), stream)) if rank == 0 else (lambda: None), st5, warm=2)
                if rank == 0:
                    c5_sharded = {"name": f"C5 (configs[4]) sharded by OBSTACLE POINT over {world} GPUs: V1 swept-volume sweep, 512^3 occupancy @ 0.1 m, mesh robot (blob, 720 faces, fast winding number), 20 pieces, "
                                          f"{M5} points, one all-reduce of {int(d5O.numel())} doubles per step ({dist.get_backend()}), lastTstar merged",
                                  "variant": "V1", "obstacle_points": int(M5), "ms_per_step": round(1e3 * el5 / st5, 5), "evals_per_s": round(M5 * st5 / el5, 1), "unit": "obstacle-point evals/s",
                                  "scaling": "strong", "steps": st5,
                                  "one_gpu_same_workload": {"ms_per_step": round(1e3 * el5one / st5, 5), "evals_per_s": round(M5 * st5 / el5one, 1),
                                                            "what": "all the points unsharded on rank 0's GPU while the other ranks wait, same run"},
                                  "speedup": round(el5one / el5, 3),
                                  "sharded_vs_unsharded_rel_diff": float((sum5 - d5O1).norm().item() / max(float(d5O1.norm().item()), 1e-300)),
                                  "lastTstar_merged_equals_unsharded_bitwise": bool(torch.equal(merged, d5ts1))}
                del e5, occ5
            except Exception as e:      # noqa: BLE001
                if rank == 0:
                    c5_sharded = {"name": "C5 sharded", "error": str(e)}
        if fused_was_on:
            eng.xchg_fuse(True)

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
    if issue is not None:
        roofline = {"bound": "valu-issue", "achieved": issue["valu_issue_busy_frac"], "peak": 1.0,
                    "unit": "VALU instructions issued per SIMD-cycle (1.0 = one on every one of the 1 024 SIMDs in every cycle at 2.4 GHz)",
                    "frac": issue["valu_issue_busy_frac"], "traffic": traffic,
                    "lane_utilisation": issue["lane_utilisation"], "useful_fp64_frac": issue["useful_fp64_frac"],
                    "fp64_share_of_valu_insts": issue["fp64_share_of_valu_insts"], "waves_per_simd_avg": issue["waves_per_simd_avg"],
                    "wait_frac_of_wave_cycles": issue["wait_frac_of_wave_cycles"],
                    "note": "C2 is ONE launch of 2 600 wavefronts for 1 024 SIMDs: at ~2 waves per SIMD it is latency-floored; the same kernel reaches its issue bound on the batch (configs[0].roofline)",
                    "definition": issue["definition"], "counters": issue["source"], "counters_measured_on_commit": issue["measured_on_commit"]}
    else:
        roofline = {"bound": "valu-issue (by the committed counters of the N = 1 workload; none for this geometry / this version of the kernel sources)",
                    "achieved": None, "peak": 1.0, "unit": "VALU instructions issued per SIMD-cycle", "frac": None, "traffic": None}
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
            extra = []

            def parity_first(e, Np, Ts, Cs, dO, oc):
                """rel. error of the FIRST trajectory of the entry's last launch vs the oracle (oc = (cfg, grid, kind, res, shape));
                outside every timed region"""
                if args.no_cpu_baseline or oc is None:
                    return None
                try:
                    ocfg, ogrid, okind, ores, oshape = oc
                    oo = graft.load_oracle().Oracle(ocfg, threads=min(30, os.cpu_count() or 1))
                    oo.set_grid(ogrid, (0, 0, 0), ores, okind); oo.set_shape(oshape)
                    c0, gT0, gC0, _ = oo.eval(Ts[:Np], Cs[:18 * Np])
                    ref = np.concatenate([[c0], gT0, gC0]); got = dO[:e.out_stride(Np)].cpu().numpy()[:ref.size]
                    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))
                except Exception as ex:      # noqa: BLE001
                    return f"oracle failed: {ex}"

            def run_cfg(name, e, n_traj, Np, Ts, Cs, steps, kk=args.kernel_size, oc=None):
                dT = torch.from_numpy(Ts).to(dev); dC = torch.from_numpy(Cs).to(dev)
                dO = torch.zeros(n_traj * e.out_stride(Np), dtype=torch.float64, device=dev)
                for _ in range(5):
                    e.eval_device(n_traj, Np, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, stream)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    e.eval_device(n_traj, Np, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, stream)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
                units = n_traj * Np * (K + 1)
                nom = units * algorithmic_bytes_per_eval(kk) / dt / 1e9
                extra.append({"name": name, "evals_per_step": units, "ms_per_step": round(dt * 1e3, 5), "evals_per_s": round(units / dt, 1),
                              "nominal_bytes_gbs": round(nom, 1), "nominal_bytes_frac": round(nom / HBM_PEAK_GBS, 4),
                              "nominal_bytes_exceeds_hbm_peak": bool(nom > HBM_PEAK_GBS),
                              "rel_err_vs_oracle": parity_first(e, Np, Ts, Cs, dO, oc),
                              "rel_err_vs_oracle_of": "trajectory 0 of the launch" if n_traj > 1 else "the launch"})
                return extra[-1]

            Ts, Cs = [], []
            ext = np.array(occ.shape) * res
            for b in range(128):
                Tb, Cf = synth.random_trajectory(ext, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
                Ts.append(Tb); Cs.append(synth.colmajor(Cf))
            oc2 = (cfg, esdf, capi.GRID_ESDF, res, shape)
            run_cfg("C3 (configs[2]): 128 trajectories x 40 pieces, shared 256^3 map, one batched launch pair", eng, 128, 40, np.concatenate(Ts), np.concatenate(Cs), 20, oc=oc2)
            _, _, T4, cm4 = build_workload(pkg, 400, args.grid, res)
            run_cfg("C4 on ONE GPU (configs[3]'s trajectory unsharded): 1 x 400 pieces", eng, 1, 400, T4, cm4, 50, oc=oc2)
            # another robot of SURVEY 8(d) on the headline geometry: the 0.4x rounded cone (three branches, a square root each)
            try:
                ec = pkg.Engine(cfg)
                ec.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
                ec.set_shape(synth.bench_rounded_cone_shape())
                run_cfg("C2 geometry (1 x 40 pieces) with the rounded-cone robot (r1 0.6, r2 0.24, h 1.8)", ec, 1, 40, T, cm, 200, oc=(cfg, esdf, capi.GRID_ESDF, res, synth.bench_rounded_cone_shape()))
                del ec
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "rounded cone", "error": str(e)})
            # 512^3 map at 0.1 m (512 MiB fp32 ESDF, built on the device), the robot and the tile halved with the voxel
            try:
                res5 = 0.1
                occ5 = synth.random_box_map((512,) * 3, res=res5, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
                cfg5 = synth.default_config(capi.V3_ESDF_TILE, device=local_rank, kernel_size=21, integral_intervs=K, safety_hor=(3 ** 0.5 / 2) * res5,
                                            weight_p=4000.0, smoothing_eps=0.01, enable_dyn=1, enable_pos=1)
                e5 = pkg.Engine(cfg5)
                e5.set_grid(occ5, (0, 0, 0), res5, capi.GRID_OCCUPANCY)
                e5.generate_esdf()
                shape5 = synth.make_shape("Box", params=(0.8, 0.15, 0.15), grad_mode=capi.GRAD_CENTRAL, bound_radius=0.83, bbox=((0, 0, 0), (0.8, 0.15, 0.15)))
                e5.set_shape(shape5)
                oc5 = None if args.no_cpu_baseline else (cfg5, e5.get_grid(capi.GRID_ESDF)[0], capi.GRID_ESDF, res5, shape5)
                ext5 = np.array(occ5.shape) * res5
                Ts, Cs = [], []
                for b in range(128):
                    Tb, Cf = synth.random_trajectory(ext5, 40, seed=900 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ5, res=res5)
                    Ts.append(Tb); Cs.append(synth.colmajor(Cf))
                run_cfg("512^3 map @ 0.1 m (512 MiB fp32 ESDF built on the device), 1 x 40 pieces, robot and tile scaled with the voxel", e5, 1, 40, Ts[0], Cs[0], 100, oc=oc5)
                run_cfg("512^3 map @ 0.1 m, batch of 128 x 40 pieces", e5, 128, 40, np.concatenate(Ts), np.concatenate(Cs), 20, oc=oc5)
                del e5, oc5
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "512^3 map", "error": str(e)})

            # ---- the tile sweep with a MESH robot (720 faces through the fast-winding-number hierarchy): scan launch -> queue of
            # 16-voxel blocks -> exact launch (quad-cooperative walks) -> per-sample sums
            try:
                shm = synth.make_mesh_shape(*synth.blob_mesh(24, 16, scale=0.9))
                em = pkg.Engine(cfg); em.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); em.set_shape(shm)
                ent = run_cfg("C2 geometry (1 x 40 pieces) with a MESH robot (blob, 720 faces, fast-winding-number hierarchy)", em, 1, 40, T, cm, 10, oc=(cfg, esdf, capi.GRID_ESDF, res, shm))
                ent["roofline"] = issue_from_counters(os.path.join(ROOT, "profiles", "pmc_counters_v3mesh.json"), kernel_source_hash())
                ent["faces"] = 720
                del em
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "V3 mesh", "error": str(e)})
            # ---- the whole-tile cull on (never `value`: it skips poses, the headline evaluates every pose)
            try:
                cfgc = synth.default_config(capi.V3_ESDF_TILE, device=local_rank, kernel_size=args.kernel_size, integral_intervs=K,
                                            safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0, smoothing_eps=0.01, enable_dyn=1, enable_pos=1, enable_cull=1)
                ecu = pkg.Engine(cfgc); ecu.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); ecu.set_shape(shape)
                run_cfg("C2 with the V3 whole-tile cull ON (trilinear esdf(pos) > bound radius + safety + sqrt(3) res skips the pose)", ecu, 1, 40, T, cm, 300, oc=(cfgc, esdf, capi.GRID_ESDF, res, shape))
                extra[-1]["poses_culled"] = ecu.stats()["culled"]
                del ecu
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "cull on", "error": str(e)})

            # ---- V1: the reference's LIVE collision term (addSaftyPenaOnSweptVolumeParallel, back_end_optimizer.hpp:557-649):
            # one unit = one obstacle point (argmin over t + back-prop).  Device resident like `value`; the oracle's OpenMP sweep
            # of the same points at 30 threads beside it.
            def run_v1(name, e1, Np, Tv, Cv, steps, oracle=None, note=None):
                Mv = e1.stats_points if hasattr(e1, "stats_points") else None
                dT = torch.from_numpy(Tv).to(dev); dC = torch.from_numpy(Cv).to(dev)
                dO = torch.zeros(e1.out_stride(Np), dtype=torch.float64, device=dev)
                for _ in range(3):
                    e1.eval_device(1, Np, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, stream)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    e1.eval_device(1, Np, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, stream)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
                st1 = e1.stats()
                Mv = st1["units"]
                # kernel split, separate pass: the search launches (scan + descent, or the single fused search) / what follows
                e1.profile_enable(1, secondary=True)
                for _ in range(min(steps, 10)):
                    e1.eval_device(1, Np, dT.data_ptr(), dC.data_ptr(), dO.data_ptr(), 0, stream)
                torch.cuda.synchronize()
                _, k_search = e1.profile_read(); k_rest = e1.profile_read_secondary()
                e1.profile_enable(0)
                ent = {"name": name, "variant": "V1", "obstacle_points": Mv, "pieces": Np, "ms_per_step": round(dt * 1e3, 5),
                       "evals_per_s": round(Mv / dt, 1), "unit": "obstacle-point evals/s",
                       "kernel_ms_search_launches": round(k_search, 5), "kernel_ms_backprop_reduce": round(k_rest, 5),
                       "active_points": st1["grad_pairs"], "intervals": st1["pairs"]}
                if note:
                    ent["note"] = note
                if oracle is not None:
                    o1, thr = oracle
                    c0, gT0, gC0, _ = o1.eval(Tv, Cv)
                    ref1 = np.concatenate([[c0], gT0, gC0]); got1 = dO.cpu().numpy()
                    ent["rel_err_vs_oracle"] = float(np.linalg.norm(got1 - ref1) / max(np.linalg.norm(ref1), 1e-300))
                    nrep = 0; tc0 = time.perf_counter()
                    while True:
                        o1.eval(Tv, Cv); nrep += 1
                        if time.perf_counter() - tc0 > 3.0:
                            break
                    dc = (time.perf_counter() - tc0) / nrep
                    ent["cpu_baseline_v1"] = {"value": round(Mv / dc, 1), "unit": "obstacle-point evals/s", "cores": thr, "kind": "port",
                                              "ms_per_step": round(dc * 1e3, 3),
                                              "sample": f"{nrep} full V1 sweeps of the same {Mv} points, oracle OpenMP schedule(dynamic) + critical-section merge like the reference"}
                extra.append(ent)
                return ent

            if not args.no_cpu_baseline:
                orc_v1 = graft.load_oracle()
            try:
                way40 = cm.reshape(3, -1).T.reshape(40, 6, 3)[1:, 0, :]
                for nm, shp in (("rounded cone (r1 0.6, r2 0.24, h 1.8)", synth.bench_rounded_cone_shape()), ("box (1.6, 0.3, 0.3)", shape)):
                    cfg1 = synth.default_config(capi.V1_SWEPT, device=local_rank, safety_hor=(3 ** 0.5 / 2) * res, weight_p=4000.0)
                    e1 = pkg.Engine(cfg1); e1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); e1.set_shape(shp)
                    e1.gather_points(way40, 1.4)
                    oracle = None
                    if not args.no_cpu_baseline:
                        thr = min(30, os.cpu_count() or 1)
                        o1 = orc_v1.Oracle(cfg1, threads=thr); o1.set_grid(occ, (0, 0, 0), res, capi.GRID_OCCUPANCY); o1.set_shape(shp); o1.set_points(e1.get_points())
                        oracle = (o1, thr)
                    ent = run_v1(f"V1 swept-volume sweep (the reference's live collision term), 40 pieces / 40 s, 256^3 occupancy, {nm}", e1, 40, T, cm, 50, oracle)
                    if nm.startswith("rounded"):
                        ent["roofline_issue"] = issue_from_counters(os.path.join(ROOT, "profiles", "pmc_counters_v1.json"), v1_source_hash(), live_kernel_us=ent["kernel_ms_search_launches"] * 1e3)
                        ent["bound"] = "valu-issue / dependent-instruction latency (SURVEY 8(d): 24 B point + trajectory per workgroup - no HBM roof); see roofline_issue"
                    del e1
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "V1 analytic", "error": str(e)})
            # configs[4]-shaped: V1 on the 512^3 occupancy map (0.1 m), concave 20-face mesh and a 720-face mesh through the
            # fast-winding-number path, obstacle points gathered on the device along a 20-piece trajectory
            try:
                res5 = 0.1
                occ5 = synth.random_box_map((512,) * 3, res=res5, occupancy=0.15, seed=12345, edge=(0.4, 2.0))
                ext5 = np.array(occ5.shape) * res5
                T5, Cf5 = synth.random_trajectory(ext5, 20, seed=780, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ5, res=res5)
                cm5 = synth.colmajor(Cf5)
                way5 = cm5.reshape(3, -1).T.reshape(20, 6, 3)[1:, 0, :]
                # SURVEY 8(d) names the reference's own Lthick.obj (concave, 20 faces) and Trefoil.obj (12 000 faces), "scaled to the same
                # bound" (tests/golden/ref_meshes.npz: isdf_read_obj's view of src/plan_manager/shapes/*.obj); the synthetic L prism and
                # the 720-face blob are the round-3 entries, kept for comparison
                meshes = [("concave L prism, 20 faces (synthetic)", synth.l_prism_mesh(), 20, "all"),
                          ("blob, 720 faces (synthetic)", synth.blob_mesh(24, 16, scale=0.9), 5, "all")]
                try:
                    meshes += [("the reference's Lthick.obj, 20 faces, scaled to a 0.83 m bound", synth.reference_mesh("Lthick", 0.83), 20, "all"),
                               ("the reference's Trefoil.obj, 12 000 faces, scaled to a 0.83 m bound", synth.reference_mesh("Trefoil", 0.83), 3, "all")]
                except Exception as e:      # noqa: BLE001
                    extra.append({"name": "reference meshes", "error": str(e)})
                for nm, (Vm, Fm), stp, omode in meshes:
                    cfg1 = synth.default_config(capi.V1_SWEPT, device=local_rank, safety_hor=(3 ** 0.5 / 2) * res5, weight_p=4000.0)
                    shp = synth.make_mesh_shape(Vm, Fm)
                    e1 = pkg.Engine(cfg1); e1.set_grid(occ5, (0, 0, 0), res5, capi.GRID_OCCUPANCY); e1.set_shape(shp)
                    e1.gather_points(way5, 0.9)
                    oracle = None
                    o1 = None
                    if not args.no_cpu_baseline:
                        thr = min(30, os.cpu_count() or 1)
                        o1 = orc_v1.Oracle(cfg1, threads=thr); o1.set_grid(occ5, (0, 0, 0), res5, capi.GRID_OCCUPANCY); o1.set_shape(shp)
                        if o1.mesh_backend() != 1:
                            o1 = None
                        elif omode == "all":
                            o1.set_points(e1.get_points()); oracle = (o1, thr)
                    ent = run_v1(f"C5 (configs[4]) shape: V1 sweep, 512^3 occupancy @ 0.1 m, mesh robot ({nm}) through the fast winding number, 20 pieces", e1, 20, T5, cm5, stp, oracle,
                                 note="both sides search their own minimisers; the device's winding number is the reference library's bit for bit (csrc/dev_mesh.hpp), so t* is identical and rel_err_vs_oracle is rounding")
                    ent["faces"] = int(len(Fm))
                    if o1 is not None and omode == "sample":
                        # the oracle's exhaustive closest-point search over 12 000 faces: a SAMPLE of the obstacle points, both sides on the same subset
                        ps = np.ascontiguousarray(e1.get_points()[::40])
                        e1.set_points(ps); o1.set_points(ps)
                        cs, gTs, gCs = e1.eval_single(T5, cm5)
                        c0, gT0, gC0, _ = o1.eval(T5, cm5)
                        ref1 = np.concatenate([[c0], gT0, gC0]); got1 = np.concatenate([[cs], gTs, gCs])
                        ent["rel_err_vs_oracle"] = float(np.linalg.norm(got1 - ref1) / max(np.linalg.norm(ref1), 1e-300))
                        ent["rel_err_vs_oracle_of"] = f"every 40th obstacle point ({len(ps)} points), both sides on that subset"
                    if "720" in nm:
                        ent["roofline_issue"] = issue_from_counters(os.path.join(ROOT, "profiles", "pmc_counters_v1mesh.json"), v1_source_hash())
                        if ent["roofline_issue"]:
                            ent["roofline_issue"]["workload_of_the_counters"] = "tools/v1_bench.py --mesh720: the same 720-face robot on the 256^3 bench map, 7 773 points"
                    del e1
                del occ5
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "C5 V1 mesh", "error": str(e)})
            # ---- the ESDF point sampler (GridMap3D::getSDFValue / getSDFValueWithGrad, A13): the one pure gather of the path -
            # measured against the HBM peak by its algorithmic bytes (24 B point + 8 x 4 B corners + 8 B value + 24 B gradient)
            try:
                npt = 20_000_000
                gen = torch.Generator(device=dev); gen.manual_seed(7)
                ext_m = args.grid * res
                m_ = npt // 4096
                a_ = torch.rand(4096, 1, 3, dtype=torch.float64, device=dev, generator=gen) * ext_m
                d_ = torch.nn.functional.normalize(torch.randn(4096, 1, 3, dtype=torch.float64, device=dev, generator=gen), dim=2)
                rnd_ = torch.rand(npt, 3, dtype=torch.float64, device=dev, generator=gen) * ext_m
                sets = (("trajectory-like points (4 096 straight lines, consecutive points 2 cm apart)", (a_ + d_ * (torch.arange(m_, dtype=torch.float64, device=dev).view(1, m_, 1) * 0.02)).reshape(-1, 3).contiguous(), False),
                        ("uniform random points", rnd_, False),
                        ("uniform random points through the BRICKED copy (isdf_esdf_sample_scattered_device: 2 x 2 x 2-cell bricks with apron, one 128-byte line per cell neighbourhood; bitwise the same results)", rnd_, True))
                plain_rnd = None
                for nm_, pts_, scat_ in sets:
                    n_ = pts_.shape[0]
                    dv_ = torch.empty(n_, dtype=torch.float64, device=dev); dg_ = torch.empty(n_, 3, dtype=torch.float64, device=dev)
                    for _ in range(2):
                        eng.esdf_sample_device(pts_.data_ptr(), n_, dv_.data_ptr(), dg_.data_ptr(), stream, scattered=scat_)
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    for _ in range(5):
                        eng.esdf_sample_device(pts_.data_ptr(), n_, dv_.data_ptr(), dg_.data_ptr(), stream, scattered=scat_)
                    ev1.record(); torch.cuda.synchronize()
                    ms_ = ev0.elapsed_time(ev1) / 5
                    ent = {"name": f"ESDF point sampler isdf_esdf_sample_device (value + gradient), {args.grid}^3 float32 ESDF, {n_:.3g} {nm_}",
                           "points": n_, "ms_per_launch": round(ms_, 4), "points_per_s": round(n_ / ms_ * 1e3, 1), "unit": "ESDF samples/s",
                           "roofline": {"bound": "hbm", "achieved": round(n_ * 88 / ms_ / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(n_ * 88 / ms_ / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_point": 88,
                                        "definition": "24 B point + 8 corner values x 4 B + 8 B value + 24 B gradient per point / launch duration (HIP events)"}}
                    if not args.no_cpu_baseline:
                        oo_ = graft.load_oracle().Oracle(cfg); oo_.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF)
                        ph = pts_[:2000].cpu().numpy()
                        rr_ = np.array([np.concatenate([[dd], gg]) for dd, gg in (oo_.esdf_trilinear(q_) for q_ in ph)])
                        got_ = np.concatenate([dv_[:2000].cpu().numpy()[:, None], dg_[:2000].cpu().numpy()], axis=1)
                        ent["rel_err_vs_oracle"] = float(np.linalg.norm(got_ - rr_) / max(np.linalg.norm(rr_), 1e-300))
                        ent["rel_err_vs_oracle_of"] = "the first 2 000 points (bitwise equal = 0.0)"
                    if nm_.startswith("uniform random points") and not scat_:
                        plain_rnd = (dv_.clone(), dg_.clone())
                    if scat_ and plain_rnd is not None:
                        ent["equals_plain_path_bitwise"] = bool(torch.equal(dv_, plain_rnd[0]) and torch.equal(dg_, plain_rnd[1]))
                    extra.append(ent)
                    del dv_, dg_
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "ESDF sampler", "error": str(e)})
            # configs[2] END TO END: the 128 trajectories optimised concurrently (isdf_optimize_lbfgs_batch: host MINCO + L-BFGS per
            # trajectory, one batched device sweep per round, PCIe included) - 30 iterations each
            try:
                heads, tails, x0b = [], [], []
                for b in range(128):
                    Tb, Cfb = synth.random_trajectory(ext, 40, seed=777 + b, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
                    c6b = Cfb.reshape(40, 6, 3)
                    hb = np.stack([c6b[0, 0], c6b[0, 1], 2.0 * c6b[0, 2]], axis=1)
                    tlb = Tb[-1]
                    pwb = lambda k: sum(np.prod([p - q for q in range(k)]) * c6b[39, p] * tlb ** (p - k) for p in range(k, 6))
                    tb = np.stack([pwb(0), pwb(1), pwb(2)], axis=1)
                    eng.set_trajectory(40, hb, tb, 1000.0)
                    x0b.append(eng.pack_variables(Tb, c6b[1:, 0, :])); heads.append(hb); tails.append(tb)
                eng.set_shard(0, 1)
                ent_b = None
                for mode_b, label_b in ((capi.MINCO_AUTO, "device"), (capi.MINCO_HOST, "host")):
                    eng.set_minco_mode(mode_b)
                    best = None
                    for _rep in range(2):             # (the first run of a mode also pays its buffers' first touch)
                        _, resb, wallb = eng.optimize_lbfgs_batch(40, np.array(heads), np.array(tails), 1000.0, np.array(x0b), max_iterations=30, g_epsilon=0.0, past=0)
                        if best is None or wallb < best[1]:
                            best = (resb, wallb)
                    resb, wallb = best
                    evb = sum(r["evaluations"] for r in resb); itb = sum(r["iterations"] for r in resb)
                    rec = {"wall_ms": round(wallb, 2), "device_rounds": resb[0]["rounds"], "iterations": itb, "evaluations": evb,
                           "evals_per_s": round(1e3 * evb * 40 * (K + 1) / wallb, 1), "trajectory_iterations_per_s": round(1e3 * itb / wallb, 1)}
                    if ent_b is None:
                        ent_b = {"name": "C3 END TO END (configs[2]): 128 trajectories x 40 pieces optimised concurrently, 30 L-BFGS iterations each (isdf_optimize_lbfgs_batch: the callback's MINCO half on the device, L-BFGS on host threads, PCIe included; best of 2 runs)", "minco": label_b}
                        ent_b.update(rec)
                    else:
                        ent_b["host_minco"] = rec
                eng.set_minco_mode(capi.MINCO_AUTO)
                extra.append(ent_b)
            except Exception as e:      # noqa: BLE001
                extra.append({"name": "C3 end to end", "error": str(e)})
            if extra and isinstance(extra[0], dict) and "error" not in extra[0]:
                extra[0]["roofline"] = issue_from_counters(os.path.join(ROOT, "profiles", "pmc_counters_c3.json"), kernel_source_hash())
            # the nominal byte model, spelled out where it stops being a bound
            roofline["nominal_bytes"]["exceeds_1_on"] = [e_["name"] for e_ in extra if isinstance(e_, dict) and e_.get("nominal_bytes_exceeds_hbm_peak")]

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
        # full objective callback (host MINCO + the device sweeps), same map / robot / piece count.  Informational.
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
                                   + ("" if world == 1 else ".  NOTE: the --gpus 1 line of this bench is ANOTHER workload (configs[1]: 40 pieces, 2 600 evals per step): "
                                      "value(N) / value(1) is not a scaling efficiency - use `speedup` (= value / one_gpu_same_workload.evals_per_s, same run) "
                                      "and configs[0] (configs[2] sharded by trajectory, with its own one-GPU figure)"),
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
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
