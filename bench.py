#!/usr/bin/env python
"""bench.py — headline benchmark of the collision cost/gradient hot path on MI355X.

Metric (BASELINE.json): SDF cost+grad point-evals/sec on a 256^3 map.  A "step" is one pass of the V3 ESDF-tile sweep
(include/isdf_accel.h: isdf_eval_device) over every constraint sample of the trajectory; one "point-eval" is one sampled
pose (piece i, sample j) incl. its kernel_size^3 ESDF tile.  Inputs are resident in HBM before the timed
region; results stay on the device (the PCIe-inclusive host-API rate is reported separately as `host_api_ms`, never as
`value`).

Workload at 1 GPU = BASELINE.json configs[1]: one trajectory, 40 pieces x 64 intervals (65 samples / piece), 256^3 fp32 ESDF
at 0.2 m, kernel_size 21, convex box robot (half extents 1.6/0.3/0.3 m, central-difference gradient).
At N GPUs (weak scaling): one trajectory of 40*N pieces, pieces sharded contiguously across ranks, one all-reduce (RCCL)
of the packed [cost | gradT | gradC] vector per step.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N>1 via torch.distributed.run (one rank per GPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pieces-per-gpu", type=int, default=40)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--kernel-size", type=int, default=21)
    ap.add_argument("--cull", type=int, default=0, help="1: enable the V3 whole-tile cull (reported separately, never as value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lbfgs-iters", type=int, default=40, help="iterations of the L-BFGS driver timed after the sweep (0: skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
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
    n_pieces = args.pieces_per_gpu * world
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
    if world > 1:
        xchg = par.XgmiExchange(eng, eng.out_stride(N), dev)
        if not xchg.ok:
            if rank == 0:
                print(f"[bench] peer-to-peer exchange not used ({xchg.why}); RCCL all_reduce", file=sys.stderr, flush=True)
            xchg = None

    def eval_only():
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, stream)

    # Preferred: the exchange INSIDE the step's one launch (isdf_xchg_fuse: the owning workgroups store their rows into the
    # peers' outputs) - used when it reproduces "step + the group's all-reduce" on every rank.
    in_kernel = bool(xchg is not None and os.environ.get("ISDF_NO_FUSED_EXCHANGE") != "1" and xchg.try_fused_step(eval_only, d_out))

    def make_step(mode):       # "fused": exchange inside the launch | "xchg": one-shot peer-to-peer kernel | "rccl"
        def step():
            eval_only()
            if world > 1 and mode != "fused":
                if mode == "xchg":
                    xchg.allreduce(d_out)
                else:
                    par.allreduce_packed(d_out)
        return step

    def timed(step):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        eng.profile_enable(8)      # HIP events around the dominant kernel on every 8th step of the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # ... and one pair of HIP events around the whole timed region on the launch stream (torch's current stream IS the
        # stream handed to isdf_eval_device): steps are serialised on it, so region / steps bounds the kernel's duration from
        # above without touching any dispatch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        n_l, k_ms = eng.profile_read()
        eng.profile_enable(0)
        return t1 - t0, n_l, (k_ms, e0.elapsed_time(e1) / args.steps)

    def any_rank(flag):
        t = torch.tensor([int(flag)], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item()) != 0

    mode = "none" if world == 1 else ("fused" if in_kernel else ("xchg" if xchg else "rccl"))
    names = {"none": "none", "fused": "in-kernel peer-to-peer exchange over xGMI (rows stored into the peers' outputs by the owning workgroups of the fused launch), verified against RCCL at start-up",
             "xchg": "xgmi peer-to-peer one-shot sum (csrc/xchg.hip), verified against RCCL at start-up", "rccl": "RCCL all_reduce"}
    elapsed, n_launch, kern_ms = timed(make_step(mode))
    # a bounded wait that timed out anywhere invalidates the run: measure again one level down
    if mode == "fused" and any_rank(eng.stats()["overflow"] != 0 or not np.isfinite(float(d_out[0].item()))):
        eng.xchg_fuse(False)
        mode = "xchg"
        elapsed, n_launch, kern_ms = timed(make_step(mode))
        names["xchg"] += " (the in-kernel exchange timed out and was abandoned)"
    if mode == "xchg" and any_rank(eng.xchg_status()):
        mode = "rccl"
        names["rccl"] = "RCCL all_reduce (the peer-to-peer exchange timed out and was abandoned)"
        elapsed, n_launch, kern_ms = timed(make_step(mode))
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

    # roofline of the dominant kernel (sweep_kernel): algorithmic bytes of the samples THIS rank's launch processed
    B = algorithmic_bytes_per_eval(args.kernel_size)
    evals_this_rank = (N // world) * (K + 1)
    # duration of the dominant kernel: two live measurements, both upper bounds - (a) start/stop events attached to the
    # dispatch itself on every 8th step (an instrumented dispatch runs 1-3 us longer than a plain one: its mean exceeds even
    # the step period when the step is a single launch), (b) the region events / steps (kernel + inter-launch gap; exact up
    # to that gap when the step is ONE launch, i.e. the fused step).  The smaller of the two is used; rocprofv3's average of
    # the same command is committed under profiles/.
    kern_ms_dispatch, kern_ms_region = kern_ms
    kern_ms = min(kern_ms_dispatch, kern_ms_region) if kern_ms_dispatch > 0 else kern_ms_region
    achieved = (evals_this_rank * B) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    # measured HBM traffic of the same kernel (bytes per launch): from the committed rocprofv3 PMC passes of this very
    # command (profiles/hbm_traffic.json, written by tools/pmc_traffic.py; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world == 1 and os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("pieces") == N and tj.get("kernel_size") == args.kernel_size and tj.get("grid") == args.grid:
                traffic = tj.get("sweep_kernel_bytes_per_launch")
        except Exception:
            traffic = None
    # SURVEY.md 8(d) asks for the COMPULSORY lower bound next to the algorithmic figure: the unique voxels that the tiles of
    # this rank's launch touch (neighbouring samples' tiles overlap heavily) x 4 B, plus the per-sample inputs / outputs
    compulsory = None
    if world == 1:
        compulsory = compulsory_bytes(T, cm, N, K, args.kernel_size, args.grid, res)
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "sweep_kernel", "kernel_ms": round(kern_ms, 5), "launches": n_launch,
                "kernel_ms_dispatch_events": round(kern_ms_dispatch, 5), "kernel_ms_region_events": round(kern_ms_region, 5),
                "algorithmic_bytes_per_eval": B, "algorithmic_bytes_per_launch": evals_this_rank * B,
                "compulsory_bytes_per_launch": compulsory,
                # secondary figure of SURVEY 8(d): robot-SDF evaluations per second inside the tiles (every pair costs one
                # evaluation, every pair with a non-zero penalty six more for the central-difference gradient)
                "pair_sdf_evals_per_s": round((stats["pairs"] + 6 * stats["grad_pairs"]) / (ms_per_step * 1e-3), 1)}

    result = None
    if rank == 0:
        # host-API (PCIe-inclusive, synchronous, accumulate) latency — informational
        host_ms = None
        if world == 1:
            eng.set_shard(0, 1)
            eng.eval_single(T, cm)
            th0 = time.perf_counter()
            for _ in range(20):
                eng.eval_single(T, cm)
            host_ms = 1e3 * (time.perf_counter() - th0) / 20

        # second half of BASELINE.json's metric: "L-BFGS iter ms" - the optimizer driver (isdf_optimize_lbfgs) on the
        # full objective callback (host MINCO + the device sweeps), same map / robot / piece count.  Informational.
        lbfgs = None
        if world == 1 and args.lbfgs_iters > 0:
            c6 = cm.reshape(3, 6 * N).T.reshape(N, 6, 3)
            head = np.stack([c6[0, 0], c6[0, 1], 2.0 * c6[0, 2]], axis=1)
            tl = T[-1]
            pw = lambda k: sum(np.prod([p - q for q in range(k)]) * c6[N - 1, p] * tl ** (p - k) for p in range(k, 6))
            tail = np.stack([pw(0), pw(1), pw(2)], axis=1)
            eng.set_trajectory(N, head, tail, 1000.0)
            x0 = eng.pack_variables(T, c6[1:, 0, :])
            c_first = eng.cost_function(x0)[0]
            xo, r = eng.optimize_lbfgs(x0, max_iterations=args.lbfgs_iters, g_epsilon=0.0, past=0)
            lbfgs = {"iterations": r["iterations"], "evaluations": r["evaluations"], "status": r["status"],
                     "ms_per_iteration": round(r["wall_ms"] / max(1, r["iterations"]), 4),
                     "ms_per_evaluation": round(r["wall_ms"] / max(1, r["evaluations"]), 4),
                     "cost_first": c_first, "cost_last": r["f"]}

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
        result = {
            "metric": "SDF cost+grad point-evals/sec", "value": round(value, 1), "unit": "point-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"V3 ESDF-tile sweep: 1 trajectory x {N} pieces x {K + 1} samples/piece, {args.grid}^3 fp32 ESDF "
                                   f"@ {res:g} m, kernel_size {args.kernel_size}, box robot (1.6,0.3,0.3) central-diff grad, "
                                   f"cull {'on' if args.cull else 'off'}",
                       "pieces": N, "samples_per_piece": K + 1, "grid": args.grid, "kernel_size": args.kernel_size,
                       "sharding": ("none" if world == 1 else ("pieces contiguous by rank; each piece's 19 rows + cost stored by its owner into every peer's output inside the step's launch" if mode == "fused" else "pieces contiguous by rank + all_reduce(sum) of 1+19N doubles")),
                       "collective": collective},
            "roofline": roofline, "cpu_baseline": cpu,
            "parity_rel_err_vs_oracle": parity, "host_api_ms": None if host_ms is None else round(host_ms, 4),
            "stats_last_step_rank0": stats, "lbfgs": lbfgs, "hbm_copy_gbs_measured": hbm_copy,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
