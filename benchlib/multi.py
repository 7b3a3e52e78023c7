"""N > 1 entries of the bench line: what the collective library saw, the SAME workload unsharded on one GPU (`one_gpu_same_workload`),
configs[2] sharded by trajectory and configs[4] sharded by obstacle point, each with its own one-GPU figure from rank 0 of the same run."""
import time

import numpy as np


def scaling_entries(B):
    """B: the bench's state (bench.py main()); collective calls inside - every rank calls it.  Returns (group facts, one_gpu, c3_sharded, c5_sharded)."""
    import torch
    import torch.distributed as dist
    args, pkg, capi, synth, par = B.args, B.pkg, B.capi, B.synth, B.par
    world, rank, local_rank, dev, stream = B.world, B.rank, B.local_rank, B.dev, B.stream
    eng, cfg, shape, occ, esdf, res, K, N, d_T, d_C, mode = B.eng, B.cfg, B.shape, B.occ, B.esdf, B.res, B.K, B.N, B.d_T, B.d_C, B.mode
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
        rccl_seen = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "devices_by_rank": [int(d.item()) for d in devs],
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
                # a step lasts milliseconds: every step timed by itself (barrier, step, synchronise; MAX over the ranks per step), the
                # MEDIAN reported - one scheduler hiccup in a five-step region had read as a 40x slowdown in a rehearsal
                def stepwise(fn, steps, warm=2):
                    for _ in range(warm):
                        fn()
                    ts_ = []
                    for _ in range(steps):
                        torch.cuda.synchronize(); dist.barrier()
                        t0_ = time.perf_counter()
                        fn()
                        torch.cuda.synchronize()
                        ts_.append(time.perf_counter() - t0_)
                    tt_ = torch.tensor(ts_, dtype=torch.float64, device=dev)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    return tt_.cpu().numpy()
                st5 = 7
                per5 = stepwise(c5_step, st5)
                el5 = float(np.median(per5)) * st5
                merged = d5ts.clone(); dist.all_reduce(merged, op=dist.ReduceOp.MAX)      # a rank's untouched entries are -1, minimisers are >= 0
                sum5 = d5O.clone()
                if rank == 0:
                    e5.set_shard(0, 1)
                    d5O1 = torch.zeros_like(d5O); d5ts1 = torch.full((M5,), -1.0, dtype=torch.float64, device=dev)
                per5one = stepwise((lambda: e5.eval_device(1, 20, d5T.data_ptr(), d5C.data_ptr(), d5O1.data_ptr(), d5ts1.data_ptr(), stream)) if rank == 0 else (lambda: None), st5)
                el5one = float(np.median(per5one)) * st5
                if rank == 0:
                    c5_sharded = {"name": f"C5 (configs[4]) sharded by OBSTACLE POINT over {world} GPUs: V1 swept-volume sweep, 512^3 occupancy @ 0.1 m, mesh robot (blob, 720 faces, fast winding number), 20 pieces, "
                                          f"{M5} points, one all-reduce of {int(d5O.numel())} doubles per step ({dist.get_backend()}), lastTstar merged",
                                  "variant": "V1", "obstacle_points": int(M5), "ms_per_step": round(1e3 * el5 / st5, 5), "evals_per_s": round(M5 * st5 / el5, 1), "unit": "obstacle-point evals/s",
                                  "scaling": "strong", "steps": st5, "ms_per_step_is": "median of the steps (each timed by itself, MAX over the ranks)",
                                  "ms_per_step_min_max": [round(1e3 * float(per5.min()), 5), round(1e3 * float(per5.max()), 5)],
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
    return rccl_seen, one_gpu, c3_sharded, c5_sharded
