"""The other single-GPU configurations of the bench line's `configs` array (world == 1): C3, C4 on one GPU, other robots, 512^3, mesh
robots on both sweeps, the whole-tile cull, V1 (the reference's live term) with its CPU baseline, configs[4]-shaped V1 x mesh, the
ESDF point sampler, configs[2] end to end.  Every entry carries its own rel_err_vs_oracle (outside every timed region)."""
import os
import time

import numpy as np

from .roofline import HBM_PEAK_GBS, ROOT, algorithmic_bytes_per_eval, issue_from_counters, kernel_source_hash, v1_source_hash
from .meshes import reference_mesh
from .workloads import build_workload


def extra_configs(B):
    """B: the bench's state (see bench.py main()).  Returns the list of entries; fills roofline['nominal_bytes']['exceeds_1_on']."""
    import torch
    args, graft, pkg, capi, synth = B.args, B.graft, B.pkg, B.capi, B.synth
    dev, stream, local_rank = B.dev, B.stream, B.local_rank
    eng, cfg, shape, occ, esdf, res, T, cm, K, roofline = B.eng, B.cfg, B.shape, B.occ, B.esdf, B.res, B.T, B.cm, B.K, B.roofline
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
    ent4 = run_cfg("C4 on ONE GPU (configs[3]'s trajectory unsharded): 1 x 400 pieces", eng, 1, 400, T4, cm4, 50, oc=oc2)
    # what an 8-GPU step of it can reach at best: every rank's share (50 pieces) by itself on this GPU - the exchange comes on top
    try:
        d4T = torch.from_numpy(T4).to(dev); d4C = torch.from_numpy(cm4).to(dev)
        d4O = torch.zeros(eng.out_stride(400), dtype=torch.float64, device=dev)
        per_rank = []
        for r8 in range(8):
            eng.set_shard(r8, 8)
            for _ in range(5):
                eng.eval_device(1, 400, d4T.data_ptr(), d4C.data_ptr(), d4O.data_ptr(), 0, stream)
            torch.cuda.synchronize(); t8 = time.perf_counter()
            for _ in range(50):
                eng.eval_device(1, 400, d4T.data_ptr(), d4C.data_ptr(), d4O.data_ptr(), 0, stream)
            torch.cuda.synchronize(); per_rank.append(round((time.perf_counter() - t8) / 50 * 1e3, 5))
        eng.set_shard(0, 1)
        ent4["eight_way_split"] = {"ms_per_step_of_each_rank_measured_alone": per_rank, "slowest_rank_ms": max(per_rank),
                                   "whole_step_over_slowest_rank": round(ent4["ms_per_step"] / max(per_rank), 3),
                                   "note": "a rank's 50 pieces are 3 250 wavefronts for 1 024 SIMDs: like C2, a latency-floored launch; the exchange of 7 601 doubles is not in these figures"}
    except Exception as e8:      # noqa: BLE001
        eng.set_shard(0, 1)
        ent4["eight_way_split"] = {"error": str(e8)}
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
            meshes += [("the reference's Lthick.obj, 20 faces, scaled to a 0.83 m bound", reference_mesh("Lthick", 0.83), 20, "all"),
                       ("the reference's Trefoil.obj, 12 000 faces, scaled to a 0.83 m bound", reference_mesh("Trefoil", 0.83), 3, "all")]
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
            if "720" in nm or "Trefoil" in nm:
                # What the ranks of an 8-GPU run would each execute, measured one after the other on this GPU: rank r's contiguous
                # eighth of the obstacle points (isdf_set_shard).  The step of the real run is its slowest rank + one all-reduce.
                try:
                    dT8 = torch.from_numpy(T5).to(dev); dC8 = torch.from_numpy(cm5).to(dev)
                    dO8 = torch.zeros(e1.out_stride(20), dtype=torch.float64, device=dev)
                    per_rank = []
                    for r8 in range(8):
                        e1.set_shard(r8, 8)
                        for _ in range(3):       # (the dispatch orders are the previous shard's at first)
                            e1.eval_device(1, 20, dT8.data_ptr(), dC8.data_ptr(), dO8.data_ptr(), 0, stream)
                        ts8 = []
                        for _ in range(5):       # every step by itself, the MEDIAN reported (one hiccup in a three-step region had read as 4.2 ms for a 2.4 ms step)
                            torch.cuda.synchronize(); t8 = time.perf_counter()
                            e1.eval_device(1, 20, dT8.data_ptr(), dC8.data_ptr(), dO8.data_ptr(), 0, stream)
                            torch.cuda.synchronize(); ts8.append(time.perf_counter() - t8)
                        per_rank.append(round(float(np.median(ts8)) * 1e3, 4))
                    e1.set_shard(0, 1)
                    ent["eight_way_split"] = {"ms_per_step_of_each_rank_measured_alone": per_rank, "slowest_rank_ms": max(per_rank),
                                              "whole_step_over_slowest_rank": round(ent["ms_per_step"] / max(per_rank), 3),
                                              "note": "a rank's 2 076 points are one generation of workgroups: its step is its longest chains (descents deep inside the robot), "
                                                      "which the whole step hides behind seven more generations of work - the bound on this configuration's strong scaling (DESIGN.md 5)"}
                except Exception as e8:      # noqa: BLE001
                    ent["eight_way_split"] = {"error": str(e8)}
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
    # configs[2] END TO END: the 128 trajectories optimised concurrently (isdf_optimize_lbfgs_batch: L-BFGS on a host thread per
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
    return extra
