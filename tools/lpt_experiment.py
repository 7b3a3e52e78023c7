#!/usr/bin/env python
"""Developer experiment: does a balanced dispatch order of the tile sweep's samples shorten the single-trajectory step?
Measures the step with the identity order, then with orders built from the PREVIOUS step's per-sample work (row-scan cycles,
candidate counts from ISDF_DEBUG_TIMING), dealt so that every workgroup gets one sample of each quartile.
Needs the statistics build: `make -C implicit-sdf-planner_amd/csrc EXTRA=-DISDF_SWEEP_STATS=1` after touching the kernel sources."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); capi, synth = pkg.capi, pkg.synth
from bench import build_workload


def step_us(eng, N, d_T, d_C, d_out, st, n=300):
    for _ in range(20):
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    shape_name = sys.argv[1] if len(sys.argv) > 1 else "box"
    res = 0.2
    occ, esdf, T, cm = build_workload(pkg, 40, 256, res)
    shape = synth.bench_box_shape() if shape_name == "box" else synth.bench_rounded_cone_shape()
    cfg = synth.default_config(capi.V3_ESDF_TILE, kernel_size=21, integral_intervs=64, safety_hor=(3 ** 0.5 / 2) * res,
                               weight_p=4000.0, smoothing_eps=0.01)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    N = T.size
    d_T = torch.from_numpy(T).to(dev); d_C = torch.from_numpy(cm).to(dev)
    st = torch.cuda.current_stream().cuda_stream

    # pass 1: per-sample work of one step (timing build of the same kernel)
    os.environ["ISDF_DEBUG_TIMING"] = "1"
    eng = pkg.Engine(cfg)
    eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    d_out = torch.zeros(eng.out_stride(N), dtype=torch.float64, device=dev)
    for _ in range(3):
        eng.eval_device(1, N, d_T.data_ptr(), d_C.data_ptr(), d_out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    lib = eng.lib
    lib.isdf_debug_timing.restype = C.c_longlong
    lib.isdf_debug_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    n = lib.isdf_debug_timing(eng.h, None, 0)
    buf = np.zeros(n, dtype=np.uint64)
    lib.isdf_debug_timing(eng.h, buf.ctypes.data_as(C.c_void_p), n)
    ns = N * 65
    sc = buf[:ns * 8].reshape(ns, 8).astype(np.int64)
    rows = np.maximum(sc[:, 4] - sc[:, 2], 0)
    cand = (sc[:, 3] >> 44) & 0xFFFFF
    rounds = sc[:, 3] & 255
    passes = (sc[:, 3] >> 8) & 255
    ref = d_out.cpu().numpy().copy()
    del eng
    os.environ["ISDF_DEBUG_TIMING"] = "0"

    eng = pkg.Engine(cfg)
    eng.set_grid(esdf, (0, 0, 0), res, capi.GRID_ESDF); eng.set_shape(shape)
    lib = eng.lib
    lib.isdf_debug_set_sample_map.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    nb = (ns + 3) // 4
    print(f"{shape_name}: identity order            {step_us(eng, N, d_T, d_C, d_out, st):7.2f} us/step")

    # the order the device builds by itself (plan_piece), then the same order fed back as a fixed table (no building)
    lib.isdf_debug_plan_map.restype = C.c_int
    lib.isdf_debug_plan_map.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    dm = np.full(4 * nb, -7, dtype=np.int32)
    kk = lib.isdf_debug_plan_map(eng.h, dm.ctypes.data_as(C.c_void_p), dm.size)
    if kk >= 3:
        lib.isdf_debug_set_sample_map(eng.h, dm.ctypes.data_as(C.c_void_p), dm.size)
        print(f"{shape_name}: the device-built order as a fixed table {step_us(eng, N, d_T, d_C, d_out, st):7.2f} us/step")
        ident = np.arange(4 * nb, dtype=np.int32); ident[ns:] = -1
        lib.isdf_debug_set_sample_map(eng.h, ident.ctypes.data_as(C.c_void_p), ident.size)
        print(f"{shape_name}: plain order as a fixed table            {step_us(eng, N, d_T, d_C, d_out, st):7.2f} us/step")

    def deal(order, snake):
        """order: sample ids, heaviest first.  rank r of the launch = workgroup r % nb, wave r // nb."""
        m = np.zeros(ns, dtype=np.int32)
        for w in range(4):
            seg = order[w * nb:(w + 1) * nb]
            if snake and (w & 1):
                seg = seg[::-1]
            m[w * nb: w * nb + seg.size] = seg
        return m

    keys = {"rows cycles": rows, "candidates": cand, "rounds*4+passes": rounds * 4 + passes, "rows + 50*cand": rows + 50 * cand,
            "chunks": (cand + 63) // 64}
    for name, key in keys.items():
        order = np.argsort(-key, kind="stable").astype(np.int32)
        for snake in (False, True):
            m = deal(order, snake)
            assert np.array_equal(np.sort(m), np.arange(ns))
            lib.isdf_debug_set_sample_map(eng.h, m.ctypes.data_as(C.c_void_p), ns)
            t = step_us(eng, N, d_T, d_C, d_out, st)
            same = np.array_equal(d_out.cpu().numpy(), ref)
            print(f"{shape_name}: by {name:18s} {'snake' if snake else 'plain'}  {t:7.2f} us/step   bitwise same: {same}")
    # identity THROUGH the table: the cost of the lookup alone
    m = np.arange(ns, dtype=np.int32)
    lib.isdf_debug_set_sample_map(eng.h, m.ctypes.data_as(C.c_void_p), ns)
    print(f"{shape_name}: identity through the table {step_us(eng, N, d_T, d_C, d_out, st):7.2f} us/step")
    # sorted within groups of 8 consecutive workgroups' samples only (keeps neighbours together)
    for G in (2, 8, 32):
        key = cand
        m = np.arange(ns, dtype=np.int32).reshape(4, nb) if ns == 4 * nb else None
        if m is not None:
            m = m.copy()
            for g0 in range(0, nb, G):
                blk = m[:, g0:g0 + G]                  # the 4 x G samples of G neighbouring workgroups
                ids = blk.reshape(-1)
                o = ids[np.argsort(-key[ids], kind="stable")]
                g = blk.shape[1]
                for w in range(4):
                    seg = o[w * g:(w + 1) * g]
                    blk[w, :] = seg[::-1] if (w & 1) else seg
            mm = m.reshape(-1).astype(np.int32)
            assert np.array_equal(np.sort(mm), np.arange(ns))
            lib.isdf_debug_set_sample_map(eng.h, mm.ctypes.data_as(C.c_void_p), ns)
            print(f"{shape_name}: balanced within {G:3d} neighbouring workgroups {step_us(eng, N, d_T, d_C, d_out, st):7.2f} us/step")
    # the workgroups beyond 2 per CU (index >= 512) get the lightest samples; trajectory order kept inside both groups
    for name, key in (("candidates", cand), ("chunks", (cand + 63) // 64), ("rows + 50*cand", rows + 50 * cand)):
        for nfull in (512, 256):
            n3 = nb - nfull
            if n3 <= 0:
                continue
            o = np.argsort(key, kind="stable")
            light = np.sort(o[:4 * n3]); heavy = np.sort(o[4 * n3:])
            m = np.zeros(ns, dtype=np.int32)
            for w in range(4):
                m[w * nb: w * nb + nfull] = heavy[w * nfull:(w + 1) * nfull]
                m[w * nb + nfull: (w + 1) * nb] = light[w * n3:(w + 1) * n3]
            assert np.array_equal(np.sort(m), np.arange(ns))
            lib.isdf_debug_set_sample_map(eng.h, m.ctypes.data_as(C.c_void_p), ns)
            t = step_us(eng, N, d_T, d_C, d_out, st)
            print(f"{shape_name}: lightest {4 * n3} samples (by {name}) to workgroups >= {nfull}, order kept: {t:7.2f} us/step  bitwise same: {np.array_equal(d_out.cpu().numpy(), ref)}")
    chunks = (cand + 63) // 64
    def zone_map(pos_order, sizes):
        """pos_order: sample ids in zone-filling order; sizes: workgroups per zone; inside a zone position q -> wave q // size, workgroup q % size"""
        m = np.full(4 * nb, -1, dtype=np.int64)
        p0 = 0; boff = 0
        for sz in sizes:
            ids = pos_order[p0:p0 + 4 * sz]
            for w in range(4):
                seg = ids[w * sz:(w + 1) * sz]
                m[w * nb + boff: w * nb + boff + seg.size] = seg
            p0 += 4 * sz; boff += sz
        return m
    def run(label, m):
        mm = m[:ns].astype(np.int32)
        if not np.array_equal(np.sort(mm), np.arange(ns)):
            print("  (not a permutation of the first ns ranks, skipped)", label); return
        lib.isdf_debug_set_sample_map(eng.h, mm.ctypes.data_as(C.c_void_p), ns)
        t = step_us(eng, N, d_T, d_C, d_out, st)
        print(f"{shape_name}: {label}: {t:7.2f} us/step  bitwise same: {np.array_equal(d_out.cpu().numpy(), ref)}")
    for name, key in (("chunks", np.minimum(chunks, 7)), ("min(chunks,3)", np.minimum(chunks, 3)), ("pairs/32", np.minimum((cand + 31) // 32, 7))):
        st_sorted = np.argsort(-key, kind="stable")          # classes descending, trajectory order inside a class
        run(f"stable sort by {name}, zones (256, rest)", zone_map(st_sorted, (256, nb - 256)))
        run(f"stable sort by {name}, zones (256, 256, rest)", zone_map(st_sorted, (256, 256, nb - 512)))
        run(f"stable sort by {name}, one zone", zone_map(st_sorted, (nb,)))
        # heaviest 1024 in trajectory order, rest in trajectory order (the earlier winner)
        o = np.argsort(key, kind="stable"); n3 = nb - 256
        order = np.concatenate([np.sort(o[4 * n3:]), np.sort(o[:4 * n3])])
        run(f"heaviest 1024 by {name} (order kept) -> workgroups < 256", zone_map(order, (256, n3)))
    # per-CU model: workgroup b -> XCD b % 8, there CU (b // 8) % 32; CUs that hold 3 workgroups should give each of them 2/3 of
    # the load a workgroup gets on a CU that holds 2
    import heapq
    for fixed in (1.0, 2.0, 3.0):
        for key_name, key in (("pairs/32", np.minimum((cand + 31) // 32, 7)), ("chunks", np.minimum(chunks, 7))):
            wgt = fixed + key.astype(float)
            per_xcd = [len(range(x, nb, 8)) for x in range(8)]
            on3 = np.zeros(nb, dtype=bool)
            for b in range(nb):
                x, j = b % 8, b // 8
                n_cu3 = per_xcd[x] - 64                     # CUs of this XCD that get a third workgroup
                on3[b] = (j % 32) < n_cu3
            L = wgt.sum() / (on3.sum() / 3.0 + (~on3).sum() / 2.0)
            target = np.where(on3, L / 3.0, L / 2.0)
            load = np.zeros(nb); cnt = np.zeros(nb, dtype=int)
            slots = [[] for _ in range(nb)]
            heap = [(-(target[b]), b) for b in range(nb)]
            heapq.heapify(heap)
            for sidx_ in np.argsort(-wgt, kind="stable"):
                while True:
                    d, b = heapq.heappop(heap)
                    if cnt[b] < 4 and abs(-d - (target[b] - load[b])) < 1e-9:
                        break
                    if cnt[b] < 4:
                        heapq.heappush(heap, (-(target[b] - load[b]), b))
                slots[b].append(int(sidx_)); cnt[b] += 1; load[b] += wgt[sidx_]
                if cnt[b] < 4:
                    heapq.heappush(heap, (-(target[b] - load[b]), b))
            m = np.full(4 * nb, -1, dtype=np.int64)
            for b in range(nb):
                for w, sidx_ in enumerate(slots[b]):
                    m[w * nb + b] = sidx_
            if np.count_nonzero(m[:ns] < 0):
                # the launch needs its empty ranks at the end: move samples from the tail ranks into the holes
                holes = np.where(m[:ns] < 0)[0]; extra = np.where(m[ns:] >= 0)[0] + ns
                m[holes] = m[extra]; m[extra] = -1
            run(f"per-CU model, weight {fixed} + {key_name}", m)
    # like weights per workgroup inside the zones (what the two-launch steps use), against the strided zones of the product
    key = np.minimum((cand + 31) // 32, 7)
    st_sorted = np.argsort(-key, kind="stable")
    def grouped(ids, b0, sz):
        """ids -> workgroups b0 .. b0+sz-1, four neighbours each"""
        out = {}
        for q, sidx_ in enumerate(ids):
            out[(q % 4) * nb + b0 + q // 4] = sidx_
        return out
    for label, z0_grouped, rest_grouped in (("zone 0 strided, rest grouped", False, True), ("zone 0 grouped, rest strided", True, False), ("both grouped", True, True)):
        m = np.full(4 * nb, -1, dtype=np.int64)
        z0 = st_sorted[:1024]; rest = st_sorted[1024:]
        if z0_grouped:
            for r, v in grouped(z0, 0, 256).items(): m[r] = v
        else:
            for w in range(4): m[w * nb: w * nb + 256] = z0[w * 256:(w + 1) * 256]
        n3 = nb - 256
        if rest_grouped:
            for r, v in grouped(rest, 256, n3).items(): m[r] = v
        else:
            for w in range(4):
                seg = rest[w * n3:(w + 1) * n3]; m[w * nb + 256: w * nb + 256 + seg.size] = seg
        run(label, m)
    # random order as a control
    rng = np.random.default_rng(1)
    m = rng.permutation(ns).astype(np.int32)
    lib.isdf_debug_set_sample_map(eng.h, m.ctypes.data_as(C.c_void_p), ns)
    print(f"{shape_name}: random order              {step_us(eng, N, d_T, d_C, d_out, st):7.2f} us/step")


if __name__ == "__main__":
    main()
