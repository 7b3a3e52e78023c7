"""Roofline side of the bench line: spec peaks, the hashes that tie committed PMC counters to the kernel sources they were measured on, the
issue-slot model (what bounds these kernels) and SURVEY 8(d)'s byte model (kept as a nominal figure)."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
