"""The ONE line bench.py prints: numbers only, <= 4 KB (the driver keeps an 8 KB tail of stdout; round 5's 21 KB line cost that round its
record).  The long form - every entry with its prose - goes to gpurun_out/bench_detail.json (bench.py writes it before printing)."""
import json

MAX_LINE = 4096

# (short name, predicate on the long entry's name)
_SHORT = (
    ("c3_batch128", lambda n: n.startswith("C3 (configs[2]):")),
    ("c3_sharded", lambda n: n.startswith("C3 (configs[2]) sharded")),
    ("c4_one_gpu", lambda n: n.startswith("C4 on ONE GPU")),
    ("c2_cone", lambda n: "rounded-cone robot" in n),
    ("c2_512", lambda n: n.startswith("512^3 map") and "1 x 40" in n),
    ("c3_512", lambda n: n.startswith("512^3 map") and "batch of 128" in n),
    ("c2_mesh720", lambda n: "MESH robot (blob, 720" in n),
    ("c2_cull", lambda n: "whole-tile cull ON" in n),
    ("v1_cone", lambda n: n.startswith("V1 swept-volume sweep") and "rounded cone" in n),
    ("v1_box", lambda n: n.startswith("V1 swept-volume sweep") and "box" in n),
    ("c5_sharded", lambda n: n.startswith("C5 (configs[4]) sharded")),
    ("c5_L20", lambda n: n.startswith("C5 (configs[4])") and "concave L prism" in n),
    ("c5_blob720", lambda n: n.startswith("C5 (configs[4])") and "blob, 720" in n),
    ("c5_lthick", lambda n: n.startswith("C5 (configs[4])") and "Lthick.obj" in n),
    ("c5_trefoil", lambda n: n.startswith("C5 (configs[4])") and "Trefoil.obj" in n),
    ("esdf_traj", lambda n: n.startswith("ESDF point sampler") and "trajectory-like" in n),
    ("esdf_rand_brick", lambda n: n.startswith("ESDF point sampler") and "BRICKED" in n),
    ("esdf_rand", lambda n: n.startswith("ESDF point sampler") and "uniform random" in n),
    ("c3_end_to_end", lambda n: n.startswith("C3 END TO END")),
)


def short_name(name):
    for s, pred in _SHORT:
        if pred(name):
            return s
    return "".join(ch if ch.isalnum() else "_" for ch in name)[:24]


def _sig(x, digits=4):
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{digits}g}")
    except Exception:       # noqa: BLE001
        return None


def _entry(e):
    """{name, ms_per_step, evals_per_s, rel_err_vs_oracle} (+ split / speedup where the entry has them)"""
    out = {"name": short_name(e.get("name", "?"))}
    if "error" in e:
        out["error"] = str(e["error"])[:60]
        return out
    ms = e.get("ms_per_step", e.get("ms_per_launch", e.get("wall_ms")))
    out["ms_per_step"] = _sig(ms, 5)
    out["evals_per_s"] = _sig(e.get("evals_per_s", e.get("points_per_s")), 4)
    r = e.get("rel_err_vs_oracle")
    out["rel_err_vs_oracle"] = _sig(r, 2) if not isinstance(r, str) else "oracle failed"
    sp = e.get("eight_way_split")
    if isinstance(sp, dict):
        out["split8"] = _sig(sp.get("whole_step_over_slowest_rank"), 3)
    if e.get("speedup") is not None:
        out["speedup"] = _sig(e["speedup"], 4)
    if isinstance(e.get("one_gpu_same_workload"), dict):
        out["one_gpu_evals_per_s"] = _sig(e["one_gpu_same_workload"].get("evals_per_s"), 4)
    return out


def compact_line(full):
    """full: bench.py's long record.  Returns the dict of the printed line; json.dumps of it is < MAX_LINE bytes (entries of
    `configs` are dropped from the end, never truncated mid-way, if a future entry list outgrows the budget)."""
    rf = full.get("roofline") or {}
    nb = rf.get("nominal_bytes") or {}
    iss = full.get("roofline_issue") or {}
    cfg = dict(full.get("config") or {})
    grp = cfg.get("group")
    if isinstance(grp, dict):
        grp = {k: grp.get(k) for k in ("backend", "ranks_seen", "distinct_devices", "all_reduce_alone_ms", "all_reduce_doubles")}
    config = {"workload": str(cfg.get("workload", ""))[:300]}
    for k in ("pieces", "samples_per_piece", "grid", "kernel_size", "rccl_all_reduce_ms_per_step"):
        if cfg.get(k) is not None:
            config[k] = cfg[k]
    for k in ("sharding", "collective"):
        if cfg.get(k) is not None:
            config[k] = str(cfg[k])[:48]
    if grp is not None:
        config["group"] = grp
    roofline = {
        # SURVEY 8(d) / BASELINE.md: algorithmic bytes per eval x evals per launch / kernel duration / HBM peak
        "bound": "hbm", "achieved": nb.get("achieved_gbs", rf.get("achieved")), "peak": 8000.0, "unit": "GB/s",
        "frac": nb.get("frac_of_hbm_peak", rf.get("frac")),
        "traffic": rf.get("traffic"),
        "bytes_per_eval": nb.get("algorithmic_bytes_per_eval"), "bytes_per_launch": nb.get("algorithmic_bytes_per_launch"),
        "compulsory_bytes_per_launch": nb.get("compulsory_bytes_per_launch"),
        "kernel": rf.get("kernel"), "kernel_ms": rf.get("kernel_ms"), "launches": rf.get("launches"),
        "hbm_frac_measured": rf.get("hbm_frac_measured"),
        "issue_frac": iss.get("valu_issue_busy_frac"), "lane_utilisation": iss.get("lane_utilisation"),
        "waves_per_simd": iss.get("waves_per_simd_avg"), "fp64_frac_of_peak": iss.get("fp64_frac_of_peak"),
    }
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        cpu = {k: (str(v)[:120] if k == "sample" else v) for k, v in cpu.items() if k in ("value", "unit", "cores", "kind", "sample", "ms_per_step")}
    lb = full.get("lbfgs")
    if isinstance(lb, dict):
        lb = {"ms_per_iteration": lb.get("ms_per_iteration"), "ms_per_evaluation": lb.get("ms_per_evaluation"), "iterations": lb.get("iterations"),
              "callback_ms_native": lb.get("callback_ms_native"), "minco": (str(lb.get("minco") or "?").split() or ["?"])[0]}
    vs = full.get("value_steady")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line.update({"config": config, "roofline": roofline, "cpu_baseline": cpu,
                 "parity_rel_err_vs_oracle": _sig(full.get("parity_rel_err_vs_oracle"), 3),
                 "value_steady": None if not isinstance(vs, dict) else vs.get("value"),
                 "host_api_ms": full.get("host_api_ms"), "lbfgs": lb, "hbm_copy_gbs_measured": full.get("hbm_copy_gbs_measured")})
    for k in ("c2_issue_frac", "c3_issue_frac", "v3mesh_issue_frac", "v1_issue_frac", "v1mesh_issue_frac"):
        line[k] = full.get(k)
    if full.get("one_gpu_same_workload") is not None:
        o = full["one_gpu_same_workload"]
        line["one_gpu_same_workload"] = {"ms_per_step": o.get("ms_per_step"), "evals_per_s": o.get("evals_per_s")}
        line["speedup"] = full.get("speedup")
    line["detail"] = full.get("detail_file")
    entries = [_entry(e) for e in (full.get("configs") or []) if isinstance(e, dict)]
    line["configs"] = entries
    while entries and len(json.dumps(line)) >= MAX_LINE:
        entries.pop()
        line["configs_dropped"] = line.get("configs_dropped", 0) + 1
    if len(json.dumps(line)) >= MAX_LINE:        # cannot happen with the keys above; never print an unparseable line
        line["config"]["workload"] = line["config"]["workload"][:120]
        if isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"].pop("sample", None)
    return line
