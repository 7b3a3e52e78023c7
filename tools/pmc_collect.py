#!/usr/bin/env python
"""Merges rocprofv3 passes into ONE json for the dominant kernel(s) of a step: average duration from the kernel trace, per-dispatch
means of every PMC counter, and a hash of the kernels' sources - bench.py uses the counters only while that hash matches the tree it
runs from.  When several kernels are named (the V1 search = scan + descent, one dispatch each per step) durations and counters are
SUMMED over them, i.e. the figures are per step.
usage: pmc_collect.py <out.json> <kernel_trace.db> <pmc.db> [<pmc.db> ...]
env:   PMC_KERNELS  comma-separated substrings of the kernel names to collect (default: sweep_kernel; first match per substring)
       PMC_HASH     sweep | v1: which source hash to stamp (bench.kernel_source_hash / bench.v1_source_hash)"""
import json, os, sqlite3, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash, v1_source_hash  # noqa: E402


def main():
    out, kt, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    subs = [x for x in os.environ.get("PMC_KERNELS", "sweep_kernel").split(",") if x]
    cur = sqlite3.connect(kt).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), avg(end-start), sum(end-start) from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    doms = []
    for sub in subs:
        m = [r for r in rows if sub in r[0]]
        if m:
            doms.append(m[0])
    names = [d[0] for d in doms]
    res = {"kernel": names if len(names) > 1 else names[0], "launches_traced": [d[1] for d in doms], "avg_kernel_us": sum(d[2] for d in doms) / 1e3,
           "per_kernel_avg_us": {d[0][:80]: d[2] / 1e3 for d in doms}, "counters": {}, "per_kernel_counters": {}, "n_dispatches": {},
           "source_hash": v1_source_hash() if os.environ.get("PMC_HASH") == "v1" else kernel_source_hash(),
           "units": "per-dispatch means (summed over the listed kernels); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles, "
           "SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles summed over the 8 XCDs, FETCH_SIZE / WRITE_SIZE KB"}
    try:
        res["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        res["commit"] = None
    seen = {}                # counter -> index of the pass it was first collected in (a counter repeated in a later pass is kept apart)
    for di, db in enumerate(dbs):
        if not db or not os.path.exists(db):
            continue
        c = sqlite3.connect(db).cursor()
        for k, ctr, n, a in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            if k in names:
                key = ctr if seen.setdefault(ctr, di) == di else f"{ctr}@pass{di}"
                res["counters"][key] = res["counters"].get(key, 0.0) + a
                res["per_kernel_counters"].setdefault(k[:80], {})[key] = a
                res["n_dispatches"][key] = n
    # lane utilisation of the VALU instructions: both counters from the SAME pass (the last one that holds SQ_THREAD_CYCLES_VALU)
    tc = res["counters"].get("SQ_THREAD_CYCLES_VALU")
    if tc:
        di = seen["SQ_THREAD_CYCLES_VALU"]
        act = res["counters"].get("SQ_ACTIVE_INST_VALU" if seen.get("SQ_ACTIVE_INST_VALU") == di else f"SQ_ACTIVE_INST_VALU@pass{di}")
        if act:
            res["lane_utilisation_valu"] = tc / (64.0 * act)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
