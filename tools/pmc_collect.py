#!/usr/bin/env python
"""Merges the rocprofv3 passes of tools/profile_round.sh into ONE json for the step's dominant kernel (the fused
sweep_kernel of the default bench): average duration from the kernel trace, per-dispatch means of every PMC counter, and a
hash of the kernel's sources - bench.py uses the counters only while that hash matches the tree it runs from.
usage: pmc_collect.py <out.json> <kernel_trace.db> <pmc.db> [<pmc.db> ...]"""
import json, os, sqlite3, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def main():
    out, kt, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    cur = sqlite3.connect(kt).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), avg(end-start), sum(end-start) from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    dom = [r for r in rows if "sweep_kernel" in r[0]][0]
    res = {"kernel": dom[0], "launches_traced": dom[1], "avg_kernel_us": dom[2] / 1e3, "counters": {}, "n_dispatches": {},
           "source_hash": kernel_source_hash(), "units": "per-dispatch means; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles, "
           "SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles summed over the 8 XCDs, FETCH_SIZE / WRITE_SIZE KB"}
    try:
        res["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        res["commit"] = None
    for db in dbs:
        if not db or not os.path.exists(db):
            continue
        c = sqlite3.connect(db).cursor()
        for k, ctr, n, a in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            if k == dom[0]:
                res["counters"][ctr] = a; res["n_dispatches"][ctr] = n
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
