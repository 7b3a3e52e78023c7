#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) run: per-kernel launch count / mean / total duration,
and per-kernel mean of a PMC counter when one was collected.  Usage: rocprof_summary.py <results.db> [counter]"""
import sqlite3, sys

def main():
    db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    tot = sum(r[3] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>10s} {'pct':>6s}")
    for n, c, a, s, mn, mx in rows:
        print(f"{n[:70]:70s} {c:7d} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {s/1e6:10.3f} {100*s/tot:6.1f}")
    if len(sys.argv) > 2:
        ctr = sys.argv[2]
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        print("\ncounter table columns:", ccols)
        try:
            q = f"select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection where counter_name like '%{ctr}%' group by kernel_name, counter_name order by sum(value) desc"
            for r in cur.execute(q).fetchall():
                print(f"{r[0][:70]:70s} {r[1]:14s} n={r[2]:6d} avg={r[3]:.4g} sum={r[4]:.4g}")
        except Exception as e:
            print("counter query failed:", e)

if __name__ == "__main__":
    main()
