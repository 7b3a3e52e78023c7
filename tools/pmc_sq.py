#!/usr/bin/env python
"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass -> text table.  usage: pmc_sq.py <results.db> <out.txt>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
cmd = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --lbfgs-iters 0"
ctrs = " ".join(sorted({r[1] for r in rows}))
lines = [f"# rocprofv3 --kernel-trace --pmc {ctrs} -- {cmd}",
         "# per-dispatch means; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY count quad-cycles (MI355X_MICROARCH.md)"]
for k, c, n, a in rows:
    if "isdf::" in k:
        lines.append(f"{k[:60]:60s} {c:20s} n={n:5d} avg={a:14.1f}")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
