#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`python bench.py --steps S --warmup W --no-cpu-baseline` into profiles/hbm_traffic.json + a text summary.
usage: pmc_traffic.py <fetch.db> <write.db> <out.json> <out.txt> [pieces kernel_size grid]"""
import json, sqlite3, sys


def per_kernel(db, ctr):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, count(*), avg(value), sum(value) from counters_collection where counter_name = ? "
         "group by kernel_name order by sum(value) desc")
    return cur.execute(q, (ctr,)).fetchall()


def main():
    fdb, wdb, out_json, out_txt = sys.argv[1:5]
    pieces, ksz, grid = (int(x) for x in (sys.argv[5:8] if len(sys.argv) >= 8 else (40, 21, 256)))
    lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --no-cpu-baseline",
             "# units: KB per dispatch as reported; gfx950 correction (MI355X_MICROARCH.md, HBM): wide coalesced reads are "
             "under-reported 2x -> traffic = 2*FETCH_SIZE + WRITE_SIZE"]
    vals = {}
    for db, ctr in ((fdb, "FETCH_SIZE"), (wdb, "WRITE_SIZE")):
        for name, n, avg, tot in per_kernel(db, ctr):
            lines.append(f"{name[:70]:70s} {ctr:12s} n={n:6d} avg_KB={avg:12.2f} sum_KB={tot:14.1f}")
            vals.setdefault(name, {})[ctr] = avg
    sweep = [k for k in vals if "sweep_kernel" in k]
    res = {"pieces": pieces, "kernel_size": ksz, "grid": grid, "unit": "bytes per launch",
           "formula": "2*FETCH_SIZE + WRITE_SIZE (KB -> bytes)"}
    if sweep:
        v = vals[sweep[0]]
        res["sweep_kernel_bytes_per_launch"] = int(round((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024))
        res["sweep_kernel_fetch_kb"] = v.get("FETCH_SIZE")
        res["sweep_kernel_write_kb"] = v.get("WRITE_SIZE")
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
    with open(out_txt, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
