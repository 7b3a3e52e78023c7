#!/usr/bin/env python
"""Where do a kernel's scratch (spill) instructions sit?  Compiles a .hip translation unit with line tables, reads the ISA and attributes every
scratch_load / scratch_store of the kernels matching <filter> to the source line it was emitted for (runs without a GPU).
usage: tools/scratch_by_source.py swept_sweep.hip 'Li15ELb0ELi1' > profiles/rN_mesh_scratch_by_source.txt"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, flt = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-gline-tables-only", "-save-temps",
                    "-c", os.path.join(ROOT, "implicit-sdf-planner_amd", "csrc", src), "-o", os.path.join(td, "x.o")], cwd=td, capture_output=True)
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(td, asm)).read().split("\n")
files, cur, fn = {}, None, None
res = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for ln in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"^(_ZN4isdf\w+):", ln)
    if m:
        fn = m.group(1); continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    if fn and flt in fn and re.match(r"^\s+[a-z_0-9]+ ", ln):
        tot[fn] += 1
        if "scratch_" in ln:
            res[fn][(cur[0] if cur else "?", "load" if "scratch_load" in ln else "store")] += 1
names = subprocess.run(["c++filt"], input="\n".join(res), capture_output=True, text=True).stdout.splitlines()
print(f"# {src}: scratch instructions of the kernels matching {flt!r}, by the source FILE of the line they were emitted for")
print("# (dev_mesh.hpp holds the hierarchy walks - the hot loops; dev_math.hpp / swept_sweep.hip the per-pass pose arithmetic and bookkeeping around them)")
for fn, nm in zip(res, names):
    print(f"\n{nm}: {tot[fn]} instructions, {sum(res[fn].values())} of them scratch")
    byfile = collections.Counter()
    for (f, k), n in res[fn].items():
        byfile[(f, k)] += n
    for (f, k), n in sorted(byfile.items()):
        print(f"    {f:28s} {k:6s} {n:4d}")
