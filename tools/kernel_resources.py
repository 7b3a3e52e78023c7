#!/usr/bin/env python
"""Register / scratch / LDS / occupancy table of every kernel of a .hip translation unit (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiled for gfx950: runs without a GPU).  usage: tools/kernel_resources.py swept_sweep.hip [filter] > profiles/rN_resources_*.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "implicit-sdf-planner_amd", "csrc")


def table(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage",
           *extra, "-c", os.path.join(CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:[^ ]+ )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?)(?: \[-Rpass|$)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}; rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(isdf::\w+(, .*)?\)$", "", re.sub(r"^void isdf::", "", o)) for o in out]


if __name__ == "__main__":
    src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = table(src)
    names = demangle([r["name"] for r in rows])
    print(f"# {src}: hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage")
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>14s} {'LDS B':>7s} {'waves/SIMD':>10s}")
    for r, n in zip(rows, names):
        if flt and not re.search(flt, n):
            continue
        print(f"{n:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('ScratchSize','?'):>14s} {r.get('LDS Size','?'):>7s} {r.get('Occupancy','?'):>10s}")
