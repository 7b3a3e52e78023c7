#!/usr/bin/env python
"""Developer tool: instruction-class breakdown of one kernel's gfx950 ISA, per top-level segment (the code before the first loop,
each outermost loop, the code between / after them).  usage: isa_breakdown.py <file.s> <mangled kernel name>
(make the .s with: hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S tile_sweep.hip -o file.s)"""
import re, sys
from collections import Counter


def classify(op, line):
    if op.startswith('v_'):
        if op.startswith(('v_readlane', 'v_writelane')): return 'v_readlane / v_writelane (SGPR spills)'
        if op.startswith('v_readfirstlane'): return 'v_readfirstlane'
        if op.startswith(('v_mov', 'v_accvgpr')): return 'v_mov (incl. DPP moves)'
        if op.startswith('v_cndmask'): return 'v_cndmask'
        if op.startswith('v_cmp'): return 'v_cmp'
        if op.startswith('v_cvt'): return 'v_cvt'
        if 'permlane' in op: return 'v_permlane swaps'
        if '_f64' in op: return 'fp64 arithmetic'
        if '_f32' in op: return 'fp32 arithmetic'
        return 'integer / logic VALU'
    if op.startswith('s_'):
        if op.startswith(('s_waitcnt', 's_nop')): return 's_waitcnt / s_nop'
        if op.startswith(('s_load', 's_buffer')): return 'scalar loads'
        if op.startswith(('s_cbranch', 's_branch')): return 'branches'
        return 'SALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')): return 'vector memory'
    return None


def main():
    txt = open(sys.argv[1]).read()
    name = sys.argv[2]
    m = re.search(r'\n' + re.escape(name) + r':(.*?)\.Lfunc_end', txt, re.S)
    body = m.group(1).splitlines()
    # outermost loops: "Loop Header: Depth=1"
    heads = [i for i, l in enumerate(body) if 'Loop Header: Depth=1' in l]
    bounds = [0] + heads + [len(body)]
    print(f"{name}: {len(body)} lines of ISA; outermost loops start at lines {heads}")
    for a, b in zip(bounds[:-1], bounds[1:]):
        c = Counter()
        for l in body[a:b]:
            p = l.split()
            if not p or p[0].startswith(('.', ';')) or p[0].endswith(':'):
                continue
            k = classify(p[0], l)
            if k:
                c[k] += 1
        valu = sum(v for k, v in c.items() if k.startswith(('v_', 'fp', 'integer')))
        f64 = c['fp64 arithmetic']
        tag = 'loop' if a in heads else 'code'
        print(f"\n  lines {a}-{b} ({tag}): {sum(c.values())} instructions, {valu} VALU of which fp64 arithmetic {f64} ({100.0 * f64 / max(valu, 1):.0f} %)")
        for k, v in c.most_common():
            print(f"      {k:42s} {v}")


if __name__ == '__main__':
    main()
