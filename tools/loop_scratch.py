#!/usr/bin/env python3
"""Scratch (spill) traffic of a kernel per LOOP, outer loops included, from a hipcc -S device listing (development tool, round 6):
   tools/loop_scratch.py listing.s <mangled kernel name substring>
For every loop header: the loop's depth, its instruction count, its VALU count and the scratch loads / stores between its header and its
back edge, split into "own" (not inside a child loop) and "with children".  tools/loop_mix.py shows the innermost loops only; the objective
backward's per-row reloads (Weak 2) sit in the row loop AROUND its innermost loop."""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
for chunk in s.split('.end_amdhsa_kernel'):
    heads = [l.split(':')[0] for l in chunk.split('\n') if l.startswith('_ZN') and ':' in l and pat in l.split(':')[0]]
    if not heads:
        continue
    name = heads[-1]
    body = chunk[chunk.index('\n' + name + ':'):].split('\n')
    print(name)
    loops = []          # (label, depth, start line)
    for n, l in enumerate(body):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:      # "=>This Loop Header: Depth=1" sits on the label's line, or (inner loops) on the comment line after "Parent Loop ..."
            d = re.search(r'Loop Header: Depth=(\d+)', l + " " + (body[n + 1] if n + 1 < len(body) and body[n + 1].lstrip().startswith(';') else ""))
            if d:
                loops.append((m.group(1), int(d.group(1)), n))
    spans = []
    for lab, depth, start in loops:
        end = start
        for n in range(start + 1, len(body)):
            t = body[n].strip()
            if t.startswith(('s_cbranch', 's_branch')) and t.split()[-1] == lab:
                end = n      # the LAST backward branch to the header closes the loop
        spans.append((lab, depth, start, end))

    def count(a, b, skip=()):
        tot = valu = ld = st = 0
        for n in range(a, b + 1):
            if any(x <= n <= y for x, y in skip):
                continue
            t = body[n].strip().split()
            if not t or t[0].startswith(('.', ';')):
                continue
            tot += 1
            valu += t[0].startswith('v_')
            ld += t[0].startswith('scratch_load')
            st += t[0].startswith('scratch_store')
        return tot, valu, ld, st

    for lab, depth, a, b in spans:
        kids = [(x, y) for l2, d2, x, y in spans if d2 == depth + 1 and a < x and y <= b]
        own, total = count(a, b, kids), count(a, b)
        print(f"  {lab:12s} depth {depth}  own: {own[0]:5d} instr {own[1]:5d} VALU  scratch ld/st {own[2]:3d}/{own[3]:3d}   |  with children: {total[0]:5d} instr  scratch ld/st {total[2]:3d}/{total[3]:3d}")
    all_ld = sum(l.strip().startswith('scratch_load') for l in body)
    all_st = sum(l.strip().startswith('scratch_store') for l in body)
    in_ld = sum(count(a, b)[2] for _, d, a, b in spans if d == 1)
    in_st = sum(count(a, b)[3] for _, d, a, b in spans if d == 1)
    print(f"  outside every loop: scratch ld/st {all_ld - in_ld}/{all_st - in_st}")
