#!/usr/bin/env python3
"""Instruction mix of the innermost loops of one kernel in a hipcc -S device listing (development tool):
   tools/loop_mix.py listing.s <mangled kernel name substring>"""
import collections, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for chunk in s.split('.end_amdhsa_kernel'):
    heads = [l.split(':')[0] + ':' for l in chunk.split('\n') if l.startswith('_ZN') and ':' in l and pat in l.split(':')[0]]
    if not heads:
        continue
    name = heads[-1].rstrip(':')
    body = chunk[chunk.index('\n' + name + ':'):].split('\n')
    print(name)
    n = 0
    while n < len(body):
        l = body[n]
        if l.startswith('.LBB') and 'Parent Loop' in l:
            lab = l.split(':')[0].strip()
            m = n + 1
            while m < len(body) and not (body[m].strip().startswith('s_cbranch') and lab in body[m]):
                m += 1
            c = collections.Counter()
            for x in body[n:m + 1]:
                t = x.strip().split()
                if t and not t[0].startswith('.') and not t[0].startswith(';'):
                    c[t[0]] += 1
            print(' ', lab, 'total', sum(c.values()), 'VALU', sum(v for k, v in c.items() if k.startswith('v_')), 'scratch', sum(v for k, v in c.items() if k.startswith('scratch')))
            print('    ' + ', '.join(f'{k} {v}' for k, v in c.most_common(24)))
            n = m
        n += 1
