#!/usr/bin/env python3
"""Instruction-class trace of one kernel from a hipcc -save-temps .s file (L = global load, S = global store, M = MFMA,
D = LDS, X = scratch, W(..) = s_waitcnt, v/s = vector/scalar ALU, run-length encoded): shows where loads are issued and
where they are waited for.  usage: isa_trace.py file.s mangled-name-prefix [--loop-only]"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(' + re.escape(name) + r'\S*):', s, re.M)
i = m.start()
j = s.index('.Lfunc_end', i)
out = []
for ln in s[i:j].split('\n'):
    t = ln.strip()
    if not t or t.startswith(';'):
        continue
    if t.startswith('.'):
        if t.startswith('.LBB'):
            out.append('\n' + t.split()[0])
        continue
    op = t.split()[0]
    if op.startswith(('global_load', 'buffer_load', 'flat_load')): k = 'L'
    elif op.startswith(('global_store', 'buffer_store', 'flat_store')): k = 'S'
    elif 'mfma' in op: k = 'M'
    elif op.startswith('s_waitcnt'): k = 'W(' + t.split(None, 1)[1].replace('cnt', '') + ')'
    elif op.startswith('scratch_'): k = 'X'
    elif op.startswith('ds_'): k = 'D'
    elif op.startswith(('s_cbranch', 's_branch')): k = 'B->' + t.split()[-1]
    elif op.startswith('s_barrier'): k = 'BAR'
    elif op.startswith('v_'): k = 'v'
    else: k = 's'
    out.append(k)
res, prev, cnt = [], None, 0
for k in out + [None]:
    if k == prev:
        cnt += 1
    else:
        if prev is not None:
            res.append(f"{prev}{cnt if cnt > 1 else ''}")
        prev, cnt = k, 1
print(m.group(1)[:60])
print(' '.join(res))
