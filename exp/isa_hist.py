#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a hipcc -save-temps .s file.
usage: isa_hist.py file.s mangled_kernel_substring [--blocks]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
show_blocks = '--blocks' in sys.argv
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if re.match(r'^(_Z\w+):', l) and key in l:
        start = i; break
assert start is not None, 'kernel not found'
hist = collections.Counter(); blocks = []; cur = None
n = 0
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end') or l.strip().startswith('.section'):
        break
    m = re.match(r'^(\.LBB\w+):', l)
    if m:
        cur = [m.group(1), collections.Counter()]; blocks.append(cur); continue
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    op = s.split()[0]
    if not re.match(r'^[vsdgb]_|^ds_|^global_|^buffer_|^flat_|^scratch_', op):
        continue
    hist[op] += 1; n += 1
    if cur: cur[1][op] += 1
def cls(op):
    if op.startswith('v_') and ('f64' in op): return 'valu_f64'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('s_'): return 'salu'
    return 'mem'
c = collections.Counter()
for op, k in hist.items(): c[cls(op)] += k
print('total', n, dict(c))
for op, k in hist.most_common(45): print(f'{k:6d} {op}')
if show_blocks:
    for name, h in blocks:
        t = sum(h.values())
        if t >= 20:
            cc = collections.Counter()
            for op, k in h.items(): cc[cls(op)] += k
            print(name, t, dict(cc))
