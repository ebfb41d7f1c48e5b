#!/usr/bin/env python3
"""Static per-phase instruction table of the pooled triangulation kernel (8 cameras, 3 tiles per wave) from hipcc's
assembly: where the vector instructions of level 0, of a screen round (tier A) and of an fp64 evaluation pass (tier B) go.
Phases are cut at landmark instructions (the tile's loads and stores, the reciprocals of the eigen-solves, the reciprocal
square roots of the reprojection passes).  usage: isa_phases.py file.s"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
key = 'p2s_tri_pool_kernelIfLi8ELi48ELb1ELi5E'
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and key in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
def find(pat, after=0, nth=1):
    n = 0
    for i in range(after, len(body)):
        if re.search(pat, body[i]):
            n += 1
            if n == nth: return i
    raise SystemExit(f'landmark {pat} not found')
def cls(op):
    if op.startswith('v_pk_'): return 'packed fp32'
    if op.startswith('v_') and 'f64' in op and not op.startswith('v_cvt') and not op.startswith('v_cmp'): return 'fp64 arithmetic'
    if op.startswith('v_') and 'f32' in op and not op.startswith('v_cvt') and not op.startswith('v_cmp'): return 'fp32 arithmetic'
    if op.startswith('v_cvt'): return 'conversions'
    if op.startswith('v_cndmask'): return 'selects'
    if op.startswith('v_cmp'): return 'compares'
    if op.startswith('v_mov') or op.startswith('v_pk_mov') or op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'moves / lane traffic'
    if op.startswith('v_'): return 'other vector (integer, bit, address)'
    if op.startswith('s_'): return 'scalar'
    return 'memory (LDS / global)'
def table(name, a, b):
    c = collections.Counter()
    for l in body[a:b]:
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.'): continue
        op = s.split()[0]
        if re.match(r'^[vs]_|^ds_|^global_|^flat_|^scratch_|^buffer_', op): c[cls(op)] += 1
    vec = sum(v for k, v in c.items() if k not in ('scalar', 'memory (LDS / global)'))
    print(f'{name}: {vec} vector instructions (+ {c["scalar"]} scalar, {c["memory (LDS / global)"]} memory), lines {a}-{b}')
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        if k not in ('scalar', 'memory (LDS / global)'): print(f'    {v:5d}  {k}')
l0 = find(r'global_load_dwordx3')                              # tile 0: first observation load
e0 = find(r'v_rcp_f64', l0)                                    # eigen-solve begins
p0 = find(r'global_load_dwordx3', e0)                          # the next tile's observations are requested: eigen-solve over
r0 = find(r'v_rsq_f64', p0)
r7 = find(r'v_rsq_f64', p0, 8)
st = find(r'global_store_dwordx4', r7)
l1 = find(r'v_cvt_f64_f32', st)                                # tile 1 begins
print('static counts: a branch that is not taken, or a loop that runs twice, is counted once\n')
table('level 0: observations -> normal matrix (8 cameras)', l0, e0)
table('level 0: eigen-solve (first pass + loop body + refinement)', e0, p0)
table('level 0: reprojection error (8 cameras)', p0, st)
table('level 0: staging, slots, stores', st, l1)
tb = find(r'ds_min_rtn_u64|ds_min_u64')
tb_rcp = max(i for i in range(tb) if re.search(r'v_rcp_f64', body[i]) and i < tb)
tb_start = max(i for i in range(tb_rcp - 1200, tb_rcp) if re.search(r'; wave barrier', body[i]) and i < tb_rcp - 200) if False else None
ta0 = find(r'v_rcp_f32')
ta_rsq_last = find(r'v_rsq_f32', ta0, 16)
ta_end = find(r'ds_bpermute_b32', ta_rsq_last)
# the screen's main block: from the base loads before the first reciprocal to the group minimum
table('tier A round: three 3x3 solves, two Rayleigh steps, 8-camera reprojection error, two candidates per lane (downdate loops and subset look-up not included)', ta0 - 40, ta_end)
