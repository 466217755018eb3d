#!/usr/bin/env python3
"""Static instruction mix of one kernel from build/isa/capi.s: per basic block (label to label) the number of VALU / SALU /
LDS / VMEM / SMEM instructions, v_readlane / v_writelane (SGPR spill traffic) and s_nop; argv: [file] substring-of-mangled-name"""
import re, sys
args = [a for a in sys.argv[1:] if a != "-b"]
path = args[0] if len(args) > 1 else "build/isa/capi.s"
key = args[-1]
s = open(path).read()
m = [x for x in re.finditer(r'\n(_Z\S+):[^\n]*\n', s) if key in x.group(1)]
name = m[0].group(1)
i = m[0].end(); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
blocks = []; cur = ['entry', {}]
def kind(op):
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'vmem'
    return 'other'
tot = {}
for ln in body:
    t = ln.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if re.match(r'^\.LBB\S+:', t):
            blocks.append(cur); cur = [t.rstrip(':'), {}]
        continue
    op = t.split()[0]
    k = kind(op)
    cur[1][k] = cur[1].get(k, 0) + 1
    tot[k] = tot.get(k, 0) + 1
blocks.append(cur)
print(name)
print('total', tot)
if '-b' in sys.argv:
    for b in blocks:
        n = sum(b[1].values())
        if n >= 8: print('%-14s %s' % (b[0], b[1]))
