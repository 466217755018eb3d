#!/usr/bin/env python3
"""VALU / SALU instructions per source line of one kernel (hipcc -gline-tables-only -S): static counts per .loc line.
usage: isa_lines.py build/isa/capi_g.s <substring of mangled name> [source-file-substring]"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]; fkey = sys.argv[3] if len(sys.argv) > 3 else 'aqc_fast'
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(m.group(1))] = (m.group(3) or m.group(2))
m = [x for x in re.finditer(r'\n(_Z\S+):[^\n]*\n', s) if key in x.group(1)][0]
i = m.end(); j = s.index('.Lfunc_end', i)
cnt = collections.defaultdict(lambda: [0, 0, 0])
cur = None
for ln in s[i:j].split('\n'):
    t = ln.strip()
    mm = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if mm:
        cur = (files.get(int(mm.group(1)), '?'), int(mm.group(2))); continue
    if not t or t[0] in ';.' : continue
    op = t.split()[0]
    if cur is None: continue
    if op.startswith('v_'): cnt[cur][0] += 1
    elif op.startswith('s_') and not op.startswith('s_waitcnt') and not op.startswith('s_nop'): cnt[cur][1] += 1
    elif op.startswith('s_nop'): cnt[cur][2] += 1
tot = [0, 0, 0]
for (f, l), c in sorted(cnt.items()):
    if fkey in f: print('%s:%d  valu %d salu %d nop %d' % (f.split('/')[-1], l, c[0], c[1], c[2]))
    for k in range(3): tot[k] += c[k]
print('total', tot)
