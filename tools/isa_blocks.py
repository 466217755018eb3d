#!/usr/bin/env python3
"""per basic block instruction counts [all, VALU, memory] of the kernel whose mangled name starts with argv[2], from argv[1] (build/isa/capi.s)"""
import sys, collections
s = open(sys.argv[1]).read()
i = s.index('\n' + sys.argv[2]); j = s.index('.end_amdhsa_kernel', i)
isins = lambda l: l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))
cur = 'entry'; cnt = collections.OrderedDict({'entry': [0, 0, 0]})
for l in s[i:j].split('\n'):
    if l.startswith('.LBB'): cur = l.split(':')[0]; cnt[cur] = [0, 0, 0]
    elif isins(l):
        op = l.split()[0]; cnt[cur][0] += 1
        if op.startswith('v_'): cnt[cur][1] += 1
        if op.startswith(('global_', 'flat_', 'ds_', 'scratch_', 'buffer_')): cnt[cur][2] += 1
print(' '.join('%s%s' % (k.replace('.LBB', 'B'), v) for k, v in cnt.items()))
print('total', [sum(v[i] for v in cnt.values()) for i in range(3)])
