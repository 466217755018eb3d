#!/usr/bin/env python3
"""register / LDS / scratch use of the lane-per-read kernel variants (or of the kernels whose mangled name contains argv[2]),
from build/isa/capi.s (hipcc -save-temps)"""
import re, sys
s = open(sys.argv[1] if len(sys.argv) > 1 else "build/isa/capi.s").read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name = m.group(1)
    if (sys.argv[2] if len(sys.argv) > 2 else 'fast_filter') not in name:
        continue
    body = m.group(2)
    g = lambda k: re.search(r'\.amdhsa_' + k + r' (\S+)', body).group(1)
    i = s.index(name + ':'); j = s.index('.end_amdhsa_kernel', i)
    sc = len(re.findall(r'\n\tscratch_', s[i:j]))
    print(name[20:60], 'vgpr', g('next_free_vgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'), 'scratch_ops', sc)
