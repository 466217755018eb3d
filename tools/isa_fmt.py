#!/usr/bin/env python3
"""registers / LDS / scratch / spilled SGPRs of the kernels whose mangled name contains argv[2] (default: fmt_), from build/isa/capi.s"""
import re, sys
s = open(sys.argv[1] if len(sys.argv) > 1 else 'build/isa/capi.s').read()
pat = sys.argv[2] if len(sys.argv) > 2 else 'fmt_'
spill = dict(re.findall(r'\.name:\s+(\S+)\n(?:.*\n){0,3}?\s+\.sgpr_spill_count:\s+(\d+)', s))
vspill = dict(re.findall(r'\.name:\s+(\S+)\n(?:.*\n){0,12}?\s+\.vgpr_spill_count:\s+(\d+)', s))
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name = m.group(1); body = m.group(2)
    g = lambda k: re.search(r'\.amdhsa_' + k + r' (\S+)', body).group(1)
    if pat in name:
        print(name[:64].ljust(64), 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'), 'sgpr_spill', spill.get(name), 'vgpr_spill', vspill.get(name))
