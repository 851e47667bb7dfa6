#!/usr/bin/env python3
"""Register / LDS / scratch usage of the compiled kernels, from the metadata of `make -C fewbit_amd/csrc asm` output.
usage: tools/kernel_resources.py [/tmp/fewbit_kernels_gfx950.s] [name-substring]"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/fewbit_kernels_gfx950.s'
flt = sys.argv[2] if len(sys.argv) > 2 else ''
s = open(path).read()
meta = s[s.index('amdhsa.kernels:'):]
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', meta, re.S):
    body = m.group(0)
    name = re.search(r'\.name:\s+(\S+)', body).group(1)
    if flt not in name:
        continue
    get = lambda k: re.search(r'\.%s:\s+(\d+)' % k, body).group(1)
    print(f"{name[:90]:90s} vgpr {get('vgpr_count'):>3} spill {get('vgpr_spill_count'):>2} sgpr {get('sgpr_count'):>3} "
          f"lds {get('group_segment_fixed_size'):>6} scratch {get('private_segment_fixed_size'):>3}")
