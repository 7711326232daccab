#!/usr/bin/env python3
"""Register / spill / scratch figures of the kernels in a hipcc -save-temps assembly file (.s), from its metadata.
usage: asm_resources.py <file.s> [name filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
md = s[s.index('amdhsa.kernels'):]
for b in md.split('  - .agpr_count:')[1:]:
  name = re.search(r'\.name:\s+(\S+)', b).group(1)
  dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0].replace('void ', '')
  if flt not in dn:
    continue
  g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
  print(dn, 'agpr', b.split('\n')[0].strip(), 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'vspill', g('vgpr_spill_count'),
        'sspill', g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'), 'lds', g('group_segment_fixed_size'))
