#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS report of libmz_hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py [name filter]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import model_based_rl_amd  # noqa: F401  (import alias)
from model_based_rl_amd import _abi
src = os.path.join(os.path.dirname(_abi.__file__), 'csrc')
flt = sys.argv[1] if len(sys.argv) > 1 else ''
# every translation unit of the library (_abi.translation_units: the host unit and one unit per search-kernel shape)
from concurrent.futures import ThreadPoolExecutor
def remarks(unit):
  name, source, defs = unit
  cmd = ['hipcc'] + list(_abi.HIPCC_FLAGS) + defs + ['-Rpass-analysis=kernel-resource-usage', '-c', source, '-o', '/tmp/_mz_res_%s.o' % name]
  return subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as pool:
  err = '\n'.join(pool.map(remarks, _abi.translation_units()))
cur = None
rows = {}
for line in err.splitlines():
  m = re.search(r'remark: +Function Name: (\S+)', line)
  if m:
    cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    cur = cur.split('(')[0].replace('void ', '')
    rows[cur] = {}
    continue
  m = re.search(r'remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (\d+)', line)
  if m and cur:
    rows[cur][m.group(1).split(' [')[0]] = int(m.group(2))
for k, v in rows.items():
  if flt in k:
    print('%-48s V %3d  A %3d  scratch %4d  lds %6d  sgpr-spill %3d' % (k[:48], v.get('VGPRs', -1), v.get('AGPRs', -1), v.get('ScratchSize', -1), v.get('LDS Size', -1), v.get('SGPRs Spill', -1)))
