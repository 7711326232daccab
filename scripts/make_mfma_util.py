#!/usr/bin/env python3
"""MFMA utilisation of the kernels from a rocprofv3 --pmc pass (its own run, counters only):
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 ... -- python3 bench.py ...
usage: make_mfma_util.py <pmc output dir> [out.json]

Per kernel: mean counter values per launch and
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * SIMDs)   (rocprofv3's MfmaUtil expression; SIMDs = 4 x 256 CUs;
              rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs -- MI355X_MICROARCH.md, DVFS note -- so one XCD's
              active cycles = the kernel's duration in shader cycles = GRBM_GUI_ACTIVE / 8; SQ_VALU_MFMA_BUSY_CYCLES counts
              cycles summed over all SIMDs)
  mfma_flop = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512                          (rocprofv3's MfmaFlopsF32 expression)
"""
import csv, glob, json, os, sys
from collections import defaultdict

SIMDS = 256 * 4
XCDS = 8


def main():
  acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
  for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    with open(f, newline='') as fh:
      for row in csv.DictReader(fh):
        name = row['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '').strip()
        a = acc[name][row['Counter_Name']]
        a[0] += 1
        a[1] += float(row['Counter_Value'])
  out = {}
  for name, cs in sorted(acc.items()):
    if not name.startswith('k_'):
      continue
    m = {c: v[1] / v[0] for c, v in cs.items()}
    e = {'launches': max(v[0] for v in cs.values()), 'mean_per_launch': m}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m.get('GRBM_GUI_ACTIVE'):
      e['active_cycles_per_xcd'] = m['GRBM_GUI_ACTIVE'] / XCDS
      e['mfma_util'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / XCDS * SIMDS)
    if 'SQ_INSTS_VALU_MFMA_MOPS_F32' in m:
      e['mfma_flop_per_launch'] = m['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512
    out[name] = e
  text = json.dumps(out, indent=1)
  if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text + '\n')
  print(text)


if __name__ == '__main__':
  main()
