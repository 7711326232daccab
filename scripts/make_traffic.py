#!/usr/bin/env python3
"""Aggregate rocprofv3 PMC passes into profiles/traffic.json (read by bench.py for `roofline.traffic`).

usage: make_traffic.py <dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass> [out.json]

HBM bytes per launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): both counters are in KB, and FETCH_SIZE is doubled
per the gfx950 correction in /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section).  The two counters
come from separate passes, as the guide prescribes.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
  acc = defaultdict(lambda: [0, 0.0])
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    with open(f, newline='') as fh:
      for row in csv.DictReader(fh):
        if row['Counter_Name'] != counter:
          continue
        name = row['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', '').strip()
        a = acc[name]
        a[0] += 1
        a[1] += float(row['Counter_Value'])
  return acc


def main():
  fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
  out = {}
  for name in sorted(set(fetch) & set(write)):
    if not name.startswith('k_'):
      continue
    nf, sf = fetch[name]
    nw, sw = write[name]
    out[name] = {'launches': nf, 'FETCH_SIZE_KB_mean': sf / nf, 'WRITE_SIZE_KB_mean': sw / nw,
                 'hbm_bytes_per_launch': 1024.0 * (2.0 * sf / nf + sw / nw),
                 'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB units; FETCH_SIZE doubled '
                         'per the gfx950 correction in MI355X_MICROARCH.md (HBM section); includes Infinity-Cache hits'}
  text = json.dumps(out, indent=1)
  if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write(text + '\n')
  print(text)


if __name__ == '__main__':
  main()
