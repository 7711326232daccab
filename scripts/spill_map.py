#!/usr/bin/env python3
"""Where the register spills of one kernel land: for a hipcc -S file and a mangled-name substring, lists every loop
(backward branch) with its instruction count, MFMA count and its spill traffic -- scratch_load / scratch_store (VGPR
spills), v_writelane / v_readlane (SGPR spills kept in VGPR lanes) -- and the totals outside all loops.  Loops are
reported innermost-first by line range; nested loops are marked.
usage: spill_map.py <file.s> <mangled-name substring> [min instructions]"""
import collections, re, sys
s = open(sys.argv[1]).read()
m = [x for x in re.finditer(r'^(_Z\S*):', s, re.M) if sys.argv[2] in x.group(1)][0]
body = s[m.end():s.index('s_endpgm', m.end())].split('\n')
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 50
labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r'^(\.LBB\S+):', l)] if mm}
def is_ins(x): return x.startswith('\t') and x.strip() and not x.strip().startswith(('.', ';'))
def count(a, b):
  c = collections.Counter()
  for x in body[a:b]:
    if not is_ins(x): continue
    op = x.split()[0]
    c['n'] += 1
    if op.startswith('v_mfma'): c['mfma'] += 1
    if op.startswith('scratch_load'): c['scratch_load'] += 1
    if op.startswith('scratch_store'): c['scratch_store'] += 1
    if op == 'v_readlane_b32': c['readlane'] += 1
    if op == 'v_writelane_b32': c['writelane'] += 1
  return c
loops = []
for i, l in enumerate(body):
  mm = re.match(r'\s+s_c?branch\S*\s+(\.LBB\S+)', l)
  if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
    loops.append((labels[mm.group(1)], i))
loops.sort(key=lambda ab: ab[1] - ab[0])
tot = count(0, len(body))
print('kernel: %d instructions, %d MFMA; spills: scratch_load %d scratch_store %d v_readlane %d v_writelane %d' %
      (tot['n'], tot['mfma'], tot['scratch_load'], tot['scratch_store'], tot['readlane'], tot['writelane']))
inloop = set()
for a, b in loops:
  c = count(a, b)
  if c['n'] < minn: continue
  outer = [1 for (x, y) in loops if x <= a and b <= y and (x, y) != (a, b)]
  print('loop lines %6d..%6d  n=%6d mfma=%5d  scratch ld/st %3d/%3d  lane rd/wr %3d/%3d  %s' %
        (a, b, c['n'], c['mfma'], c['scratch_load'], c['scratch_store'], c['readlane'], c['writelane'],
         'nested in %d loop(s)' % len(outer) if outer else 'outermost'))
