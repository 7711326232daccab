#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S assembly file: for every backward branch, the instruction
classes between its target label and the branch.  usage: asm_loops.py <file.s> <mangled-name substring>"""
import collections, re, sys
s = open(sys.argv[1]).read()
m = [x for x in re.finditer(r'^(_Z\S*):', s, re.M) if sys.argv[2] in x.group(1)][0]
body = s[m.end():s.index('s_endpgm', m.end())].split('\n')
labels = {}
for i, l in enumerate(body):
  mm = re.match(r'^(\.LBB\S+):', l)
  if mm:
    labels[mm.group(1)] = i
def cls(l):
  op = l.split()[0]
  if op.startswith('v_mfma'): return 'mfma'
  if op.startswith('scratch_'): return op
  if op in ('v_readlane_b32', 'v_writelane_b32'): return 'lane'
  if op.startswith('v_accvgpr'): return 'accmov'
  if op.startswith('s_nop'): return 's_nop'
  if op.startswith('s_waitcnt'): return 'waitcnt'
  if op.startswith('ds_'): return 'ds'
  if op.startswith('buffer_load') or op.startswith('global_load') or op.startswith('flat_load'): return 'vmem_ld'
  if op.startswith('buffer_store') or op.startswith('global_store') or op.startswith('flat_store'): return 'vmem_st'
  if op.startswith('v_'): return 'valu'
  if op.startswith('s_'): return 'salu'
  return 'other'
for i, l in enumerate(body):
  mm = re.match(r'\s+s_cbranch\S*\s+(\.LBB\S+)', l) or re.match(r'\s+s_branch\s+(\.LBB\S+)', l)
  if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
    a = labels[mm.group(1)]
    c = collections.Counter(cls(x) for x in body[a:i] if x.startswith('\t') and not x.strip().startswith(('.', ';')) and x.strip())
    if sum(c.values()) > int(sys.argv[3] if len(sys.argv) > 3 else 200):
      print('loop lines %d..%d  n=%d ' % (a, i, sum(c.values())), dict(c.most_common()))
