#!/usr/bin/env python3
"""Opcode-level diff of one kernel's largest loop between two hipcc -S files (kernel development: what a source change
did to the simulation loop).  usage: asm_diff.py <old.s> <new.s> <mangled-name substring>"""
import collections, difflib, re, sys
def body(f, name):
  s = open(f).read()
  m = [x for x in re.finditer(r'^(_Z\S*):', s, re.M) if name in x.group(1)][0]
  return s[m.end():s.index('s_endpgm', m.end())].split('\n')
def is_ins(l): return l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))
def simloop(lines):
  labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r'^(\.LBB\S+):', l)] if mm}
  best = None
  for i, l in enumerate(lines):
    mm = re.match(r'\s+s_c?branch\S*\s+(\.LBB\S+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
      a = labels[mm.group(1)]
      nm = sum(1 for x in lines[a:i] if is_ins(x) and x.split()[0].startswith('v_mfma'))
      n = sum(1 for x in lines[a:i] if is_ins(x))
      # the simulation loop: the largest loop that is itself nested (has all 720 MFMAs but is not the move loop)
      if nm >= 700 and (best is None or n < best[2]) and n > 1900: best = (a, i, n)
  return lines[best[0]:best[1]]
ins = lambda lines: [re.sub(r'\s+', ' ', l.strip()) for l in lines if is_ins(l)]
a, b = ins(simloop(body(sys.argv[1], sys.argv[3]))), ins(simloop(body(sys.argv[2], sys.argv[3])))
oa, ob = [x.split()[0] for x in a], [x.split()[0] for x in b]
print('instructions: %d -> %d' % (len(a), len(b)))
ca, cb = collections.Counter(oa), collections.Counter(ob)
for k in sorted(set(ca) | set(cb)):
  if ca[k] != cb[k]: print('  %-28s %4d -> %4d' % (k, ca[k], cb[k]))
if len(sys.argv) > 4:
  sm = difflib.SequenceMatcher(None, oa, ob, autojunk=False)
  for tag, i1, i2, j1, j2 in sm.get_opcodes():
    if tag != 'equal':
      print(tag, i1, i2, j1, j2)
      for x in a[i1:i2]: print('   -', x)
      for x in b[j1:j2]: print('   +', x)
