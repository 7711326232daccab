"""Where does the fused tree code leave the CPU restatement on injected outputs?  Runs the random-output case of
tests/test_gpu_fused_exact.py one simulation at a time through (a) oracle, (b) the stand-alone tree kernels, (c) the fused
kernel with mz_sim_io inject, and prints the first simulation / tree where each differs, with the scores of that decision."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from model_based_rl_amd.engine import Engine
from tests.test_gpu_fused_exact import random_weights, env_switches

A, sims, two, bounds, B, ivs = 4, 30, False, (None, None), 4096, 0.0
variant = sys.argv[1] if len(sys.argv) > 1 else 'lds'
flags = set(sys.argv[2:])        # 'noval0', 'norew0', 'noties'
rng = np.random.RandomState(11 + A)
logits = (rng.standard_normal((B, A)) * 2).astype(np.float32)
legal = (rng.uniform(size=(B, A)) < 0.8).astype(np.uint8)
legal[np.arange(B), rng.randint(0, A, B)] = 1
noise = rng.dirichlet([0.25] * A, size=B) * legal
noise /= noise.sum(1, keepdims=True)
tp = np.ones(B, np.int8)
v0 = rng.standard_normal(B).astype(np.float32)
val = (rng.standard_normal((B, sims)) * 3).astype(np.float32)
rew = rng.standard_normal((B, sims)).astype(np.float32)
lg = (rng.standard_normal((B, sims, A)) * 2).astype(np.float32)
m1, m2, m3 = rng.uniform(size=(B, sims)) < 0.1, rng.uniform(size=(B, sims)) < 0.05, rng.uniform(size=(B, sims)) < 0.3
if 'noties' not in flags: lg[m1] = 0.5
if 'noval0' not in flags: val[m2] = 0.0
if 'norew0' not in flags: rew[m3] = 0.0
vals = np.zeros((B, sims + 1, 2 + A), np.float32)
vals[:, 1:, 0], vals[:, 1:, 1], vals[:, 1:, 2:] = val, rew, lg

def engine(sw):
  with env_switches(**sw):
    e = Engine(B, 8, A, sims, two_players=two, known_bounds=bounds, discount=0.997, init_value_score=ivs)
  e.set_weights(random_weights(8, A))
  e.root_load(v0, logits)
  e.root_prepare(tp, legal, noise)
  return e

fused = engine({'MZ_NO_LDS_TREES': '1'} if variant == 'pool' else {})
fused.sim_io('inject', values=vals)
print(fused.search_kernel_info())
fused.search()
ef = fused.export_tree()
alone = engine({'MZ_NO_FUSED': '1'})
for s in range(sims):
  alone.select(); alone.expand_backup(val[:, s], rew[:, s], lg[:, s])
ea = alone.export_tree()

def run_oracle(upto):
  t = orc.Trees(orc.tree_cfg(A, sims, two, bounds, 0.997, init_value_score=ivs), B)
  t.root_expand(tp, logits, legal); t.add_noise(noise, 0.25)
  sel = None
  for s in range(upto):
    sel = t.select() + [t.paths()]
    t.expand_backup(val[:, s], rew[:, s], lg[:, s])
  return t, sel
t, _ = run_oracle(sims)
eo = t.export()
EX = eo['EX'].astype(bool)
for name, ex in (('stand-alone', ea), ('fused', ef)):
  same = np.all((ex['N'] == eo['N']) | ~EX, 1) & np.all((ex['E'] == eo['E']) | ~EX, 1) & np.all((ex['W'] == eo['W']) | ~EX, 1)
  print('%s: %d of %d trees differ' % (name, (~same).sum(), B))
  if same.all():
    continue
  # first divergent simulation of every bad tree: the smallest expansion index whose node differs
  firsts = []
  for b in np.flatnonzero(~same):
    e_dev = {int(e): int(n) for n, e in enumerate(ex['E'][b]) if e > 0}
    e_orc = {int(e): int(n) for n, e in enumerate(eo['E'][b]) if e > 0}
    firsts.append(min(e for e in range(1, sims + 1) if e_dev.get(e) != e_orc.get(e)))
  firsts = np.array(firsts)
  print('  first divergent simulation (0-based) histogram:', np.bincount(firsts - 1))
  for b, e in list(zip(np.flatnonzero(~same), firsts))[:4]:
    s = int(e) - 1                      # simulation s expanded a different leaf: its descent (after backup s - 1) differed
    to, sel = run_oracle(s + 1)
    leaf, slot, act, depth, paths = sel
    st, _ = run_oracle(s)               # the tree the descent of simulation s looked at
    eb = st.export()
    dev_leaf = [n for n, x in enumerate(ex['E'][b]) if x == e][0]
    print('  tree %d sim %d: oracle path %s leaf %d; device expanded node %d; margin of the tree %.3g' %
          (b, s, paths[b][:depth[b] + 1], leaf[b], dev_leaf, t.margin()[b]))
    mn, mx = eb['minmax'][b]
    print('    minmax before', mn, mx)
    node = 0
    for lvl in range(depth[b]):
      ee = eb['E'][b][node]
      ch = 1 + ee * A + np.arange(A)
      Np = eb['N'][b][node]
      sc = []
      for c in ch:
        Nc = eb['N'][b][c]
        pb = (np.log((Np + 19652 + 1) / 19652) + 1.25) * (np.sqrt(Np) / (Nc + 1))
        if Nc > 0:
          x = eb['R'][b][c] + 0.997 * (eb['W'][b][c] / Nc)
          vs = (x - mn) / (mx - mn) if mx > mn else (1.0 if mx == mn else x)
        else:
          vs = ivs
        sc.append(pb * eb['P'][b][c] + vs)
      print('    level %d node %d children N %s P %s R %s W %s\n      scores %s' % (lvl, node, eb['N'][b][ch], eb['P'][b][ch], eb['R'][b][ch], eb['W'][b][ch], ['%.17g' % x for x in sc]))
      node = paths[b][lvl + 1]
    print('    inputs of simulation %d: value %r reward %r logits %r' % (s - 1, val[b, s - 1], rew[b, s - 1], lg[b, s - 1]))
