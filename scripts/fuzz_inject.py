"""Randomised sweep of the fused kernels' tree code over the whole dispatch table: random configurations (action count 1..32,
1..62 simulations, tree counts that are not multiples of 16, one / two players, known bounds, init_value_score, illegal root
actions, exact ties, long chains) with injected 'network' outputs (mz_sim_io) against the oracle's tree fed the same numbers;
every field of every tree must be identical (priors within 8 ulp).  usage: fuzz_inject.py [n_configs] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from model_based_rl_amd.engine import Engine
from tests.parity_util import env_switches, random_weights, ulp_diff

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
A_CHOICES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 16, 17, 21, 22, 32]
S_CHOICES = [1, 2, 3, 5, 17, 30, 33, 50, 61, 62]
done = 0
t0 = time.time()
for it in range(n_cfg):
  A = int(rng.choice(A_CHOICES)); sims = int(rng.choice(S_CHOICES)); two = bool(rng.randint(2))
  B = int(rng.choice([1, 5, 16, 17, 48, 100, 256]))
  bounds = [(None, None), (-1.0, 1.0), (-3.0, 3.0), (None, 2.0)][rng.randint(4)]
  ivs = float(rng.choice([0.0, 0.0, 0.25]))
  variant = ['lds', 'pool', 'nohybrid', 'split'][rng.randint(4)]
  split = variant == 'split' and A <= 13
  sw = {'pool': {'MZ_NO_LDS_TREES': '1'}, 'nohybrid': {'MZ_NO_LDS_HYBRID': '1'}}.get(variant, {})
  with env_switches(**sw):
    eng = Engine(B, 8, A, sims, two_players=two, known_bounds=bounds, discount=0.997, init_value_score=ivs, split_f16=split)
  eng.set_weights(random_weights(8, A))
  t = orc.Trees(orc.tree_cfg(A, sims, two, bounds, 0.997, init_value_score=ivs), B)
  logits = (rng.standard_normal((B, A)) * 2).astype(np.float32)
  legal = (rng.uniform(size=(B, A)) < 0.8).astype(np.uint8)
  legal[np.arange(B), rng.randint(0, A, B)] = 1
  noise = rng.dirichlet([0.25] * A, size=B) * legal
  noise /= noise.sum(1, keepdims=True)
  tp = rng.choice([-1, 1], size=B).astype(np.int8) if two else np.ones(B, np.int8)
  v0 = rng.standard_normal(B).astype(np.float32)
  val = (rng.standard_normal((B, sims)) * 3).astype(np.float32)
  rew = rng.standard_normal((B, sims)).astype(np.float32)
  lg = (rng.standard_normal((B, sims, A)) * 2).astype(np.float32)
  lg[rng.uniform(size=(B, sims)) < 0.1] = 0.5
  val[rng.uniform(size=(B, sims)) < 0.05] = 0.0
  rew[rng.uniform(size=(B, sims)) < 0.3] = 0.0
  nch = max(1, B // 3)                          # chains: one action carries the prior mass -> paths of up to sims + 1 nodes
  lg[:nch] = (rng.standard_normal((nch, sims, A)) * 0.1).astype(np.float32)
  lg[:nch, :, rng.randint(0, A)] += 9.0
  logits[:nch] = lg[:nch, 0]; legal[:nch] = 1; noise[:nch] = rng.dirichlet([0.25] * A, size=nch)
  eng.root_load(v0, logits); eng.root_prepare(tp, legal, noise)
  vals = np.zeros((B, sims + 1, 2 + A), np.float32)
  vals[:, 1:, 0], vals[:, 1:, 1], vals[:, 1:, 2:] = val, rew, lg
  eng.sim_io('inject', values=vals)
  info = eng.search_kernel_info()
  if info['kind'] == 'standalone':             # (num_simulations > 62 or no scalable weights: not this test's subject)
    eng.close(); continue
  eng.search()
  t.root_expand(tp, logits, legal); t.add_noise(noise, 0.25)
  for s in range(sims):
    t.select(); t.expand_backup(val[:, s], rew[:, s], lg[:, s])
  ex, eo = eng.export_tree(), t.export()
  EX = eo['EX'].astype(bool)
  tag = (it, 'A', A, 'sims', sims, 'B', B, 'two', two, bounds, ivs, variant, info)
  assert np.array_equal(ex['EX'].astype(bool), EX), tag
  for k in ('N', 'E', 'TP', 'W'):
    assert np.array_equal(ex[k][EX], eo[k][EX]), (k,) + tag
  assert np.array_equal(ex['R'].astype(np.float64)[EX], eo['R'][EX]), tag
  assert np.array_equal(ex['minmax'], eo['minmax']), tag
  assert ulp_diff(ex['P'][EX], eo['P'][EX]).max() <= 8, tag
  temp = rng.choice([1.0, 0.0], size=B); u = rng.uniform(size=B)
  out = {k: v.cpu().numpy() for k, v in eng.finalize(temp, u).items()}
  action, cv, rv, vc = t.finalize(temp, u)
  assert np.array_equal(out['visit_counts'], vc) and np.array_equal(out['action'], action) and np.array_equal(out['root_value'], rv), tag
  eng.sim_io('off'); eng.close()
  done += 1
  print('ok', *tag, 'deepest path', int((eo['N'] > 0).sum(1).max()), flush=True)
print('%d configurations identical (%d skipped: stand-alone kernels), %.0f s' % (done, n_cfg - done, time.time() - t0))
