"""HIP tree kernels (through the C ABI, external-inference mode) vs the reference's golden traces.

The recorded network outputs and Dirichlet draws of tests/golden/g2_tree_*.npz / g3_game_*.npz are fed
to mz_root_load / mz_root_prepare / mz_select / mz_expand_backup / mz_finalize.  Bar:
  bit-exact : selected paths, leaf depth, actions, N, expansion index, to_play, node existence,
              W (value sums), MinMax bounds, rewards, visit distributions, root value, error, action
  <= 4 ulp  : priors P (device exp() vs glibc exp(); everything downstream of P that is an integer
              or a sum of network values is still exact)
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'g2_tree_*.npz')) +
               glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'g3_game_*.npz')))


def make_engine(g, B):
  from model_based_rl_amd.engine import Engine
  kb = [None if np.isnan(x) else float(x) for x in g['known_bounds']]
  return Engine(B, int(g['O']), int(g['A']), int(g['sims']), two_players=bool(g['two_players']),
                known_bounds=tuple(kb), discount=float(g['discount']), pb_c_base=float(g['pb_c_base']),
                pb_c_init=float(g['pb_c_init']), init_value_score=float(g['init_value_score']),
                root_dirichlet_alpha=float(g['alpha']), root_exploration_fraction=float(g['frac']))


def ulp_diff(a, b):
  a = np.ascontiguousarray(a, np.float64).view(np.int64)
  b = np.ascontiguousarray(b, np.float64).view(np.int64)
  return np.abs(a - b)


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_tree_parity(path):
  import torch
  g = np.load(path)
  A, sims = int(g['A']), int(g['sims'])
  M = g['action'].shape[0]
  eng = make_engine(g, M)
  eng.root_load(g['root_value'], g['root_logits'], g['root_hidden'])
  eng.root_prepare(g['to_play'], g['legal'], g['noise'])
  for s in range(sims):
    leaf, slot, act, depth = [x.cpu().numpy() for x in eng.select()]
    assert np.array_equal(depth, g['leaf_depth'][:, s]), 'depth sim %d' % s
    assert np.array_equal(act, g['sim_action'][:, s]), 'action sim %d' % s
    hid = eng.gather_hidden().cpu().numpy()
    assert np.array_equal(hid, g['sim_parent_hidden'][:, s]), 'parent hidden sim %d' % s
    eng.expand_backup(g['sim_value'][:, s], g['sim_reward'][:, s], g['sim_logits'][:, s], g['sim_hidden'][:, s])
  ex = eng.export_tree()
  EX = g['tree_EX'].astype(bool)
  assert np.array_equal(ex['EX'].astype(bool), EX)
  for k in ('N', 'E', 'TP'):
    assert np.array_equal(ex[k][EX], g['tree_' + k][EX]), k
  assert np.array_equal(ex['W'][EX], g['tree_W'][EX]), 'W'
  assert np.array_equal(ex['R'].astype(np.float64)[EX], g['tree_R'][EX]), 'R'
  assert np.array_equal(ex['minmax'], g['minmax']), 'minmax'
  assert ulp_diff(ex['P'][EX], g['tree_P'][EX]).max() <= 4, 'P'
  u = np.where(g['uniform'] < 0, 0.0, g['uniform'])
  out = eng.finalize(g['temperature'], u)
  out = {k: v.cpu().numpy() for k, v in out.items()}
  assert np.array_equal(out['child_visits'], g['child_visits'])
  assert np.array_equal(out['root_value'], g['final_root_value'])
  assert np.array_equal(out['error'], g['error'])
  sampled = g['temperature'] != 0
  assert np.array_equal(out['action'][sampled], g['action'][sampled])
  for bi in np.where(~sampled)[0]:
    assert out['visit_counts'][bi, g['action'][bi]] == out['visit_counts'][bi].max()
  eng.close()


def test_tree_vs_oracle_random_large():
  """4096 trees, random 'network' outputs: HIP tree kernels vs the CPU oracle, same bar as above."""
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine
  rng = np.random.RandomState(7)
  for (A, sims, two, bounds) in [(4, 30, False, (None, None)), (9, 30, True, (-1.0, 1.0)), (6, 50, False, (None, None)),
                                 (18, 20, False, (None, None))]:
    B = 4096 if A <= 6 else 1024
    eng = Engine(B, 8, A, sims, two_players=two, known_bounds=bounds, discount=0.997)
    t = orc.Trees(orc.tree_cfg(A, sims, two, bounds, 0.997), B)
    logits = (rng.standard_normal((B, A)) * 2).astype(np.float32)
    legal = (rng.uniform(size=(B, A)) < 0.8).astype(np.uint8)
    legal[np.arange(B), rng.randint(0, A, B)] = 1
    noise = rng.dirichlet([0.25] * A, size=B) * legal
    noise /= noise.sum(1, keepdims=True)
    tp = rng.choice([-1, 1], size=B).astype(np.int8) if two else np.ones(B, np.int8)
    v0 = rng.standard_normal(B).astype(np.float32)
    eng.root_load(v0, logits)
    eng.root_prepare(tp, legal, noise)
    t.root_expand(tp, logits, legal)
    t.add_noise(noise, 0.25)
    for s in range(sims):
      got = [x.cpu().numpy() for x in eng.select()]
      want = t.select()
      for gg, ww, nm in zip(got, want, ('leaf', 'slot', 'action', 'depth')):
        assert np.array_equal(gg, ww), (nm, s)
      val = (rng.standard_normal(B) * 3).astype(np.float32)
      rew = (rng.standard_normal(B)).astype(np.float32)
      lg = (rng.standard_normal((B, A)) * 2).astype(np.float32)
      lg[rng.uniform(size=B) < 0.1] = 0.5          # exact ties
      eng.expand_backup(val, rew, lg)
      t.expand_backup(val, rew, lg)
    ex, eo = eng.export_tree(), t.export()
    EX = eo['EX'].astype(bool)
    assert np.array_equal(ex['EX'].astype(bool), EX)
    for k in ('N', 'E', 'TP', 'W'):
      assert np.array_equal(ex[k][EX], eo[k][EX]), k
    assert np.array_equal(ex['minmax'], eo['minmax'])
    assert ulp_diff(ex["P"][EX], eo["P"][EX]).max() <= 8   # up to 18 exp() terms in the normaliser
    temp = rng.choice([1.0, 0.5, 0.25, 0.0], size=B)
    u = rng.uniform(size=B)
    out = {k: v.cpu().numpy() for k, v in eng.finalize(temp, u).items()}
    action, cv, rv, vc = t.finalize(temp, u)
    assert np.array_equal(out['action'], action)
    assert np.array_equal(out['child_visits'], cv)
    assert np.array_equal(out['root_value'], rv)
    assert np.array_equal(out['visit_counts'], vc)
    eng.close()


def test_fused_step_vs_oracle_random_large():
  """mz_expand_backup_select (k_tree_step_ext: a simulation's expand + backup and the next descent in one launch) against
  the CPU oracle stepped the reference's way (mcts.py:83-99): every selection it hands back and the final trees, same bar."""
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine
  rng = np.random.RandomState(11)
  for (A, sims, two, bounds) in [(4, 30, False, (None, None)), (9, 25, True, (-1.0, 1.0)), (6, 50, False, (None, None)),
                                 (18, 12, False, (None, None))]:
    B = 2048 if A <= 6 else 512
    eng = Engine(B, 8, A, sims, two_players=two, known_bounds=bounds, discount=0.997)
    t = orc.Trees(orc.tree_cfg(A, sims, two, bounds, 0.997), B)
    logits = (rng.standard_normal((B, A)) * 2).astype(np.float32)
    legal = (rng.uniform(size=(B, A)) < 0.8).astype(np.uint8)
    legal[np.arange(B), rng.randint(0, A, B)] = 1
    noise = rng.dirichlet([0.25] * A, size=B) * legal
    noise /= noise.sum(1, keepdims=True)
    tp = rng.choice([-1, 1], size=B).astype(np.int8) if two else np.ones(B, np.int8)
    eng.root_load(rng.standard_normal(B).astype(np.float32), logits)
    eng.root_prepare(tp, legal, noise)
    t.root_expand(tp, logits, legal)
    t.add_noise(noise, 0.25)
    sel = eng.select()
    for s in range(sims):
      want = t.select()
      for gg, ww, nm in zip(sel, want, ('leaf', 'slot', 'action', 'depth')):
        assert np.array_equal(gg.cpu().numpy(), ww), (nm, s)
      val = (rng.standard_normal(B) * 3).astype(np.float32)
      rew = (rng.standard_normal(B)).astype(np.float32)
      lg = (rng.standard_normal((B, A)) * 2).astype(np.float32)
      lg[rng.uniform(size=B) < 0.1] = 0.5          # exact ties
      sel = eng.expand_backup_select(val, rew, lg, last=(s + 1 == sims))
      t.expand_backup(val, rew, lg)
    assert sel is None
    ex, eo = eng.export_tree(), t.export()
    EX = eo['EX'].astype(bool)
    assert np.array_equal(ex['EX'].astype(bool), EX)
    for k in ('N', 'E', 'TP', 'W'):
      assert np.array_equal(ex[k][EX], eo[k][EX]), k
    assert np.array_equal(ex['minmax'], eo['minmax'])
    temp = rng.choice([1.0, 0.5, 0.0], size=B)
    u = rng.uniform(size=B)
    out = {k: v.cpu().numpy() for k, v in eng.finalize(temp, u).items()}
    action, cv, rv, vc = t.finalize(temp, u)
    assert np.array_equal(out['action'], action) and np.array_equal(out['child_visits'], cv) and np.array_equal(out['visit_counts'], vc)
    eng.close()
