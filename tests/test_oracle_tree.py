"""Oracle (oracle/mz_oracle.c) vs the reference's golden vectors: TREE logic, bit-exact.

Fixtures: tests/golden/g2_tree_*.npz and g3_game_*.npz (made by oracle/make_goldens.py from the
unmodified reference).  The recorded network outputs and Dirichlet noise are replayed into the oracle;
every node's N / W / P / R / expansion index / to_play, MinMax, root value, visit distribution and the
sampled action must equal the reference's Python doubles exactly (0 ulp).
"""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as orc

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'g2_tree_*.npz')) +
               glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'g3_game_*.npz')))


def replay_moves(g, batched):
  A, sims = int(g['A']), int(g['sims'])
  M = g['action'].shape[0]
  cfg = orc.tree_cfg(A, sims, bool(g['two_players']), tuple(g['known_bounds']), float(g['discount']),
                     float(g['pb_c_base']), float(g['pb_c_init']), float(g['init_value_score']))
  groups = [np.arange(M)] if batched else [np.array([m]) for m in range(M)]
  for idx in groups:
    B = len(idx)
    t = orc.Trees(cfg, B)
    t.root_expand(g['to_play'][idx], g['root_logits'][idx], g['legal'][idx])
    ex = t.export()
    assert np.array_equal(ex['P'][:, 1:1 + A], g['prior_pre'][idx]), 'root priors (mcts.py:52-55)'
    t.add_noise(g['noise'][idx], float(g['frac']))
    for s in range(sims):
      leaf, slot, act, depth = t.select()
      assert np.array_equal(depth, g['leaf_depth'][idx, s])
      assert np.array_equal(act, g['sim_action'][idx, s])
      paths = t.paths()
      # path actions: node -> action = (node-1) % A
      for bi, m in enumerate(idx):
        d = depth[bi]
        acts = (paths[bi, 1:d + 1] - 1) % A
        assert np.array_equal(acts, g['path_actions'][m, s, :d])
      t.expand_backup(g['sim_value'][idx, s], g['sim_reward'][idx, s], g['sim_logits'][idx, s])
    ex = t.export()
    for k in ('N', 'W', 'P', 'R', 'E', 'TP', 'EX'):
      assert np.array_equal(ex[k], g['tree_' + k][idx]), k
    assert np.array_equal(ex['minmax'], g['minmax'][idx])
    # the smallest top-2 score gap over all select_child decisions of the move: the reference's own number, bit for bit
    # (the GPU parity tests demand 100 % identity in every tree whose margin is above the network's float32 noise)
    assert np.array_equal(t.margin(), g['min_margin'][idx])
    temp = g['temperature'][idx]
    u = np.where(g['uniform'][idx] < 0, 0.0, g['uniform'][idx])
    action, cv, rv, vc = t.finalize(temp, u)
    assert np.array_equal(cv, g['child_visits'][idx])
    assert np.array_equal(rv, g['final_root_value'][idx])
    err = rv - g['root_value'][idx].astype(np.float64)
    assert np.array_equal(err, g['error'][idx])
    sampled = temp != 0
    assert np.array_equal(action[sampled], g['action'][idx][sampled])
    # temperature 0: the reference draws the tie with numpy's own integer sampler; check membership
    for bi in np.where(~sampled)[0]:
      assert vc[bi, g['action'][idx][bi]] == vc[bi].max()


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
@pytest.mark.parametrize('batched', [False, True], ids=['b1', 'batched'])
def test_tree_bit_exact(path, batched):
  replay_moves(np.load(path), batched)


def test_threaded_search_equals_one_batch():
  """oracle.search_fc_threads (what the large GPU parity tests run) splits the batch over host threads: same trees,
  same margins as one Trees object over the whole batch."""
  g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g1_net_lunar.npz'))
  w = orc.load_weights(g)
  B, O, A, sims = 37, 8, 4, 12
  rng = np.random.RandomState(1)
  obs = rng.standard_normal((B, O)).astype(np.float32)
  noise = rng.dirichlet([0.25] * A, size=B)
  u = rng.uniform(size=B)
  cfg = orc.tree_cfg(A, sims)
  net = orc.FCNet(w, O, A)
  t = orc.Trees(cfg, B)
  hpool, v0 = t.search_fc(net, obs, np.ones(B, np.int8), None, noise, 0.25)
  action, cv, rv, vc = t.finalize(1.0, u)
  r = orc.search_fc_threads(cfg, net, obs, noise=noise, uniform=u, threads=3)
  assert np.array_equal(r['action'], action) and np.array_equal(r['visit_counts'], vc) and np.array_equal(r['root_value'], rv)
  assert np.array_equal(r['margin'], t.margin()) and np.array_equal(r['hpool'], hpool) and np.array_equal(r['v0'], v0)
  ex = t.export()
  for k in ex:
    assert np.array_equal(r['tree'][k], ex[k]), k
  assert np.all(r['margin'] >= 0) and np.isfinite(r['margin']).all()


def test_slot_is_parent_hidden():
  """parent_slot returned by select indexes the hidden state the reference passes to
  recurrent_inference (mcts.py:94-96)."""
  g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g2_tree_lunar_fc.npz'))
  A, sims = int(g['A']), int(g['sims'])
  cfg = orc.tree_cfg(A, sims, False, (None, None), float(g['discount']))
  M = g['action'].shape[0]
  t = orc.Trees(cfg, M)
  t.root_expand(g['to_play'], g['root_logits'], g['legal'])
  t.add_noise(g['noise'], float(g['frac']))
  pool = np.zeros((M, sims + 1, 50), np.float32)
  pool[:, 0] = g['root_hidden']
  for s in range(sims):
    leaf, slot, act, depth = t.select()
    assert np.array_equal(pool[np.arange(M), slot], g['sim_parent_hidden'][:, s])
    pool[:, s + 1] = g['sim_hidden'][:, s]
    t.expand_backup(g['sim_value'][:, s], g['sim_reward'][:, s], g['sim_logits'][:, s])
