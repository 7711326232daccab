"""The TREE code inside the fused search kernels, proven exactly -- no tie margin, no excluded trees.

`mz_tree_expand_f` / `mz_tree_backup_select_f` (csrc/mz_tree.hip.h: LDS placements 0 / 1 / 2, X cache, reciprocal
normalise, table division, DPP arg-max) are different code from the stand-alone `mz_tree_select` /
`mz_tree_expand_backup` that tests/test_gpu_tree.py feeds the goldens to.  Everything a simulation's tree step consumes
from the network is three things: the value and reward scalars and the A policy logits.  `mz_sim_io` (test
instrumentation, zeros in the production kernel arguments) makes the PRODUCTION instantiations of `k_search_fused` /
`k_search_h2` either

  inject: read those three from the caller -- the reference's recorded network outputs (tests/golden g2 / g3, with their
          exact ties, known bounds, two players, init_value_score != 0) reach the fused kernels' own tree code; bar =
          test_gpu_tree.py's: N, E, to_play, W, R, MinMax, visit distributions, root value, error, action bit-exact, priors
          <= 4 ulp (device exp vs glibc);
  log:    store them -- every move of the whole-moves (HEAD) launch bench.py times, at its size (4096 environments, 16
          moves per launch), is replayed through oracle/mz_oracle.c's tree (orc_root_expand / orc_select /
          orc_expand_backup / orc_finalize, mcts.py:47-61,78-143, config.py:70-81) on the device's own network outputs,
          Dirichlet draws and uniforms: records (visit distribution, action, root value, error as float64) identical in
          100 % of the trees of every move, and every field of the exported trees of the last move.
"""
import glob
import os

import numpy as np
import pytest

from tests.parity_util import env_switches, philox_action_uniform, random_weights, replay_move, ulp_diff

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
TREE_FILES = sorted(glob.glob(os.path.join(G, 'g2_tree_*.npz')) + glob.glob(os.path.join(G, 'g3_game_*.npz')))


VARIANTS = {
    # name: (environment switches at mz_create, split_f16, expected kernel kind, expected LDS placement or None = by fit)
    'lds': ({}, False, 'fused', None),
    'pool': ({'MZ_NO_LDS_TREES': '1'}, False, 'fused', 0),
    'split_f16': ({}, True, 'split_f16', None),
}


def golden_engine(g, B, variant):
  from model_based_rl_amd.engine import Engine
  sw, split, kind, lt = VARIANTS[variant]
  kb = [None if np.isnan(x) else float(x) for x in g['known_bounds']]
  with env_switches(**sw):
    eng = Engine(B, int(g['O']), int(g['A']), int(g['sims']), two_players=bool(g['two_players']),
                 known_bounds=tuple(kb), discount=float(g['discount']), pb_c_base=float(g['pb_c_base']),
                 pb_c_init=float(g['pb_c_init']), init_value_score=float(g['init_value_score']),
                 root_dirichlet_alpha=float(g['alpha']), root_exploration_fraction=float(g['frac']), split_f16=split)
  eng.set_weights(random_weights(int(g['O']), int(g['A'])))
  return eng, kind, lt


@pytest.mark.parametrize('variant', sorted(VARIANTS))
@pytest.mark.parametrize('path', TREE_FILES, ids=[os.path.basename(f)[:-4] for f in TREE_FILES])
def test_goldens_injected_into_the_fused_kernels(path, variant):
  """the reference's recorded network outputs through mz_search's own kernel: the non-HEAD production instantiation of
  this shape (what Actor / mz_search run), trees in LDS (whole or compact, by fit), in the node pool, and k_search_h2"""
  g = np.load(path)
  A, sims = int(g['A']), int(g['sims'])
  if variant == 'split_f16' and A > 13:
    pytest.skip('k_search_h2 covers action_space <= 13 (MZ_H2_MAXA)')
  M = g['action'].shape[0]
  eng, kind, lt = golden_engine(g, M, variant)
  eng.root_load(g['root_value'], g['root_logits'])       # (no hidden state: the injected run never looks at the network)
  eng.root_prepare(g['to_play'], g['legal'], g['noise'])
  vals = np.zeros((M, sims + 1, 2 + A), np.float32)
  vals[:, 1:, 0] = g['sim_value']
  vals[:, 1:, 1] = g['sim_reward']
  vals[:, 1:, 2:] = g['sim_logits']
  eng.sim_io('inject', values=vals)
  info = eng.search_kernel_info()
  assert info['kind'] == kind, info                       # the fused kernel is what mz_search launches now
  if lt is not None:
    assert info['lt'] == lt, info
  eng.search()
  with pytest.raises(RuntimeError, match='inject mode applies to mz_search only'):      # (the self-play loop refuses it loudly)
    eng.selfplay_reset(4, 1.0)
    eng.selfplay_steps(1)
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
  out = {k: v.cpu().numpy() for k, v in eng.finalize(g['temperature'], u).items()}
  assert np.array_equal(out['child_visits'], g['child_visits'])
  assert np.array_equal(out['root_value'], g['final_root_value'])
  assert np.array_equal(out['error'], g['error'])
  sampled = g['temperature'] != 0
  assert np.array_equal(out['action'][sampled], g['action'][sampled])
  for bi in np.where(~sampled)[0]:
    assert out['visit_counts'][bi, g['action'][bi]] == out['visit_counts'][bi].max()
  eng.sim_io('off')
  eng.close()


RANDOM_SHAPES = [
    # A, sims, two players, known bounds, B (full grid where the trees are small), init_value_score
    (4, 30, False, (None, None), 4096, 0.0),      # <14,1,4>, whole trees in LDS
    (6, 50, False, (None, None), 4096, 0.0),      # <14,1,8>, compact LDS trees (Pong-ram shapes)
    (9, 30, True, (-1.0, 1.0), 4096, 0.0),        # <15,1,16> two players, known bounds
    (8, 30, True, (None, None), 1024, 0.25),      # <15,1,8>, init_value_score != 0: the descent's general branch
    (12, 20, False, (None, None), 1024, 0.0),     # <16,1,16>
    (16, 20, False, (-3.0, 3.0), 1024, 0.0),      # <18,1,16>
    (18, 20, False, (None, None), 1024, 0.0),     # <18,2,32>: 32 lanes per tree, two passes
    (30, 10, True, (None, None), 512, 0.0),       # <21,2,32>
]


@pytest.mark.parametrize('variant', ['lds', 'pool', 'split_f16'])
@pytest.mark.parametrize('shape', RANDOM_SHAPES, ids=['A%d_s%d%s' % (s[0], s[1], '_2p' if s[2] else '') for s in RANDOM_SHAPES])
def test_fused_tree_code_vs_oracle_on_injected_random_outputs(shape, variant):
  """every fused instantiation of the dispatch table (launch_fused) on random 'network' outputs with exact ties, illegal
  root actions and both players, against the oracle's tree fed the same numbers: whole trees identical"""
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine
  A, sims, two, bounds, B, ivs = shape
  sw, split, kind, lt = VARIANTS[variant]
  if split and A > 13:
    pytest.skip('k_search_h2 covers action_space <= 13 (MZ_H2_MAXA)')
  rng = np.random.RandomState(11 + A)
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
  lg[rng.uniform(size=(B, sims)) < 0.1] = 0.5              # exact ties among the priors
  val[rng.uniform(size=(B, sims)) < 0.05] = 0.0
  rew[rng.uniform(size=(B, sims)) < 0.3] = 0.0             # (value 0 / reward 0: exact score ties between siblings)
  # chains: in the first 128 trees one action carries nearly all the prior at every node, so the search keeps extending ONE
  # path -- up to num_simulations + 1 nodes, more than the 16 (32) lanes a tree's backup handles per round (the r04 bug:
  # MinMaxStats lost the first round's nodes of such a path)
  lg[:128] = (rng.standard_normal((128, sims, A)) * 0.1).astype(np.float32)
  lg[:128, :, rng.randint(0, A)] += 9.0
  logits[:128] = lg[:128, 0]
  legal[:128] = 1
  noise[:128] = rng.dirichlet([0.25] * A, size=128)
  eng.root_load(v0, logits)
  eng.root_prepare(tp, legal, noise)
  vals = np.zeros((B, sims + 1, 2 + A), np.float32)
  vals[:, 1:, 0], vals[:, 1:, 1], vals[:, 1:, 2:] = val, rew, lg
  eng.sim_io('inject', values=vals)
  info = eng.search_kernel_info()
  assert info['kind'] == kind and (lt is None or info['lt'] == lt), info
  eng.search()
  t.root_expand(tp, logits, legal)
  t.add_noise(noise, 0.25)
  for s in range(sims):
    t.select()
    t.expand_backup(val[:, s], rew[:, s], lg[:, s])
  ex, eo = eng.export_tree(), t.export()
  EX = eo['EX'].astype(bool)
  assert t.margin().min() == 0.0                            # (the ties are there)
  assert (eo['N'][:128] > 0).sum(1).max() > min(sims, 17)   # (and the long chains: more visited nodes than one backup round)
  assert np.array_equal(ex['EX'].astype(bool), EX)
  for k in ('N', 'E', 'TP', 'W'):
    assert np.array_equal(ex[k][EX], eo[k][EX]), (k, info)
  assert np.array_equal(ex['R'].astype(np.float64)[EX], eo['R'][EX])
  assert np.array_equal(ex['minmax'], eo['minmax'])
  assert ulp_diff(ex['P'][EX], eo['P'][EX]).max() <= 8      # up to 30 exp() terms in the normaliser
  temp = rng.choice([1.0, 0.0], size=B)
  u = rng.uniform(size=B)
  out = {k: v.cpu().numpy() for k, v in eng.finalize(temp, u).items()}
  action, cv, rv, vc = t.finalize(temp, u)
  assert np.array_equal(out['visit_counts'], vc)
  assert np.array_equal(out['action'], action)
  assert np.array_equal(out['child_visits'], cv)
  assert np.array_equal(out['root_value'], rv)
  eng.sim_io('off')
  eng.close()


# ---------------------------------------------------------------------------------------------------------------------
LOG_SHAPES = {
    # BASELINE configs[1] / configs[3] shapes, configs[0]'s game on the device, and the wide (32 lanes per tree) path
    'lunar': dict(gold='g1_net_lunar', O=8, A=4, sims=30, u8=False, info=dict(kind='fused', lt=1, ks1=14, G=4)),
    'pong': dict(gold='g1_net_pong', O=128, A=6, sims=50, u8=True, info=dict(kind='fused', lt=2, ks1=14, G=8)),
    'ttt': dict(gold='g1_net_ttt', O=9, A=9, sims=30, u8=False, game=True, info=dict(kind='fused', lt=2, ks1=15, G=16)),
    'wide18': dict(gold=None, O=8, A=18, sims=20, u8=False, info=dict(kind='fused', lt=2, ks1=18, G=32)),
}


@pytest.mark.parametrize('shape,mode', [('lunar', 'exact'), ('lunar', 'split_f16'), ('pong', 'exact'), ('pong', 'split_f16'),
                                        ('ttt', 'exact'), ('wide18', 'exact'), ('lunar', 'pool'), ('pong', 'no_persist')])
def test_benchmarked_launch_replayed_exactly_on_its_logged_outputs(shape, mode):
  """Zero excluded trees.  The launch structure bench.py times -- whole moves inside one launch, 4096 environments = 256
  workgroups, 3 launches of 16 moves, a weight update before the third -- with the simulation log on; EVERY move of EVERY
  tree is replayed through the oracle's tree with the device's own network outputs, Dirichlet draw (noise log) and
  select_action uniform (Philox key restated).  'pool' / 'no_persist': the same loop as a hipGraph of root + search
  kernels per move (trees in the node pool resp. in LDS) -- the non-HEAD instantiations with the end of the move fused."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine, records_view
  sh = LOG_SHAPES[shape]
  O, A, sims = sh['O'], sh['A'], sh['sims']
  game = bool(sh.get('game'))
  B, T, seed, chunk, launches = 4096, 11, 4321, 16, 3
  split = mode == 'split_f16'
  sw = {'pool': {'MZ_NO_LDS_TREES': '1'}, 'no_persist': {'MZ_NO_PERSIST': '1'}}.get(mode, {})
  w0 = orc.load_weights(np.load(os.path.join(G, sh['gold'] + '.npz'))) if sh['gold'] else random_weights(O, A, 3)
  rng = np.random.RandomState(5)
  w1 = {k: (v * (1 + 0.01 * rng.standard_normal(v.shape))).astype(np.float32) for k, v in w0.items()}
  kw = dict(two_players=True, known_bounds=(-1.0, 1.0), discount=1.0) if game else {}
  with env_switches(**sw):
    eng = Engine(B, O, A, sims, seed=seed, split_f16=split, **kw)
  if game:
    eng.selfplay_set_env('tictactoe')
  eng.set_weights(w0)
  info = eng.search_kernel_info()
  want = dict(sh['info'])
  if split:
    want['kind'] = 'split_f16'
  if mode == 'pool':
    want['lt'] = 0
  assert info == want, (info, want)
  assert eng.selfplay_moves_per_launch() == (16 if mode in ('exact', 'split_f16') else 0)
  if sh['u8']:
    eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
  eng.selfplay_noise_log(True)
  eng.selfplay_reset(T, 1.0, stagger=True)
  nm = launches * chunk
  log = eng.sim_io('log', keep_moves=nm)
  for k in range(launches):
    if k == launches - 1:
      eng.set_weights(w1)
      eng.selfplay_export_trees(True)                      # (a run-time flag of the same instantiation: the last move's trees)
    eng.selfplay_steps(chunk)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  assert n == nm
  rec = buf[:n].numpy().copy()
  rv = records_view(rec, O, A)
  tree = eng.export_tree()
  io_all = log.cpu().numpy()
  frac = 0.25
  cfg = orc.tree_cfg(A, sims, two_players=game, known_bounds=(-1.0, 1.0) if game else (None, None),
                     discount=1.0 if game else 0.997)
  below = 0
  for m in range(nm):
    io = io_all[m]
    noise = eng.selfplay_noise(m)
    u = philox_action_uniform(seed, np.arange(B), m)
    if game:        # tic_tac_toe.py:24,27-28: observation = turn * board, legal = the empty cells; the mover travels in the record
      legal = (rec[m, :, :O] == 0).astype(np.uint8)
      to_play = rv['to_play'][m].astype(np.int8)
      assert np.all((noise > 0) == (legal > 0))
    else:
      legal, to_play = None, np.ones(B, np.int8)
    ref = replay_move(cfg, B, A, sims, io, noise, frac, to_play, legal, 1.0, u, want_tree=(m == nm - 1))
    below += int((ref['margin'] <= 1e-4).sum())
    assert np.array_equal(rv['child_visits'][m], ref['child_visits'].astype(np.float32)), (m, 'visit distribution')
    assert np.array_equal(rv['action'][m], ref['action']), (m, 'action')
    assert np.array_equal(rv['root_value'][m], ref['root_value']), (m, 'root value')            # float64, bit for bit
    assert np.array_equal(rv['error'][m], ref['root_value'] - ref['v0'].astype(np.float64)), (m, 'error')
    if ref['tree'] is not None:
      eo = ref['tree']
      EX = eo['EX'].astype(bool)
      assert np.array_equal(tree['EX'].astype(bool), EX)
      for k in ('N', 'E', 'TP', 'W'):
        assert np.array_equal(tree[k][EX], eo[k][EX]), k
      assert np.array_equal(tree['R'].astype(np.float64)[EX], eo['R'][EX])
      assert np.array_equal(tree['minmax'], eo['minmax'])
      assert ulp_diff(tree['P'][EX], eo['P'][EX]).max() <= 8
      assert np.array_equal(tree['noise'], noise)
  print('%s %s: %d moves x %d trees replayed on their logged network outputs, all identical (%d of them had a select_child '
        'decision within 1e-4 of a tie)' % (shape, mode, nm, B, below))
  out = os.environ.get('MZ_PARITY_REPORT')
  if out:
    with open(out, 'a') as f:
      f.write('%s %s: %d moves x %d trees identical on logged outputs; %d trees below the 1e-4 margin among them\n' % (shape, mode, nm, B, below))
  eng.sim_io('off')
  eng.close()


def test_random_sweep_of_the_dispatch_table():
  """scripts/fuzz_inject.py: 80 random configurations -- 1..32 actions, 1..62 simulations (search paths of up to 63 nodes: four
  rounds of the fused backup), tree counts that are not multiples of 16, one / two players, known bounds, init_value_score,
  illegal root actions, ties, chains; LDS whole / compact / pool placements and k_search_h2 -- every tree identical to the
  oracle's on the injected outputs."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'fuzz_inject.py'), '80', '3'], capture_output=True, text=True,
                       timeout=1200, cwd=root)
  assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
  assert 'configurations identical' in out.stdout.splitlines()[-1] and out.stdout.count('\nok ') + out.stdout.startswith('ok ') >= 70


def test_random_sweep_of_the_selfplay_loop():
  """scripts/fuzz_selfplay_log.py: 60 random shapes of the device self-play loop (1..32 actions, observations of 1..200 floats /
  bytes / packed bytes, 2..61 simulations, ragged tree counts, whole-moves launches and kernel-per-phase graphs, LDS and pool
  trees, exact and split-f16, temperature 1 and 0) -- every move of every tree equals the oracle's tree on its logged outputs."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'fuzz_selfplay_log.py'), '60', '5'], capture_output=True,
                       text=True, timeout=1200, cwd=root)
  assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
  assert 'configurations identical' in out.stdout.splitlines()[-1]
