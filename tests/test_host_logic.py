"""CPU tests of the host-side mirror of the reference interface: TicTacToe rules and Game bookkeeping
against the reference's recorded games (goldens g3), config helpers, the ray shim, shared storage."""
import glob
import os
import types

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), 'golden')
FILES = sorted(glob.glob(os.path.join(G, 'g3_game_*.npz')))


def make_cfg(**kw):
  from model_based_rl_amd.config import make_config
  argv = ['--environment', 'TicTacToe', '--two_players', '--known_bounds', '-1', '1', '--discount', '1', '--seed', '0']
  cfg = make_config(argv)
  for k, v in kw.items():
    setattr(cfg, k, v)
  return cfg


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_tictactoe_and_game_reproduce_reference_histories(path):
  """replaying the reference's recorded actions through envs.TicTacToe + game.Game gives the reference's
  observations, legal sets, to_play, rewards, dones, steps and history slices (flush rules included)."""
  from model_based_rl_amd.envs import TicTacToe
  from model_based_rl_amd.game import Game
  g = np.load(path)
  cfg = make_cfg(max_history_length=int(g['max_history_length']))
  env = TicTacToe()
  moves = g['action'].shape[0]
  m = 0
  flush_k = 0
  while m < moves:
    game = Game(env, cfg)
    in_game = 0
    while not game.terminal and m < moves:
      obs = np.float32(game.get_observation(-1))
      assert np.array_equal(obs, g['obs'][m])
      legal = np.isin(np.arange(9), env.legal_actions()).astype(np.uint8)
      assert np.array_equal(legal, g['legal'][m])
      assert game.to_play == int(g['to_play'][m])
      game.history.errors.append(float(g['error'][m]))
      game.apply(int(g['action'][m]))
      game.store_search_statistics(g['child_visits'][m], float(g['final_root_value'][m]))
      m += 1
      in_game += 1
      if (game.history_idx - game.previous_collect_to) == cfg.max_history_length or game.done or game.terminal:
        overlap = cfg.num_unroll_steps + cfg.td_steps
        collect_from = max(0, game.previous_collect_to - overlap) if not game.history.dones[game.previous_collect_to - 1] \
            else game.previous_collect_to
        h = game.get_history_sequence(collect_from)
        meta = g['flush_meta'][flush_k]
        assert collect_from == int(meta[2]) and in_game == int(meta[1])
        assert (-1 if game.done else overlap) == int(meta[3]) and int(game.terminal) == int(meta[4])
        pre = 'flush%d_' % flush_k
        assert np.array_equal(np.stack([np.float32(o) for o in h.observations]), g[pre + 'observations'])
        for f in ('actions', 'rewards', 'dones', 'steps', 'to_play', 'root_values', 'errors'):
          assert np.array_equal(np.asarray(getattr(h, f), g[pre + f].dtype), g[pre + f]), f
        assert np.array_equal(np.asarray(h.child_visits), g[pre + 'child_visits'])
        flush_k += 1
  assert flush_k == int(g['n_flushes'])


def test_config_defaults_and_temperature_schedule():
  from model_based_rl_amd.config import make_config
  c = make_config([])
  assert (c.num_simulations, c.discount, c.pb_c_base, c.pb_c_init) == (30, 0.997, 19652, 1.25)
  assert (c.root_dirichlet_alpha, c.root_exploration_fraction, c.max_history_length) == (0.25, 0.25, 500)
  assert c.value_support_size == 31 and c.reward_support_range[0] == -15
  assert c.known_bounds == [None, None] and c.action_space == 4 and c.obs_space == (8,)
  assert [c.visit_softmax_temperature(s) for s in (0, 15000, 15001, 30000, 30001)] == [1.0, 1.0, 0.5, 0.5, 0.25]


def test_select_action_matches_numpy_stream():
  from model_based_rl_amd.config import Config
  from model_based_rl_amd.mcts import Node
  root = Node(0)
  for a, n in zip((0, 2, 5), (3, 20, 7)):
    root.children[a] = Node(0.1); root.children[a].visit_count = n
  np.random.seed(5); a1 = Config.select_action(root, 1.0)
  np.random.seed(5)
  p = np.array([3, 20, 7]) / 30.0
  assert a1 == (0, 2, 5)[np.random.choice(3, p=p)]
  assert Config.select_action(root, 0) == 2


def test_rayshim_runs_calls_in_order_and_propagates_errors():
  from model_based_rl_amd import rayshim as ray

  class Counter(object):
    def __init__(self, start): self.v = start
    def add(self, k): self.v += k; return self.v
    def boom(self): raise ValueError('x')

  h = ray.remote(Counter).remote(10)
  refs = [h.add.remote(i) for i in range(5)]
  assert ray.get(refs) == [10, 11, 13, 16, 20]
  done, pending = ray.wait(refs, num_returns=5)
  assert len(done) == 5 and not pending
  with pytest.raises(ValueError):
    ray.get(h.boom.remote())


def test_shared_storage_surface():
  from model_based_rl_amd.shared_storage import SharedStorage
  s = SharedStorage(types.SimpleNamespace(num_actors=3))
  assert not s.is_ready()
  s.store_weights({'w': 1}, 7)
  w, step = s.get_weights(games=5, actor_key=2)
  assert s.is_ready() and w == {'w': 1} and step == 7
  assert s.get_stats('actor_games') == {0: 0, 1: 0, 2: 5} and s.get_stats()['training_step'] == 7


def test_store_search_statistics_takes_a_node_like_the_reference():
  """game.py:106-115: store_search_statistics(root) normalises the root children's visit counts over the whole action
  space (0 for illegal actions) and keeps root.value(); the batched actor's (child_visits, root_value) form stores the same."""
  from model_based_rl_amd.game import Game
  from model_based_rl_amd.mcts import Node
  from model_based_rl_amd.envs import TicTacToe
  cfg = make_cfg()
  root = Node(0)
  for a, n in ((0, 3), (4, 12), (8, 15)):
    root.children[a] = Node(0.1)
    root.children[a].visit_count = n
  root.visit_count, root.value_sum = 30, 7.5
  g1, g2 = Game(TicTacToe(), cfg), Game(TicTacToe(), cfg)
  g1.store_search_statistics(root)
  want = [0.1, 0, 0, 0, 0.4, 0, 0, 0, 0.5]
  assert g1.history.child_visits[-1] == want and g1.history.root_values[-1] == 0.25
  g2.store_search_statistics(want, 0.25)
  assert g2.history.child_visits == g1.history.child_visits and g2.history.root_values == g1.history.root_values
  assert g1.sum_values == 0.25 and g1.max_value == 0.25


def test_bench_times_the_products_entry_points():
  """bench.py's timed regions contain no loop of their own: between the two clock reads of `timed_regions` there is ONE call
  of Actor.launch (the product's self-play loop, reference actors.py:87-124) and the barriers; the learner line times ONE
  call of Learner.launch (learners.py:115-153) per run.  No engine / replay primitive is called from a timed region."""
  import ast
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  tree = ast.parse(open(os.path.join(root, 'bench.py')).read())
  fn = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == 'timed_regions']
  assert len(fn) == 1
  loops = [n for n in ast.walk(fn[0]) if isinstance(n, (ast.For, ast.While))]
  assert len(loops) == 1 and isinstance(loops[0], ast.For)                 # `for _ in range(n_runs)`: one region per pass
  calls = [n.func.attr for n in ast.walk(fn[0]) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)]
  assert calls.count('launch') == 1
  # (all_gather / zeros_like: the per-rank frames and clocks of the N > 1 line, collected AFTER the region's second clock read)
  assert set(calls) <= {'launch', 'barrier', 'frames', 'perf_counter', 'process_time', 'tensor', 'all_reduce', 'all_gather', 'zeros_like',
                        'item', 'append'}, set(calls)
  src = open(os.path.join(root, 'bench.py')).read()
  assert 'class Pipeline' not in src and 'ingest_records(' not in src and 'selfplay_steps(' not in src
  # the learner line: the timed statement is learner.launch.remote(updates) through the handle
  lsrc = open(os.path.join(root, 'bench_learner.py')).read()
  assert 'learner.launch.remote(updates)' in lsrc and 'update_weights' not in lsrc and 'learner_graph_speed' not in lsrc


def test_load_state_overrides_accept_both_flag_forms(tmp_path, capsys):
  """train --load_state: `--training_steps=8` overrides the checkpoint's config like `--training_steps 8` does, and a flag
  that cannot override it (the checkpoint's config is the run's config, reference train.py:130-134) is reported, not
  silently dropped (r04 advice).  The parsing only: no GPU."""
  import types
  import torch
  from model_based_rl_amd import train
  from model_based_rl_amd.config import make_config
  saved = make_config(['--environment', 'TicTacToe', '--seed', '1', '--training_steps', '100', '--batch_size', '32'])
  path = str(tmp_path / 'ckpt')
  torch.save({'config': saved, 'training_step': 0}, path)
  seen = {}
  orig = train.launch
  train.launch = lambda cfg, *a, **k: seen.setdefault('cfg', cfg)
  try:
    train.main(['--environment', 'TicTacToe', '--load_state', path, '--training_steps=8', '--batch_size', '64'])
  finally:
    train.launch = orig
  assert seen['cfg'].training_steps == 8 and seen['cfg'].batch_size == 32
  assert '--batch_size' in capsys.readouterr().err


def test_chunk_schedule_and_vectorised_game_statistics():
  """Host logic of Actor.run_selfplay's device loop (no GPU): the chunk schedule (equal chunks of at most `chunk` moves; the
  rank without an actor, train._CollectiveOnly, walks the same one) and the vectorised game statistics of a chunk of records
  (Actor._log_games: games/{return,length,avg_value,max_value}, reference actors.py:99-117) against the move-by-move loop it
  replaced, on random chunks with none, few and many game ends."""
  import types
  from model_based_rl_amd.actors import Actor, chunk_schedule, selfplay_chunk
  assert list(chunk_schedule(20, 16)) == [10, 10] and list(chunk_schedule(48, 16)) == [16, 16, 16] and list(chunk_schedule(0, 16)) == []
  assert sum(chunk_schedule(3001, 16)) == 3001 and max(chunk_schedule(3001, 16)) <= 16
  it = chunk_schedule(None, 8)
  assert [next(it) for _ in range(3)] == [8, 8, 8]
  assert selfplay_chunk(types.SimpleNamespace(architecture='FCNetwork')) == 16
  assert selfplay_chunk(types.SimpleNamespace(architecture='MuZeroNetwork')) == 1
  assert selfplay_chunk(types.SimpleNamespace(architecture='FCNetwork', selfplay_chunk=4)) == 4

  def sequential(self, rv):          # the loop of rounds 1-4, one move at a time
    B = rv['done'].shape[1]
    if self._game_stats is None:
      self._game_stats = {'ret': np.zeros(B), 'len': np.zeros(B), 'sumv': np.zeros(B), 'maxv': np.full(B, -np.inf)}
    st = self._game_stats
    f = max(1, self.config.actor_log_frequency)
    for m in range(rv['done'].shape[0]):
      st['ret'] += rv['reward'][m]; st['len'] += 1; st['sumv'] += rv['root_value'][m]
      st['maxv'] = np.maximum(st['maxv'], rv['root_value'][m])
      d = rv['done'][m] != 0
      k = int(d.sum())
      if k:
        self.games_played += k
        if self.games_played // f != (self.games_played - k) // f:
          self.log_points([('games/return', self.games_played, st['ret'][d].mean()), ('games/length', self.games_played, st['len'][d].mean()),
                           ('games/avg_value', self.games_played, (st['sumv'][d] / st['len'][d]).mean()),
                           ('games/max_value', self.games_played, st['maxv'][d].mean())])
        st['ret'][d] = 0; st['len'][d] = 0; st['sumv'][d] = 0; st['maxv'][d] = -np.inf

  class Fake(object):
    def __init__(self, f):
      self._game_stats, self.games_played, self.log = None, 0, []
      self.config = types.SimpleNamespace(actor_log_frequency=f)
    def log_points(self, pts):
      self.log += [(t, int(i), float(v)) for t, i, v in pts]

  rng = np.random.RandomState(0)
  for trial in range(120):
    B, f, p = int(rng.randint(1, 40)), int(rng.choice([1, 3, 50])), float(rng.choice([0.0, 0.05, 0.3, 0.9]))
    a, b = Fake(f), Fake(f)
    for _ in range(5):
      M = int(rng.randint(1, 17))
      rv = {'done': (rng.rand(M, B) < p).astype(np.int32), 'reward': rng.randn(M, B).astype(np.float32), 'root_value': rng.randn(M, B)}
      sequential(a, rv)
      Actor._log_games(b, rv)
    assert a.games_played == b.games_played and len(a.log) == len(b.log)
    for x, y in zip(a.log, b.log):
      assert x[0] == y[0] and x[1] == y[1] and abs(x[2] - y[2]) <= 1e-9 * (1 + abs(x[2])), (x, y)
    for k in a._game_stats:
      assert np.allclose(a._game_stats[k], b._game_stats[k], rtol=1e-12, atol=1e-12), k


def test_replay_threads_default_follows_the_batch_size():
  """replay_buffer.default_ingest_threads: --ingest_threads wins; else 4, and 8 from batch size 1024 up (sampling and the priority
  refresh of such a batch are dealt to the replay's threads: the learner loop's host side), never more than the usable cores less one"""
  import types
  from model_based_rl_amd.replay_buffer import default_ingest_threads
  from model_based_rl_amd.distributed import usable_cores
  cap = max(1, usable_cores() - 1)
  assert default_ingest_threads(types.SimpleNamespace(ingest_threads=3, batch_size=4096)) == 3
  assert default_ingest_threads(types.SimpleNamespace(ingest_threads=None, batch_size=256)) == min(4, cap)
  assert default_ingest_threads(types.SimpleNamespace(ingest_threads=None, batch_size=2048)) == min(8, cap)
  assert default_ingest_threads(None) == min(4, cap)
