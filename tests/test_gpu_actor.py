"""GPU: the reference-shaped surface on top of the engine.

test_actor_reproduces_reference_games -- a one-environment Actor in parity mode (numpy's global stream, the
reference's seeds) replays the reference's recorded TicTacToe self-play games (goldens g3): same actions, same
visit distributions, same history slices reaching the replay buffer, same replay tree totals.  This is the
whole path a1-a18 of SURVEY.md s8 against the reference itself.
"""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


class RecordingReplay(object):
  def __init__(self, inner):
    self.inner, self.calls = inner, []

  def save_history(self, history, ignore=None, terminal=False):
    self.calls.append((history, ignore, terminal))
    self.inner.save_history(history, ignore=ignore, terminal=terminal)


@pytest.mark.parametrize('gi,temp,envs', [(0, 1.0, 1), (1, 0.1, 1), (2, 0.0, 1), (3, 1.0, 1), (0, 1.0, 3)])
def test_actor_reproduces_reference_games(gi, temp, envs):
  # (envs = 3: Actor.play_game(game) with ONE Game, the reference's call (actors.py:126), on an actor whose engine searches
  # three environments in lock-step -- the rows without a game of their own are ignored)
  from oracle import oracle as orc
  from model_based_rl_amd.actors import Actor
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  from model_based_rl_amd.shared_storage import SharedStorage
  g = np.load(os.path.join(G, 'g3_game_ttt_%d.npz' % gi))
  cfg = make_config(['--environment', 'TicTacToe', '--two_players', '--known_bounds', '-1', '1', '--discount', '1',
                     '--num_simulations', '30', '--seed', '0', '--num_envs', str(envs), '--parity_rng', '--window_size', '60000',
                     '--max_history_length', str(int(g['max_history_length'])), '--fixed_temperatures', str(temp)])
  storage = SharedStorage(cfg)
  storage.store_weights(orc.load_weights(g), 0)
  replay = RecordingReplay(PrioritizedReplay(cfg))
  actor = Actor(0, cfg, storage, replay)
  actor.sync_weights(force=True)
  moves = g['action'].shape[0]
  m = 0
  same_moves = 0
  diverged_at = None
  for game_no in range(3):
    game = cfg.new_game(actor.environments[0])
    actor.play_game(game)
    h = game.history
    n = len(h.actions)
    if diverged_at is None:
      for i in range(n):      # move by move: visit distribution and action are the reference's, bit for bit
        if h.actions[i] != int(g['action'][m + i]) or not np.array_equal(np.asarray(h.child_visits[i]), g['child_visits'][m + i]):
          diverged_at = m + i
          break
        same_moves += 1
      if diverged_at is None:
        # root values / errors: one step of the float32 inverse-transform staircase (tests/test_oracle_net.py)
        assert np.abs(np.asarray(h.root_values) - g['final_root_value'][m:m + n]).max() <= 5e-4
        assert np.abs(np.asarray(h.errors) - g['error'][m:m + n]).max() <= 5e-4
    m += n
    if m >= moves:
      break
  print('g3_game_ttt_%d: %d of %d moves identical%s' % (gi, same_moves, moves, '' if diverged_at is None else
        ', first difference at move %d (min top-2 margin there %.3g)' % (diverged_at, g['min_margin'][diverged_at])))
  # The engine's network agrees with PyTorch-CPU to ~1e-6: a game can only leave the reference's at a decision whose two
  # best scores were closer than that noise can bridge (a flipped near-tie changes the game and the numpy stream from
  # there on).  The goldens carry the smallest top-2 score gap of every move (MCTS.select_child, mcts.py:104-113).
  if diverged_at is not None:
    assert g['min_margin'][diverged_at] < 1e-5, (diverged_at, g['min_margin'][diverged_at])      # (measured: no divergence at all)
  safe_prefix = int(np.argmax(g['min_margin'] < 1e-4)) if np.any(g['min_margin'] < 1e-4) else moves
  assert same_moves >= safe_prefix, (same_moves, safe_prefix)          # 100 % identity up to the first near-tie
  if diverged_at is None:
    assert same_moves == moves
    assert len(replay.calls) == int(g['n_flushes'])
    for k, (hist, ignore, terminal) in enumerate(replay.calls):
      meta = g['flush_meta'][k]
      assert (-1 if ignore is None else ignore) == int(meta[3]) and int(terminal) == int(meta[4])
      assert np.array_equal(np.asarray(hist.actions), g['flush%d_actions' % k])
    assert replay.inner.size() == int(g['flush_meta'][-1][5])
    # total priority = sum(|error| + eps): every error may sit one staircase step (1.5e-4 (1 + |v|), twice: the root
    # value and the initial value both come out of the float32 inverse transform) away from the reference's
    bound = float(np.sum(2 * 1.5e-4 * (1 + np.abs(g['final_root_value']))))
    assert abs(replay.inner.tree.total_priority - float(g['replay_total'])) <= bound, bound
  actor.engine.close()


class ReplayNet(object):
  """network object that returns the reference's recorded outputs (a fake for MCTS.run)."""
  def __init__(self, g, m):
    from model_based_rl_amd.networks import NetworkOutput
    import torch
    self.g, self.m, self.s, self.NO, self.torch = g, m, 0, NetworkOutput, torch

  def recurrent_inference(self, hidden, action):
    g, m, s, t = self.g, self.m, self.s, self.torch
    assert int(action[0]) == int(g['sim_action'][m, s])
    assert np.array_equal(hidden.numpy().reshape(-1), g['sim_parent_hidden'][m, s])
    self.s += 1
    return self.NO(t.tensor([[g['sim_value'][m, s]]]), t.tensor([[g['sim_reward'][m, s]]]),
                   t.from_numpy(g['sim_logits'][m, s][None].copy()), t.from_numpy(g['sim_hidden'][m, s][None].copy()))


@pytest.mark.parametrize('name', ['g2_tree_ttt_fc', 'g2_tree_fake_1p', 'g2_tree_fake_bounds'])
def test_mcts_run_frontend_matches_reference_tree(name):
  """MCTS(config).run(root, network) with Node objects: visit counts, value sums, MinMax, search paths."""
  import torch
  from model_based_rl_amd.mcts import MCTS, Node
  from model_based_rl_amd.networks import NetworkOutput
  g = np.load(os.path.join(G, name + '.npz'))
  A, sims = int(g['A']), int(g['sims'])
  kb = [None if np.isnan(x) else float(x) for x in g['known_bounds']]
  cfg = types.SimpleNamespace(num_simulations=sims, action_space=A, two_players=bool(g['two_players']),
                              known_bounds=kb, discount=float(g['discount']), pb_c_base=float(g['pb_c_base']),
                              pb_c_init=float(g['pb_c_init']), init_value_score=float(g['init_value_score']))
  mcts = MCTS(cfg)
  for m in range(0, g['action'].shape[0], 5):
    root = Node(0)
    init = NetworkOutput(torch.tensor([[g['root_value'][m]]]), 0, torch.from_numpy(g['root_logits'][m][None].copy()),
                         torch.from_numpy(g['root_hidden'][m][None].copy()))
    legal = np.flatnonzero(g['legal'][m])
    root.expand(init, int(g['to_play'][m]), legal)
    for a in legal:                                   # the recorded Dirichlet draw instead of a fresh one
      root.children[int(a)].prior = root.children[int(a)].prior * (1 - float(g['frac'])) + g['noise'][m, a] * float(g['frac'])
    paths = mcts.run(root, ReplayNet(g, m))
    assert [len(p) - 1 for p in paths] == list(g['leaf_depth'][m])
    counts = np.zeros(A, np.int64)
    for a, ch in root.children.items():
      counts[a] = ch.visit_count
    assert np.array_equal(counts, g['tree_N'][m, 1:1 + A])
    assert root.value() == float(g['final_root_value'][m])
    assert (mcts.min_max_stats.minimum, mcts.min_max_stats.maximum) == tuple(g['minmax'][m])
    assert paths[-1][-1].expanded() and paths[0][0] is root


def test_device_selfplay_records_and_sharding():
  """on-device self-play loop: record contents vs the synthetic-env definition, episode bookkeeping, and
  shard invariance: envs [16,32) searched by a second engine with env_id_offset 16 produce the records the
  one-engine run produced for those env ids."""
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine, records_view, REC_EXTRA
  g = np.load(os.path.join(G, 'g1_net_lunar.npz'))
  w = orc.load_weights(g)
  O, A, T, moves = 8, 4, 5, 12

  def run(B, off):
    eng = Engine(B, O, A, 30, seed=77, env_id_offset=off)
    eng.set_weights(w)
    eng.selfplay_export_trees(True)
    eng.selfplay_reset(T, 1.0, stagger=False)
    eng.selfplay_steps(moves)
    buf, n = eng.selfplay_drain()
    import torch
    torch.cuda.synchronize()
    return eng, buf[:n].numpy().copy()

  eng, rec = run(32, 0)
  assert rec.shape == (moves, 32, O + A + REC_EXTRA)
  rv = records_view(rec, O, A)
  for m in range(moves):
    assert np.array_equal(rv['step'][m], np.full(32, m % T))
    assert np.array_equal(rv['episode'][m], np.full(32, m // T))
    assert np.array_equal(rv['done'][m], np.full(32, int(m % T == T - 1)))
    assert np.array_equal(rv['env_id'][m], np.arange(32))
  for (m, b) in [(0, 0), (3, 7), (7, 31), (11, 16)]:
    obs, rew = eng.synth_obs(b, m // T, m % T)
    assert np.array_equal(rec[m, b, :O], obs) and rv['reward'][m, b] == np.float32(rew)
  cv = rv['child_visits']
  assert np.allclose(cv.sum(-1), 1.0, atol=1e-6) and np.all(cv * 30 == np.round(cv * 30))
  acts = rv['action']
  assert np.all(np.take_along_axis(cv, acts[..., None], -1) > 0)
  assert 0.15 < np.abs(rec[..., :O]).mean() / 0.8 < 1.5                 # roughly unit-variance observations
  # last move against the oracle, with the Dirichlet draw the device used
  noise = eng.export_tree()['noise']
  t = orc.Trees(orc.tree_cfg(A, 30), 32)
  t.search_fc(orc.FCNet(w, O, A), rec[-1, :, :O], np.ones(32, np.int8), None, noise, 0.25)
  _, cvo, rvo, _ = t.finalize(1.0, np.zeros(32))
  # margin rule (tests/test_gpu_search.py): 100 % identity in every tree whose closest select_child decision was
  # further than 1e-4 from a tie; the rest is counted
  wide = t.margin() > 1e-4
  same = np.all(cvo.astype(np.float32) == cv[-1], axis=1)
  print('device self-play, last move vs oracle: %d of 32 trees above the margin, all identical: %s; below: %d of %d identical'
        % (wide.sum(), bool(np.all(same[wide])), (same & ~wide).sum(), (~wide).sum()))
  assert wide.sum() >= 16 and np.all(same[wide]), (np.flatnonzero(wide & ~same), t.margin()[wide & ~same])
  assert np.abs(rvo - rv['root_value'][-1])[wide].max() < 5e-4
  eng.close()
  eng2, rec2 = run(16, 16)
  assert np.array_equal(rec2.view(np.int32), rec[:, 16:32].view(np.int32))      # (bit patterns: the float64 halves may read as NaN)
  eng2.close()


def test_device_dirichlet_matches_numpy_distribution():
  """throughput mode draws the root noise on the device (Marsaglia-Tsang gamma on Philox, float32 transcendentals): it
  is Dirichlet(alpha * 1_A) like mcts.py:59's np.random.dirichlet -- sums to 1, mean 1/A, variance and quantiles of
  the marginal Beta(alpha, alpha (A - 1)) against numpy's own sampler."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine
  g = np.load(os.path.join(G, 'g1_net_lunar.npz'))
  w = orc.load_weights(g)
  B, A, alpha = 4096, 4, 0.25
  eng = Engine(B, 8, A, 30, seed=11)
  eng.set_weights(w)
  xs = []
  for m in range(8):
    eng.initial_inference(torch.randn(B, 8, device='cuda'))
    eng.root_prepare(None, None, None, device_rng=True, move=m)
    xs.append(eng.export_tree()['noise'])
  eng.close()
  x = np.concatenate(xs)
  assert not np.isnan(x).any() and np.abs(x.sum(1) - 1).max() < 1e-12 and x.min() >= 0
  assert np.abs(x.mean(0) - 1 / A).max() < 0.01
  assert np.abs(x.var(0) - (1 / A) * (1 - 1 / A) / (alpha * A + 1)).max() < 0.005
  ref = np.random.RandomState(0).dirichlet([alpha] * A, size=x.shape[0])
  for q in (0.25, 0.5, 0.75, 0.9, 0.99):
    assert abs(np.quantile(x[:, 1], q) - np.quantile(ref[:, 1], q)) < 0.02, q
  assert np.abs(np.corrcoef(x[:-B, 0], x[B:, 0])[0, 1]) < 0.03       # consecutive moves of an env: independent draws


def test_timed_launches_are_the_same_moves():
  """mz_selfplay_steps_timed (eager launches, events around every search dispatch -- bench.py's roofline clock) and
  mz_search_timed produce exactly what the untimed entry points produce, and report plausible durations."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine
  g = np.load(os.path.join(G, 'g1_net_lunar.npz'))
  w = orc.load_weights(g)
  recs = []
  for timed in (False, True):
    eng = Engine(64, 8, 4, 30, seed=5)
    eng.set_weights(w)
    eng.selfplay_reset(7, 1.0, stagger=True)
    if timed:
      ms = eng.selfplay_steps_timed(6)
      assert len(ms) == 6 and all(0.01 < x < 50.0 for x in ms), ms
    else:
      eng.selfplay_steps(6)
    buf, n = eng.selfplay_drain()
    torch.cuda.synchronize()
    recs.append(buf[:n].numpy().copy())
    eng.close()
  assert np.array_equal(recs[0].view(np.int32), recs[1].view(np.int32))
  trees = []
  obs = np.random.RandomState(2).standard_normal((64, 8)).astype(np.float32)
  noise = np.random.RandomState(3).dirichlet([0.25] * 4, size=64)
  for timed in (False, True):
    eng = Engine(64, 8, 4, 30)
    eng.set_weights(w)
    eng.initial_inference(obs)
    eng.root_prepare(None, None, noise)
    if timed:
      assert 0.01 < eng.search_timed() < 50.0
    else:
      eng.search()
    trees.append(eng.export_tree())
    eng.close()
  for k in ('N', 'W', 'E', 'minmax'):
    assert np.array_equal(trees[0][k], trees[1][k]), k


@pytest.mark.parametrize('A,sims,T,temp,ns,split', [(4, 30, 5, 1.0, False, False), (6, 12, 4, 0.5, False, False),
                                                    (3, 9, 3, 0.0, False, False), (18, 8, 3, 1.0, False, False),
                                                    (4, 30, 5, 1.0, True, False), (4, 30, 5, 1.0, False, True),
                                                    (6, 12, 4, 0.5, False, True)])
def test_selfplay_loop_equals_stepwise_abi(A, sims, T, temp, ns, split):
  """The fused per-move kernels of the device loop (root kernel with in-kernel observation + Dirichlet + first
  descent; search kernel whose tail samples the action, steps the env and writes the record) against the
  stepwise C ABI on the same engine configuration: mz_initial_inference -> mz_root_prepare(device RNG) ->
  mz_search -> mz_finalize, fed with mz_synth_obs.  Same device functions, same keys => every record field is
  bit-identical."""
  import torch
  from model_based_rl_amd.engine import Engine
  from model_based_rl_amd.networks import FCNetwork
  from model_based_rl_amd.engine import flatten_weights, records_view
  O, B, moves = 8, 48, 7
  torch.manual_seed(5)
  cfg = types.SimpleNamespace(value_support=(-15, 15), reward_support=(-15, 15), no_support=ns,
                              no_target_transform=False)
  net = FCNetwork(O, A, torch.device('cpu'), cfg)
  flat = flatten_weights(net.state_dict())
  loop = Engine(B, O, A, sims, seed=99, env_id_offset=7, no_support=ns, split_f16=split)
  loop.set_weights(flat)
  loop.selfplay_reset(T, temp, stagger=False)
  loop.selfplay_steps(moves)
  buf, n = loop.selfplay_drain()
  torch.cuda.synchronize()
  rec = buf[:n].numpy().copy()
  loop.close()
  rv = records_view(rec, O, A)

  step = Engine(B, O, A, sims, seed=99, env_id_offset=7, no_support=ns, split_f16=split)
  step.set_weights(flat)
  for m in range(moves):
    obs = np.stack([step.synth_obs(7 + b, m // T, m % T)[0] for b in range(B)])
    assert np.array_equal(rec[m, :, :O], obs)
    step.initial_inference(torch.from_numpy(obs).cuda())
    step.root_prepare(None, None, None, device_rng=True, move=m)
    step.search()
    out = step.finalize(temp, None, move=m)
    torch.cuda.synchronize()
    assert np.array_equal(rv['action'][m], out['action'].cpu().numpy())
    assert np.array_equal(rec[m, :, O:O + A], out['child_visits'].cpu().numpy().astype(np.float32))
    # root value and error travel as float64 (actors.py:147-148 keeps Python floats): nothing is narrowed
    assert np.array_equal(rv['root_value'][m], out['root_value'].cpu().numpy())
    assert np.array_equal(rv['error'][m], out['error'].cpu().numpy())
  step.close()


def test_selfplay_temperature_takes_effect_at_episode_start():
  """actors.py:128-129 evaluates config.visit_softmax_temperature(training_step) once per game: a temperature set in
  the middle of a run (mz_selfplay_set_temperature) reaches every environment at ITS next episode start -- envs are
  staggered, so they switch at different moves -- and the sampled actions equal the stepwise ABI run with exactly
  those per-env temperatures."""
  import torch
  from model_based_rl_amd.engine import Engine, records_view
  from oracle import oracle as orc
  g = np.load(os.path.join(G, 'g1_net_lunar.npz'))
  w = orc.load_weights(g)
  O, A, B, sims, T = 8, 4, 40, 30, 6
  loop = Engine(B, O, A, sims, seed=21)
  loop.set_weights(w)
  loop.selfplay_reset(T, 1.0, stagger=True)
  loop.selfplay_steps(4)
  loop.selfplay_set_temperature(0.25)            # "training_step crossed a boundary of the schedule" (config.py:41-49)
  loop.selfplay_steps(9)
  loop.selfplay_set_temperature(0.0)
  loop.selfplay_steps(7)
  buf, n = loop.selfplay_drain()
  torch.cuda.synchronize()
  rec = buf[:n].numpy().copy()
  loop.close()
  rv = records_view(rec, O, A)
  moves = rec.shape[0]
  # the temperature env b plays move m with: the value in force when the episode containing m started
  in_force = np.array([1.0] * 4 + [0.25] * 9 + [0.0] * 7)
  temp = np.zeros((moves, B))
  cur = np.full(B, 1.0)
  for m in range(moves):
    temp[m] = cur
    cur = np.where(rv['done'][m] != 0, in_force[m], cur)      # the setter is ordered before move m+1's kernels
  assert len({tuple(np.unique(temp[:, b], return_index=True)[1]) for b in range(B)}) > 1    # envs switch at different moves
  step = Engine(B, O, A, sims, seed=21)
  step.set_weights(w)
  for m in range(moves):
    step.initial_inference(torch.from_numpy(rec[m, :, :O].copy()).cuda())
    step.root_prepare(None, None, None, device_rng=True, move=m)
    step.search()
    out = step.finalize(temp[m], None, move=m)
    torch.cuda.synchronize()
    assert np.array_equal(rv['action'][m], out['action'].cpu().numpy()), m
  # and it matters: with the wrong temperatures the actions differ somewhere
  step.close()
  sharp = rv['child_visits'][temp == 0.0]
  acts = rv['action'][temp == 0.0]
  assert np.all(np.take_along_axis(sharp, acts[:, None], -1)[:, 0] == sharp.max(-1))     # T = 0: arg-max of the visits


def test_selfplay_uint8_observations_and_norm_obs():
  """The -ram- shapes: byte-valued synthetic observations and --norm_obs 0 255 inside the root
  kernel (actors.py:55-58,134-137: float32 (obs - min) / range); the record keeps the raw bytes.  Checked against
  the oracle's initial inference on host-normalised inputs and against the stepwise ABI."""
  import torch
  from model_based_rl_amd.engine import Engine, records_view
  from oracle import oracle as orc
  g = np.load(os.path.join(G, 'g1_net_pong.npz'))
  w = orc.load_weights(g)
  O, A, B, sims, T = 128, 6, 32, 10, 4
  eng = Engine(B, O, A, sims, seed=8)
  eng.set_weights(w)
  eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
  eng.selfplay_reset(T, 1.0, stagger=False)
  eng.selfplay_steps(3)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  rec = buf[:n].numpy().copy()
  rv = records_view(rec, O, A)
  raw = rec[..., :O]
  assert np.array_equal(raw, np.round(raw)) and raw.min() >= 0 and raw.max() <= 255 and raw.max() > 200
  assert abs(raw.mean() - 127.5) < 4.0
  for (m, b) in [(0, 0), (1, 5), (2, 31)]:
    assert np.array_equal(raw[m, b], eng.synth_obs(b, 0, m)[0])
  # root of the last move: value / logits the kernel produced vs the oracle on (raw - 0) / 255 in float32
  v_dev, lg_dev, h_dev = [x.cpu().numpy() for x in eng.root_outputs()]
  norm = (raw[-1] - np.float32(0.0)) / np.float32(255.0)
  h, v, lg = orc.FCNet(w, O, A).initial(norm)
  assert np.abs(h - h_dev).max() < 1e-5 and np.abs(lg - lg_dev).max() < 1e-5
  assert (np.abs(v - v_dev) < 1e-5).mean() >= 0.9 and np.abs(v - v_dev).max() < 1e-3
  # the same move through the stepwise ABI fed the host-normalised observation: identical action / visits
  step = Engine(B, O, A, sims, seed=8)
  step.set_weights(w)
  step.initial_inference(torch.from_numpy(norm).cuda())
  step.root_prepare(None, None, None, device_rng=True, move=2)
  step.search()
  out = step.finalize(1.0, None, move=2)
  assert np.array_equal(out['action'].cpu().numpy(), rv['action'][-1])
  assert np.array_equal(out['root_value'].cpu().numpy(), rv['root_value'][-1])
  eng.close(); step.close()


def test_small_ring_drain_on_copy_stream_is_ordered():
  """A drain on a copy stream only enqueues its D2H copy; when the ring is too small to keep the next moves out of
  the slots being copied, mz_selfplay_steps must order itself behind the copy (the event the drain records).  Wide
  records shrink the ring to its minimum; the records of a run with overlapped drains must equal those of a run
  that synchronises after every drain."""
  import torch
  from model_based_rl_amd.engine import Engine
  from model_based_rl_amd.networks import FCNetwork
  from model_based_rl_amd.engine import flatten_weights
  O, A, B, sims, T = 4000, 3, 2100, 2, 5
  torch.manual_seed(1)
  cfg = types.SimpleNamespace()
  flat = flatten_weights(FCNetwork(O, A, torch.device('cpu'), cfg).state_dict())

  def run(overlap):
    eng = Engine(B, O, A, sims, seed=4)
    eng.set_weights(flat)
    eng.selfplay_reset(T, 1.0, stagger=False)
    ring = eng.ring_moves
    chunk = ring // 2 + 3                      # two chunks do not fit the ring: the second reuses slots of the first
    cs = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    pinned = [torch.empty(chunk, B, eng.rec_floats, dtype=torch.float32).pin_memory() for _ in range(3)]
    for k in range(3):
      eng.selfplay_steps(chunk)
      # overlap == 2: the drains alternate between two copy streams (every drain has its own event; a drain on another
      # stream is chained behind the previous one, and a launch waits for the oldest drain that still covers its slots)
      buf, n = eng.selfplay_drain(pinned[k], chunk, copy_stream=(cs[k % overlap] if overlap else None))
      assert n == chunk
      if not overlap:
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    outs = [p.numpy()[:, ::97, :64].copy().view(np.int32) for p in pinned] + \
           [p.numpy()[:, ::97, -12:].copy().view(np.int32) for p in pinned]
    eng.close()
    return ring, outs

  ring, a = run(0)
  assert ring == 32
  for streams in (1, 2):
    _, b = run(streams)
    for x, y in zip(a, b):
      assert np.array_equal(x, y), streams


@pytest.mark.parametrize('O,A,B,sims,T,no_persist,ttt', [(8, 4, 512, 30, 7, False, False), (128, 6, 130, 12, 5, False, False),
                                                        (8, 4, 64, 10, 4, True, False), (9, 9, 48, 10, 9, False, True)])
def test_records_stored_into_pinned_memory_equal_the_drained_ring(O, A, B, sims, T, no_persist, ttt, monkeypatch):
  """mz_selfplay_steps_into (the kernels store each record through the device mapping of the caller's pinned buffer) is
  mz_selfplay_steps + mz_selfplay_drain bit for bit -- whole moves in one launch, the two kernels per move (MZ_NO_PERSIST)
  and the device TicTacToe alike -- may alternate with the ring on one engine, rejects pageable memory and refuses to
  overtake undrained moves."""
  import torch
  from model_based_rl_amd.engine import Engine, flatten_weights
  from model_based_rl_amd.networks import FCNetwork
  if no_persist:
    monkeypatch.setenv('MZ_NO_PERSIST', '1')
  torch.manual_seed(2)
  flat = flatten_weights(FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).state_dict())

  def make():
    eng = Engine(B, O, A, sims, seed=9, **({'two_players': True, 'discount': 1.0, 'known_bounds': (-1.0, 1.0)} if ttt else {}))
    eng.set_weights(flat)
    if ttt:
      eng.selfplay_set_env('tictactoe')
    eng.selfplay_reset(T, 1.0, stagger=True)
    return eng
  ring, direct = make(), make()
  chunks = [16, 5, 16, 1, 23]
  for i, k in enumerate(chunks):
    ring.selfplay_steps(k)
    want, n = ring.selfplay_drain(max_moves=k)
    assert n == k
    got = torch.full((k + 1, B, direct.rec_floats), 7.0).pin_memory()
    if i == 3:              # the ring in between: both paths on one engine
      direct.selfplay_steps(k)
      direct.selfplay_drain(got, k)
    else:
      direct.selfplay_steps_into(got, k)
    torch.cuda.synchronize()
    assert np.array_equal(want.numpy()[:k].view(np.int32), got.numpy()[:k].view(np.int32)), (i, k)
    assert (got.numpy()[k] == 7.0).all()          # nothing past the launch's moves
  with pytest.raises(ValueError):
    direct.selfplay_steps_into(torch.empty(4, B, direct.rec_floats), 4)
  import ctypes as C
  pageable = np.zeros((2, B, direct.rec_floats), np.float32)
  assert direct.lib.mz_selfplay_steps_into(direct._h, 2, pageable.ctypes.data_as(C.c_void_p), direct.stream) != 0
  assert b'page-locked' in direct.lib.mz_last_error()
  direct.selfplay_steps(3)
  with pytest.raises(RuntimeError, match='drain them first'):
    direct.selfplay_steps_into(torch.empty(3, B, direct.rec_floats).pin_memory(), 3)
  ring.close(); direct.close()


@pytest.mark.parametrize('base', [(1 << 30) - 20, (1 << 32) - 20])
def test_move_counter_across_its_word_boundaries(base, monkeypatch):
  """mz_selfplay_set_moves puts every environment's move counter close to 2^30 (where mz_selfplay_steps_into's 2^30-slot ring
  would wrap inside a launch: the host switches to its second modulus) and to 2^32 (the counter's high word: RNG keys, record
  slots): records stored into pinned memory == the drained device ring == the kernel-per-phase launch structure, bit for bit."""
  import torch
  from model_based_rl_amd.engine import Engine, flatten_weights
  from model_based_rl_amd.networks import FCNetwork
  O, A, B, sims, T = 8, 4, 80, 10, 7
  torch.manual_seed(3)
  flat = flatten_weights(FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).state_dict())

  def make():
    eng = Engine(B, O, A, sims, seed=11)
    eng.set_weights(flat)
    eng.selfplay_reset(T, 1.0, stagger=True)
    eng.selfplay_set_moves(base)
    return eng
  ring, direct = make(), make()
  monkeypatch.setenv('MZ_NO_PERSIST', '1')
  phases = make()
  assert ring.selfplay_moves_per_launch() == 16 and phases.selfplay_moves_per_launch() == 0
  for k in (16, 16, 7):                      # the second chunk crosses the boundary
    got = torch.zeros(k, B, direct.rec_floats).pin_memory()
    direct.selfplay_steps_into(got, k)
    outs = []
    for eng in (ring, phases):
      eng.selfplay_steps(k)
      rec, n = eng.selfplay_drain(max_moves=k)
      assert n == k
      outs.append(rec)
    torch.cuda.synchronize()
    for rec in outs:
      assert np.array_equal(rec.numpy()[:k].view(np.int32), got.numpy().view(np.int32)), k
  ring.selfplay_steps(2)
  with pytest.raises(RuntimeError, match='drain them first'):
    ring.selfplay_set_moves(5)
  for eng in (ring, direct, phases):
    eng.close()


def test_actor_load_state_metrics_and_run_dirs(tmp_path):
  """Actor.load_state (actors.py:75-79): weights, training step and this actor's game count come back from a learner
  checkpoint dictionary; the actor's scalars (actors.py:105-117) land in <run>/<worker>/metrics.csv and the game count is
  reported to the storage with the weight pulls (actors.py:82, shared_storage.py:12-14)."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.actors import Actor
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.logger import read_metrics
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  from model_based_rl_amd.shared_storage import SharedStorage
  g = np.load(os.path.join(G, 'g1_net_lunar.npz'))
  w = orc.load_weights(g)
  cfg = make_config(['--environment', 'LunarLander-v2', '--num_envs', '32', '--num_simulations', '6', '--episode_length',
                     '4', '--seed', '1', '--window_size', '4096', '--weight_sync_frequency', '8', '--runs_dir',
                     str(tmp_path / 'runs'), '--run_tag', 'r', '--num_actors', '2'])
  storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
  storage.store_weights({k: torch.from_numpy(v) for k, v in w.items()}, 3)
  actor = Actor(1, cfg, storage, replay)
  actor.launch(max_moves=16)
  assert actor.training_step == 3 and actor.games_played == 32 * 4 == storage.get_stats('actor_games')[1]
  base = tmp_path / 'runs' / 'LunarLander-v2' / 'r'
  m = read_metrics(str(base / 'actor-1' / 'metrics.csv'))
  assert set(m) == {'games/return', 'games/length', 'games/avg_value', 'games/max_value'}
  # (staggered starts: the first game of an environment is partial, every later one has episode_length moves)
  assert m['games/length'][-1][1] == 4.0 and all(v <= 4.0 for _, v in m['games/length'])
  assert m['games/return'][-1][0] == actor.games_played
  assert os.path.isfile(base / 'config' / 'config.json')
  # resume: a checkpoint dictionary as Learner.save_state writes it
  w2 = {k: torch.from_numpy(v * np.float32(0.5)) for k, v in w.items()}
  state = {'weights': w2, 'training_step': 41, 'actor_games': {0: 7, 1: 19}}
  resumed = Actor(1, cfg, storage, replay, state=state)
  assert resumed.training_step == 41 and resumed.games_played == 19
  assert resumed.dirs['base'] == str(base / 'resumed' / '41')
  obs = np.random.RandomState(0).standard_normal((32, 8)).astype(np.float32)
  resumed.engine.initial_inference(obs)
  v, lg, h = [x.cpu().numpy() for x in resumed.engine.root_outputs()]
  ho, vo, lgo = orc.FCNet({k: t.numpy() for k, t in w2.items()}, 8, 4).initial(obs)
  assert np.abs(h - ho).max() < 1e-5 and np.abs(lg - lgo).max() < 1e-5
  actor.engine.close(); resumed.engine.close()


def test_train_driver_selfplay_only():
  from model_based_rl_amd import train
  thr = train.main(['--environment', 'LunarLander-v2', '--num_envs', '64', '--num_simulations', '8', '--seed', '3',
                    '--episode_length', '6', '--max_moves', '24', '--window_size', '4096', '--selfplay_only'])
  assert thr['games'] >= 64 * 3 and 64 * 18 <= thr['frames'] <= 64 * 24


def test_train_driver_with_learner():
  """actors + native replay + learner in one process: the learner publishes weights, waits for
  stored_before_train experiences, trains a few steps on sampled batches and refreshes priorities."""
  from model_based_rl_amd import train
  thr = train.main(['--environment', 'LunarLander-v2', '--num_envs', '64', '--num_simulations', '8', '--seed', '3',
                    '--episode_length', '6', '--max_moves', '48', '--window_size', '8192', '--stored_before_train', '512',
                    '--batch_size', '32', '--learner_steps', '5', '--send_weights_frequency', '2', '--use_gpu_for',
                    'actors', 'learner', '--gpu_turns'])      # (one GPU for both: they take turns, gpu_turns.py)
  assert thr['frames'] >= 64 * 36
  lt = thr['learner']                 # the reference's own throughput scalars (learners.py:88-113)
  assert lt['frames_per_second'] > 0 and lt['updates_per_second'] > 0
  assert abs(lt['replay_ratio'] - lt['updates_per_second'] / lt['frames_per_second']) < 1e-9


def test_train_resumes_from_a_checkpoint(tmp_path):
  """`train --load_state <checkpoint>` (train.py:130-134): the run continues from the learner's checkpoint -- its config, weights,
  optimiser, training step and frame / game totals (learners.py:62-83), the actors' weights, step and game counts
  (actors.py:75-79) -- in a run directory `<run>/resumed/<step>` (learners.py:63)."""
  import glob
  from model_based_rl_amd import train
  common = ['--environment', 'LunarLander-v2', '--num_envs', '64', '--num_simulations', '8', '--seed', '3', '--episode_length', '6',
            '--window_size', '8192', '--stored_before_train', '512', '--batch_size', '32', '--send_weights_frequency', '2',
            '--save_state_frequency', '4', '--use_gpu_for', 'actors', 'learner', '--runs_dir', str(tmp_path), '--run_tag', 'r']
  thr = train.main(common + ['--max_moves', '-1', '--training_steps', '4'])
  ck = glob.glob(os.path.join(str(tmp_path), '**', 'saves', '4'), recursive=True)
  assert len(ck) == 1 and thr['frames'] >= 512
  thr2 = train.main(['--load_state', ck[0], '--max_moves', '-1', '--training_steps', '8'])
  ck2 = glob.glob(os.path.join(str(tmp_path), '**', 'resumed', '4', 'saves', '8'), recursive=True)
  assert len(ck2) == 1, glob.glob(os.path.join(str(tmp_path), '**', 'saves', '*'), recursive=True)
  import torch
  state4 = torch.load(ck[0], map_location='cpu', weights_only=False)
  state = torch.load(ck2[0], map_location='cpu', weights_only=False)
  assert state4['training_step'] == 4 and state['training_step'] == 8
  assert state['total_frames'] > state4['total_frames'] > 0 and state['total_games'] > state4['total_games']      # totals carried over (add_initial_throughput)
  assert 0 in state['actor_games'] and thr2['learner']['updates_per_second'] > 0 and thr2['games'] > thr['games'] // 2


def test_device_records_to_sampled_batch():
  """Device loop -> drain -> native ingest -> PrioritizedReplay.sample_batch: every sampled position is traced back to the
  experience record it came from (observations are unique) and its targets are recomputed from the raw records with
  replay_buffer.py:124-198 written out in numpy -- policy = the stored visit distribution or zeros past the end of the
  game (absorbing, 195-198), reward = the previous step's, value = discounted float32 reward sum + discount^td * root
  value at the bootstrap step (176-189), actions padded with draws past the end (150-151), IS weights
  (N * p / total)^-beta / max (157-160)."""
  import random
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine, records_view
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  g = np.load(os.path.join(G, 'g1_net_lunar.npz'))
  O, A, B, T, moves, K, td, disc = 8, 4, 48, 7, 40, 5, 10, 0.997
  eng = Engine(B, O, A, 12, seed=31)
  eng.set_weights(orc.load_weights(g))
  eng.selfplay_reset(T, 1.0, stagger=True)
  cfg = types.SimpleNamespace(batch_size=64, epsilon=0.01, alpha=1.0, beta=0.7, obs_space=(O,), action_space=A, window_size=4096,
                              window_step=None, num_unroll_steps=K, td_steps=td, max_history_length=500, discount=disc, seed=2)
  rep = PrioritizedReplay(cfg)
  recs = []
  for _ in range(moves // 8):
    eng.selfplay_steps(8)
    buf, n = eng.selfplay_drain()
    torch.cuda.synchronize()
    recs.append(buf[:n].numpy().copy())
    rep.ingest_records(recs[-1], n, B)
  eng.close()
  rec = np.concatenate(recs, 0)
  rv = records_view(rec, O, A)
  random.seed(9); np.random.seed(10)
  total, size, beta = rep.tree.total_priority, rep.size(), min(1.0, 0.7 + 0.001)
  (obs, actions, (t_rew, t_val, t_pol)), idxs, isw = rep.sample_batch()
  actions = np.asarray(actions)
  pri = rep.tree.leaves()[np.asarray(idxs) - (4096 - 1)]
  w = np.power(size * pri / total, -beta)
  assert np.allclose(isw, w / w.max(), rtol=0, atol=1e-15)
  key = {rec[m, b, :O].tobytes(): (m, b) for m in range(moves) for b in range(B)}
  discf = np.array([disc ** n for n in range(K + td)], np.float32)
  checked_absorbing = 0
  for i in range(64):
    m, b = key[obs[i].tobytes()]
    # the game this step belongs to: from the step after the previous done to the next done (histories are whole games here)
    end = m
    while not rv['done'][end, b]:
      end += 1                                  # (a sampled step always belongs to a finished, flushed game)
    n_game = end + 1                            # exclusive end, in move indices
    start = m
    while start > 0 and not rv['done'][start - 1, b]:
      start -= 1                                # first recorded move of that game = step 0 of its history slice
    assert abs(pri[i] - (abs(rv['error'][m, b]) + 0.01)) == 0
    for k in range(K):
      if m + k < n_game:
        assert actions[i, k] == rv['action'][m + k, b]
      else:
        assert 0 <= actions[i, k] < A
    for j in range(K + 1):
      cur = m + j
      last_reward = rv['reward'][cur - 1, b] if start < cur <= n_game else np.float32(0)      # replay_buffer.py:170-173
      assert t_rew[i, j] == last_reward, (i, j)
      if cur < n_game:
        boot = cur + td
        value = rv['root_value'][boot, b] * disc ** td if boot < n_game else 0.0
        hi = min(boot, n_game)
        acc = np.float32(0)
        for q in range(cur, hi):
          acc = np.float32(acc + np.float32(rv['reward'][q, b] * discf[q - cur]))
        assert abs(t_val[i, j] - np.float32(value + float(acc))) <= 1e-6
        assert np.array_equal(t_pol[i, j], rv['child_visits'][cur, b])
      else:
        checked_absorbing += 1
        assert t_val[i, j] == 0 and not t_pol[i, j].any()
  assert checked_absorbing > 0


@pytest.mark.parametrize('O,A,sims,u8,split', [(8, 4, 30, False, False), (128, 6, 50, True, False), (5, 2, 6, False, False),
                                               (8, 4, 30, False, True), (60, 6, 50, True, True),
                                               (9, 10, 12, False, False), (12, 7, 20, False, False)])
def test_persistent_selfplay_launch_equals_graph_of_kernels(O, A, sims, u8, split, monkeypatch):
  """Single-player self-play runs as whole moves inside ONE launch of the search kernel (its HEAD instantiation: root,
  simulations and end of every move, trees never leaving LDS between root and search); MZ_NO_PERSIST=1 keeps the
  hipGraph of root + search kernels per move.  Same device functions, same keys: every record is bit-identical --
  across launch boundaries, a weight update, a temperature change, ragged B, byte observations with --norm_obs, trees
  in LDS (LunarLander shapes), descent fields only in LDS (Pong-ram shapes) and trees smaller than the root's working
  set, the exact-f32 kernel and the split-f16 one; with tree export on, the exported trees agree too."""
  import torch
  from model_based_rl_amd.engine import Engine, flatten_weights
  from model_based_rl_amd.networks import FCNetwork
  B = 40
  torch.manual_seed(11)
  cfg = types.SimpleNamespace(value_support=(-15, 15), reward_support=(-15, 15), no_support=False, no_target_transform=False)
  w0 = flatten_weights(FCNetwork(O, A, torch.device('cpu'), cfg).state_dict())
  w1 = flatten_weights(FCNetwork(O, A, torch.device('cpu'), cfg).state_dict())
  out = []
  for persist in (True, False):
    if persist:
      monkeypatch.delenv('MZ_NO_PERSIST', raising=False)
    else:
      monkeypatch.setenv('MZ_NO_PERSIST', '1')
    eng = Engine(B, O, A, sims, seed=21, env_id_offset=3, split_f16=split)
    assert (eng.selfplay_moves_per_launch() > 0) == persist
    eng.set_weights(w0)
    if u8:
      eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
    eng.selfplay_reset(9, 1.0, stagger=True)
    recs = []
    for chunk, k in enumerate((16, 5, 1, 20)):          # 20 > 16: two launches inside one call
      if chunk == 1:
        eng.selfplay_set_temperature(0.5)
      if chunk == 2:
        eng.set_weights(w1)
      if chunk == 3:
        eng.selfplay_export_trees(True)
      eng.selfplay_steps(k)
      buf, n = eng.selfplay_drain()
      torch.cuda.synchronize()
      assert n == k
      recs.append(buf[:n].numpy().copy())
    tree = eng.export_tree()
    if persist and not eng.split_f16:
      ph = eng.selfplay_phase_profile(4)
      assert set(ph) == set(Engine.SELFPLAY_PHASES) and all(v > 0 for v in ph.values()), ph
      assert ph['simulations'] > 5 * ph['root_prediction']
    else:
      with pytest.raises(RuntimeError):
        eng.selfplay_phase_profile(4)
    eng.close()
    out.append((np.concatenate(recs, 0), tree))
  assert np.array_equal(out[0][0].view(np.int32), out[1][0].view(np.int32))
  for k in ('N', 'W', 'E', 'R', 'EX', 'minmax'):
    assert np.array_equal(out[0][1][k], out[1][1][k]), k
  ex = out[0][1]['EX'].astype(bool)            # (priors exist only where a node does)
  assert np.array_equal(out[0][1]['P'][ex], out[1][1]['P'][ex])


@pytest.mark.parametrize('record_copy', [False, True])
def test_actor_pipeline_equals_a_plain_engine_loop(tmp_path, monkeypatch, record_copy):
  """(record_copy: MZ_RECORD_COPY=1, the device ring + D2H copy on a copy stream instead of the kernels' stores into pinned memory)
  Actor.launch (the product's device loop: _RecordPipe -- launch-ahead chunks, pinned buffers, ingest on a worker thread) hands
  the replay exactly the records a plain engine loop produces, whatever the chunking: one launch of 48 moves, and 20 + 28 moves in
  two calls (the second continues episodes, record ring and pipeline), give the same record stream bit for bit, the same replay
  (every leaf priority, frames, games) as a replay fed by direct ingest, and the same game statistics."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.actors import Actor
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.engine import Engine
  from model_based_rl_amd.logger import read_metrics
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  from model_based_rl_amd.shared_storage import SharedStorage
  g = np.load(os.path.join(G, 'g1_net_lunar.npz'))
  w = {k: torch.from_numpy(v) for k, v in orc.load_weights(g).items()}
  B, moves = 96, 48
  if record_copy:
    monkeypatch.setenv('MZ_RECORD_COPY', '1')

  def cfg_of(tag):
    return make_config(['--environment', 'LunarLander-v2', '--num_envs', str(B), '--num_simulations', '10', '--episode_length', '7',
                        '--seed', '5', '--window_size', '16384', '--weight_sync_frequency', '16', '--runs_dir', str(tmp_path / tag),
                        '--run_tag', 'r', '--actor_log_frequency', '1'])

  def run(tag, calls):
    cfg = cfg_of(tag)
    storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
    storage.store_weights(w, 1)
    actor = Actor(0, cfg, storage, replay)
    seen = []
    actor.record_tap = lambda v: seen.append(v.copy())
    for n in calls:
      actor.launch(max_moves=n)
    rec = np.concatenate(seen, 0)
    m = read_metrics(os.path.join(actor.dirs['worker'], 'metrics.csv'))
    out = (rec, replay.tree.leaves(), replay.tree.total_priority, replay.get_throughput(), actor.games_played, m, actor.move_counter)
    actor.close(); actor.engine.close()
    return out

  one = run('a', [moves])
  two = run('b', [20, 28])
  assert one[0].shape == (moves, B, 8 + 4 + 10) and one[6] == two[6] == moves
  assert np.array_equal(one[0].view(np.int32), two[0].view(np.int32))
  assert np.array_equal(one[1], two[1]) and one[2] == two[2] and one[3] == two[3] and one[4] == two[4]
  for tag in ('games/return', 'games/length', 'games/avg_value', 'games/max_value'):
    assert len(one[5][tag]) == len(two[5][tag]) > 0
    assert all(a[0] == b[0] and abs(a[1] - b[1]) <= 1e-9 * (1 + abs(a[1])) for a, b in zip(one[5][tag], two[5][tag]))
  # the plain loop: engine, drain, direct ingest
  cfg = cfg_of('c')
  eng = Engine.from_config(cfg, B, seed=cfg.seed, env_id_offset=0)
  eng.set_weights(w)
  eng.selfplay_reset(cfg.episode_length, 1.0, stagger=True)
  replay = PrioritizedReplay(cfg)
  recs = []
  for _ in range(moves // 8):
    eng.selfplay_steps(8)
    buf, n = eng.selfplay_drain()
    torch.cuda.synchronize()
    recs.append(buf[:n].numpy().copy())
    replay.ingest_records(buf, n, B)
  eng.close()
  rec = np.concatenate(recs, 0)
  assert np.array_equal(rec.view(np.int32), one[0].view(np.int32))
  assert np.array_equal(replay.tree.leaves(), one[1]) and replay.tree.total_priority == one[2] and replay.get_throughput() == one[3]
  assert one[4] == int((rec[..., 8 + 4 + 5:].view(np.int32)[..., 1] & 1).sum())      # games_played = done records


def test_train_selfplay_only_with_priming():
  """train --selfplay_only --prime_moves: the priming moves are part of --max_moves, the printed rate covers the rest"""
  from model_based_rl_amd import train
  thr = train.main(['--environment', 'LunarLander-v2', '--num_envs', '64', '--num_simulations', '8', '--seed', '3',
                    '--episode_length', '6', '--max_moves', '40', '--prime_moves', '16', '--window_size', '8192', '--selfplay_only'])
  assert 64 * 30 <= thr['frames'] <= 64 * 40 and thr['env_steps_per_s'] > 0
