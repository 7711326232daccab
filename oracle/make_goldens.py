#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, runs only in the build container).

Imports the *unmodified* reference (JimOhman/model-based-rl at /root/reference) and
records inputs/outputs of its self-play search path into small .npz fixtures under
tests/golden/.  The reference itself never travels; only these vectors do.

What is driven (all imported, nothing restated except the 30-line move loop of
actors.py:131-173, which cannot be imported because actors.py needs ray/gym):
  mcts.MCTS / mcts.Node / mcts.MinMaxStats      (mcts.py:6-143)
  networks.FCNetwork                            (networks.py:122-180)
  config.Config (make_config, select_action)    (config.py:7-84, 87-231)
  game.Game                                     (game.py:54-126)
  custom_environments.tic_tac_toe.TicTacToe     (tic_tac_toe.py:5-76)
  replay_buffer.PrioritizedReplay / SumTree     (replay_buffer.py:6-210, with a ray stub)

Node numbering used in the tree dumps (this is the build's SoA numbering, imposed on the
reference's object tree so the two can be compared index by index):
  expansion index e: root = 0, the leaf expanded by simulation s = s + 1
  node index:        root = 0, child `a` of the node with expansion index e = 1 + e*A + a

Usage:  python oracle/make_goldens.py [outdir]     (default tests/golden: G1-G3 + the replay batch, 18 files)
        python oracle/make_goldens.py convnets     (G6: MuZeroNetwork / TinyNetwork forward checksums)
        python oracle/make_goldens.py learner      (G5: two Learner.update_weights steps of the reference -- TicTacToe batch 16 on the
                                                    recorded batch; LunarLander shapes batch 256, K = 5 and Pong-ram shapes with
                                                    --norm_obs on seeded synthetic batches)
        python oracle/make_goldens.py refresh      (G4b: PrioritizedReplay.update with the learner's float32 errors)
Every mode is deterministic: a re-run leaves every fixture byte-identical.
"""
import os
import random
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
H = 50


def _import_reference():
  if not os.path.isdir(REF):
    raise SystemExit('reference not mounted at %s (goldens are generated in the build container only)' % REF)
  ray = types.ModuleType('ray')
  ray.remote = lambda c: c
  sys.modules.setdefault('ray', ray)
  sys.path.insert(0, REF)
  import config as rconfig
  import game as rgame
  import mcts as rmcts
  import networks as rnetworks
  import replay_buffer as rreplay
  from custom_environments.tic_tac_toe import TicTacToe
  return types.SimpleNamespace(config=rconfig, game=rgame, mcts=rmcts, networks=rnetworks,
                               replay=rreplay, TicTacToe=TicTacToe)


def make_ref_config(ref, argv, action_space, obs_space):
  """config.make_config() + the scalarisation train.config_generator/launch perform
  (train.py:66-68, 105-120)."""
  old = sys.argv
  sys.argv = ['train.py'] + list(argv)
  try:
    cfg = ref.config.make_config()
  finally:
    sys.argv = old
  for key in ('seed', 'num_actors', 'lr_init', 'discount', 'window_size', 'window_step',
              'batch_size', 'num_simulations', 'num_unroll_steps', 'td_steps'):
    setattr(cfg, key, getattr(cfg, key)[0])
  cfg.action_space = action_space
  cfg.obs_space = obs_space
  return cfg


def set_all_seeds(seed):
  """utils.set_all_seeds (utils.py:136-144) without the cudnn switches."""
  torch.manual_seed(seed)
  random.seed(seed + 2)
  np.random.seed(seed + 3)


WEIGHT_KEYS = None


def weights_of(net):
  sd = net.state_dict()
  return {('w.' + k): v.detach().cpu().numpy().copy() for k, v in sd.items()}


class SyntheticEnv(object):
  """Shape-faithful stand-in for gym envs that are not installed (LunarLander / -ram shapes).
  Fixed-length episodes, all actions legal, obs ~ N(0,1), reward ~ U(-1,1)."""

  def __init__(self, obs_dim, n_actions, length, seed):
    self.obs_dim, self.n, self.length = obs_dim, n_actions, length
    self.rng = np.random.RandomState(seed)
    self._elapsed_steps = 0
    self.was_real_done = False
    self.last_reward = 0.0

  def seed(self, seed):
    return

  def legal_actions(self):
    return np.arange(self.n)

  def reset(self):
    self._elapsed_steps = 0
    return self.rng.standard_normal(self.obs_dim).astype(np.float32)

  def step(self, action):
    self._elapsed_steps += 1
    done = self._elapsed_steps >= self.length
    reward = float(self.rng.uniform(-1, 1))
    self.last_reward = reward
    self.was_real_done = done
    return self.rng.standard_normal(self.obs_dim).astype(np.float32), reward, done, {"result": None}


class RecordingNet(object):
  """Forwards to the reference network and records every recurrent_inference I/O."""

  def __init__(self, net):
    self.net = net
    self.calls = []

  def initial_inference(self, obs):
    return self.net.initial_inference(obs)

  def recurrent_inference(self, hidden, action):
    out = self.net.recurrent_inference(hidden, action)
    self.calls.append((hidden.detach().numpy().copy().reshape(-1), int(action[0]),
                       float(out.value.item()), float(out.reward.item()),
                       out.policy_logits.detach().numpy().copy().reshape(-1),
                       out.hidden_state.detach().numpy().copy().reshape(-1)))
    return out


class FakeNet(object):
  """Random 'network' for tree-only traces: exercises rewards, large values, equal logits (ties)."""

  def __init__(self, ref, A, rng, scale, tie_prob):
    self.ref, self.A, self.rng, self.scale, self.tie_prob = ref, A, rng, scale, tie_prob
    self.calls = []

  def _out(self, with_reward):
    v = np.float32(self.rng.standard_normal() * self.scale)
    r = np.float32(self.rng.standard_normal() * self.scale * 0.5) if with_reward else 0
    if self.rng.uniform() < 0.1 and with_reward:
      r = np.float32(0.0)
    logits = self.rng.standard_normal(self.A).astype(np.float32) * 2
    if self.rng.uniform() < self.tie_prob:
      logits[:] = np.float32(self.rng.standard_normal())
    elif self.rng.uniform() < self.tie_prob:
      logits[self.rng.randint(self.A)] = logits[self.rng.randint(self.A)]
    h = self.rng.standard_normal(H).astype(np.float32)
    NO = self.ref.networks.NetworkOutput
    value_t = torch.tensor([[v]])
    reward_t = torch.tensor([[r]]) if with_reward else 0
    return NO(value_t, reward_t, torch.from_numpy(logits)[None], torch.from_numpy(h)[None]), (v, r, logits, h)

  def initial_inference(self, obs):
    out, _ = self._out(False)
    return out

  def recurrent_inference(self, hidden, action):
    out, (v, r, logits, h) = self._out(True)
    self.calls.append((hidden.detach().numpy().copy().reshape(-1), int(action[0]), float(v), float(r),
                       logits.copy(), h.copy()))
    return out


def dump_tree(root, search_paths, A, sims):
  """Walks the reference's Node objects and lays them out in the build's SoA numbering."""
  NN = 1 + (sims + 1) * A
  N = np.zeros(NN, np.int32)
  W = np.zeros(NN, np.float64)
  P = np.zeros(NN, np.float64)
  R = np.zeros(NN, np.float64)
  E = np.full(NN, -1, np.int32)
  TP = np.zeros(NN, np.int8)
  EX = np.zeros(NN, np.uint8)     # node exists (was created by an expand)
  eidx = {id(root): 0}
  for s, path in enumerate(search_paths):
    eidx[id(path[-1])] = s + 1

  def visit(node, idx):
    N[idx] = node.visit_count
    W[idx] = node.value_sum
    P[idx] = node.prior
    R[idx] = node.reward
    TP[idx] = node.to_play
    EX[idx] = 1
    if node.expanded():
      e = eidx[id(node)]
      E[idx] = e
      for a, child in node.children.items():
        visit(child, 1 + e * A + int(a))

  visit(root, 0)
  return dict(N=N, W=W, P=P, R=R, E=E, TP=TP, EX=EX)


def choice_uniform_and_action(ref_cfg, root, temperature):
  """config.select_action (config.py:70-81) + capture of the single uniform np.random.choice
  consumes when temperature > 0 (legacy RandomState.choice: cdf.searchsorted(u, 'right'))."""
  state = np.random.get_state()
  action = ref_cfg.select_action(root, temperature)
  after = np.random.get_state()
  u = -1.0
  if temperature:
    np.random.set_state(state)
    u = float(np.random.random_sample())
    np.random.set_state(after)
  return int(action), u


def margin_mcts(ref, cfg):
  """The reference's MCTS with select_child wrapped (subclass in the generator, reference file untouched): before
  delegating, it evaluates the children's scores the way select_child does (mcts.py:104-113) and keeps the smallest gap
  between the best and the second-best score over all decisions of a search -- how close the move came to a tie that
  float32 network noise could flip ('each record carries the min top-2 UCB margin')."""

  class MarginMCTS(ref.mcts.MCTS):
    min_margin = float('inf')

    def select_child(self, node):
      if len(node.children) > 1:
        if node.visit_count == 0:
          scores = sorted((c.prior for c in node.children.values()), reverse=True)
        else:
          scores = sorted((self.ucb_score(node, c) for c in node.children.values()), reverse=True)
        self.min_margin = min(self.min_margin, scores[0] - scores[1])
      return super().select_child(node)

  return MarginMCTS(cfg)


def play_and_record(ref, cfg, env, net, temperature, n_moves, recording=True):
  """The move loop of actors.py:131-173 around imported reference objects; returns per-move
  records plus the history flushes it would have sent to the replay buffer."""
  A, sims = cfg.action_space, cfg.num_simulations
  mcts = margin_mcts(ref, cfg)
  game = cfg.new_game(env)
  moves, flushes = [], []
  while not game.terminal and len(moves) < n_moves:
    root = ref.mcts.Node(0)
    obs = np.float32(game.get_observation(-1))
    obs_t = torch.from_numpy(np.ascontiguousarray(obs))
    with torch.inference_mode():
      init = net.initial_inference(obs_t.unsqueeze(0))
    legal = np.asarray(game.environment.legal_actions())
    root.expand(init, game.to_play, legal)
    prior_pre = np.zeros(A, np.float64)
    for a, ch in root.children.items():
      prior_pre[int(a)] = ch.prior
    rng_state = np.random.get_state()
    root.add_exploration_noise(cfg.root_dirichlet_alpha, cfg.root_exploration_fraction)
    after = np.random.get_state()
    np.random.set_state(rng_state)
    noise_legal = np.random.dirichlet([cfg.root_dirichlet_alpha] * len(legal))
    np.random.set_state(after)
    noise = np.zeros(A, np.float64)
    noise[legal] = noise_legal

    net.calls = []
    mcts.min_margin = float('inf')
    with torch.inference_mode():
      paths = mcts.run(root, net)
    error = root.value() - init.value.item()
    game.history.errors.append(error)
    action, u = choice_uniform_and_action(cfg, root, temperature)

    rec = dict(obs=obs.astype(np.float32).reshape(-1), legal=np.isin(np.arange(A), legal).astype(np.uint8),
               to_play=np.int8(game.to_play), root_value=np.float32(init.value.item()),
               root_logits=init.policy_logits.detach().numpy().reshape(-1).astype(np.float32),
               root_hidden=init.hidden_state.detach().numpy().reshape(-1).astype(np.float32),
               prior_pre=prior_pre, noise=noise,
               minmax=np.array([mcts.min_max_stats.minimum, mcts.min_max_stats.maximum]),
               final_root_value=np.float64(root.value()), error=np.float64(error),
               uniform=np.float64(u), action=np.int32(action), temperature=np.float64(temperature))
    depth = np.zeros(sims, np.int32)
    pact = np.full((sims, sims + 1), -1, np.int32)
    for s, path in enumerate(paths):
      depth[s] = len(path) - 1
      node = root
      for d, nxt in enumerate(path[1:]):
        a = [k for k, c in node.children.items() if c is nxt][0]
        pact[s, d] = int(a)
        node = nxt
    rec['leaf_depth'] = depth
    rec['path_actions'] = pact
    rec['sim_parent_hidden'] = np.stack([c[0] for c in net.calls]).astype(np.float32)
    rec['sim_action'] = np.array([c[1] for c in net.calls], np.int32)
    rec['sim_value'] = np.array([c[2] for c in net.calls], np.float32)
    rec['sim_reward'] = np.array([c[3] for c in net.calls], np.float32)
    rec['sim_logits'] = np.stack([c[4] for c in net.calls]).astype(np.float32)
    rec['sim_hidden'] = np.stack([c[5] for c in net.calls]).astype(np.float32)
    tree = dump_tree(root, paths, A, sims)
    for k, v in tree.items():
      rec['tree_' + k] = v
    rec['min_margin'] = np.float64(mcts.min_margin)     # smallest top-2 score gap over all select_child decisions
    moves.append(rec)

    game.apply(action)
    game.store_search_statistics(root)
    rec['child_visits'] = np.asarray(game.history.child_visits[-1], np.float64)

    save_history = (game.history_idx - game.previous_collect_to) == cfg.max_history_length
    if save_history or game.done or game.terminal:
      overlap = cfg.num_unroll_steps + cfg.td_steps
      if not game.history.dones[game.previous_collect_to - 1]:
        collect_from = max(0, game.previous_collect_to - overlap)
      else:
        collect_from = game.previous_collect_to
      history = game.get_history_sequence(collect_from)
      ignore = overlap if not game.done else None
      flushes.append(dict(collect_from=collect_from, ignore=-1 if ignore is None else ignore,
                          terminal=bool(game.terminal), at_move=len(moves), history=history))
    if game.step >= cfg.max_steps:
      break
  return moves, flushes, game


def stack_moves(moves, prefix=''):
  out = {}
  for k in moves[0]:
    out[prefix + k] = np.stack([np.asarray(m[k]) for m in moves])
  return out


def history_arrays(h, A, prefix):
  return {
      prefix + 'observations': np.stack([np.asarray(o, np.float32).reshape(-1) for o in h.observations]),
      prefix + 'child_visits': np.asarray(h.child_visits, np.float64).reshape(-1, A),
      prefix + 'root_values': np.asarray(h.root_values, np.float64),
      prefix + 'actions': np.asarray(h.actions, np.int32),
      prefix + 'rewards': np.asarray(h.rewards, np.float64),
      prefix + 'errors': np.asarray(h.errors, np.float64),
      prefix + 'dones': np.asarray(h.dones, np.uint8),
      prefix + 'steps': np.asarray(h.steps, np.int32),
      prefix + 'to_play': np.asarray(h.to_play, np.int8),
  }


# ----------------------------------------------------------------------------- G1: network I/O
def gen_net(ref, outdir, name, argv, O, A, seed, save_weights=True):
  cfg = make_ref_config(ref, argv, A, (O,))
  set_all_seeds(seed)
  net = ref.networks.FCNetwork(O, A, torch.device('cpu'), cfg)
  net.eval()
  # perturb LN affine + biases so the fixture does not sit on PyTorch's defaults (1, 0)
  g = torch.Generator().manual_seed(seed + 100)
  with torch.no_grad():
    net.LN.weight.add_(0.2 * torch.randn(H, generator=g))
    net.LN.bias.add_(0.1 * torch.randn(H, generator=g))
  rng = np.random.RandomState(seed + 7)
  rows = 64
  obs = rng.standard_normal((rows, O)).astype(np.float32)
  if 'TicTacToe' in argv:
    obs = rng.randint(-1, 2, size=(rows, O)).astype(np.float32)
  acts = rng.randint(0, A, size=rows).astype(np.int32)
  out = weights_of(net) if save_weights else {'weights_file': np.array('g1_net_lunar.npz')}
  with torch.inference_mode():
    init_v, init_l, init_h = [], [], []
    rec_v, rec_r, rec_l, rec_h = [], [], [], []
    for i in range(rows):          # batch-1, exactly like the reference's actors
      o = net.initial_inference(torch.from_numpy(obs[i:i + 1]))
      init_v.append(o.value.item()); init_l.append(o.policy_logits.numpy()[0].copy())
      init_h.append(o.hidden_state.numpy()[0].copy())
      r = net.recurrent_inference(o.hidden_state, [int(acts[i])])
      rec_v.append(r.value.item()); rec_r.append(r.reward.item())
      rec_l.append(r.policy_logits.numpy()[0].copy()); rec_h.append(r.hidden_state.numpy()[0].copy())
    # raw (un-transformed) support logits for the value head, to pin the inverse transform alone
    vlogits = net.value_head(torch.from_numpy(np.stack(init_h))).numpy().copy()
    vinv = (vlogits if cfg.no_support else cfg.inverse_value_transform(torch.from_numpy(vlogits)).numpy()).reshape(-1).copy()
  out.update(obs=obs, actions=acts, init_value=np.array(init_v, np.float32), init_logits=np.stack(init_l),
             init_hidden=np.stack(init_h), rec_value=np.array(rec_v, np.float32),
             rec_reward=np.array(rec_r, np.float32), rec_logits=np.stack(rec_l), rec_hidden=np.stack(rec_h),
             support_logits=vlogits.astype(np.float32), support_inverse=vinv.astype(np.float32),
             O=np.int32(O), A=np.int32(A), seed=np.int32(seed))
  np.savez_compressed(os.path.join(outdir, name), **out)
  return net, cfg


# ----------------------------------------------------------------------------- G2: tree traces
def gen_tree_traces(ref, outdir, nets, only=None):
  specs = [
      # name, argv, O, A, fake(scale, tie_prob) or None, temperature, moves
      ('g2_tree_ttt_fc', ['--environment', 'TicTacToe', '--two_players', '--known_bounds', '-1', '1',
                          '--discount', '1', '--num_simulations', '30', '--seed', '0'], 9, 9, None, 1.0, 32),
      ('g2_tree_lunar_fc', ['--num_simulations', '30', '--seed', '1'], 8, 4, None, 1.0, 32),
      ('g2_tree_pong_fc', ['--num_simulations', '50', '--seed', '2'], 128, 6, None, 0.5, 16),
      ('g2_tree_fake_1p', ['--num_simulations', '30', '--seed', '3'], 8, 4, (3.0, 0.15), 1.0, 32),
      ('g2_tree_fake_2p', ['--environment', 'TicTacToe', '--two_players', '--discount', '0.997',
                           '--num_simulations', '30', '--seed', '4'], 9, 9, (2.0, 0.15), 0.25, 32),
      ('g2_tree_fake_bounds', ['--known_bounds', '-2', '2', '--num_simulations', '50', '--seed', '5',
                               '--init_value_score', '0.5'], 8, 6, (4.0, 0.3), 1.0, 16),
      # 18 actions: more than one 16-lane child group -- the 32-lane two-pass select_child of the kernels meets a REFERENCE-made
      # tree, not only the oracle's (VERDICT r05 item 5 iii)
      ('g2_tree_fake_a18', ['--num_simulations', '20', '--seed', '6'], 8, 18, (3.0, 0.15), 1.0, 16),
  ]
  for name, argv, O, A, fake, temp, n_moves in specs:
    if only is not None and name not in only:
      continue
    cfg = make_ref_config(ref, argv, A, (O,))
    set_all_seeds(cfg.seed)
    if fake is None:
      net = RecordingNet(nets[(O, A)][0])
    else:
      net = FakeNet(ref, A, np.random.RandomState(cfg.seed + 11), *fake)
    moves = []
    ep = 0
    while len(moves) < n_moves:
      if 'TicTacToe' in argv:
        env = ref.TicTacToe()
      else:
        env = SyntheticEnv(O, A, 12, cfg.seed * 100 + ep)
      m, _, _ = play_and_record(ref, cfg, env, net, temp, n_moves - len(moves))
      moves += m
      ep += 1
    out = stack_moves(moves)
    out.update(A=np.int32(A), O=np.int32(O), sims=np.int32(cfg.num_simulations),
               two_players=np.int32(bool(cfg.two_players)), discount=np.float64(cfg.discount),
               pb_c_base=np.float64(cfg.pb_c_base), pb_c_init=np.float64(cfg.pb_c_init),
               init_value_score=np.float64(cfg.init_value_score),
               known_bounds=np.array([np.nan if b is None else b for b in cfg.known_bounds], np.float64),
               alpha=np.float64(cfg.root_dirichlet_alpha), frac=np.float64(cfg.root_exploration_fraction))
    if fake is None:
      out['weights_file'] = np.array(nets[(O, A)][1])
    np.savez_compressed(os.path.join(outdir, name), **out)
    print(name, 'moves', len(moves), 'mean depth %.2f max %d' % (out['leaf_depth'].mean(), out['leaf_depth'].max()))


# ----------------------------------------------------------------------------- G3: full games + G4 replay
def gen_games(ref, outdir, nets):
  argv = ['--environment', 'TicTacToe', '--two_players', '--known_bounds', '-1', '1', '--discount', '1',
          '--num_simulations', '30', '--seed', '0', '--window_size', '60000']
  for gi, (temp, max_hist) in enumerate([(1.0, 500), (0.1, 500), (0.0, 500), (1.0, 4)]):
    cfg = make_ref_config(ref, argv + ['--max_history_length', str(max_hist)], 9, (9,))
    env = ref.TicTacToe()
    replay = ref.replay.PrioritizedReplay(cfg)   # seeds np.random/random (replay_buffer.py:104-106) in ITS process
    set_all_seeds(cfg.seed + 0)                  # actor_key 0 (actors.py:20); separate process in the reference
    net = RecordingNet(nets[(9, 9)][0])
    out = {}
    all_moves = []
    flush_meta = []
    n_games = 3
    for g in range(n_games):
      moves, flushes, game = play_and_record(ref, cfg, env, net, temp, 1000)
      for m in moves:
        m['game'] = np.int32(g)
      all_moves += moves
      for f in flushes:
        k = len(flush_meta)
        st = (np.random.get_state(), random.getstate())
        replay.save_history(f['history'], ignore=None if f['ignore'] < 0 else f['ignore'], terminal=f['terminal'])
        np.random.set_state(st[0]); random.setstate(st[1])
        out.update(history_arrays(f['history'], 9, 'flush%d_' % k))
        flush_meta.append([g, f['at_move'], f['collect_from'], f['ignore'], int(f['terminal']),
                           replay.tree.num_memories, replay.throughput['frames'], replay.throughput['games']])
        out['flush%d_total_priority' % k] = np.float64(replay.tree.total_priority)
    out.update(stack_moves(all_moves))
    out['flush_meta'] = np.asarray(flush_meta, np.int64)
    out['n_flushes'] = np.int32(len(flush_meta))
    out['replay_leaves'] = replay.tree.tree[replay.tree.max_capacity - 1:
                                            replay.tree.max_capacity - 1 + replay.tree.num_memories].copy()
    out['replay_total'] = np.float64(replay.tree.total_priority)
    # G4: one sample_batch under a fixed `random` seed; records the stratified draws it consumed
    replay.batch_size = 16
    random.seed(1234); np.random.seed(4321)
    rs = random.getstate()
    (bobs, bact, (trew, tval, tpol)), idxs, isw = replay.sample_batch()
    random.setstate(rs)
    seg = replay.tree.total_priority / 16
    draws = np.array([random.uniform(seg * i, seg * (i + 1)) for i in range(16)])
    out.update(sample_obs=bobs, sample_actions=np.asarray(bact, np.int32), sample_target_rewards=trew,
               sample_target_values=tval, sample_target_policies=tpol, sample_idxs=np.asarray(idxs, np.int64),
               sample_is_weights=np.asarray(isw, np.float64), sample_draws=draws,
               sample_np_seed=np.int32(4321), max_capacity=np.int64(replay.tree.max_capacity))
    out['weights_file'] = np.array(nets[(9, 9)][1])
    out.update(A=np.int32(9), O=np.int32(9), sims=np.int32(30), two_players=np.int32(1),
               discount=np.float64(cfg.discount), pb_c_base=np.float64(cfg.pb_c_base),
               pb_c_init=np.float64(cfg.pb_c_init), init_value_score=np.float64(cfg.init_value_score),
               known_bounds=np.array(cfg.known_bounds, np.float64), alpha=np.float64(0.25), frac=np.float64(0.25),
               temperature_cfg=np.float64(temp), max_history_length=np.int32(max_hist),
               num_unroll_steps=np.int32(cfg.num_unroll_steps), td_steps=np.int32(cfg.td_steps),
               epsilon=np.float64(cfg.epsilon), per_alpha=np.float64(cfg.alpha), per_beta=np.float64(cfg.beta))
    np.savez_compressed(os.path.join(outdir, 'g3_game_ttt_%d' % gi), **out)
    print('g3_game_ttt_%d' % gi, 'moves', len(all_moves), 'flushes', len(flush_meta))


# ----------------------------------------------------------------------------- G6: conv networks (config 5)
def perturb_norm_layers(net):
  """Deterministic, non-trivial BatchNorm statistics / affine parameters (a fresh BatchNorm is the identity up to
  eps): the same formulas are applied by the tests to the build's network, so nothing but the formulas travels."""
  i = 0
  with torch.no_grad():
    for m in net.modules():
      if isinstance(m, torch.nn.BatchNorm2d):
        c = np.arange(m.num_features, dtype=np.float64)
        m.running_mean.copy_(torch.from_numpy(0.05 * np.sin(c + i)).float())
        m.running_var.copy_(torch.from_numpy(1.0 + 0.5 * np.cos(0.37 * c + i) ** 2).float())
        m.weight.copy_(torch.from_numpy(1.0 + 0.1 * np.sin(0.11 * c - i)).float())
        m.bias.copy_(torch.from_numpy(0.05 * np.cos(0.23 * c + 2 * i)).float())
        i += 1
  return net


def gen_convnets(ref, outdir):
  """MuZeroNetwork / TinyNetwork of the unmodified reference (networks.py:498-555, 657-718), default init under
  torch.manual_seed(seed): the weights themselves (94 MB / 16 MB) do not travel -- the build's definitions construct
  their modules in the reference's order, so the same seed draws the same weights; the fixture pins that with
  per-tensor checksums and pins the forward passes with a small batch of inputs and outputs."""
  for name, cls, C, A, seed in (('g6_net_muzero', 'MuZeroNetwork', 4, 4, 11), ('g6_net_tiny', 'TinyNetwork', 4, 4, 12),
                                ('g6_net_tiny_a6', 'TinyNetwork', 2, 6, 13)):
    cfg = make_ref_config(ref, ['--seed', str(seed)], A, (C, 96, 96))
    torch.manual_seed(seed)
    net = getattr(ref.networks, cls)(C, A, torch.device('cpu'), cfg).eval()
    perturb_norm_layers(net)
    rng = np.random.RandomState(seed)
    rows = 3
    obs = (rng.randint(0, 256, size=(rows, C, 96, 96)).astype(np.float32) / np.float32(255.0))
    acts = np.array([1, A - 1, 0], np.int32)
    with torch.inference_mode():
      o = net.initial_inference(torch.from_numpy(obs))
      r = net.recurrent_inference(o.hidden_state, [int(a) for a in acts])
      r2 = net.recurrent_inference(r.hidden_state, [int(a) for a in acts[::-1]])
    sd = net.state_dict()
    keys = list(sd.keys())
    out = dict(obs_u8=np.round(obs * 255).astype(np.uint8), actions=acts, seed=np.int32(seed), C=np.int32(C), A=np.int32(A),
               arch=np.array(cls), keys=np.array(keys),
               # (numpy's pairwise float64 sums: independent of torch's thread count)
               key_sums=np.array([sd[k].numpy().astype(np.float64).sum() for k in keys]),
               key_abs_sums=np.array([np.abs(sd[k].numpy().astype(np.float64)).sum() for k in keys]),
               init_value=o.value.numpy().reshape(-1), init_logits=o.policy_logits.numpy(),
               init_hidden=o.hidden_state.numpy(),
               rec_value=r.value.numpy().reshape(-1), rec_reward=r.reward.numpy().reshape(-1),
               rec_logits=r.policy_logits.numpy(), rec_hidden=r.hidden_state.numpy(),
               rec2_value=r2.value.numpy().reshape(-1), rec2_reward=r2.reward.numpy().reshape(-1),
               rec2_logits=r2.policy_logits.numpy(), rec2_hidden_sum=np.float64(r2.hidden_state.double().sum()))
    np.savez_compressed(os.path.join(outdir, name), **out)
    print(name, 'params', sum(p.numel() for p in net.parameters()), 'value', out['init_value'], out['rec_value'])


def main():
  if len(sys.argv) > 2 and sys.argv[1] == 'tree':          # one tree-trace fixture by name (the fake-network ones need no weights)
    torch.set_num_threads(1)
    gen_tree_traces(_import_reference(), os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden')), {}, only=sys.argv[2:])
    return
  if len(sys.argv) > 1 and sys.argv[1] == 'convnets':
    torch.set_num_threads(1)
    gen_convnets(_import_reference(), os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden')))
    return
  outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden')
  outdir = os.path.abspath(outdir)
  os.makedirs(outdir, exist_ok=True)
  torch.set_num_threads(1)
  ref = _import_reference()
  nets = {}
  nets[(9, 9)] = (gen_net(ref, outdir, 'g1_net_ttt', ['--environment', 'TicTacToe', '--two_players'], 9, 9, 0)[0],
                  'g1_net_ttt.npz')
  nets[(8, 4)] = (gen_net(ref, outdir, 'g1_net_lunar', [], 8, 4, 1)[0], 'g1_net_lunar.npz')
  nets[(128, 6)] = (gen_net(ref, outdir, 'g1_net_pong', [], 128, 6, 2)[0], 'g1_net_pong.npz')
  gen_net(ref, outdir, 'g1_net_lunar_notransform', ['--no_target_transform'], 8, 4, 1, save_weights=False)
  gen_net(ref, outdir, 'g1_net_lunar_nosupport', ['--no_support'], 8, 4, 3)
  gen_tree_traces(ref, outdir, nets)
  gen_games(ref, outdir, nets)
  gen_convnets(ref, outdir)
  total = sum(os.path.getsize(os.path.join(outdir, f)) for f in os.listdir(outdir) if f.endswith('.npz'))
  print('total fixture bytes', total)


if __name__ == '__main__' and (len(sys.argv) < 2 or sys.argv[-1] not in ('learner', 'refresh', 'learner_conv')):
  main()


# ----------------------------------------------------------------------------- G5: learner step
def _stand_in_modules():
  """learners.py / utils.py / wrappers.py / logger.py import ray, gym, cv2 and tensorboard, none of which is installed: empty
  stand-in MODULES for the import only (nothing of them runs on the update_weights path)"""
  ray = types.ModuleType('ray'); ray.remote = lambda c: c; ray.get = lambda x: x
  sys.modules['ray'] = ray
  gym = types.ModuleType('gym')

  class _W(object):
    def __init__(self, env=None):
      self.env = env
  gym.Wrapper = gym.ObservationWrapper = gym.RewardWrapper = gym.ActionWrapper = _W
  spaces = types.ModuleType('gym.spaces'); spaces.Box = object; gym.spaces = spaces
  sys.modules['gym'] = gym; sys.modules['gym.spaces'] = spaces
  cv2 = types.ModuleType('cv2'); cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda x: None)
  sys.modules['cv2'] = cv2
  tb = types.ModuleType('torch.utils.tensorboard')

  class SW(object):
    def __init__(self, *a, **k): pass
    def add_scalar(self, *a, **k): pass
    def add_scalars(self, *a, **k): pass
    def add_histogram(self, *a, **k): pass
  tb.SummaryWriter = SW
  sys.modules['torch.utils.tensorboard'] = tb
  if REF not in sys.path:
    sys.path.insert(0, REF)


def gen_learner(outdir):
  """One Learner.update_weights step (learners.py:164-230) of the UNMODIFIED reference on the batch recorded in
  g3_game_ttt_0 (G4).  learners.py / utils.py / wrappers.py / logger.py import ray, gym, cv2 and tensorboard, none of
  which is installed; they are replaced by empty stand-in modules for the IMPORT only (nothing of them is executed on
  this path: the environment is the reference's own TicTacToe, logging calls are no-ops)."""
  import tempfile
  _stand_in_modules()
  import config as rconfig
  import learners as rlearners
  g = np.load(os.path.join(outdir, 'g3_game_ttt_0.npz'))
  old = sys.argv
  sys.argv = ['train.py', '--environment', 'TicTacToe', '--two_players', '--known_bounds', '-1', '1', '--discount', '1',
              '--seed', '0', '--batch_size', '16', '--window_size', '60000', '--run_tag', 'g5', '--group_tag', 'g5']
  try:
    cfg = rconfig.make_config()
  finally:
    sys.argv = old
  for key in ('seed', 'num_actors', 'lr_init', 'discount', 'window_size', 'window_step', 'batch_size', 'num_simulations',
              'num_unroll_steps', 'td_steps'):
    setattr(cfg, key, getattr(cfg, key)[0])
  cfg.action_space, cfg.obs_space = 9, (9,)

  class Sink(object):            # replay / storage handles: record what the learner sends
    def __init__(self): self.calls = []
    def __getattr__(self, name):
      sink = self
      class M(object):
        def remote(self_, *a, **k):
          sink.calls.append((name, a, k)); return None
      return M()
  replay, storage = Sink(), Sink()
  cwd = os.getcwd()
  os.chdir(tempfile.mkdtemp())
  try:
    learner = rlearners.Learner(cfg, storage, replay)
  finally:
    os.chdir(cwd)
  w0 = {('w0.' + k): v.detach().numpy().copy() for k, v in learner.network.state_dict().items()}
  batch = ((g['sample_obs'].copy(), g['sample_actions'].tolist(),
            (g['sample_target_rewards'].copy(), g['sample_target_values'].copy(), g['sample_target_policies'].copy())),
           g['sample_idxs'].tolist(), g['sample_is_weights'].copy())
  out = dict(w0)
  for step in range(2):
    learner.update_weights(batch)
    out.update({('w%d.' % (step + 1) + k): v.detach().numpy().copy() for k, v in learner.network.state_dict().items()})
  out['losses'] = np.array([learner.losses_to_log['reward'], learner.losses_to_log['value'], learner.losses_to_log['policy']])
  upd = [c for c in replay.calls if c[0] == 'update']
  out['new_errors'] = np.stack([np.asarray(c[1][1], np.float64) for c in upd])
  out['lr_init'] = np.float64(cfg.lr_init); out['weight_decay'] = np.float64(cfg.weight_decay)
  np.savez_compressed(os.path.join(outdir, 'g5_learner_ttt'), **out)
  print('g5_learner_ttt: losses (sum of 2 steps)', out['losses'])


def gen_learner_synth(outdir, name, env_argv, O, A, bs, seed):
  """Two Learner.update_weights steps (learners.py:164-230) of the UNMODIFIED reference on a seeded synthetic batch at the
  shapes the learner bench runs (LunarLander: obs 8, 4 actions, batch 256, K = 5) and at the Pong-ram shapes (obs 128 bytes
  with --norm_obs, 6 actions).  gen_learner's stand-in modules are in place (it runs first).  The environment probe of
  utils.get_network (gym.make: not installed) is replaced by the direct FCNetwork construction SURVEY.md s8c prescribes
  (networks.py:124: FCNetwork(input_dim, action_space, device, config)); everything else -- optimiser, losses, hooks,
  update_weights -- is the reference's."""
  import tempfile
  import config as rconfig
  import learners as rlearners
  import networks as rnetworks
  old = sys.argv
  sys.argv = ['train.py'] + env_argv + ['--seed', '0', '--batch_size', str(bs), '--window_size', '60000', '--run_tag', 'g5', '--group_tag', 'g5']
  try:
    cfg = rconfig.make_config()
  finally:
    sys.argv = old
  for key in ('seed', 'num_actors', 'lr_init', 'discount', 'window_size', 'window_step', 'batch_size', 'num_simulations',
              'num_unroll_steps', 'td_steps'):
    setattr(cfg, key, getattr(cfg, key)[0])
  cfg.action_space, cfg.obs_space = A, (O,)

  class Sink(object):
    def __init__(self): self.calls = []
    def __getattr__(self, nm):
      sink = self
      class M(object):
        def remote(self_, *a, **k):
          sink.calls.append((nm, a, k)); return None
      return M()
  replay, storage = Sink(), Sink()
  probe = rlearners.get_network
  rlearners.get_network = lambda config, device=None: rnetworks.FCNetwork(O, A, device, config)
  cwd = os.getcwd()
  os.chdir(tempfile.mkdtemp())
  try:
    learner = rlearners.Learner(cfg, storage, replay)
  finally:
    os.chdir(cwd)
    rlearners.get_network = probe
  K = cfg.num_unroll_steps
  rng = np.random.RandomState(seed)
  if cfg.norm_obs:
    obs = rng.randint(0, 256, size=(bs, O)).astype(np.float32)       # the -ram- bytes as the replay hands them over (np.float32(obs))
  else:
    obs = rng.standard_normal((bs, O)).astype(np.float32)
  actions = rng.randint(0, A, size=(bs, K))
  t_rew = rng.uniform(-1, 1, size=(bs, K + 1)).astype(np.float32)
  t_val = rng.uniform(-4, 4, size=(bs, K + 1)).astype(np.float32)
  t_pol = rng.dirichlet([0.5] * A, size=(bs, K + 1)).astype(np.float32)
  t_pol[rng.rand(bs, K + 1) < 0.05] = 0.0                            # absorbing steps (replay_buffer.py:195-198)
  w = rng.uniform(0.2, 1.0, size=bs); w /= w.max()
  idxs = rng.randint(59999, 119999, size=bs)
  out = {('w0.' + k): v.detach().numpy().copy() for k, v in learner.network.state_dict().items()}
  out.update(sample_obs=obs, sample_actions=actions, sample_target_rewards=t_rew, sample_target_values=t_val,
             sample_target_policies=t_pol, sample_is_weights=w, sample_idxs=idxs)
  batch = ((obs.copy(), actions.tolist(), (t_rew.copy(), t_val.copy(), t_pol.copy())), idxs.tolist(), w.copy())
  for step in range(2):
    learner.update_weights(batch)
    out.update({('w%d.' % (step + 1) + k): v.detach().numpy().copy() for k, v in learner.network.state_dict().items()})
  out['losses'] = np.array([learner.losses_to_log['reward'], learner.losses_to_log['value'], learner.losses_to_log['policy']])
  upd = [c for c in replay.calls if c[0] == 'update']
  out['new_errors'] = np.stack([np.asarray(c[1][1], np.float64) for c in upd])
  out['new_errors_dtype'] = np.array(str(np.asarray(upd[0][1][1]).dtype))
  out['lr_init'] = np.float64(cfg.lr_init); out['weight_decay'] = np.float64(cfg.weight_decay)
  np.savez_compressed(os.path.join(outdir, name), **out)
  print(name, ': losses (sum of 2 steps)', out['losses'], 'errors dtype', out['new_errors_dtype'])


def gen_refresh(outdir):
  """G4b: PrioritizedReplay.update (replay_buffer.py:200-203) of the unmodified reference with the FLOAT32 errors its learner
  sends (learners.py:181-182) and with the same errors as float64: leaves and total priority after each."""
  ref = _import_reference()
  import replay_buffer as rrb
  cfg = types.SimpleNamespace(batch_size=16, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(9,), action_space=9, window_size=1000,
                              window_step=None, num_unroll_steps=5, td_steps=10, max_history_length=500, discount=1.0, seed=0,
                              beta_increment_per_sampling=0.001, stored_before_train=10)
  rep = rrb.PrioritizedReplay(cfg)
  rng = np.random.RandomState(0)
  pri = rng.uniform(0.01, 3, size=500)
  rep.tree.add(list(pri), types.SimpleNamespace())
  out = {'priorities': pri, 'total0': np.float64(rep.tree.total_priority)}
  idxs = rng.randint(999, 999 + 500, size=64)
  err = rng.standard_normal(64).astype(np.float32)
  rep.update(list(idxs), err)
  out.update(idxs=idxs, errors32=err, leaves_after_f32=np.array(rep.tree.tree[999:999 + 500], np.float64),
             total_after_f32=np.float64(rep.tree.total_priority))
  rep.update(list(idxs), err.astype(np.float64))
  out.update(leaves_after_f64=np.asarray(rep.tree.tree[999:999 + 500], np.float64), total_after_f64=np.float64(rep.tree.total_priority))
  np.savez_compressed(os.path.join(outdir, 'g4_refresh'), **out)
  print('g4_refresh: totals', out['total0'], out['total_after_f32'], out['total_after_f64'])


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[-1] == 'refresh':
  gen_refresh(os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden')))

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[-1] == 'learner':
  _out = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
  gen_learner(_out)
  gen_learner_synth(_out, 'g5_learner_lunar', ['--environment', 'LunarLander-v2'], 8, 4, 256, 11)
  gen_learner_synth(_out, 'g5_learner_pong', ['--environment', 'Pong-ramNoFrameskip-v4', '--norm_obs', '--obs_range', '0', '255'], 128, 6, 256, 12)      # (batch 256: the size the configs run; 64 until r05)


def conv_batch(seed, bs, C, A, K):
  """the seeded synthetic image batch of g5_learner_conv (tests/test_learner.py regenerates it from the seed: numpy's legacy
  RandomState stream is stable by NEP 19): byte frames as the replay hands them over (np.float32(obs), learners.py:170-172
  normalises), actions, targets with absorbing steps, importance weights"""
  rng = np.random.RandomState(seed)
  obs = rng.randint(0, 256, size=(bs, C, 96, 96)).astype(np.float32)
  actions = rng.randint(0, A, size=(bs, K))
  t_rew = rng.uniform(-1, 1, size=(bs, K + 1)).astype(np.float32)
  t_val = rng.uniform(-4, 4, size=(bs, K + 1)).astype(np.float32)
  t_pol = rng.dirichlet([0.5] * A, size=(bs, K + 1)).astype(np.float32)
  t_pol[rng.rand(bs, K + 1) < 0.05] = 0.0
  w = rng.uniform(0.2, 1.0, size=bs); w /= w.max()
  idxs = rng.randint(59999, 119999, size=bs)
  return obs, actions, t_rew, t_val, t_pol, w, idxs


def gen_learner_conv(outdir, name='g5_learner_conv', arch='TinyNetwork', C=4, A=4, bs=8, seed=5, batch_seed=21):
  """Two Learner.update_weights steps (learners.py:164-230) of the UNMODIFIED reference with a conv network (TinyNetwork,
  networks.py:657-718; BASELINE.json configs[4] names the conv networks) on a seeded synthetic image batch.  16 MB of weights
  do not travel (the G6 recipe): the build constructs its modules in the reference's order under the same seed
  (set_all_seeds(config.seed), learners.py:18, utils.py:136-144), pinned here by per-tensor checksums of the INITIAL weights;
  after each step: per-tensor float64 sums and absolute sums of every state_dict entry (BatchNorm statistics and batch counters
  included; tensors of <= 4096 elements in full), the three loss sums, the priorities' new errors."""
  import tempfile
  _stand_in_modules()
  import config as rconfig
  import learners as rlearners
  import networks as rnetworks
  old = sys.argv
  sys.argv = ['train.py', '--environment', 'BreakoutNoFrameskip-v4', '--architecture', arch, '--stack_obs', str(C), '--norm_obs', '--obs_range', '0', '255',
              '--seed', str(seed), '--batch_size', str(bs), '--window_size', '60000', '--run_tag', 'g5', '--group_tag', 'g5']
  try:
    cfg = rconfig.make_config()
  finally:
    sys.argv = old
  for key in ('seed', 'num_actors', 'lr_init', 'discount', 'window_size', 'window_step', 'batch_size', 'num_simulations',
              'num_unroll_steps', 'td_steps'):
    setattr(cfg, key, getattr(cfg, key)[0])
  cfg.action_space, cfg.obs_space = A, (C, 96, 96)

  class Sink(object):
    def __init__(self): self.calls = []
    def __getattr__(self, nm):
      sink = self
      class M(object):
        def remote(self_, *a, **k):
          sink.calls.append((nm, a, k)); return None
      return M()
  replay, storage = Sink(), Sink()
  probe = rlearners.get_network
  # (utils.get_network probes the environment through gym.make -- not installed; the direct construction SURVEY.md s8c prescribes)
  rlearners.get_network = lambda config, device=None: getattr(rnetworks, arch)(C, A, device, config)
  cwd = os.getcwd()
  os.chdir(tempfile.mkdtemp())
  try:
    learner = rlearners.Learner(cfg, storage, replay)
  finally:
    os.chdir(cwd)
    rlearners.get_network = probe
  K = cfg.num_unroll_steps
  obs, actions, t_rew, t_val, t_pol, w, idxs = conv_batch(batch_seed, bs, C, A, K)

  def sums(tag, out):
    sd = learner.network.state_dict()
    keys = list(sd.keys())
    out['keys'] = np.array(keys)
    out[tag + '.sums'] = np.array([sd[k].detach().numpy().astype(np.float64).sum() for k in keys])
    out[tag + '.abs_sums'] = np.array([np.abs(sd[k].detach().numpy().astype(np.float64)).sum() for k in keys])
    out['numel'] = np.array([sd[k].numel() for k in keys], np.int64)
    for k in keys:
      if sd[k].numel() <= 4096:
        out[tag + '.full.' + k] = sd[k].detach().numpy().copy()
  out = dict(arch=np.array(arch), C=np.int32(C), A=np.int32(A), bs=np.int32(bs), K=np.int32(K), seed=np.int32(seed), batch_seed=np.int32(batch_seed),
             obs_sum=np.float64(obs.astype(np.float64).sum()), lr_init=np.float64(cfg.lr_init), weight_decay=np.float64(cfg.weight_decay))
  sums('w0', out)
  batch = ((obs.copy(), actions.tolist(), (t_rew.copy(), t_val.copy(), t_pol.copy())), idxs.tolist(), w.copy())
  step_losses = []
  for step in range(2):
    before = dict(learner.losses_to_log)
    learner.update_weights(batch)
    step_losses.append([learner.losses_to_log[k] - before[k] for k in ('reward', 'value', 'policy')])
    sums('w%d' % (step + 1), out)
  out['losses'] = np.array([learner.losses_to_log['reward'], learner.losses_to_log['value'], learner.losses_to_log['policy']])
  out['step_losses'] = np.array(step_losses)
  upd = [c for c in replay.calls if c[0] == 'update']
  out['new_errors'] = np.stack([np.asarray(c[1][1], np.float64) for c in upd])
  np.savez_compressed(os.path.join(outdir, name), **out)
  print(name, ': losses (sum of 2 steps)', out['losses'], 'per step', out['step_losses'].tolist(), 'tensors', len(out['keys']),
        'bytes', os.path.getsize(os.path.join(outdir, name + '.npz')))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[-1] == 'learner_conv':
  torch.set_num_threads(1)
  gen_learner_conv(os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden')))
