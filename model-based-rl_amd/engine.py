"""Batched device engine: B search trees in lock-step on one MI355X, through the C ABI.

torch is used for device memory and streams only; every computation below is a HIP kernel in
csrc/ reached through libmz_hip.so.  Mirrors, batched over B environments, what one reference Actor
does per move (actors.py:131-153): initial inference, root expand + noise, MCTS.run, select_action."""
import ctypes as C
import os

import numpy as np
import torch

from . import _abi

H = 50

WEIGHT_ORDER = [
    'representation_head.fc1.weight', 'representation_head.fc1.bias',
    'representation_head.out.weight', 'representation_head.out.bias',
    'value_head.fc1.weight', 'value_head.fc1.bias', 'value_head.value.weight', 'value_head.value.bias',
    'policy_head.fc1.weight', 'policy_head.fc1.bias', 'policy_head.policy.weight', 'policy_head.policy.bias',
    'reward_head.fc1.weight', 'reward_head.fc1.bias', 'reward_head.reward.weight', 'reward_head.reward.bias',
    'transition_head.fc1.weight', 'transition_head.fc1.bias', 'transition_head.out.weight', 'transition_head.out.bias',
    'LN.weight', 'LN.bias',
]


def flatten_weights(weights):
  """state_dict (reference key names, networks.py:137-144) -> one float32 vector in ABI order.
  Copied with numpy: torch.cat parallelises over its intra-op pool once a tensor passes 32 K elements (the 128 x 512 first
  layer of the -ram- shapes), and the pool's threads spin after every call -- 7 of 16 host cores busy at a pull every 0.1 s."""
  parts = []
  for k in WEIGHT_ORDER:
    v = weights[k]
    parts.append(v.detach().to('cpu', torch.float32).numpy().reshape(-1) if torch.is_tensor(v)
                 else np.ascontiguousarray(v, np.float32).reshape(-1))
  out = np.empty(sum(p.size for p in parts), np.float32)
  pos = 0
  for p in parts:
    out[pos:pos + p.size] = p
    pos += p.size
  return torch.from_numpy(out)


def weights_scale_ok(flat_host, obs_dim, action_space, value_outputs, reward_outputs):
  """mz_weights_scale_ok (include/mz_engine.h): does this weight set admit the search kernel's clamp-ReLU scale?  Host
  arithmetic on a host copy of the flat weights (engine.WEIGHT_ORDER); no GPU."""
  w = flat_host if torch.is_tensor(flat_host) else torch.from_numpy(np.ascontiguousarray(flat_host, np.float32))
  w = w.detach().to('cpu', torch.float32).contiguous().reshape(-1)
  rc = _abi.load().mz_weights_scale_ok(C.c_void_p(w.data_ptr()), w.numel(), int(obs_dim), int(action_space), int(value_outputs),
                                       int(reward_outputs))
  if rc < 0:
    _abi.check(rc, 'mz_weights_scale_ok')
  return int(rc)


def config_scale_check(config):
  """flat host weights -> mz_weights_scale_ok for a Config's FCNetwork shapes (None for the torch networks)"""
  if getattr(config, 'architecture', 'FCNetwork') != 'FCNetwork':
    return None
  O, A = int(np.prod(config.obs_space)), int(config.action_space)
  if getattr(config, 'no_support', False):
    sv = sr = 1
  else:
    sv = int(config.value_support[1]) - int(config.value_support[0]) + 1
    sr = int(config.reward_support[1]) - int(config.reward_support[0]) + 1
  return lambda flat: weights_scale_ok(flat, O, A, sv, sr)


REC_EXTRA = 10      # include/mz_engine.h MZ_REC_EXTRA


def records_view(rec, O, A, obs_u8=False):
  """Named views into experience records (numpy float32, layout of include/mz_engine.h / include/mz_replay.h):
  obs, child_visits, root_value / error (float64), reward, action / done / step / env_id / episode (int32).
  obs_u8: the observation is O bytes packed into ceil(O / 4) float slots (image frames, torch_search.TorchSelfplay)."""
  rec = np.asarray(rec)
  OS = (O + 3) // 4 if obs_u8 else O
  assert rec.dtype == np.float32 and rec.shape[-1] == OS + A + REC_EXTRA, (rec.dtype, rec.shape)
  ints = rec[..., OS + A + 5:].view(np.int32)
  flags = ints[..., 1]                   # bit 0 done, bit 1 the mover was player -1 (History.to_play, game.py:100-101)
  obs = np.ascontiguousarray(rec[..., :OS]).view(np.uint8)[..., :O] if obs_u8 else rec[..., :O]
  return dict(obs=obs, child_visits=rec[..., OS:OS + A],
              root_value=np.ascontiguousarray(rec[..., OS + A:OS + A + 2]).view(np.float64)[..., 0],
              error=np.ascontiguousarray(rec[..., OS + A + 2:OS + A + 4]).view(np.float64)[..., 0],
              reward=rec[..., OS + A + 4], action=ints[..., 0], done=flags & 1, to_play=1 - (flags & 2), step=ints[..., 2],
              env_id=ints[..., 3], episode=ints[..., 4])


def _ptr(t):
  return None if t is None else C.c_void_p(t.data_ptr())


class Engine(object):

  def __init__(self, num_envs, obs_dim, action_space, num_simulations, two_players=False,
               known_bounds=(None, None), value_support=(-15, 15), reward_support=(-15, 15),
               no_target_transform=False, no_support=False, discount=0.997, pb_c_base=19652, pb_c_init=1.25, init_value_score=0.0,
               root_dirichlet_alpha=0.25, root_exploration_fraction=0.25, seed=0, env_id_offset=0, device=None,
               split_f16=False):
    if not torch.cuda.is_available():
      raise RuntimeError('model_based_rl_amd.Engine needs a HIP device (torch.cuda.is_available() is False); '
                         'there is no CPU path.')
    self.lib = _abi.load()
    self.device = torch.device(device if device is not None else 'cuda:0')
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    torch.cuda.set_device(self.device)
    lo, hi = known_bounds
    self.B, self.O, self.A, self.sims = int(num_envs), int(obs_dim), int(action_space), int(num_simulations)
    # outputs of the value / reward heads: the support sizes, one scalar each with --no_support (networks.py:135-136)
    self._outputs = (1, 1) if no_support else (int(value_support[1]) - int(value_support[0]) + 1,
                                                int(reward_support[1]) - int(reward_support[0]) + 1)
    self.cfg = _abi.MzConfig(
        self.B, self.O, self.A, self.sims, int(bool(two_players)), int(lo is not None), int(hi is not None),
        int(value_support[0]), int(value_support[1]), int(reward_support[0]), int(reward_support[1]),
        int(bool(no_target_transform)), 0.0 if lo is None else float(lo), 0.0 if hi is None else float(hi),
        float(discount), float(pb_c_base), float(pb_c_init), float(init_value_score), float(root_dirichlet_alpha),
        float(root_exploration_fraction), int(seed), int(env_id_offset), int(bool(no_support)), int(bool(split_f16)))
    # (the library also honours MZ_SPLIT_F16=1, for running the parity suite on that kernel)
    self.split_f16 = bool(split_f16) or os.environ.get('MZ_SPLIT_F16', '0')[:1] == '1'
    h = C.c_void_p()
    _abi.check(self.lib.mz_create(C.byref(self.cfg), C.byref(h)), 'mz_create')
    self._h = h
    self.NN = self.lib.mz_nodes_per_tree(self._h)
    self.num_weights = self.lib.mz_num_weights(self._h)
    self._keep = []
    self.obs_packed = False

  @classmethod
  def from_config(cls, config, num_envs, device=None, seed=None, env_id_offset=0):
    """Build from a reference-style Config (config.py:87-231 attribute names)."""
    obs_dim = int(np.prod(config.obs_space))
    return cls(num_envs, obs_dim, config.action_space, config.num_simulations,
               two_players=getattr(config, 'two_players', False),
               known_bounds=tuple(getattr(config, 'known_bounds', (None, None))),
               value_support=tuple(getattr(config, 'value_support', (-15, 15))),
               reward_support=tuple(getattr(config, 'reward_support', (-15, 15))),
               no_target_transform=getattr(config, 'no_target_transform', False),
               no_support=getattr(config, 'no_support', False), discount=config.discount,
               pb_c_base=config.pb_c_base, pb_c_init=config.pb_c_init,
               init_value_score=getattr(config, 'init_value_score', 0.0),
               root_dirichlet_alpha=config.root_dirichlet_alpha,
               root_exploration_fraction=config.root_exploration_fraction,
               seed=(config.seed if seed is None else seed) or 0, env_id_offset=env_id_offset, device=device,
               split_f16=getattr(config, 'split_f16', False))

  def close(self):
    if getattr(self, '_h', None):
      self.lib.mz_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  @property
  def stream(self):
    return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def _dev(self, x, dtype):
    if x is None:
      return None
    if not torch.is_tensor(x):
      x = torch.as_tensor(np.ascontiguousarray(x))
    x = x.to(self.device, dtype).contiguous()
    self._keep.append(x)
    if len(self._keep) > 64:
      del self._keep[:32]
    return x

  # ---- weights (networks.py:36-37, actors.py:81-85)
  def set_weights(self, weights, scale_ok=None, sync=False):
    """Weights on the HOST (a state_dict, a numpy array, a CPU tensor): mz_set_weights_async -- the repack is queued in
    stream order and the call returns without waiting for the moves queued before it (which kernel set the weights run on
    is decided on the host copy, mz_weights_scale_ok).  Weights on the DEVICE (the buffer a broadcast filled): the same
    when the caller passes `scale_ok` (the learner rank's mz_weights_scale_ok on its host copy; the tensor must stay valid
    until the stream has passed the call), else mz_set_weights, which reads the decision back and so waits for the stream.
    sync: force mz_set_weights."""
    if isinstance(weights, dict):
      weights = flatten_weights(weights)
    if not torch.is_tensor(weights):
      weights = torch.from_numpy(np.ascontiguousarray(weights, np.float32))
    weights = weights.reshape(-1)
    if weights.numel() != self.num_weights:
      raise ValueError('expected %d weights, got %d' % (self.num_weights, weights.numel()))
    on_dev = weights.is_cuda
    w = weights.to(torch.float32).contiguous()
    sync = sync or os.environ.get('MZ_SYNC_WEIGHTS', '0')[:1] == '1'      # (A/B runs against the synchronous pull)
    if not on_dev and scale_ok is None and not sync:
      scale_ok = self.weights_scale_ok(w)
    if scale_ok is None or sync:
      _abi.check(self.lib.mz_set_weights(self._h, _ptr(w), w.numel(), int(on_dev), self.stream), 'mz_set_weights')
    else:
      _abi.check(self.lib.mz_set_weights_async(self._h, _ptr(w), w.numel(), int(on_dev), int(bool(scale_ok)), self.stream),
                 'mz_set_weights_async')
    self._weights_dev = w if on_dev else None      # (the one device buffer of the last pull stays alive; nothing accumulates)

  def weights_scale_ok(self, flat_host):
    """mz_weights_scale_ok for this engine's shapes on a host copy of the flat weights"""
    return weights_scale_ok(flat_host, self.O, self.A, self._outputs[0], self._outputs[1])

  def weight_scale(self):
    """(1, 2^-k, 2^k, chosen) of the last set_weights (mz_weight_scale): the power of two the search kernel's weight
    stream is scaled by so that its ReLU can be a clamp."""
    out = np.zeros(4, np.float32)
    _abi.check(self.lib.mz_weight_scale(self._h, out.ctypes.data, self.stream), 'mz_weight_scale')
    return out

  # ---- network
  def initial_inference(self, obs):
    obs = self._dev(obs, torch.float32).reshape(self.B, self.O)
    _abi.check(self.lib.mz_initial_inference(self._h, _ptr(obs), self.stream), 'mz_initial_inference')

  def root_load(self, value, logits, hidden=None):
    value = self._dev(value, torch.float32); logits = self._dev(logits, torch.float32)
    hidden = self._dev(hidden, torch.float32)
    _abi.check(self.lib.mz_root_load(self._h, _ptr(hidden), _ptr(value), _ptr(logits), self.stream), 'mz_root_load')

  def root_outputs(self):
    v = torch.empty(self.B, dtype=torch.float32, device=self.device)
    lg = torch.empty(self.B, self.A, dtype=torch.float32, device=self.device)
    h = torch.empty(self.B, H, dtype=torch.float32, device=self.device)
    _abi.check(self.lib.mz_root_outputs(self._h, _ptr(v), _ptr(lg), _ptr(h), self.stream), 'mz_root_outputs')
    return v, lg, h

  def recurrent_inference(self, hidden, action):
    hidden = self._dev(hidden, torch.float32).reshape(-1, H)
    action = self._dev(action, torch.int32)
    n = hidden.shape[0]
    ho = torch.empty(n, H, dtype=torch.float32, device=self.device)
    r = torch.empty(n, dtype=torch.float32, device=self.device)
    v = torch.empty(n, dtype=torch.float32, device=self.device)
    lg = torch.empty(n, self.A, dtype=torch.float32, device=self.device)
    _abi.check(self.lib.mz_recurrent_inference(self._h, _ptr(hidden), _ptr(action), n, _ptr(ho), _ptr(r), _ptr(v),
                                               _ptr(lg), self.stream), 'mz_recurrent_inference')
    return ho, r, v, lg

  # ---- tree
  def root_prepare(self, to_play=None, legal=None, noise=None, device_rng=False, move=0):
    to_play = self._dev(to_play, torch.int8)
    legal = self._dev(legal, torch.uint8)
    noise = self._dev(noise, torch.float64)
    _abi.check(self.lib.mz_root_prepare(self._h, _ptr(to_play), _ptr(legal), _ptr(noise), int(device_rng), int(move),
                                        self.stream), 'mz_root_prepare')

  def root_set_priors(self, priors, to_play=None, legal=None):
    to_play = self._dev(to_play, torch.int8); legal = self._dev(legal, torch.uint8)
    priors = self._dev(priors, torch.float64)
    _abi.check(self.lib.mz_root_set_priors(self._h, _ptr(to_play), _ptr(legal), _ptr(priors), self.stream),
               'mz_root_set_priors')

  def last_paths(self):
    paths = torch.empty(self.B, self.sims + 2, dtype=torch.int32, device=self.device)
    lens = torch.empty(self.B, dtype=torch.int32, device=self.device)
    _abi.check(self.lib.mz_last_paths(self._h, _ptr(paths), _ptr(lens), self.stream), 'mz_last_paths')
    return paths, lens

  def search(self, num_simulations=None):
    n = self.sims if num_simulations is None else int(num_simulations)
    _abi.check(self.lib.mz_search(self._h, n, self.stream), 'mz_search')

  def search_profiled(self, num_simulations=None):
    """-> (ms of all recurrent-inference launches, ms of all tree-step launches) for one search."""
    n = self.sims if num_simulations is None else int(num_simulations)
    ms = (C.c_float * 2)()
    _abi.check(self.lib.mz_search_profiled(self._h, n, ms, self.stream), 'mz_search_profiled')
    return float(ms[0]), float(ms[1])

  def search_timed(self, num_simulations=None):
    """mz_search with events around the search kernel's own dispatch; returns milliseconds (synchronous)."""
    n = self.sims if num_simulations is None else int(num_simulations)
    ms = (C.c_float * 1)()
    _abi.check(self.lib.mz_search_timed(self._h, n, ms, self.stream), 'mz_search_timed')
    return float(ms[0])

  def search_phase_profile(self, num_simulations=None):
    n = self.sims if num_simulations is None else int(num_simulations)
    out = np.zeros((4, 14), np.uint64)
    _abi.check(self.lib.mz_search_phase_profile(self._h, n, out.ctypes.data_as(C.c_void_p), self.stream),
               'mz_search_phase_profile')
    return out

  def search_phase_spread(self):
    """(mean, min, max) over the workgroups of a workgroup's total cycles in the last search_phase_profile"""
    out = (C.c_double * 3)()
    _abi.check(self.lib.mz_search_phase_spread(self._h, out), 'mz_search_phase_spread')
    return tuple(float(x) for x in out)

  def select(self):
    out = [torch.empty(self.B, dtype=torch.int32, device=self.device) for _ in range(4)]
    _abi.check(self.lib.mz_select(self._h, *[_ptr(o) for o in out], self.stream), 'mz_select')
    return out   # leaf_node, parent_slot, action, depth

  def tree_pair_timed(self, value, reward, logits):
    """one simulation's select + expand_backup with events around each kernel's own dispatch; returns (ms, ms), synchronous"""
    value = self._dev(value, torch.float32); reward = self._dev(reward, torch.float32); logits = self._dev(logits, torch.float32)
    ms = (C.c_float * 2)()
    _abi.check(self.lib.mz_tree_pair_timed(self._h, _ptr(value), _ptr(reward), _ptr(logits), ms, self.stream), 'mz_tree_pair_timed')
    return float(ms[0]), float(ms[1])

  def gather_hidden(self):
    h = torch.empty(self.B, H, dtype=torch.float32, device=self.device)
    _abi.check(self.lib.mz_gather_hidden(self._h, _ptr(h), self.stream), 'mz_gather_hidden')
    return h

  def expand_backup(self, value, reward, logits, hidden=None):
    value = self._dev(value, torch.float32); reward = self._dev(reward, torch.float32)
    logits = self._dev(logits, torch.float32); hidden = self._dev(hidden, torch.float32)
    _abi.check(self.lib.mz_expand_backup(self._h, _ptr(value), _ptr(reward), _ptr(logits), _ptr(hidden), self.stream),
               'mz_expand_backup')

  def expand_backup_select(self, value, reward, logits, hidden=None, last=False):
    """expand_backup of this simulation and select of the next in one launch (mz_expand_backup_select); returns what
    select() returns, or None after the move's last simulation (`last`: the caller's count of simulations)"""
    value = self._dev(value, torch.float32); reward = self._dev(reward, torch.float32)
    logits = self._dev(logits, torch.float32); hidden = self._dev(hidden, torch.float32)
    out = None if last else [torch.empty(self.B, dtype=torch.int32, device=self.device) for _ in range(4)]
    ptrs = [None] * 4 if last else [_ptr(o) for o in out]
    _abi.check(self.lib.mz_expand_backup_select(self._h, _ptr(value), _ptr(reward), _ptr(logits), _ptr(hidden), *ptrs, self.stream),
               'mz_expand_backup_select')
    return out

  def finalize(self, temperature, uniform=None, move=0):
    if torch.is_tensor(temperature):
      t = temperature.to(self.device, torch.float64).expand(self.B)
    else:
      t = torch.as_tensor(np.broadcast_to(np.asarray(temperature, np.float64), (self.B,)).copy())
    t = self._dev(t, torch.float64)
    u = self._dev(uniform, torch.float64)
    action = torch.empty(self.B, dtype=torch.int32, device=self.device)
    cv = torch.empty(self.B, self.A, dtype=torch.float64, device=self.device)
    rv = torch.empty(self.B, dtype=torch.float64, device=self.device)
    err = torch.empty(self.B, dtype=torch.float64, device=self.device)
    vc = torch.empty(self.B, self.A, dtype=torch.int32, device=self.device)
    _abi.check(self.lib.mz_finalize(self._h, _ptr(t), _ptr(u), int(move), _ptr(action), _ptr(cv), _ptr(rv), _ptr(err),
                                    _ptr(vc), self.stream), 'mz_finalize')
    return dict(action=action, child_visits=cv, root_value=rv, error=err, visit_counts=vc)

  def export_tree(self, hidden=False):
    B, NN, A = self.B, self.NN, self.A
    d = dict(N=np.zeros((B, NN), np.int32), W=np.zeros((B, NN)), P=np.zeros((B, NN)),
             R=np.zeros((B, NN), np.float32), E=np.zeros((B, NN), np.int32), TP=np.zeros((B, NN), np.int8),
             legal=np.zeros(B, np.uint32), minmax=np.zeros((B, 2)), noise=np.zeros((B, A)))
    hp = np.zeros((B, self.sims + 1, H), np.float32) if hidden else None
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _abi.check(self.lib.mz_export_tree(self._h, p(d['N']), p(d['W']), p(d['P']), p(d['R']), p(d['E']), p(d['TP']),
                                       p(d['legal']), p(d['minmax']), p(d['noise']), p(hp)), 'mz_export_tree')
    if hidden:
      d['hidden'] = hp
    # EX (node exists): root, legal root children, and every child slab of an expanded node
    EX = np.zeros((B, NN), np.uint8)
    EX[:, 0] = 1
    for a in range(A):
      EX[:, 1 + a] = (d['legal'] >> a) & 1
    exp_idx = d['E']
    for b in range(B):
      for e in np.unique(exp_idx[b][exp_idx[b] > 0]):
        EX[b, 1 + e * A:1 + (e + 1) * A] = 1
    d['EX'] = EX
    return d

  # ---- on-device self-play (actors.py:126-176 on synthetic envs)
  def selfplay_reset(self, episode_len, temperature=1.0, stagger=False):
    _abi.check(self.lib.mz_selfplay_reset(self._h, int(episode_len), float(temperature), int(stagger), self.stream),
               'mz_selfplay_reset')
    self.rec_floats = self.lib.mz_selfplay_rec_floats(self._h)
    self.ring_moves = self.lib.mz_selfplay_ring_moves(self._h)

  def selfplay_set_temperature(self, temperature):
    """Temperature every environment's next game starts with (actors.py:128-129); games in progress keep theirs."""
    _abi.check(self.lib.mz_selfplay_set_temperature(self._h, float(temperature), self.stream),
               'mz_selfplay_set_temperature')

  def selfplay_set_moves(self, moves):
    """every environment's move counter (RNG key, record placement); the device ring must be drained"""
    _abi.check(self.lib.mz_selfplay_set_moves(self._h, int(moves)), 'mz_selfplay_set_moves')

  def selfplay_set_obs(self, uint8_obs=False, obs_min=None, obs_range=None, packed=False):
    """Synthetic observations as bytes (the -ram- envs) and / or --norm_obs (actors.py:55-58,134-137): obs_min and
    obs_range are broadcast to obs_dim like numpy does in the reference's (obs - min) / range.  packed: the experience
    records carry the byte observations four per float slot (records_view(..., obs_u8=True), a replay with obs_u8)."""
    mn = rg = None
    if obs_min is not None:
      mn = np.ascontiguousarray(np.broadcast_to(np.asarray(obs_min, np.float32).reshape(-1), (self.O,)))
      rg = np.ascontiguousarray(np.broadcast_to(np.asarray(obs_range, np.float32).reshape(-1), (self.O,)))
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    if packed and not uint8_obs:
      raise ValueError('packed records need byte-valued observations (uint8_obs=True)')
    _abi.check(self.lib.mz_selfplay_set_obs(self._h, 2 if packed else int(bool(uint8_obs)), p(mn), p(rg)), 'mz_selfplay_set_obs')
    self.obs_packed = bool(packed)
    self.rec_floats = self.lib.mz_selfplay_rec_floats(self._h)

  ENVS = {'synthetic': 0, 'tictactoe': 1}

  def selfplay_set_env(self, kind):
    """'synthetic' (default) or 'tictactoe' (the reference's custom_environments/tic_tac_toe.py on the device); before selfplay_reset"""
    _abi.check(self.lib.mz_selfplay_set_env(self._h, self.ENVS[kind] if isinstance(kind, str) else int(kind)), 'mz_selfplay_set_env')

  def selfplay_set_draws(self, noise=None, uniform=None):
    """parity runs of a game environment: the next moves use these Dirichlet draws [B, A] / select_action uniforms [B]"""
    noise = self._dev(noise, torch.float64); uniform = self._dev(uniform, torch.float64)
    _abi.check(self.lib.mz_selfplay_set_draws(self._h, _ptr(noise), _ptr(uniform), self.stream), 'mz_selfplay_set_draws')

  def selfplay_export_trees(self, keep=True):
    _abi.check(self.lib.mz_selfplay_export_trees(self._h, int(bool(keep))), 'mz_selfplay_export_trees')

  def selfplay_noise_log(self, keep=True):
    """Keep every move's Dirichlet draw (test instrumentation; selfplay_noise(move) reads one move's draws)."""
    _abi.check(self.lib.mz_selfplay_noise_log(self._h, int(bool(keep))), 'mz_selfplay_noise_log')

  def selfplay_noise(self, move):
    out = np.zeros((self.B, self.A), np.float64)
    _abi.check(self.lib.mz_selfplay_read_noise(self._h, int(move), out.ctypes.data_as(C.c_void_p)), 'mz_selfplay_read_noise')
    return out

  SIM_IO = {'off': 0, 'log': 1, 'inject': 2}

  def sim_io(self, mode, keep_moves=1, values=None):
    """Test instrumentation of the fused search kernels (mz_sim_io): 'log' -> returns the device tensor
    [keep_moves, B, sims + 1, 2 + A] float32 the kernels fill with (value, reward, logits) per tree and simulation (slot 0 =
    the root of a self-play move); 'inject' -> `values` ([B, sims + 1, 2 + A] or [1, B, ...]) are what mz_search's simulations
    consume instead of their own network outputs; 'off' -> production state."""
    m = self.SIM_IO[mode] if isinstance(mode, str) else int(mode)
    buf = None
    if m == 1:
      buf = torch.zeros(int(keep_moves), self.B, self.sims + 1, 2 + self.A, dtype=torch.float32, device=self.device)
    elif m == 2:
      buf = torch.as_tensor(np.ascontiguousarray(values, np.float32)).reshape(1, self.B, self.sims + 1, 2 + self.A)
      buf = buf.to(self.device).contiguous()
      keep_moves = 1
    torch.cuda.synchronize(self.device)
    _abi.check(self.lib.mz_sim_io(self._h, m, _ptr(buf), int(keep_moves)), 'mz_sim_io')
    self._sim_io_buf = buf          # (the engine reads / writes it until the mode changes)
    return buf

  def search_kernel_info(self):
    """dict(kind='standalone' | 'fused' | 'split_f16', lt=LDS placement of the trees, ks1, G) of the kernel mz_search /
    mz_selfplay_steps launch right now (mz_search_kernel_info)"""
    out = (C.c_int * 4)()
    _abi.check(self.lib.mz_search_kernel_info(self._h, out), 'mz_search_kernel_info')
    return dict(kind=('standalone', 'fused', 'split_f16')[out[0]], lt=int(out[1]), ks1=int(out[2]), G=int(out[3]))

  def selfplay_steps(self, moves):
    _abi.check(self.lib.mz_selfplay_steps(self._h, int(moves), self.stream), 'mz_selfplay_steps')

  def selfplay_steps_into(self, out, moves):
    """moves self-play moves whose records the kernels write straight into `out` (pinned host tensor
    [>= moves, B, rec_floats]); complete when the work queued on the current stream behind this call is."""
    if not out.is_pinned() or out.dtype != torch.float32 or not out.is_contiguous() \
        or out.numel() < int(moves) * self.B * self.rec_floats:
      raise ValueError('selfplay_steps_into: out must be a pinned contiguous float32 tensor of [moves, B, rec_floats]')
    _abi.check(self.lib.mz_selfplay_steps_into(self._h, int(moves), C.c_void_p(out.data_ptr()), self.stream),
               'mz_selfplay_steps_into')

  def selfplay_steps_timed(self, moves):
    """moves self-play moves launched eagerly back to back, events around every search-kernel dispatch;
    returns the durations in milliseconds (synchronous)."""
    ms = (C.c_float * int(moves))()
    _abi.check(self.lib.mz_selfplay_steps_timed(self._h, int(moves), ms, self.stream), 'mz_selfplay_steps_timed')
    return [float(x) for x in ms]

  def selfplay_moves_per_launch(self):
    """16 where selfplay_steps plays whole moves inside one launch of the search kernel, 0 where every move is a root
    kernel + a search kernel (two-player games, trees in the global pool, MZ_NO_PERSIST)."""
    return int(self.lib.mz_selfplay_moves_per_launch(self._h))

  SELFPLAY_PHASES = ('root_stage0', 'root_rep_ln', 'root_prediction', 'root_tree', 'resident+tree_setup', 'ring+barrier',
                     'simulations', 'end_of_move')

  def selfplay_phase_profile(self, moves=16):
    """Shader cycles per move and phase of the persistent self-play launch (diagnostic, synchronous): dict phase -> cycles."""
    out = (C.c_double * 8)()
    _abi.check(self.lib.mz_selfplay_phase_profile(self._h, int(moves), out, self.stream), 'mz_selfplay_phase_profile')
    return dict(zip(self.SELFPLAY_PHASES, [float(x) for x in out]))

  def selfplay_drain(self, out=None, max_moves=None, copy_stream=None):
    """Asynchronous D2H of the records produced since the last drain into pinned memory; returns
    (host tensor [n_moves, B, rec_floats], n_moves).  Synchronise the stream before reading.
    copy_stream: issue the copy there, ordered after the work queued on the current stream so far -- the next
    moves on the current stream then overlap the copy (the ring keeps them in different slots)."""
    max_moves = self.ring_moves if max_moves is None else int(max_moves)
    if out is None:
      out = torch.empty(max_moves, self.B, self.rec_floats, dtype=torch.float32).pin_memory()
    n = C.c_int(0)
    stream = self.stream
    if copy_stream is not None:
      copy_stream.wait_stream(torch.cuda.current_stream(self.device))
      stream = C.c_void_p(copy_stream.cuda_stream)
    _abi.check(self.lib.mz_selfplay_drain(self._h, C.c_void_p(out.data_ptr()), max_moves, C.byref(n), stream),
               'mz_selfplay_drain')
    return out, n.value

  def synth_obs(self, env, episode, t):
    obs = np.zeros(self.O, np.float32)
    r = C.c_float(0)
    _abi.check(self.lib.mz_synth_obs(self._h, env, episode, t, obs.ctypes.data_as(C.c_void_p), C.byref(r)),
               'mz_synth_obs')
    return obs, float(r.value)
