"""Host-side prioritized replay with the reference's surface (replay_buffer.py:6-210), backed by the
native library libmz_replay.so (include/mz_replay.h): SumTree arithmetic and update order are the
reference's, so sums are bit-identical for the same sequence of histories.

PrioritizedReplay.{save_history, size, get_throughput, add_initial_throughput, get_priorities, update}
keep the reference's signatures; `ingest_records` is the bulk path used by the GPU actor pool (records
from Engine.selfplay_drain) and applies the actor-side flush rules (actors.py:160-169) per environment.
"""
import ctypes as C
import random

import numpy as np

from . import _abi


def _p(a):
  """the array's address for a void * parameter (cheaper than ndarray.ctypes.data_as: these calls sit in the sampling loop)"""
  return None if a is None else C.c_void_p(a.__array_interface__['data'][0])


class SumTree(object):
  """replay_buffer.py:6-66 over the native tree (payload handling lives in PrioritizedReplay)."""

  def __init__(self, max_capacity, capacity_step, _owner=None, obs_dim=1, action_space=1):
    self.max_capacity, self.capacity_step = int(max_capacity), int(capacity_step)
    self.lib = _abi.load_replay()
    if _owner is None:
      cfg = _abi.MzrConfig(self.max_capacity, self.capacity_step, obs_dim, action_space, 5, 10, 500, 256, 0.01, 1.0,
                           1.0, 0.001, 0.997, 0)
      h = C.c_void_p()
      _abi.check_replay(self.lib.mzr_create(C.byref(cfg), C.byref(h)), 'mzr_create')
      self._h, self._own = h, True
    else:
      self._h, self._own = _owner, False
    self._rows = (False, int(obs_dim), int(action_space))      # (byte observations, obs_dim, action_space) of the record rows

  def __del__(self):
    if getattr(self, '_own', False) and getattr(self, '_h', None):
      self.lib.mzr_destroy(self._h)
      self._h = None

  def add(self, priorities, history=None):
    pri = np.ascontiguousarray(priorities, np.float64)
    pos = np.zeros(pri.size, np.int64)
    _abi.check_replay(self.lib.mzr_tree_add(self._h, _p(pri), pri.size, _p(pos)), 'mzr_tree_add')
    return pos

  def update(self, idx, priority):
    idx = np.ascontiguousarray(np.atleast_1d(idx), np.int64)
    pri = np.ascontiguousarray(np.atleast_1d(priority), np.float64)
    _abi.check_replay(self.lib.mzr_tree_update(self._h, _p(idx), _p(pri), idx.size), 'mzr_tree_update')

  def get_leaf_index(self, value):
    return int(self.lib.mzr_tree_get_leaf(self._h, float(value)))

  def get_leaf(self, value):
    """replay_buffer.py:42-62: (leaf_index, priority, step, history) -- the history comes back as a HistorySlice
    (game.py:5-16) rebuilt from the record rows the native replay keeps (None for a priorities-only leaf); like the
    stored slice it has as many observations as steps (the reference's History keeps one more, game.py:93-96)."""
    idx = self.get_leaf_index(value)
    pri, step, n, has = C.c_double(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
    _abi.check_replay(self.lib.mzr_leaf_info(self._h, idx, C.byref(pri), C.byref(step), C.byref(n), C.byref(has)), 'mzr_leaf_info')
    history = None
    if has.value:
      from .engine import REC_EXTRA, records_view
      from .game import HistorySlice
      u8, O, A = self._rows
      rows = np.zeros((n.value, ((O + 3) // 4 if u8 else O) + A + REC_EXTRA), np.float32)
      _abi.check_replay(self.lib.mzr_leaf_history(self._h, idx, _p(rows), n.value), 'mzr_leaf_history')
      v = records_view(rows, O, A, obs_u8=u8)
      history = HistorySlice(list(v['obs']), [list(map(float, c)) for c in v['child_visits']], v['root_value'].tolist(),
                             v['action'].tolist(), v['reward'].tolist(), v['error'].tolist(),
                             [bool(d) for d in v['done']], v['step'].tolist(), [], v['to_play'].tolist())
    return idx, pri.value, int(step.value), history

  @property
  def total_priority(self):
    return float(self.lib.mzr_total_priority(self._h))

  @property
  def num_memories(self):
    return int(self.lib.mzr_size(self._h))

  def leaves(self, n=None):
    n = self.num_memories if n is None else int(n)
    out = np.zeros(n, np.float64)
    _abi.check_replay(self.lib.mzr_tree_leaves(self._h, n, _p(out)), 'mzr_tree_leaves')
    return out


def default_ingest_threads(config=None):
  """--ingest_threads, else 4 -- 8 from batch size 1024 up, where sampling a batch and refreshing its priorities (both dealt to these
  threads) are most of the learner loop's host time -- bounded by the CPUs this process may use (affinity mask and cgroup quota; one is
  left to the launching thread)"""
  n = getattr(config, 'ingest_threads', None)
  if n:
    return max(1, int(n))
  from .distributed import usable_cores
  want = 8 if int(getattr(config, 'batch_size', 0) or 0) >= 1024 else 4
  return max(1, min(want, usable_cores() - 1))


def native_config(config):
  """mzr_config (include/mz_replay.h) of a run's Config: what the native replay -- and a producing rank's assembler -- is made from"""
  capacity = int(config.window_size)
  step = int(config.window_step) if getattr(config, 'window_step', None) is not None else capacity
  return _abi.MzrConfig(capacity, step, int(np.prod(config.obs_space)), int(config.action_space), int(config.num_unroll_steps),
                        int(config.td_steps), int(getattr(config, 'max_history_length', 500)), int(config.batch_size),
                        float(config.epsilon), float(config.alpha), float(config.beta),
                        float(getattr(config, 'beta_increment_per_sampling', 0.001)), float(config.discount),
                        int(config.seed or 0), int(bool(getattr(config, 'two_players', False))),
                        int(bool(getattr(config, 'episode_life', False))), default_ingest_threads(config),
                        int(bool(getattr(config, 'obs_u8', False))))


class PrioritizedReplay(object):

  def __init__(self, config):
    self.config = config
    self.batch_size = config.batch_size
    self.epsilon, self.alpha, self.beta = config.epsilon, config.alpha, config.beta
    self.obs_dim = int(np.prod(config.obs_space))
    self.action_space = int(config.action_space)
    capacity = int(config.window_size)
    step = int(config.window_step) if getattr(config, 'window_step', None) is not None else capacity
    self.lib = _abi.load_replay()
    cfg = native_config(config)
    h = C.c_void_p()
    _abi.check_replay(self.lib.mzr_create(C.byref(cfg), C.byref(h)), 'mzr_create')
    self._h = h
    self.tree = SumTree(capacity, step, _owner=self._h, obs_dim=self.obs_dim, action_space=self.action_space)
    self.tree._rows = (bool(getattr(config, 'obs_u8', False)), self.obs_dim, self.action_space)
    if config.seed is not None:          # replay_buffer.py:104-106
      np.random.seed(config.seed)
      random.seed(config.seed + 1)

  def __del__(self):
    if getattr(self, '_h', None):
      self.lib.mzr_destroy(self._h)
      self._h = None

  # replay_buffer.py:106-108
  def add_initial_throughput(self, frames, games):
    _abi.check_replay(self.lib.mzr_add_initial_throughput(self._h, int(frames), int(games)))

  # replay_buffer.py:110-111
  def get_priorities(self, errors):
    if isinstance(errors, np.ndarray) and errors.dtype == np.float32:
      # the learner's refresh (learners.py:181-182: a float32 array): numpy computes (|e| + epsilon) ** alpha in float32 here
      errors = np.ascontiguousarray(errors)
      out = np.empty_like(errors)
      _abi.check_replay(self.lib.mzr_priorities_f32(self._h, _p(errors), errors.size, _p(out)))
      return out
    errors = np.ascontiguousarray(errors, np.float64)
    out = np.zeros_like(errors)
    _abi.check_replay(self.lib.mzr_priorities(self._h, _p(errors), errors.size, _p(out)))
    return out

  # replay_buffer.py:113-122; `history` is a HistorySlice-like object (game.py:5-16)
  def save_history(self, history, ignore=None, terminal=False):
    n = len(history.errors)
    errors = np.ascontiguousarray(history.errors, np.float64)
    O, A = self.obs_dim, self.action_space
    obs = np.ascontiguousarray(np.asarray(history.observations[:n], np.float32).reshape(n, O)) if n else None
    cv = np.ascontiguousarray(np.asarray(history.child_visits, np.float32).reshape(n, A)) if n else None
    rv = np.ascontiguousarray(history.root_values, np.float64)
    rew = np.ascontiguousarray(history.rewards, np.float32)
    act = np.ascontiguousarray(history.actions, np.int32)
    dn = np.ascontiguousarray(history.dones, np.uint8)
    tp = np.ascontiguousarray(history.to_play, np.int8)
    _abi.check_replay(self.lib.mzr_save_history(self._h, n, _p(errors), -1 if ignore is None else int(ignore),
                                                int(bool(terminal)), _p(obs), _p(cv), _p(rv), _p(rew), _p(act), _p(dn),
                                                _p(tp)), 'mzr_save_history')

  def ingest_records(self, records, n_moves, B, env_base=0, env_major=False):
    """records: float32 [n_moves, B, rec_floats] host array/tensor from Engine.selfplay_drain.  env_base: index of
    the records' first environment among this replay's environments (one replay fed by several actor ranks: rank r
    passes r * B -- every reference actor sends to the one replay buffer, train.py:71-72, actors.py:169).
    env_major: the chunk was packed by the producing rank (mzr_pack_env_major: [B, n_moves, rec_floats])."""
    if hasattr(records, 'data_ptr'):
      ptr, rec = C.c_void_p(records.data_ptr()), int(records.shape[-1])
    else:
      records = np.ascontiguousarray(records, np.float32)
      ptr, rec = _p(records), int(records.shape[-1])
    fn = self.lib.mzr_ingest_records_packed if env_major else self.lib.mzr_ingest_records_from
    _abi.check_replay(fn(self._h, ptr, int(n_moves), int(B), rec, int(env_base)), 'mzr_ingest_records')

  def ingest_slices(self, blob, nbytes, env_base=0):
    """finished history slices assembled by a PRODUCING rank (distributed.RingReplay -> mz_assembler; include/mz_replay.h): one
    sequential copy per slice and the insertion of its leaves -- the one-replay layout's hand-off (train --ranks N)"""
    ptr = C.c_void_p(blob.data_ptr()) if hasattr(blob, 'data_ptr') else C.c_void_p(np.asarray(blob).ctypes.data)
    _abi.check_replay(self.lib.mzr_ingest_slices(self._h, ptr, int(nbytes), int(env_base)), 'mzr_ingest_slices')

  # replay_buffer.py:124-163 (+ insert_target 165-198 inside the native call)
  def sample_batch_arrays(self):
    """sample_batch as the arrays the learner step consumes, no Python lists in between: (dict obs float32 [bs, ...], act
    int32 [bs, K], t_rew / t_val float32 [bs, K + 1], t_pol float32 [bs, K + 1, A], w float64 [bs]), idxs int64 [bs].  The
    same draws as sample_batch (which wraps this): stratified random.uniform segments in the reference's order."""
    return self.sample_batches_arrays(1)[0]

  def sample_batches_arrays(self, n):
    """n consecutive sample_batch_arrays calls in one native call (the learner samples a few batches ahead of its updates
    anyway, learners.py:124): the same generator words, beta steps and weights as n calls in a row would give with no
    priority refresh in between."""
    bs, K, A, O = self.batch_size, int(self.config.num_unroll_steps), self.action_space, self.obs_dim
    betas = np.empty(n, np.float64)
    for j in range(n):
      if self.beta < 1:
        self.beta = np.float64(min(1., self.beta + getattr(self.config, 'beta_increment_per_sampling', 0.001)))
      betas[j] = self.beta
    # random.uniform(a, b) is a + (b - a) * random.random() (CPython's random.py), and random.random() is two consecutive
    # 32-bit Mersenne Twister outputs a, b -> ((a >> 5) * 2**26 + (b >> 6)) / 2**53 (_randommodule.c): the bs draws of the
    # reference's loop (replay_buffer.py:138-140) come, bit for bit and from the same generator state, out of ONE
    # getrandbits call (its words come out least significant first); the arithmetic runs inside the native call
    words = np.frombuffer(random.getrandbits(64 * bs * n).to_bytes(8 * bs * n, 'little'), np.uint32)
    obs = np.empty((n, bs) + tuple(self.config.obs_space), np.float32)      # (the native call writes every element)
    actions = np.empty((n, bs, K), np.int32)
    t_rew = np.empty((n, bs, K + 1), np.float32); t_val = np.empty((n, bs, K + 1), np.float32)
    t_pol = np.empty((n, bs, K + 1, A), np.float32)
    idxs = np.empty((n, bs), np.int64); probs = np.empty((n, bs), np.float64); info = np.empty((n, 2), np.int64)
    _abi.check_replay(self.lib.mzr_sample_batches_words(self._h, _p(words), n, bs, _p(obs), _p(actions), _p(t_rew), _p(t_val),
                                                        _p(t_pol), _p(idxs), _p(probs), _p(info)), 'mzr_sample_batches_words')
    if info[:, 0].any():
      for j in range(n):
        for i_, k in zip(*np.nonzero(actions[j] < 0)):      # replay_buffer.py:150-151, in the reference's draw order
          actions[j, i_, k] = np.random.randint(A)
    is_weights = np.power(info[:, 1:2] * probs, -betas[:, None])
    is_weights /= is_weights.max(axis=1, keepdims=True)
    return [({'obs': obs[j], 'act': actions[j], 't_rew': t_rew[j], 't_val': t_val[j], 't_pol': t_pol[j], 'w': is_weights[j]}, idxs[j])
            for j in range(n)]

  def sample_batch(self):
    b, idxs = self.sample_batch_arrays()
    return (b['obs'], b['act'].tolist(), (b['t_rew'], b['t_val'], b['t_pol'])), idxs.tolist(), b['w']

  # replay_buffer.py:200-203
  def update(self, idxs, errors):
    self.tree.update(np.asarray(idxs, np.int64), self.get_priorities(errors))

  def set_ingest_threads(self, threads):
    _abi.check_replay(self.lib.mzr_set_ingest_threads(self._h, int(threads)), 'mzr_set_ingest_threads')

  @property
  def ingest_threads(self):
    return int(self.lib.mzr_ingest_threads(self._h))

  def get_ingest_threads(self):
    return self.ingest_threads

  def size(self):
    return int(self.lib.mzr_size(self._h))

  def get_throughput(self):
    return {'frames': int(self.lib.mzr_frames(self._h)), 'games': int(self.lib.mzr_games(self._h))}
