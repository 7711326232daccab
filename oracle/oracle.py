"""ctypes front-end of the CPU oracle (oracle/mz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.  See the header of mz_oracle.c for what pins it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
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


class TreeCfg(C.Structure):
  _fields_ = [('A', C.c_int32), ('sims', C.c_int32), ('two_players', C.c_int32),
              ('has_min', C.c_int32), ('has_max', C.c_int32),
              ('min_bound', C.c_double), ('max_bound', C.c_double),
              ('discount', C.c_double), ('pb_c_base', C.c_double), ('pb_c_init', C.c_double),
              ('init_value_score', C.c_double)]


_FP = C.POINTER(C.c_float)


class FC(C.Structure):
  _fields_ = [('O', C.c_int32), ('A', C.c_int32), ('S', C.c_int32), ('support_min', C.c_int32),
              ('no_target_transform', C.c_int32)] + [(n, _FP) for n in (
                  'rep_w1', 'rep_b1', 'rep_w2', 'rep_b2', 'val_w1', 'val_b1', 'val_w2', 'val_b2',
                  'pol_w1', 'pol_b1', 'pol_w2', 'pol_b2', 'rew_w1', 'rew_b1', 'rew_w2', 'rew_b2',
                  'tr_w1', 'tr_b1', 'tr_w2', 'tr_b2', 'ln_w', 'ln_b')]


def build(force=False):
  if os.environ.get('MZ_ORACLE_LIB'):      # a sanitizer build of the same source (tests/test_sanitizers.py)
    return os.environ['MZ_ORACLE_LIB']
  so = os.path.join(_HERE, 'libmz_oracle.so')
  src = os.path.join(_HERE, 'mz_oracle.c')
  if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(['make', '-s', '-C', _HERE, 'libmz_oracle.so'])
  return so


def lib():
  global _LIB
  if _LIB is None:
    _LIB = C.CDLL(build())
    _LIB.orc_trees_create.restype = C.c_void_p
    _LIB.orc_sumtree_create.restype = C.c_void_p
    _LIB.orc_sumtree_total.restype = C.c_double
    _LIB.orc_sumtree_num_memories.restype = C.c_int64
    _LIB.orc_sumtree_get_leaf.restype = C.c_int64
    _LIB.orc_inverse_transform_row.restype = C.c_float
  return _LIB


def _p(a):
  return a.ctypes.data_as(C.c_void_p) if a is not None else None


def tree_cfg(A, sims, two_players=False, known_bounds=(None, None), discount=0.997, pb_c_base=19652,
             pb_c_init=1.25, init_value_score=0.0):
  lo, hi = known_bounds
  lo = None if (lo is None or (isinstance(lo, float) and np.isnan(lo))) else float(lo)
  hi = None if (hi is None or (isinstance(hi, float) and np.isnan(hi))) else float(hi)
  return TreeCfg(int(A), int(sims), int(bool(two_players)), int(lo is not None), int(hi is not None),
                 0.0 if lo is None else lo, 0.0 if hi is None else hi, float(discount), float(pb_c_base),
                 float(pb_c_init), float(init_value_score))


class Trees(object):
  """B independent search trees (the reference's Node/MCTS/MinMaxStats objects, SoA)."""

  def __init__(self, cfg, B):
    self.cfg, self.B, self.A, self.sims = cfg, B, cfg.A, cfg.sims
    self.NN = 1 + (cfg.sims + 1) * cfg.A
    self._h = C.c_void_p(lib().orc_trees_create(C.byref(cfg), C.c_int32(B)))

  def __del__(self):
    if getattr(self, '_h', None):
      lib().orc_trees_destroy(self._h)
      self._h = None

  def root_expand(self, to_play, logits, legal=None):
    to_play = np.ascontiguousarray(to_play, np.int8)
    logits = np.ascontiguousarray(logits, np.float32)
    legal = None if legal is None else np.ascontiguousarray(legal, np.uint8)
    lib().orc_root_expand(self._h, _p(to_play), _p(logits), _p(legal))

  def set_prior_override(self, priors):
    """test hook: every later Node.expand takes its children's priors from priors [B][NN] (a device tree's export; root children
    post-noise: skip add_noise then) instead of exp / sum on the logits; None switches it off"""
    self._prior_override = None if priors is None else np.ascontiguousarray(priors, np.float64)      # (kept alive here)
    lib().orc_set_prior_override(self._h, _p(self._prior_override))

  def add_noise(self, noise, frac):
    noise = np.ascontiguousarray(noise, np.float64)
    lib().orc_add_noise(self._h, _p(noise), C.c_double(frac))

  def select(self):
    out = [np.zeros(self.B, np.int32) for _ in range(4)]
    lib().orc_select(self._h, *[_p(o) for o in out])
    return out   # leaf_node, parent_slot, action, depth

  def paths(self):
    out = np.zeros((self.B, self.sims + 2), np.int32)
    lib().orc_get_paths(self._h, _p(out))
    return out

  def margin(self):
    """per tree: the smallest gap between the best and the second-best score over all select_child decisions since
    root_expand (mcts.py:104-113) -- what make_goldens.py records from the reference as `min_margin`."""
    out = np.zeros(self.B, np.float64)
    lib().orc_get_margin(self._h, _p(out))
    return out

  def expand_backup(self, value, reward, logits):
    value = np.ascontiguousarray(value, np.float32)
    reward = np.ascontiguousarray(reward, np.float32)
    logits = np.ascontiguousarray(logits, np.float32)
    lib().orc_expand_backup(self._h, _p(value), _p(reward), _p(logits))

  def finalize(self, temperature, uniform):
    temperature = np.ascontiguousarray(np.broadcast_to(temperature, (self.B,)), np.float64)
    uniform = np.ascontiguousarray(np.broadcast_to(uniform, (self.B,)), np.float64)
    action = np.zeros(self.B, np.int32)
    cv = np.zeros((self.B, self.A), np.float64)
    rv = np.zeros(self.B, np.float64)
    vc = np.zeros((self.B, self.A), np.int32)
    lib().orc_finalize(self._h, _p(temperature), _p(uniform), _p(action), _p(cv), _p(rv), _p(vc))
    return action, cv, rv, vc

  def export(self):
    B, NN = self.B, self.NN
    d = dict(N=np.zeros((B, NN), np.int32), W=np.zeros((B, NN)), P=np.zeros((B, NN)), R=np.zeros((B, NN)),
             E=np.zeros((B, NN), np.int32), TP=np.zeros((B, NN), np.int8), EX=np.zeros((B, NN), np.uint8),
             minmax=np.zeros((B, 2)))
    lib().orc_export(self._h, *[_p(d[k]) for k in ('N', 'W', 'P', 'R', 'E', 'TP', 'EX', 'minmax')])
    return d

  def search_fc(self, net, obs, to_play, legal, noise, frac):
    obs = np.ascontiguousarray(obs, np.float32)
    to_play = np.ascontiguousarray(to_play, np.int8)
    legal = None if legal is None else np.ascontiguousarray(legal, np.uint8)
    noise = None if noise is None else np.ascontiguousarray(noise, np.float64)
    hpool = np.zeros((self.B, self.sims + 1, H), np.float32)
    v0 = np.zeros(self.B, np.float32)
    lib().orc_search_fc(self._h, C.byref(net.c), _p(obs), _p(to_play), _p(legal), _p(noise), C.c_double(frac),
                        _p(hpool), _p(v0))
    return hpool, v0


def search_fc_threads(cfg, net, obs, to_play=None, legal=None, noise=None, frac=0.25, temperature=1.0, uniform=None,
                      threads=None, tree=True):
  """Trees.search_fc + finalize (+ export) for a large batch, the trees split over `threads` host threads (the
  trees are independent and ctypes releases the GIL).  Returns a dict: action, child_visits, root_value, visit_counts,
  margin, v0, hpool and, with tree=True, 'tree' = Trees.export()."""
  from concurrent.futures import ThreadPoolExecutor
  obs = np.ascontiguousarray(obs, np.float32)
  B = obs.shape[0]
  threads = max(1, min(threads or len(os.sched_getaffinity(0)), B))
  to_play = np.ones(B, np.int8) if to_play is None else np.ascontiguousarray(to_play, np.int8)
  temperature = np.ascontiguousarray(np.broadcast_to(temperature, (B,)), np.float64)
  uniform = np.zeros(B) if uniform is None else np.ascontiguousarray(np.broadcast_to(uniform, (B,)), np.float64)
  cuts = np.linspace(0, B, threads + 1).astype(int)

  def part(i):
    sl = slice(cuts[i], cuts[i + 1])
    t = Trees(cfg, cuts[i + 1] - cuts[i])
    hpool, v0 = t.search_fc(net, obs[sl], to_play[sl], None if legal is None else legal[sl],
                            None if noise is None else noise[sl], frac)
    action, cv, rv, vc = t.finalize(temperature[sl], uniform[sl])
    return dict(action=action, child_visits=cv, root_value=rv, visit_counts=vc, margin=t.margin(), v0=v0, hpool=hpool,
                tree=t.export() if tree else None)

  with ThreadPoolExecutor(threads) as ex:
    parts = list(ex.map(part, range(threads)))
  out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0] if k != 'tree'}
  if tree:
    out['tree'] = {k: np.concatenate([p['tree'][k] for p in parts]) for k in parts[0]['tree']}
  return out


class FCNet(object):
  """float32 FCNetwork forward (networks.py:122-180) over a dict of reference-named weights."""

  def __init__(self, weights, O, A, support=(-15, 15), no_target_transform=False, no_support=False):
    self.O, self.A = int(O), int(A)
    self.S = 1 if no_support else support[1] - support[0] + 1      # networks.py:135-136
    self._keep = [np.ascontiguousarray(weights[k], np.float32) for k in WEIGHT_ORDER]
    self.c = FC(self.O, self.A, self.S, int(support[0]), 2 if no_support else int(bool(no_target_transform)),
                *[a.ctypes.data_as(_FP) for a in self._keep])

  def initial(self, obs):
    obs = np.ascontiguousarray(obs, np.float32).reshape(-1, self.O)
    n = obs.shape[0]
    h = np.zeros((n, H), np.float32); v = np.zeros(n, np.float32); lg = np.zeros((n, self.A), np.float32)
    lib().orc_fc_initial(C.byref(self.c), _p(obs), C.c_int(n), _p(h), _p(v), _p(lg))
    return h, v, lg

  def recurrent(self, hidden, action):
    hidden = np.ascontiguousarray(hidden, np.float32).reshape(-1, H)
    action = np.ascontiguousarray(action, np.int32)
    n = hidden.shape[0]
    h = np.zeros((n, H), np.float32); r = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    lg = np.zeros((n, self.A), np.float32)
    lib().orc_fc_recurrent(C.byref(self.c), _p(hidden), _p(action), C.c_int(n), _p(h), _p(r), _p(v), _p(lg))
    return h, r, v, lg


def inverse_transform(logits, support_min=-15, no_target_transform=False):
  logits = np.ascontiguousarray(logits, np.float32)
  out = np.zeros(logits.shape[0], np.float32)
  lib().orc_inverse_transform(_p(logits), C.c_int(logits.shape[0]), C.c_int(logits.shape[1]),
                              C.c_int(support_min), C.c_int(int(no_target_transform)), _p(out))
  return out


def priorities(errors, epsilon=0.01, alpha=1.0):
  errors = np.ascontiguousarray(errors, np.float64)
  out = np.zeros_like(errors)
  lib().orc_priorities(_p(errors), C.c_int(errors.size), C.c_double(epsilon), C.c_double(alpha), _p(out))
  return out


class SumTree(object):
  def __init__(self, max_capacity, capacity_step):
    self.max_capacity = max_capacity
    self._h = C.c_void_p(lib().orc_sumtree_create(C.c_int64(max_capacity), C.c_int64(capacity_step)))

  def __del__(self):
    if getattr(self, '_h', None):
      lib().orc_sumtree_destroy(self._h)
      self._h = None

  def add(self, pri):
    pri = np.ascontiguousarray(pri, np.float64)
    pos = np.zeros(pri.size, np.int64)
    lib().orc_sumtree_add(self._h, _p(pri), C.c_int(pri.size), _p(pos))
    return pos

  def update(self, idxs, pri):
    idxs = np.ascontiguousarray(idxs, np.int64); pri = np.ascontiguousarray(pri, np.float64)
    lib().orc_sumtree_update(self._h, _p(idxs), _p(pri), C.c_int(pri.size))

  def get_leaf(self, value):
    return int(lib().orc_sumtree_get_leaf(self._h, C.c_double(value)))

  @property
  def total(self):
    return float(lib().orc_sumtree_total(self._h))

  @property
  def num_memories(self):
    return int(lib().orc_sumtree_num_memories(self._h))

  def leaves(self, n):
    out = np.zeros(n, np.float64)
    lib().orc_sumtree_leaves(self._h, C.c_int64(n), _p(out))
    return out


def load_weights(npz):
  """weights dict {reference state_dict key: array} from a golden file (follows 'weights_file')."""
  if 'weights_file' in npz.files:
    npz = np.load(os.path.join(os.path.dirname(npz.fid.name), str(npz['weights_file'])))
  return {k[2:]: npz[k] for k in npz.files if k.startswith('w.')}
