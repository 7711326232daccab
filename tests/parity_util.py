"""Helpers shared by the GPU parity tests (no test functions here)."""
import os

import numpy as np


def philox_action_uniform(seed, env, move):
  """The uniform the device's select_action consumes for (env, move): Philox4x32-10 keyed by the engine seed, counter
  (env, move lo, move hi, MZ_RNG_ACTION << 24), 53 bits of (x, y) (csrc/mz_rng.h, csrc/mz_tree.hip.h:mz_finalize_tree) --
  integer arithmetic, restated here so that the oracle's Config.select_action gets the draw the device used."""
  env = np.asarray(env, np.uint64)
  c = [env & np.uint64(0xFFFFFFFF), np.full_like(env, move & 0xFFFFFFFF), np.full_like(env, move >> 32),
       np.full_like(env, 4 << 24)]
  k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
  M = np.uint64(0xFFFFFFFF)
  for _ in range(10):
    p0 = np.uint64(0xD2511F53) * c[0]
    p1 = np.uint64(0xCD9E8D57) * c[2]
    c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & M, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & M]
    k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
  v = ((c[0] << np.uint64(32)) | c[1]) >> np.uint64(11)
  return v.astype(np.float64) * (1.0 / 9007199254740992.0)


def ulp_diff(a, b):
  a = np.ascontiguousarray(a, np.float64).view(np.int64)
  b = np.ascontiguousarray(b, np.float64).view(np.int64)
  return np.abs(a - b)


def random_weights(O, A, seed=0, scale=1.0):
  """FCNetwork-shaped weights (reference key names) with PyTorch-like magnitudes: the injected runs never look at what
  the network computes, they only need a weight set the fused kernel accepts (mz_set_weights' clamp-ReLU scale)."""
  from oracle import oracle as orc
  rng = np.random.RandomState(seed)
  shapes = {'representation_head.fc1': (512, O), 'representation_head.out': (50, 512), 'value_head.fc1': (512, 50),
            'value_head.value': (31, 512), 'policy_head.fc1': (512, 50), 'policy_head.policy': (A, 512),
            'reward_head.fc1': (512, 50 + A), 'reward_head.reward': (31, 512), 'transition_head.fc1': (512, 50 + A),
            'transition_head.out': (50, 512)}
  w = {}
  for k, (n, m) in shapes.items():
    w[k + '.weight'] = (rng.uniform(-1, 1, (n, m)) * scale / np.sqrt(m)).astype(np.float32)
    w[k + '.bias'] = (rng.uniform(-1, 1, n) * scale / np.sqrt(m)).astype(np.float32)
  w['LN.weight'] = np.ones(50, np.float32)
  w['LN.bias'] = np.zeros(50, np.float32)
  assert set(w) == set(orc.WEIGHT_ORDER)
  return w


class env_switches(object):
  """MZ_* switches are read by mz_create: set them around the construction of an engine"""

  def __init__(self, **kv):
    self.kv = {k: v for k, v in kv.items() if v is not None}

  def __enter__(self):
    self.old = {k: os.environ.get(k) for k in ('MZ_NO_LDS_TREES', 'MZ_NO_LDS_HYBRID', 'MZ_SPLIT_F16', 'MZ_NO_PERSIST', 'MZ_NO_FUSED')}
    for k in self.old:
      os.environ.pop(k, None)
    os.environ.update(self.kv)

  def __exit__(self, *a):
    for k, v in self.old.items():
      os.environ.pop(k, None)
      if v is not None:
        os.environ[k] = v


def replay_move(cfg, B, A, sims, io, noise, frac, to_play, legal, temperature, uniform, want_tree=False):
  """one move of B trees through the oracle's TREE on logged network outputs io [B, sims + 1, 2 + A]"""
  from oracle import oracle as orc
  t = orc.Trees(cfg, B)
  t.root_expand(to_play, io[:, 0, 2:], legal)
  t.add_noise(noise, frac)
  for s in range(sims):
    t.select()
    t.expand_backup(io[:, 1 + s, 0], io[:, 1 + s, 1], io[:, 1 + s, 2:])
  action, cv, rv, vc = t.finalize(temperature, uniform)
  return dict(action=action, child_visits=cv, root_value=rv, visit_counts=vc, v0=io[:, 0, 0], margin=t.margin(),
              tree=t.export() if want_tree else None)
