"""Reference-SHAPED CPU baseline: a batch-1 PyTorch-CPU restatement of one self-play move of the reference
(actors.py:131-153 around mcts.py:78-143, networks.py:122-180, config.py:27-33,70-81), with the reference's
granularity -- one environment, one Python object per node, one tiny torch op per network layer, `.item()` round trips
per simulation -- so that its speed is the speed of the reference's own path on whatever CPU it runs on.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (SURVEY.md s8d "CPU baseline beside it"): the reference itself cannot travel to
the GPU box, this file can.  It is timed by bench.py's `cpu_baseline` leg and by scripts/ref_shaped_ratio.py, which runs
it next to the imported reference in the build container and records the ratio of the two rates (accepted: 1.0 +- 0.1)
under profiles/.  It is never imported by the product package and it is not the parity oracle (that is mz_oracle.c).

  python oracle/ref_shaped.py --obs 8 --actions 4 --sims 30 --moves 200     # prints one JSON line
"""
import argparse
import json
import math
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HIDDEN, WIDTH = 50, 512


class _Head(nn.Module):
  """Linear -> ReLU -> Linear, the shape of every FC head (networks.py:55-119)."""

  def __init__(self, n_in, n_out):
    super().__init__()
    self.fc1 = nn.Linear(n_in, WIDTH)
    self.out = nn.Linear(WIDTH, n_out)

  def forward(self, x):
    return self.out(F.relu(self.fc1(x)))


class ShapedFC(nn.Module):
  """FCNetwork in eval mode (networks.py:122-180): same layers, same op sequence per call."""

  def __init__(self, obs_dim, actions, support=(-15, 15)):
    super().__init__()
    self.actions = actions
    self.support_range = list(range(support[0], support[1] + 1))
    n = len(self.support_range)
    self.representation_head = _Head(obs_dim, HIDDEN)
    self.value_head = _Head(HIDDEN, n)
    self.policy_head = _Head(HIDDEN, actions)
    self.reward_head = _Head(HIDDEN + actions, n)
    self.transition_head = _Head(HIDDEN + actions, HIDDEN)
    self.LN = nn.LayerNorm([HIDDEN])

  def inverse_transform(self, logits):                                  # config.py:27-33, op for op
    p = torch.softmax(logits, dim=1)
    support = torch.tensor(self.support_range, dtype=torch.float, device=p.device).expand(p.shape)
    v = torch.sum(support * p, dim=1, keepdim=True)
    return torch.sign(v) * (((torch.sqrt(1 + 4 * 0.001 * (torch.abs(v) + 1 + 0.001)) - 1) / (2 * 0.001)) ** 2 - 1)

  def prediction(self, h):                                              # networks.py:151-156
    return self.policy_head(h), self.inverse_transform(self.value_head(h))

  def initial_inference(self, obs):                                     # networks.py:26-29,146-149
    x = obs.view(obs.shape[0], -1)
    h = F.relu(self.LN(self.representation_head(x)))
    logits, value = self.prediction(h)
    return value, 0, logits, h

  def recurrent_inference(self, h, action):                             # networks.py:31-34,158-174
    a = torch.from_numpy(np.array(action, dtype=np.int64)[:, np.newaxis])
    one_hot = torch.zeros((len(action), self.actions), dtype=torch.float32)
    one_hot.scatter_(1, a, 1.0)
    x = torch.cat((h, one_hot), dim=1)
    reward = self.inverse_transform(self.reward_head(x))
    h2 = F.relu(self.LN(self.transition_head(x)))
    logits, value = self.prediction(h2)
    return value, reward, logits, h2


class _Node(object):                                                    # mcts.py:28-45
  __slots__ = ('hidden', 'n', 'w', 'reward', 'children', 'prior', 'to_play')

  def __init__(self, prior):
    self.hidden, self.n, self.w, self.reward, self.children, self.prior, self.to_play = None, 0, 0, 0, {}, prior, 1

  def value(self):
    return self.w / self.n if self.n else 0

  def expand(self, reward, logits, hidden, to_play, actions):           # mcts.py:47-55
    self.to_play, self.hidden = to_play, hidden
    if torch.is_tensor(reward) and reward:
      self.reward = reward.item()
    pol = {a: math.exp(logits[0][a].item()) for a in actions}
    tot = sum(pol.values())
    for a, p in pol.items():
      self.children[a] = _Node(p / tot)


class ShapedSearch(object):
  """MCTS.run for one tree (mcts.py:78-143), single player."""

  def __init__(self, actions, sims, discount=0.997, base=19652, init=1.25):
    self.actions, self.sims, self.g, self.base, self.init = range(actions), sims, discount, base, init

  def ucb(self, parent, child, lo, hi):                                 # mcts.py:115-124
    pb = math.log((parent.n + self.base + 1) / self.base) + self.init
    pb *= math.sqrt(parent.n) / (child.n + 1)
    score = pb * child.prior
    if child.n > 0:
      v = child.reward + self.g * child.value()
      if hi > lo:
        v = (v - lo) / (hi - lo)
      elif hi == lo:
        v = 1.0
      return score + v
    return score

  def run(self, root, net):
    lo, hi = float('inf'), -float('inf')
    for _ in range(self.sims):
      node, path = root, [root]
      while node.children:                                              # mcts.py:87-92, 104-113
        if node.n == 0:
          _, action, child = max((c.prior, a, c) for a, c in node.children.items())
        else:
          _, action, child = max((self.ucb(node, c, lo, hi), a, c) for a, c in node.children.items())
        node = child
        path.append(node)
      value, reward, logits, hidden = net.recurrent_inference(path[-2].hidden, [action])
      node.expand(reward, logits, hidden, 1, self.actions)
      v = value.item()
      for idx, nd in enumerate(reversed(path)):                         # mcts.py:126-143
        nd.w += v
        nd.n += 1
        if idx < len(path) - 1:
          q = nd.reward + self.g * nd.value()
          lo, hi = min(lo, q), max(hi, q)
        v = nd.reward + self.g * v
    return root


def play_moves(net, search, obs_dim, actions, moves, rng):
  """actors.py:131-153 per move, search only (no env, no replay): observation -> initial inference -> root expand +
  Dirichlet noise -> MCTS.run -> root error -> select_action at temperature 1."""
  done = 0
  with torch.inference_mode():
    for _ in range(moves):
      obs = torch.from_numpy(np.float32(rng.standard_normal(obs_dim))).unsqueeze(0)
      value, _, logits, hidden = net.initial_inference(obs)
      root = _Node(0)
      root.expand(0, logits, hidden, 1, range(actions))
      noise = np.random.dirichlet([0.25] * actions)                     # mcts.py:57-61
      for a, nz in zip(list(root.children), noise):
        root.children[a].prior = root.children[a].prior * 0.75 + nz * 0.25
      search.run(root, net)
      _ = root.value() - value.item()                                   # actors.py:147-148
      counts = np.array([c.n for c in root.children.values()])          # config.py:70-81
      dist = counts ** 1.0
      dist = dist / dist.sum()
      np.random.choice(actions, p=dist)
      done += 1
  return done


def measure(obs_dim, actions, sims, moves, seed=0):
  torch.set_num_threads(1)                                              # train.py:63: OMP_NUM_THREADS=1 per actor process
  torch.manual_seed(seed)
  np.random.seed(seed + 3)
  net = ShapedFC(obs_dim, actions).eval()
  search = ShapedSearch(actions, sims)
  rng = np.random.RandomState(seed)
  play_moves(net, search, obs_dim, actions, 3, rng)                     # warm-up
  t0 = time.perf_counter()
  n = play_moves(net, search, obs_dim, actions, moves, rng)
  dt = time.perf_counter() - t0
  return {'env_steps_per_s': n / dt, 'moves': n, 'seconds': dt, 'sims': sims, 'obs': obs_dim, 'actions': actions}


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--obs', type=int, default=8)
  ap.add_argument('--actions', type=int, default=4)
  ap.add_argument('--sims', type=int, default=30)
  ap.add_argument('--moves', type=int, default=200)
  ap.add_argument('--seed', type=int, default=0)
  a = ap.parse_args()
  print(json.dumps(measure(a.obs, a.actions, a.sims, a.moves, a.seed)))
  sys.stdout.flush()
