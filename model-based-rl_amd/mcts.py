"""The reference's search surface -- MinMaxStats, Node, MCTS(config).run(root, network) -> search_paths
(mcts.py:6-143) -- as a front-end of the device engine.

`MCTS.run` keeps the reference's calling convention (a caller-built root Node that was already expanded and
noised, any network object with `recurrent_inference(hidden_state, [action])`), runs every tree operation
(select_child / ucb_score / expand / backpropagate / MinMaxStats) in the HIP tree kernels through the
external-inference entry points of the C ABI, and materialises the result back into Node objects so that
`config.select_action(root, T)`, `root.value()`, `root.children[a].visit_count` and the tree walkers of
evaluate.py / visualize_mcts.py see what they expect.  The batched path used by the GPU actor
(engine.Engine.search) never builds Node objects.
"""
import math

import numpy as np
import torch

from .engine import Engine


class MinMaxStats(object):
  """mcts.py:6-25 (host mirror; during a search the device keeps the pair per tree)."""

  def __init__(self, minimum_bound=None, maximum_bound=None):
    self.reset(minimum_bound, maximum_bound)

  def reset(self, minimum_bound=None, maximum_bound=None):
    self.minimum = float('inf') if minimum_bound is None else minimum_bound
    self.maximum = -float('inf') if maximum_bound is None else maximum_bound

  def update(self, value):
    self.minimum, self.maximum = min(self.minimum, value), max(self.maximum, value)

  def normalize(self, value):
    if self.maximum > self.minimum:
      return (value - self.minimum) / (self.maximum - self.minimum)
    return 1.0 if self.maximum == self.minimum else value


class Node(object):
  """mcts.py:28-61."""

  def __init__(self, prior):
    self.hidden_state = None
    self.visit_count = 0
    self.value_sum = 0
    self.reward = 0
    self.children = {}
    self.prior = prior
    self.to_play = 1

  def expanded(self):
    return len(self.children) > 0

  def value(self):
    return self.value_sum / self.visit_count if self.visit_count else 0

  def expand(self, network_output, to_play, actions):
    self.to_play = to_play
    self.hidden_state = network_output.hidden_state
    if torch.is_tensor(network_output.reward):
      self.reward = network_output.reward.item()
    logits = network_output.policy_logits.detach().cpu().numpy().reshape(-1)
    p = {int(a): math.exp(float(logits[a])) for a in actions}
    total = sum(p.values())
    for a, v in p.items():
      self.children[a] = Node(v / total)

  def add_exploration_noise(self, dirichlet_alpha, frac):
    actions = list(self.children.keys())
    noise = np.random.dirichlet([dirichlet_alpha] * len(actions))
    for a, n in zip(actions, noise):
      self.children[a].prior = self.children[a].prior * (1 - frac) + n * frac


class MCTS(object):

  def __init__(self, config, device=None):
    self.config = config
    self.num_simulations = config.num_simulations
    self.action_space = range(config.action_space)
    self.two_players = config.two_players
    self.known_bounds = config.known_bounds
    self.min_max_stats = MinMaxStats(*config.known_bounds)
    self._device = device
    self._engine = None

  def _eng(self):
    if self._engine is None:
      c = self.config
      self._engine = Engine(1, 1, c.action_space, c.num_simulations, two_players=c.two_players,
                            known_bounds=tuple(c.known_bounds), discount=c.discount, pb_c_base=c.pb_c_base,
                            pb_c_init=c.pb_c_init, init_value_score=getattr(c, 'init_value_score', 0.0),
                            device=self._device)
    return self._engine

  def run(self, root, network):
    eng, A = self._eng(), self.config.action_space
    self.min_max_stats.reset(*self.known_bounds)
    priors = np.zeros((1, A)); legal = np.zeros((1, A), np.uint8)
    for a, child in root.children.items():
      priors[0, a], legal[0, a] = child.prior, 1
    eng.root_set_priors(priors, to_play=np.array([root.to_play], np.int8), legal=legal)
    hidden = {0: root.hidden_state}       # expansion slot -> network hidden state (any shape)
    paths = []
    for s in range(self.num_simulations):
      leaf, slot, act, depth = [int(x.item()) for x in eng.select()]
      p, n = eng.last_paths()
      paths.append(p[0, :int(n.item())].cpu().numpy().copy())
      out = network.recurrent_inference(hidden[slot], [act])
      hidden[s + 1] = out.hidden_state
      eng.expand_backup(out.value.reshape(1).float(), out.reward.reshape(1).float(),
                        out.policy_logits.reshape(1, A).float())
    ex = eng.export_tree()
    self.min_max_stats.minimum, self.min_max_stats.maximum = float(ex['minmax'][0, 0]), float(ex['minmax'][0, 1])
    nodes = {0: root}

    def fill(node, idx):
      node.visit_count = int(ex['N'][0, idx]); node.value_sum = float(ex['W'][0, idx])
      node.reward = float(ex['R'][0, idx]); node.to_play = int(ex['TP'][0, idx])
      e = int(ex['E'][0, idx])
      if e >= 0:
        node.hidden_state = hidden[e]
        for a in (root.children.keys() if idx == 0 else self.action_space):
          ci = 1 + e * A + int(a)
          child = node.children.get(a) if idx == 0 else None
          if child is None:
            child = Node(float(ex['P'][0, ci]))
            node.children[int(a)] = child
          child.prior = float(ex['P'][0, ci])
          nodes[ci] = child
          fill(child, ci)

    fill(root, 0)
    return [[nodes[int(i)] for i in path] for path in paths]
