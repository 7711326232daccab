"""PyTorch definitions of the networks (training side + weight source).  Only FCNetwork's *definition*
lives here; its inference inside the search is the fused HIP kernel in csrc/mz_net.hip.h.  Parameter names
equal the reference's state_dict keys (networks.py:137-144) so checkpoints and the flat weight order
(engine.WEIGHT_ORDER) are interchangeable.  Surface: initial_inference / recurrent_inference ->
NetworkOutput(value, reward, policy_logits, hidden_state), load_weights / get_weights (networks.py:9-52).
"""
from collections import namedtuple

import torch
from torch import nn

NetworkOutput = namedtuple('network_output', ('value', 'reward', 'policy_logits', 'hidden_state'))

HIDDEN = 50
WIDTH = 512


class _TwoLayer(nn.Module):
  """Linear(in, 512) -> ReLU -> Linear(512, out); the second layer's attribute name varies per head."""

  def __init__(self, n_in, n_out, out_name):
    super().__init__()
    self.fc1 = nn.Linear(n_in, WIDTH)
    setattr(self, out_name, nn.Linear(WIDTH, n_out))
    self._out_name = out_name

  def forward(self, x):
    return getattr(self, self._out_name)(torch.relu(self.fc1(x.flatten(1))))


def support_to_scalar(logits, support_min, no_target_transform=False):
  """softmax expectation over the integer support + inverse of h(x)=sign(x)(sqrt(|x|+1)-1)+0.001x
  (reference config.py:27-33), float32."""
  p = torch.softmax(logits, dim=1)
  support = torch.arange(support_min, support_min + logits.shape[1], dtype=torch.float32, device=logits.device)
  x = (p * support).sum(1, keepdim=True)
  if no_target_transform:
    return x
  return torch.sign(x) * (((torch.sqrt(1 + 4 * 0.001 * (torch.abs(x) + 1 + 0.001)) - 1) / (2 * 0.001)) ** 2 - 1)


class FCNetwork(nn.Module):

  def __init__(self, input_dim, action_space, device, config):
    super().__init__()
    self.device = device
    self.action_space = int(action_space)
    self.no_support = bool(getattr(config, 'no_support', False))
    self.no_target_transform = bool(getattr(config, 'no_target_transform', False))
    vs = tuple(getattr(config, 'value_support', (-15, 15)))
    rs = tuple(getattr(config, 'reward_support', (-15, 15)))
    self.value_support_min, self.reward_support_min = vs[0], rs[0]
    v_out = 1 if self.no_support else vs[1] - vs[0] + 1
    r_out = 1 if self.no_support else rs[1] - rs[0] + 1
    self.representation_head = _TwoLayer(int(input_dim), HIDDEN, 'out')
    self.value_head = _TwoLayer(HIDDEN, v_out, 'value')
    self.policy_head = _TwoLayer(HIDDEN, self.action_space, 'policy')
    self.reward_head = _TwoLayer(HIDDEN + self.action_space, r_out, 'reward')
    self.transition_head = _TwoLayer(HIDDEN + self.action_space, HIDDEN, 'out')
    self.LN = nn.LayerNorm([HIDDEN], elementwise_affine=True)
    self.to(device)

  def representation(self, observation):
    return torch.relu(self.LN(self.representation_head(observation)))

  def prediction(self, hidden_state):
    value = self.value_head(hidden_state)
    if not self.training and not self.no_support:
      value = support_to_scalar(value, self.value_support_min, self.no_target_transform)
    return self.policy_head(hidden_state), value

  def dynamics(self, hidden_state, action):
    a = torch.as_tensor(action, dtype=torch.int64, device=hidden_state.device).reshape(-1)
    x = torch.cat((hidden_state, torch.nn.functional.one_hot(a, self.action_space).to(hidden_state.dtype)), dim=1)
    reward = self.reward_head(x)
    if not self.training and not self.no_support:
      reward = support_to_scalar(reward, self.reward_support_min, self.no_target_transform)
    return torch.relu(self.LN(self.transition_head(x))), reward

  def initial_inference(self, observation):
    hidden_state = self.representation(observation)
    policy_logits, value = self.prediction(hidden_state)
    return NetworkOutput(value, 0, policy_logits, hidden_state)

  def recurrent_inference(self, hidden_state, action):
    hidden_state, reward = self.dynamics(hidden_state, action)
    policy_logits, value = self.prediction(hidden_state)
    return NetworkOutput(value, reward, policy_logits, hidden_state)

  def load_weights(self, weights):
    self.load_state_dict(weights)

  def get_weights(self):
    return {k: v.cpu() for k, v in self.state_dict().items()}
