"""Learner with the reference's surface (learners.py:14-235): K-step unrolled training step on batches from the
prioritized replay, priority refresh, weight publication, checkpoints.  The training step is stock PyTorch
(on ROCm when `--use_gpu_for learner`); only its semantics follow the reference: initial inference + K
recurrent steps, 0.5 gradient scale on the hidden state per step (learners.py:200), importance-sampling
weighted cross-entropy losses on the categorical supports, 1/K gradient scale on the total loss (214),
AdamW with eps 1.5e-4 (utils.py:85-97)."""
import os
import time
from copy import deepcopy

import numpy as np
import torch

from .actors import _call, set_all_seeds
from .networks import get_network, support_to_scalar


def scalar_transform(x):
  """h(x) = sign(x)(sqrt(|x|+1)-1) + 0.001x  (config.py:51-54)."""
  return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + 0.001 * x


def scalar_to_support(x, lo, hi):
  """two-hot projection of scalars onto the integer support [lo, hi] (config.py:56-68)."""
  x = x.clamp(lo, hi)
  low, high = x.floor(), x.ceil()
  p_high = x - low
  out = torch.zeros(*x.shape, hi - lo + 1, device=x.device)
  out.scatter_(2, (high - lo).long().unsqueeze(-1), p_high.unsqueeze(-1))
  out.scatter_(2, (low - lo).long().unsqueeze(-1), (1 - p_high).unsqueeze(-1))
  return out


def soft_cross_entropy(logits, target):
  return (-target * torch.log_softmax(logits, dim=1)).sum(1)


def make_optimizer(config, params):
  name = getattr(config, 'optimizer', 'AdamW')
  lr, wd = config.lr_init, getattr(config, 'weight_decay', 1e-4)
  if name == 'AdamW':
    return torch.optim.AdamW(params, lr=lr, weight_decay=wd, eps=0.00015)
  if name == 'Adam':
    return torch.optim.Adam(params, lr=lr, weight_decay=wd, eps=0.00015)
  if name == 'RMSprop':
    return torch.optim.RMSprop(params, lr=lr, momentum=getattr(config, 'momentum', 0.9), eps=0.01, weight_decay=wd)
  if name == 'SGD':
    return torch.optim.SGD(params, lr=lr, momentum=getattr(config, 'momentum', 0.9), weight_decay=wd)
  raise NotImplementedError(name)


class Learner(object):

  def __init__(self, config, storage, replay_buffer, state=None):
    set_all_seeds(config.seed)
    self.config = deepcopy(config)
    self.storage, self.replay_buffer = storage, replay_buffer
    use_gpu = 'learner' in getattr(config, 'use_gpu_for', []) and torch.cuda.is_available()
    self.device = torch.device('cuda' if use_gpu else 'cpu')
    self.network = get_network(config, self.device)          # utils.get_network (utils.py:21-37)
    self.network.train()
    self.optimizer = make_optimizer(config, self.network.parameters())
    self.training_step = 0
    self.losses_to_log = {'reward': 0., 'value': 0., 'policy': 0.}
    self.throughput = {'total_frames': 0, 'total_games': 0}
    if getattr(config, 'norm_obs', False):
      self.obs_min = np.array(config.obs_range[::2], dtype=np.float32)
      self.obs_range = np.array(config.obs_range[1::2], dtype=np.float32) - self.obs_min
    self.saves_dir = os.path.join('runs', str(config.environment), str(config.group_tag), str(config.run_tag), 'saves')
    if state is not None:
      self.load_state(state)

  # learners.py:62-70
  def load_state(self, state):
    self.network.load_state_dict(state['weights'])
    self.optimizer.load_state_dict(state['optimizer'])
    _call(self.replay_buffer, 'add_initial_throughput', state['total_frames'], state['total_games'])
    self.throughput['total_frames'] = state['total_frames']
    self.training_step = state['training_step']

  # learners.py:72-83 (same dictionary keys)
  def save_state(self, path=None):
    state = {'dirs': {'saves': self.saves_dir}, 'config': self.config, 'weights': self.network.get_weights(),
             'optimizer': self.optimizer.state_dict(), 'training_step': self.training_step,
             'total_games': self.throughput['total_games'], 'total_frames': self.throughput['total_frames'],
             'actor_games': _call(self.storage, 'get_stats', 'actor_games')}
    path = path or os.path.join(self.saves_dir, str(self.training_step))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(state, path)
    return path

  # learners.py:85-86
  def send_weights(self):
    _call(self.storage, 'store_weights', self.network.get_weights(), self.training_step)

  # learners.py:164-230
  def update_weights(self, batch):
    (observations, actions, (target_rewards, target_values, target_policies)), idxs, is_weights = batch
    cfg, dev = self.config, self.device
    if getattr(cfg, 'norm_obs', False):
      observations = (observations - self.obs_min) / self.obs_range
    obs = torch.from_numpy(np.ascontiguousarray(observations)).to(dev)
    value, _, policy_logits, hidden = self.network.initial_inference(obs)
    with torch.no_grad():
      t_pol = torch.from_numpy(target_policies).to(dev)
      t_val = torch.from_numpy(target_values).to(dev)
      t_rew = torch.from_numpy(target_rewards).to(dev)
      w = torch.from_numpy(np.asarray(is_weights)).to(dev)
      no_support = getattr(cfg, 'no_support', False)
      init_value = value if no_support else support_to_scalar(value, cfg.value_support_min, cfg.no_target_transform)
      new_errors = (init_value.squeeze() - t_val[:, 0]).cpu().numpy()
      _call(self.replay_buffer, 'update', idxs, new_errors)
      if not cfg.no_target_transform:
        t_val, t_rew = scalar_transform(t_val), scalar_transform(t_rew)
      if not no_support:
        t_val = scalar_to_support(t_val, cfg.value_support_min, cfg.value_support_max)
        t_rew = scalar_to_support(t_rew, cfg.reward_support_min, cfg.reward_support_max)
    scalar_loss = soft_cross_entropy if not no_support else (lambda a, b: (a - b) ** 2)
    reward_loss = 0
    value_loss = scalar_loss(value.squeeze(), t_val[:, 0])
    policy_loss = soft_cross_entropy(policy_logits.squeeze(), t_pol[:, 0])
    for i, action in enumerate(zip(*actions), 1):
      value, reward, policy_logits, hidden = self.network.recurrent_inference(hidden, action)
      hidden.register_hook(lambda grad: grad * 0.5)
      reward_loss = reward_loss + scalar_loss(reward.squeeze(), t_rew[:, i])
      value_loss = value_loss + scalar_loss(value.squeeze(), t_val[:, i])
      policy_loss = policy_loss + soft_cross_entropy(policy_logits.squeeze(), t_pol[:, i])
    reward_loss, value_loss, policy_loss = (w * reward_loss).mean(), (w * value_loss).mean(), (w * policy_loss).mean()
    total = reward_loss + value_loss + policy_loss
    total.register_hook(lambda grad: grad * (1 / cfg.num_unroll_steps))
    self.optimizer.zero_grad()
    total.backward()
    if getattr(cfg, 'clip_grad', 0):
      torch.nn.utils.clip_grad_norm_(self.network.parameters(), cfg.clip_grad)
    self.optimizer.step()
    self.losses_to_log['reward'] += reward_loss.item()
    self.losses_to_log['value'] += value_loss.item()
    self.losses_to_log['policy'] += policy_loss.item()

  # learners.py:115-136 (logging left out)
  def learn(self, max_steps=None):
    cfg = self.config
    self.send_weights()
    while _call(self.replay_buffer, 'size') < cfg.stored_before_train:
      time.sleep(0.05)
    last = cfg.training_steps if max_steps is None else min(cfg.training_steps, self.training_step + max_steps)
    while self.training_step < last:
      self.update_weights(_call(self.replay_buffer, 'sample_batch'))
      self.training_step += 1
      if self.training_step % cfg.send_weights_frequency == 0:
        self.send_weights()
      if self.training_step % getattr(cfg, 'save_state_frequency', 1000) == 0:
        self.save_state()
    self.send_weights()

  def launch(self, max_steps=None):
    print('Learner is online on {}.'.format(self.device))
    self.learn(max_steps)
