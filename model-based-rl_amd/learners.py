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
from .logger import Logger
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


class MuZeroLR(object):
  """lr_init * decay_rate ** (step / decay_steps) (utils.py:85-99)"""

  def __init__(self, optimizer, config):
    self.optimizer, self.lr_init = optimizer, config.lr_init
    self.rate, self.steps, self.lr_step, self.lr = config.lr_decay_rate, config.lr_decay_steps, 0, config.lr_init

  def step(self):
    self.lr_step += 1
    self.lr = self.lr_init * self.rate ** (self.lr_step / self.steps)
    for g in self.optimizer.param_groups:
      g['lr'] = self.lr


class WarmUpLR(object):
  """linear ramp to lr_init over 5000 steps, then constant (utils.py:102-118)"""

  def __init__(self, optimizer, config, warm_up_steps=5000):
    self.optimizer, self.max_lr, self.n, self.lr_step = optimizer, config.lr_init, warm_up_steps, 0
    self._set(1 / self.n * self.max_lr)

  def _set(self, lr):
    self.lr = lr
    for g in self.optimizer.param_groups:
      g['lr'] = lr

  def step(self):
    self.lr_step += 1
    if self.lr_step <= self.n:
      self._set(self.lr_step / self.n * self.max_lr)


def make_lr_scheduler(config, optimizer):
  """utils.get_lr_scheduler (utils.py:121-132)"""
  name = getattr(config, 'lr_scheduler', None)
  if name is None:
    return None
  if name == 'ExponentialLR':
    sched = torch.optim.lr_scheduler.ExponentialLR(optimizer, config.lr_decay_rate)
    sched.lr = config.lr_init
    return sched
  if name == 'MuZeroLR':
    return MuZeroLR(optimizer, config)
  if name == 'WarmUpLR':
    return WarmUpLR(optimizer, config)
  raise NotImplementedError(name)


class Learner(Logger):

  def __init__(self, config, storage, replay_buffer, state=None):
    set_all_seeds(config.seed)
    self.config = deepcopy(config)
    self.run_tag, self.group_tag = getattr(config, 'run_tag', None) or 'run', getattr(config, 'group_tag', None)
    self.worker_id = 'learner'
    self.storage, self.replay_buffer = storage, replay_buffer
    if 'learner' in getattr(config, 'use_gpu_for', []):          # learners.py:27-37
      if not torch.cuda.is_available():
        raise RuntimeError('GPU was requested but torch.cuda.is_available() is False.')
      dev_id = getattr(config, 'learner_gpu_device_id', None)
      self.device = torch.device('cuda', dev_id if dev_id is not None else torch.cuda.current_device())
    else:
      self.device = torch.device('cpu')
    self.network = get_network(config, self.device)          # utils.get_network (utils.py:21-37)
    self.network.train()
    self.optimizer = make_optimizer(config, self.network.parameters())
    self.lr_scheduler = make_lr_scheduler(config, self.optimizer)
    if getattr(config, 'scalar_loss', 'MSE') not in ('MSE', 'Huber'):
      raise NotImplementedError(config.scalar_loss)
    self.training_step = 0
    self.losses_to_log = {'reward': 0., 'value': 0., 'policy': 0.}
    self.throughput = {'total_frames': 0, 'total_games': 0, 'training_step': 0, 'time': {'ups': 0, 'fps': 0}}
    self.last_throughput = {}
    if getattr(config, 'norm_obs', False):
      self.obs_min = np.array(config.obs_range[::2], dtype=np.float32)
      self.obs_range = np.array(config.obs_range[1::2], dtype=np.float32) - self.obs_min
    if state is not None:
      self.load_state(state)
    Logger.__init__(self)
    self.saves_dir = self.dirs['saves']

  # learners.py:62-70
  def load_state(self, state):
    self.run_tag = os.path.join(str(self.run_tag), 'resumed', '{}'.format(state['training_step']))
    self.network.load_state_dict(state['weights'])
    self.optimizer.load_state_dict(state['optimizer'])
    _call(self.replay_buffer, 'add_initial_throughput', state['total_frames'], state['total_games'])
    self.throughput['total_frames'] = state['total_frames']
    self.throughput['training_step'] = state['training_step']
    self.training_step = state['training_step']

  # learners.py:72-83 (same dictionary keys)
  def save_state(self, path=None):
    thr = _call(self.replay_buffer, 'get_throughput')        # (the reference refreshes these in log_throughput)
    self.throughput['total_games'] = thr['games']
    self.throughput['total_frames'] = max(self.throughput['total_frames'], thr['frames'])
    state = {'dirs': self.dirs, 'config': self.config, 'weights': self.network.get_weights(),
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
    if not no_support:
      scalar_loss = soft_cross_entropy
    elif getattr(cfg, 'scalar_loss', 'MSE') == 'Huber':        # utils.py:62-70
      scalar_loss = torch.nn.SmoothL1Loss(reduction='none')
    else:
      scalar_loss = torch.nn.MSELoss(reduction='none')
    reward_loss = 0
    value_loss = scalar_loss(value.squeeze(), t_val[:, 0])
    policy_loss = soft_cross_entropy(policy_logits.squeeze(), t_pol[:, 0])
    # the K action columns go to the device ONCE (the reference hands recurrent_inference a Python tuple per unroll step,
    # learners.py:196-197: one pageable host-to-device copy per step, each of which waits behind whatever else the GPU
    # runs -- 3 ms per copy beside a self-play loop on the same GPU)
    act = torch.as_tensor(np.asarray(actions, np.int64), device=dev)
    for i in range(1, act.shape[1] + 1):
      action = act[:, i - 1]
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
    if self.lr_scheduler is not None:             # learners.py:225-226
      self.lr_scheduler.step()
    self.losses_to_log['reward'] += reward_loss.item()
    self.losses_to_log['value'] += value_loss.item()
    self.losses_to_log['policy'] += policy_loss.item()

  # learners.py:88-113: the reference's own throughput scalars -- frames_per_second is its env-steps/sec metric
  def log_throughput(self, force=False):
    data = _call(self.replay_buffer, 'get_throughput')
    self.throughput['total_games'] = data['games']
    self.log_scalar(tag='games/finished', value=data['games'], i=self.training_step)
    new_frames = data['frames'] - self.throughput['total_frames']
    if new_frames > getattr(self.config, 'frames_before_fps_log', 10000) or (force and new_frames > 0):
      now = time.time()
      new_updates = self.training_step - self.throughput['training_step']
      ups = new_updates / max(1e-9, now - self.throughput['time']['ups'])
      fps = new_frames / max(1e-9, now - self.throughput['time']['fps'])
      replay_ratio = ups / fps
      self.throughput['total_frames'] = data['frames']
      self.throughput['training_step'] = self.training_step
      self.throughput['time']['ups'] = self.throughput['time']['fps'] = now
      self.last_throughput = {'frames_per_second': fps, 'updates_per_second': ups, 'replay_ratio': replay_ratio,
                              'sample_ratio': self.config.batch_size * replay_ratio, 'total_frames': data['frames']}
      for k, v in self.last_throughput.items():
        self.log_scalar(tag='throughput/' + k, value=v, i=self.training_step)

  # learners.py:115-153
  def learn(self, max_steps=None):
    cfg = self.config
    self.send_weights()
    self.throughput['time']['fps'] = time.time()
    while _call(self.replay_buffer, 'size') < cfg.stored_before_train:
      time.sleep(0.05)
    self.throughput['time']['ups'] = time.time()
    last = cfg.training_steps if max_steps is None else min(cfg.training_steps, self.training_step + max_steps)
    log_every = max(1, getattr(cfg, 'learner_log_frequency', 100))
    from . import gpu_turns
    if self.device.type == 'cuda':
      gpu_turns.register(self.device, 'learner')
    while self.training_step < last:
      batch = _call(self.replay_buffer, 'sample_batch')
      turn = gpu_turns.turn(self.device)
      with turn:         # (an actor on the same GPU: one update per turn, see gpu_turns.py)
        self.update_weights(batch)
        if turn is not gpu_turns.NO_TURNS:
          torch.cuda.current_stream(self.device).synchronize()
      self.training_step += 1
      if self.training_step % cfg.send_weights_frequency == 0:
        self.send_weights()
      if self.training_step % getattr(cfg, 'save_state_frequency', 1000) == 0:
        self.save_state()
      if self.training_step % log_every == 0:
        for k in ('reward', 'value', 'policy'):
          self.log_scalar(tag='loss/' + k, value=self.losses_to_log[k] / log_every, i=self.training_step)
          self.losses_to_log[k] = 0
        self.log_throughput()
        if self.lr_scheduler is not None:
          self.log_scalar(tag='loss/learning_rate', value=self.optimizer.param_groups[0]['lr'], i=self.training_step)      # (what the optimizer really uses, every scheduler)
    self.log_throughput(force=True)
    self.send_weights()

  def get_last_throughput(self):
    return dict(self.last_throughput)

  def launch(self, max_steps=None):
    print('Learner is online on {}.'.format(self.device))
    self.learn(max_steps)
