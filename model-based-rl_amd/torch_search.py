"""Batched search for ANY torch network: MCTS.run (reference mcts.py:78-102) for B trees in lock-step with the network
behind `initial_inference` / `recurrent_inference` -- the path of the reference's network-agnostic actor
(actors.py:45-47,139-145; utils.get_network, utils.py:21-37) for MuZeroNetwork / TinyNetwork (BASELINE.json configs[4]).

Per simulation, with no host synchronisation anywhere in the loop:
    mz_select            the descent of every tree (HIP tree kernel)            mcts.py:83-94
    index_select         parent hidden states out of a device-resident pool [B, sims+1, ...] keyed by parent_slot
    recurrent_inference  ONE batched call of the torch network (PyTorch-ROCm / MIOpen)   mcts.py:96
    mz_expand_backup     expand + backpropagate of every tree (HIP tree kernel)  mcts.py:97-99
    (mz_expand_backup_select: that and the next simulation's mz_select in one launch)
The tree arithmetic is the engine's (IEEE double, bit-exact against the CPU restatement of the reference given the same network outputs); the
network arithmetic is PyTorch's.  The engine's own FCNetwork kernels are not involved (its weight buffer stays unset).

`TorchSelfplay` is the move loop of Actor.play_game (actors.py:126-176) on top of it for B synthetic image
environments: observation -> initial inference -> root -> search -> select_action -> env step -> experience record in
the layout of include/mz_engine.h, handed to the native replay.
"""
import numpy as np
import torch

from .engine import Engine, REC_EXTRA


class BatchedSearch(object):

  def __init__(self, config, network, num_envs, device, seed=0, env_id_offset=0):
    self.network = network
    self.device = torch.device(device)
    self.B, self.A, self.sims = int(num_envs), int(config.action_space), int(config.num_simulations)
    # the engine only runs tree kernels here: obs_dim 1 keeps its (unused) FCNetwork buffers minimal
    self.engine = Engine(self.B, 1, self.A, self.sims, two_players=getattr(config, 'two_players', False),
                         known_bounds=tuple(getattr(config, 'known_bounds', (None, None))), discount=config.discount,
                         pb_c_base=config.pb_c_base, pb_c_init=config.pb_c_init,
                         init_value_score=getattr(config, 'init_value_score', 0.0),
                         root_dirichlet_alpha=config.root_dirichlet_alpha,
                         root_exploration_fraction=config.root_exploration_fraction, seed=seed,
                         env_id_offset=env_id_offset, device=self.device)
    self.pool = None                      # [B, sims+1, *hidden_shape] hidden states of expanded nodes (mcts.py:48)
    self._row0 = torch.arange(self.B, device=self.device, dtype=torch.int64) * (self.sims + 1)
    self.on_simulation = None             # test hook: called with (s, leaf, slot, action, depth, network_output)

  def close(self):
    self.engine.close()

  def _pool_for(self, hidden):
    shape = (self.B, self.sims + 1) + tuple(hidden.shape[1:])
    if self.pool is None or tuple(self.pool.shape) != shape or self.pool.dtype != hidden.dtype:
      self.pool = torch.empty(shape, dtype=hidden.dtype, device=self.device)
      self._flat = self.pool.view(self.B * (self.sims + 1), -1)
      self._hshape = tuple(hidden.shape[1:])
    return self.pool

  @torch.inference_mode()
  def run(self, observation, to_play=None, legal=None, noise=None, device_rng=True, move=0):
    """actors.py:139-145: initial inference, root.expand + add_exploration_noise, MCTS.run.  observation
    [B, ...] device tensor.  noise: the Dirichlet draw (parity runs) or None with device_rng (throughput runs) or
    None without (evaluation: no noise).  Returns the initial-inference output."""
    eng, net, B = self.engine, self.network, self.B
    init = net.initial_inference(observation)
    pool = self._pool_for(init.hidden_state)
    pool[:, 0].copy_(init.hidden_state)
    eng.root_load(init.value.reshape(B).float(), init.policy_logits.reshape(B, self.A).float())
    eng.root_prepare(to_play, legal, noise, device_rng=device_rng, move=move)
    sel = eng.select()                                                           # [B] int32 each, on the device
    for s in range(self.sims):
      leaf, slot, action, depth = sel
      parent = self._flat.index_select(0, self._row0 + slot.long()).view((B,) + self._hshape)
      out = net.recurrent_inference(parent, action)
      pool[:, s + 1].copy_(out.hidden_state)
      # expand + backup of this simulation and the descent of the next: one tree launch per simulation
      sel = eng.expand_backup_select(out.value.reshape(B).float(), out.reward.reshape(B).float(),
                                     out.policy_logits.reshape(B, self.A).float(), last=(s + 1 == self.sims))
      if self.on_simulation is not None:
        self.on_simulation(s, leaf, slot, action, depth, out)
    return init

  def finalize(self, temperature, uniform=None, move=0):
    """Config.select_action + Game.store_search_statistics + root error (config.py:70-81, game.py:106-115,
    actors.py:147-148) for all trees; device tensors."""
    return self.engine.finalize(temperature, uniform, move=move)


class SyntheticImageEnvs(object):
  """B shape-faithful stand-ins for the Atari image environments (gym / ALE are not installed on either box):
  observations uint8 [C, H, W] (what wrap_atari's frame stack emits, wrappers.py:422-444), all actions legal,
  reward ~ U(-1, 1), fixed-length episodes with staggered starts.  Lives on the device; stepping never touches the host."""

  def __init__(self, num_envs, obs_shape, episode_len, device, seed=0, env_id_offset=0, stagger=True):
    self.B, self.shape, self.T, self.device = int(num_envs), tuple(obs_shape), int(episode_len), torch.device(device)
    self.gen = torch.Generator(device=self.device)
    self.gen.manual_seed(int(seed) * 1000003 + int(env_id_offset))
    ids = torch.arange(env_id_offset, env_id_offset + self.B, dtype=torch.int64)
    t0 = ((ids * 2654435761) % (1 << 32) >> 8) % self.T if stagger else torch.zeros(self.B, dtype=torch.int64)
    self.t = t0.to(self.device, torch.int32)
    self.episode = torch.zeros(self.B, dtype=torch.int32, device=self.device)
    self.env_id = ids.to(self.device, torch.int32)
    self.obs = self._draw()

  def _draw(self):
    return torch.randint(0, 256, (self.B,) + self.shape, dtype=torch.uint8, device=self.device, generator=self.gen)

  def step(self):
    """-> (reward [B] f32, done [B] i32, step [B] i32 (pre-step), episode [B] i32); advances to the next observation."""
    reward = torch.rand(self.B, device=self.device, generator=self.gen) * 2 - 1
    step, episode = self.t.clone(), self.episode.clone()
    done = (self.t + 1 >= self.T)
    self.t = torch.where(done, torch.zeros_like(self.t), self.t + 1)
    self.episode = self.episode + done.to(torch.int32)
    self.obs = self._draw()
    return reward, done.to(torch.int32), step, episode


class TorchSelfplay(object):
  """Actor.play_game's move loop (actors.py:126-176) for B synthetic image envs and a torch network."""

  def __init__(self, config, network, num_envs, device, seed=0, env_id_offset=0, norm=None):
    self.config, self.device = config, torch.device(device)
    self.search = BatchedSearch(config, network, num_envs, device, seed=seed, env_id_offset=env_id_offset)
    self.B, self.A = self.search.B, self.search.A
    self.envs = SyntheticImageEnvs(num_envs, tuple(config.obs_space), int(config.episode_length), device, seed=seed,
                                   env_id_offset=env_id_offset)
    self.O = int(np.prod(config.obs_space))
    # the frames are bytes and stay bytes in the experience record (game.py:93-96 keeps the raw observation): O bytes in
    # ceil(O / 4) float slots -- a quarter of the D2H traffic and of the replay's memory (include/mz_replay.h obs_u8)
    self.obs_u8 = self.envs.obs.dtype == torch.uint8
    self.OS = (self.O + 3) // 4 if self.obs_u8 else self.O
    self.rec_floats = self.OS + self.A + REC_EXTRA
    self.norm = norm                   # (obs_min, obs_range) device tensors or None (actors.py:134-137)
    self.move = 0
    self.temperature = torch.full((self.B,), 1.0, dtype=torch.float64, device=self.device)
    self.next_temperature = 1.0

  def close(self):
    self.search.close()

  def set_temperature(self, temperature):
    """takes effect per env at its next episode start (actors.py:128-129)"""
    self.next_temperature = float(temperature)

  @torch.inference_mode()
  def play_move(self, out):
    """One move of every environment; the experience records go to out [B, rec_floats] (device float32 tensor)."""
    envs, B, O, A = self.envs, self.B, self.OS, self.A
    raw = envs.obs
    obs = raw.to(torch.float32)                                     # actors.py:134
    if self.norm is not None:
      obs = (obs - self.norm[0]) / self.norm[1]
    self.search.run(obs, move=self.move)
    fin = self.search.finalize(self.temperature, None, move=self.move)
    reward, done, step, episode = envs.step()
    if self.obs_u8:                                                 # History keeps the raw observation (game.py:93-96)
      out.view(torch.uint8)[:, :self.O].copy_(raw.reshape(B, self.O))
    else:
      out[:, :O].copy_(raw.reshape(B, self.O))
    out[:, O:O + A].copy_(fin['child_visits'])
    out[:, O + A:O + A + 4].copy_(torch.stack((fin['root_value'], fin['error']), 1).view(torch.float32))
    out[:, O + A + 4].copy_(reward)
    out[:, O + A + 5:].copy_(torch.stack((fin['action'], done, step, envs.env_id, episode), 1).view(torch.float32))
    self.temperature = torch.where(done.bool(), torch.full_like(self.temperature, self.next_temperature), self.temperature)
    self.move += 1
    return fin
