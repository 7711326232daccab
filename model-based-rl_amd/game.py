"""Episode bookkeeping on the host (the reference's game.py:5-126): what Actor.play_game appends per move and
the history slices it hands to the replay buffer.  Used by the host-environment path of the actor (real
envs such as TicTacToe); the synthetic-env path keeps the same record on the device."""
from typing import NamedTuple

import numpy as np


class HistorySlice(NamedTuple):
  observations: list
  child_visits: list
  root_values: list
  actions: list
  rewards: list
  errors: list
  dones: list
  steps: list
  env_states: list
  to_play: list


_FIELDS = HistorySlice._fields


class History(object):

  def __init__(self):
    for f in _FIELDS:
      setattr(self, f, [])

  def get_slice(self, collect_from):
    return HistorySlice(*[getattr(self, f)[collect_from:] for f in _FIELDS])


class Game(object):

  def __init__(self, environment, config):
    self.environment = environment
    self.episode_life = getattr(config, 'episode_life', False)
    self.clip_rewards = getattr(config, 'clip_rewards', False)
    self.two_players = config.two_players
    self.action_space = range(config.action_space)
    self.history = History()
    self.terminal = self.done = False
    self.previous_collect_to = 0
    self.history_idx = 0
    self.sum_rewards = 0
    self.sum_values = 0
    self.max_value = -np.inf
    self.step = 0
    self.to_play = 1
    self.info = {}

  def get_observation(self, index):
    if not self.history.observations:
      self.history.observations.append(self.environment.reset())
    return self.history.observations[index]

  # game.py:79-104
  def apply(self, action):
    h, env = self.history, self.environment
    h.steps.append(self.step)
    obs, reward, done, info = env.step(action)
    self.sum_rewards += env.last_reward if self.clip_rewards else reward
    self.step = env._elapsed_steps
    self.history_idx += 1
    self.terminal = env.was_real_done if self.episode_life else done
    self.done = done
    if done:
      obs = env.reset()
    h.observations.append(obs)
    h.actions.append(action)
    h.dones.append(done)
    h.rewards.append(reward)
    h.to_play.append(self.to_play)
    self.info = info
    if self.two_players:
      self.to_play = -self.to_play

  # game.py:106-115.  Reference call style: store_search_statistics(root) with a searched Node; the batched actor passes
  # what the engine's finalize step already computed: store_search_statistics(child_visits, root_value).
  def store_search_statistics(self, root, root_value=None):
    if root_value is None and hasattr(root, 'children'):
      total = sum(child.visit_count for child in root.children.values())
      child_visits = [root.children[a].visit_count / total if a in root.children else 0 for a in self.action_space]
      root_value = root.value()
    else:
      child_visits = root
    self.history.child_visits.append(list(child_visits))
    self.history.root_values.append(root_value)
    self.sum_values += root_value
    self.max_value = max(self.max_value, root_value)

  def get_history_sequence(self, collect_from):
    out = self.history.get_slice(collect_from)
    self.previous_collect_to = self.history_idx
    return out
