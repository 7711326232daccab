"""Environments that exist without gym.  TicTacToe follows the reference's custom environment
(custom_environments/tic_tac_toe.py:5-76: gym-0.x API, observation = turn * board, reward 1 for the
winning move, draw after nine moves).  Every other reference environment (Box2D, ALE) is unavailable on
both boxes; its SHAPE is served by the on-device synthetic env (csrc/mz_selfplay.hip.h)."""
from types import SimpleNamespace

import numpy as np

_LINES = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [0, 3, 6], [1, 4, 7], [2, 5, 8], [0, 4, 8], [2, 4, 6]])


class TicTacToe(object):

  def __init__(self):
    self.action_space = SimpleNamespace(n=9)
    self.observation_space = np.zeros(9, dtype=np.int32)
    self.reset()

  def seed(self, seed):
    return

  def reset(self):
    self.board = np.zeros(9, dtype=np.int32)
    self.turn = 1
    self._elapsed_steps = 0
    return self.board.copy()

  def legal_actions(self):
    return np.flatnonzero(self.board == 0)

  def step(self, action):
    self.board[action] = self.turn
    sums = self.board[_LINES].sum(axis=1)
    touched = (_LINES == action).any(axis=1)
    won = bool(np.any(np.abs(sums[touched]) == 3))
    done = won or self._elapsed_steps == 8
    result = None
    if won:
      result = 'player 1 wins' if self.turn == 1 else 'player 2 wins'
    elif done:
      result = 'draw'
    self._elapsed_steps += 1
    self.turn = -self.turn
    return self.turn * self.board.copy(), int(won), done, {'result': result}


def get_environment(config):
  if config.environment == 'TicTacToe':
    return TicTacToe()
  raise NotImplementedError('%s needs gym/ALE/Box2D, which are not installed; the GPU actor serves its shape with '
                            'the synthetic on-device environment' % config.environment)
