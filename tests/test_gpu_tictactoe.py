"""Two-player games in the device loop (SURVEY.md s8d Config 1): TicTacToe with the reference's rules
(custom_environments/tic_tac_toe.py:5-76) as the environment of mz_selfplay_steps -- observation, legal moves, wins,
draws, alternating to_play and the experience record all on the device.

test_device_tictactoe_reproduces_reference_games: a one-environment device loop fed the goldens' Dirichlet draws and
select_action uniforms (numpy's stream as the reference consumed it) replays the reference's recorded self-play games
(g3, three games each): same observations, movers, actions, visit distributions, rewards and game ends, and the replay the
records are ingested into ends up with the reference's leaves."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
LINES = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [0, 3, 6], [1, 4, 7], [2, 5, 8], [0, 4, 8], [2, 4, 6]])


def ttt_engine(B, seed=0, split=False):
  from model_based_rl_amd.engine import Engine
  eng = Engine(B, 9, 9, 30, two_players=True, known_bounds=(-1.0, 1.0), discount=1.0, seed=seed, split_f16=split)
  return eng


@pytest.mark.parametrize('gi', [0, 1, 2, 3])
def test_device_tictactoe_reproduces_reference_games(gi):
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import records_view
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  g = np.load(os.path.join(G, 'g3_game_ttt_%d.npz' % gi))
  M = g['action'].shape[0]
  eng = ttt_engine(1)
  eng.set_weights(orc.load_weights(g))
  eng.selfplay_set_env('tictactoe')
  eng.selfplay_reset(9, float(g['temperature'][0]))
  rep = PrioritizedReplay(types.SimpleNamespace(batch_size=16, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(9,), action_space=9,
                                                window_size=int(g['max_capacity']), window_step=None, num_unroll_steps=5, td_steps=10,
                                                max_history_length=int(g['max_history_length']), discount=1.0, seed=0, two_players=True))
  same = 0
  for m in range(M):
    u = float(g['uniform'][m])
    if g['temperature'][m] == 0:        # config.py:79: numpy's own tie draw -- steer the device's floor(u * n_ties) to the same child
      vc = g['child_visits'][m]
      ties = np.flatnonzero(vc == vc.max())
      u = (int(np.flatnonzero(ties == g['action'][m])[0]) + 0.5) / len(ties)
    eng.selfplay_set_draws(g['noise'][m][None], np.array([u]))
    eng.selfplay_steps(1)
    buf, n = eng.selfplay_drain()
    torch.cuda.synchronize()
    assert n == 1
    rec = buf[:1].numpy().copy()
    rv = records_view(rec, 9, 9)
    # the position the device played from is the reference's: observation = turn * board, the mover
    assert np.array_equal(rv['obs'][0, 0], g['obs'][m]) and int(rv['to_play'][0, 0]) == int(g['to_play'][m]), m
    eq = np.array_equal(rv['child_visits'][0, 0], g['child_visits'][m].astype(np.float32))
    if g['min_margin'][m] > 1e-4:       # the margin rule of tests/test_gpu_search.py
      assert eq, (m, g['min_margin'][m])
    assert eq or g['min_margin'][m] < 1e-5, (m, g['min_margin'][m])
    if not eq:
      break                              # (a flipped near-tie: the game leaves the recorded one; measured: never)
    assert int(rv['action'][0, 0]) == int(g['action'][m]), m
    assert abs(rv['root_value'][0, 0] - g['final_root_value'][m]) <= 5e-4 and abs(rv['error'][0, 0] - g['error'][m]) <= 5e-4
    last = m + 1 == M or g['game'][m + 1] != g['game'][m]
    assert int(rv['done'][0, 0]) == int(last), m
    rep.ingest_records(rec, 1, 1)
    same += 1
  eng.close()
  print('g3_game_ttt_%d on the device: %d of %d moves identical' % (gi, same, M))
  assert same == M
  # rewards / dones / movers as the reference's histories have them, and the replay built from the device's records
  k = 0
  for f in range(int(g['n_flushes'])):
    n = len(g['flush%d_actions' % f])
    k += n
  assert rep.get_throughput() == {'frames': int(g['flush_meta'][-1][6]), 'games': int(g['flush_meta'][-1][7])}
  bound = float(np.sum(2 * 1.5e-4 * (1 + np.abs(g['final_root_value']))))       # (the float32 staircase of the two values in every error)
  assert abs(rep.tree.total_priority - float(g['replay_total'])) <= bound


def test_device_tictactoe_rules_at_4096_envs():
  """4096 games at once with the device RNG: every record obeys the rules -- the mover alternates from +1 inside a game,
  the action lands on an empty cell, the observation is turn * board replayed from the actions, reward 1 exactly for a
  move that completes a line, done on a win or on the ninth move, the step counter restarts with every game."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import records_view
  g = np.load(os.path.join(G, 'g1_net_ttt.npz'))
  B, moves = 4096, 24
  eng = ttt_engine(B, seed=5)
  eng.set_weights(orc.load_weights(g))
  eng.selfplay_set_env('tictactoe')
  eng.selfplay_reset(9, 1.0)
  eng.selfplay_steps(16); eng.selfplay_steps(8)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  assert n == moves
  rv = records_view(buf[:n].numpy().copy(), 9, 9)
  eng.close()
  board = np.zeros((B, 9), np.int64); turn = np.ones(B, np.int64); t = np.zeros(B, np.int64)
  ar = np.arange(B)
  wins = draws = 0
  for m in range(moves):
    assert np.array_equal(rv['obs'][m], (turn[:, None] * board).astype(np.float32)), m
    assert np.array_equal(rv['to_play'][m], turn) and np.array_equal(rv['step'][m], t)
    a = rv['action'][m]
    assert np.all(board[ar, a] == 0)                                         # a legal move
    assert np.all(rv['child_visits'][m][board != 0] == 0) and np.allclose(rv['child_visits'][m].sum(1), 1, atol=1e-6)
    board[ar, a] = turn
    won = np.any(np.abs(board[:, LINES].sum(-1)) == 3, axis=1)
    done = won | (t == 8)
    assert np.array_equal(rv['reward'][m], won.astype(np.float32)) and np.array_equal(rv['done'][m], done.astype(np.int32))
    wins += int(won.sum()); draws += int((done & ~won).sum())
    board[done] = 0
    turn = np.where(done, 1, -turn); t = np.where(done, 0, t + 1)
  assert wins > B and draws > 0                  # ~2.7 games per environment, both endings occur
  assert np.array_equal(rv['episode'][-1], np.cumsum(rv['done'], 0)[-1] - rv['done'][-1])


def test_actor_plays_tictactoe_on_the_device_and_feeds_the_replay():
  """Actor with --environment TicTacToe and a pool of environments: the games run on the device, the records (with the
  mover) reach the two-player replay, games are counted, and a sampled batch carries sign-flipped n-step targets."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.actors import Actor
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  from model_based_rl_amd.shared_storage import SharedStorage
  g = np.load(os.path.join(G, 'g1_net_ttt.npz'))
  cfg = make_config(['--environment', 'TicTacToe', '--two_players', '--known_bounds', '-1', '1', '--discount', '1',
                     '--num_simulations', '30', '--seed', '4', '--num_envs', '256', '--window_size', '16384',
                     '--weight_sync_frequency', '8', '--batch_size', '64', '--td_steps', '3'])
  storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
  storage.store_weights({k: torch.from_numpy(v) for k, v in orc.load_weights(g).items()}, 0)
  actor = Actor(0, cfg, storage, replay)
  assert not actor.host_env
  actor.launch(max_moves=32)
  thr = replay.get_throughput()
  assert actor.games_played == thr['games'] and thr['games'] >= 256 * 3 and thr['frames'] >= 256 * 20
  (obs, actions, (t_rew, t_val, t_pol)), idxs, isw = replay.sample_batch()
  assert obs.shape == (64, 9) and set(np.unique(obs)) <= {-1.0, 0.0, 1.0}
  assert set(np.unique(t_rew)) <= {0.0, 1.0} and (t_rew == 1).any()
  assert (t_val < 0).any() and (t_val > 0).any()          # a win two plies ahead is a loss for the player to move (replay_buffer.py:187-189)
  actor.engine.close()
