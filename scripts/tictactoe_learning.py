"""End-to-end check that the stack LEARNS: MuZero on TicTacToe with the reference's recipe (README: --two_players
--td_steps 10 --discount 1 --known_bounds -1 1, FCNetwork, 30 simulations) -- the games played by the device environment
(whole moves inside the two-player launch), the native replay with sign-flipped n-step targets, the stock-PyTorch learner
on the same GPU -- then the trained network's MCTS agent (temperature 0, no exploration noise) against a uniformly random
opponent on the host's TicTacToe rules, 512 games as each side, next to the untrained network.

  python scripts/tictactoe_learning.py [--training_steps 3000] [--num_envs 1024] [--out profiles/r03_tictactoe_learning.json]"""
import argparse, json, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LINES = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [0, 3, 6], [1, 4, 7], [2, 5, 8], [0, 4, 8], [2, 4, 6]])


def play_vs_random(weights, agent_side, games=512, sims=30, seed=0):
  """agent (MCTS, T = 0, no noise) vs a uniformly random opponent, `games` boards in lock-step through the stepwise ABI;
  returns (wins, draws, losses) of the agent.  agent_side: +1 moves first, -1 second."""
  from model_based_rl_amd.engine import Engine
  rng = np.random.RandomState(seed)
  eng = Engine(games, 9, 9, sims, two_players=True, known_bounds=(-1.0, 1.0), discount=1.0, seed=seed)
  eng.set_weights(weights)
  board = np.zeros((games, 9), np.int64); turn = np.ones(games, np.int64); live = np.ones(games, bool)
  result = np.zeros(games, np.int64)          # +1 agent won, -1 agent lost, 0 draw
  for ply in range(9):
    if not live.any():
      break
    legal = (board == 0)
    act = np.zeros(games, np.int64)
    agent_moves = live & (turn == agent_side)
    if agent_moves.any():
      obs = (turn[:, None] * board).astype(np.float32)
      lg = legal.astype(np.uint8); lg[~live] = 1               # (finished boards: any mask, their result is not read)
      eng.initial_inference(obs)
      eng.root_prepare(turn.astype(np.int8), lg, None, device_rng=False)
      eng.search()
      out = eng.finalize(0.0, np.full(games, 0.5))
      act = out['action'].cpu().numpy().astype(np.int64)
    rnd = np.array([rng.choice(np.flatnonzero(legal[i])) if live[i] and legal[i].any() else 0 for i in range(games)])
    act = np.where(agent_moves, act, rnd)
    idx = np.flatnonzero(live)
    assert np.all(board[idx, act[idx]] == 0)
    board[idx, act[idx]] = turn[idx]
    won = np.any(np.abs(board[:, LINES].sum(-1)) == 3, axis=1) & live
    result[won] = np.where(turn[won] == agent_side, 1, -1)
    full = ~(board == 0).any(1)
    live &= ~won & ~full
    turn = -turn
  eng.close()
  return int((result == 1).sum()), int((result == 0).sum()), int((result == -1).sum())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--training_steps', type=int, default=3000)
  ap.add_argument('--num_envs', type=int, default=1024)
  ap.add_argument('--out', default=None)
  a = ap.parse_args()
  from model_based_rl_amd import train
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.networks import get_network
  from model_based_rl_amd.engine import flatten_weights
  base = ['--environment', 'TicTacToe', '--two_players', '--architecture', 'FCNetwork', '--td_steps', '10', '--discount', '1',
          '--known_bounds', '-1', '1', '--num_simulations', '30', '--seed', '0', '--num_envs', str(a.num_envs)]
  torch.manual_seed(0)
  untrained = flatten_weights(get_network(make_config(base), torch.device('cpu')).state_dict())
  before = {side: play_vs_random(untrained, side) for side in (1, -1)}
  saves = os.path.join('/tmp', 'mz_ttt_learning_%d' % os.getpid())
  t0 = time.time()
  thr = train.main(base + ['--max_moves', '-1', '--training_steps', str(a.training_steps), '--stored_before_train', '20000',
                           '--batch_size', '256', '--window_size', '200000', '--send_weights_frequency', '100',
                           '--weight_sync_frequency', '16', '--use_gpu_for', 'actors', 'learner', '--gpu_turns', '--runs_dir', saves,
                           '--run_tag', 'learn', '--save_state_frequency', str(a.training_steps), '--learner_log_frequency', '500'])
  seconds = time.time() - t0
  import glob
  ck = sorted(glob.glob(os.path.join(saves, '**', 'saves', '*'), recursive=True), key=os.path.getmtime)[-1]
  state = torch.load(ck, map_location='cpu', weights_only=False)
  trained = flatten_weights(state['weights'])
  after = {side: play_vs_random(trained, side) for side in (1, -1)}
  out = {'recipe': ' '.join(base), 'training_steps': int(state['training_step']), 'train_seconds': seconds,
         'selfplay_frames': thr['frames'], 'selfplay_games': thr['games'], 'learner': thr.get('learner'),
         'vs_random_512_games': {'untrained': {'agent_first (win, draw, loss)': before[1], 'agent_second': before[-1]},
                                 'trained': {'agent_first (win, draw, loss)': after[1], 'agent_second': after[-1]}}}
  print(json.dumps(out))
  if a.out:
    json.dump(out, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
  main()
