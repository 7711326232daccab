"""Cycles per phase of a MOVE of the device TicTacToe environment inside the two-player whole-moves launch
(mz_selfplay_phase_profile on the <15,1,16> GAME instantiation; kernel development).  usage: ttt_phase_profile.py [out.json]"""
import json, os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
torch.manual_seed(0)
net = FCNetwork(9, 9, torch.device('cpu'), types.SimpleNamespace()).eval()
eng = Engine(4096, 9, 9, 30, two_players=True, known_bounds=(-1.0, 1.0), discount=1.0, seed=1)
eng.selfplay_set_env('tictactoe')
eng.set_weights(flatten_weights(net.state_dict()))
eng.selfplay_reset(9, 1.0)
eng.selfplay_steps(32); eng.selfplay_drain()
for _ in range(3):
  ph = eng.selfplay_phase_profile(16); eng.selfplay_drain()
tot = sum(ph.values())
for k in Engine.SELFPLAY_PHASES:
  print('%-22s %9.0f  %5.2f %%' % (k, ph[k], 100 * ph[k] / tot))
print('total', tot, ' per simulation', ph['simulations'] / 30)
if len(sys.argv) > 1:
  json.dump({'what': 'device TicTacToe, 4096 games, two-player whole-moves launch: shader cycles per phase of a move', 'cycles_per_move': ph, 'total': tot}, open(sys.argv[1], 'w'), indent=1)
