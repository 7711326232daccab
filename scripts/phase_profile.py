"""In-kernel phase stamps of the fused search kernel (mz_search_phase_profile: a diagnostic build of k_search_fused with
s_memtime stamps between its phases; never used for timing claims -- the stamps pin the schedule, the sum runs ~1 %
above the production build).  Prints cycles per simulation per wave and writes profiles/phase_cycles_<tag>.json.

  python scripts/phase_profile.py <obs> <actions> <sims> [tag]
"""
import json, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
torch.manual_seed(0)
O, A, SIMS = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 4, 30)
tag = sys.argv[4] if len(sys.argv) > 4 else None
net = FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).eval()
two = bool(os.environ.get('MZ_PP_TWO'))          # two players, known bounds (-1, 1), discount 1: the TicTacToe recipe's tree
eng = Engine(4096, O, A, SIMS, seed=1, **(dict(two_players=True, known_bounds=(-1.0, 1.0), discount=1.0) if two else {}))
eng.set_weights(net.state_dict())
obs = torch.randn(4096, O, device='cuda')
names = ['gather', 'bar', 'dyn_fc1', 'dyn_fc2', 'comb1', 'ln/rew', 'pred_fc1', 'pred_fc2', 'comb2', 'val/lg', 't_expand',
         't_backup', 't_select', 't_rest']
for it in range(3):
  eng.initial_inference(obs); eng.root_prepare(None, None, None, device_rng=True, move=it)
  c = eng.search_phase_profile()
print('cycles per sim (avg over WGs), per wave:')
for p in range(14):
  print('%-9s' % names[p], ' '.join('%8.0f' % (c[w, p] / SIMS) for w in range(4)))
print('total    ', ' '.join('%8.0f' % (c[w].sum() / SIMS) for w in range(4)))
spread = eng.search_phase_spread()
print('workgroup totals per sim: mean %.0f  min %.0f  max %.0f  (max / mean %.3f)' %
      (spread[0] / SIMS, spread[1] / SIMS, spread[2] / SIMS, spread[2] / spread[0]))
# the persistent self-play launch (whole moves inside one launch of the exact-f32 kernel): cycles per phase of a MOVE
move_phases = None
if eng.selfplay_moves_per_launch() > 0 and not eng.split_f16:
  if O > 64:
    eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
  eng.selfplay_reset(256, 1.0, stagger=True)
  eng.selfplay_steps(32)
  for it in range(3):
    move_phases = eng.selfplay_phase_profile(16)
    eng.selfplay_drain()
  tot = sum(move_phases.values())
  print('persistent self-play launch, cycles per move (mean over waves):')
  for k in Engine.SELFPLAY_PHASES:
    print('  %-20s %9.0f  %5.1f %%' % (k, move_phases[k], 100 * move_phases[k] / tot))
  print('  %-20s %9.0f' % ('total', tot))
if tag:
  out = {'what': 'k_search_fused phase stamps (s_memtime, 100 MHz-independent shader clock cycles), cycles per simulation, '
                 'averaged over the 256 workgroups, per wave; 4096 trees',
         'obs': O, 'actions': A, 'sims': SIMS, 'phases': names,
         'cycles_per_sim_per_wave': {names[p]: [float(c[w, p]) / SIMS for w in range(4)] for p in range(14)},
         'total_per_wave': [float(c[w].sum()) / SIMS for w in range(4)],
         'workgroup_total_per_sim': {'mean': spread[0] / SIMS, 'min': spread[1] / SIMS, 'max': spread[2] / SIMS},
         'mfma_stage_cycles_wave0': float(sum(c[0, p] for p in (2, 3, 6, 7))) / SIMS,
         'note': 'waves wait for each other at the four barriers of a simulation (end of gather, partials of the two out '
                 'layers, LayerNorm): the time a wave spends waiting shows up in the phase that ENDS with the barrier '
                 '(bar, comb1, ln/rew, comb2); per-wave differences inside t_* are the lock-step cost of unequal tree depths'}
  if move_phases is not None:
    out['selfplay_move'] = {'what': 'mz_selfplay_phase_profile: shader cycles per phase of one MOVE inside the persistent '
                                    'self-play launch (16 moves per launch), mean over all waves; the stamps pin the schedule',
                            'cycles_per_move': move_phases, 'total': sum(move_phases.values())}
  json.dump(out, open(os.path.join(ROOT, 'profiles', 'phase_cycles_%s.json' % tag), 'w'), indent=1)
