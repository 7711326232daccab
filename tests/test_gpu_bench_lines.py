"""The secondary modes of bench.py at toy sizes: each prints one well-formed JSON line with the contract's keys and a
`roofline` object (so that the lines committed under profiles/ cannot rot unnoticed).  Child processes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline')


def line_of(args):
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
  rows = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(rows) == 1, out.stdout[-2000:]
  line = json.loads(rows[0])
  for k in KEYS:
    assert k in line, k
  r = line['roofline']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
    assert k in r, k
  assert line['value'] > 0 and r['achieved'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
  assert 'workload' in line['config'] and line['n_gpus'] == 1
  return line


def test_tictactoe_line():
  line = line_of(['--workload', 'tictactoe', '--envs', '256', '--steps', '16', '--warmup', '4', '--no-cpu-baseline', '--min-seconds', '0.2'])
  assert 'TicTacToe on the device' in line['config']['environment'] and line['roofline']['bound'] == 'mfma'
  assert line['roofline']['moves_per_launch'] == 16         # whole moves inside the two-player launch
  assert 0.5 * 256 * line['timed_steps'] < line['value'] * line['timed_seconds'] < 1.5 * 256 * line['timed_steps']


def test_tree_line():
  line = line_of(['--workload', 'tree', '--envs', '512', '--steps', '4', '--warmup', '1'])
  r = line['roofline']
  assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and set(r['per_kernel']) == {'k_tree_select', 'k_tree_expand_backup'}
  assert all(v['us_per_launch'] > 1.0 for v in r['per_kernel'].values()) and line['config']['mean_leaf_depth'] > 1.5


def test_breakout_line():
  line = line_of(['--workload', 'breakout', '--envs', '32', '--steps', '2', '--warmup', '1'])
  assert line['config']['host_syncs_in_simulation_loop'] == 0 and line['record_bytes_per_env_step'] == 4 * (4 * 96 * 96 // 4 + 4 + 10)
  assert line['network_calls']['recurrent_inference_ms'] > 0


def test_pong_split_line():
  line = line_of(['--workload', 'pong', '--split-f16', '--envs', '256', '--steps', '16', '--warmup', '4', '--no-cpu-baseline',
                  '--min-seconds', '0.2'])
  assert line.get('secondary_line') and line['roofline']['kernel'] == 'k_search_h2' and 'f16x2' in line['dtype']


def test_learner_line():
  line = line_of(['--workload', 'learner', '--steps', '60', '--runs', '1', '--batch', '256,1024'])
  assert line['metric'] == 'learner_updates_per_second' and line['roofline']['bound'] == 'mfma' and line.get('secondary')
  assert line['roofline']['flop_per_update'] == 1425801216        # batch 256, K = 5, LunarLander shapes: 3 x forward
  assert 20 < line['roofline']['us_per_update'] < 400              # the native step (the PyTorch graph: ~900)
  assert line['torch_graph']['gpu_ms_per_update'] > 2e-3 * line['roofline']['us_per_update']
  # the timed call is Learner.launch, and its stretches between Python boundaries ran in the native loop (mz_fcl_run)
  assert line['config']['native_loop'] and 'Learner.launch' in line['config']['workload']
  assert line['config']['native_loop_host_us_per_update']['updates'] >= 60
  # the batch sweep: updates/s, samples/s, roofline fraction and host microseconds per update for every batch
  sw = line['batch_sweep']
  assert [p_['batch'] for p_ in sw] == [256, 1024]
  for p_ in sw:
    assert p_['updates_per_s'] > 0 and abs(p_['samples_per_s'] - p_['updates_per_s'] * p_['batch']) < 1e-6 * p_['samples_per_s']
    assert 0 < p_['frac'] < 1 and p_['us_per_update_gpu'] > 10 and set(p_['host_us_per_update']) >= {'sample_us', 'refresh_us', 'launch_us'}
  # (no ordering of the two fractions: through the product's loop batch 1024 is bound by the ONE host thread that samples and refreshes)
  assert sw[1]['flop_per_update'] == 4 * sw[0]['flop_per_update']


def test_depth_sensitivity_block_and_policy_gain():
  """--policy_gain g (policy-head output layer x g: sharper priors, deeper trees) and the headline's depth_sensitivity block
  (VERDICT r05 item 4): a sharper policy deepens the search, the block carries rate, roofline fraction, depths and the cycles of the two
  depth-dependent phases per gain."""
  if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
  from bench import sharpened
  import torch
  w = {'policy_head.policy.weight': torch.ones(4, 8), 'policy_head.policy.bias': torch.ones(4), 'x': torch.ones(2)}
  s4 = sharpened(w, 4)
  assert float(s4['policy_head.policy.weight'][0, 0]) == 4.0 and float(s4['policy_head.policy.bias'][0]) == 4.0 and float(s4['x'][0]) == 1.0
  line = line_of(['--steps', '32', '--warmup', '16', '--runs', '1', '--no-cpu-baseline', '--no-live-traffic', '--min-seconds', '0.2'])
  rows = line['depth_sensitivity']['rows']
  assert [r_['policy_gain'] for r_ in rows] == [1, 4, 24, 96]
  assert rows[2]['mean_leaf_depth'] > 3.5 and rows[3]['mean_leaf_depth'] > 5.5 and rows[0]['mean_leaf_depth'] < 3.0      # depth ~4 and ~6 beside the headline's 2.5
  for r_ in rows:
    assert r_['env_steps_per_s'] > 1e6 and 0.2 < r_['frac'] < 1 and r_['cycles_per_sim_wave0']['t_select'] > 0
  assert rows[-1]['cycles_per_sim_wave0']['t_select'] > rows[0]['cycles_per_sim_wave0']['t_select']
  assert 'share_of_added' in rows[-1]
  g = line_of(['--steps', '32', '--warmup', '16', '--runs', '1', '--no-cpu-baseline', '--no-live-traffic', '--min-seconds', '0.2', '--policy_gain', '4'])
  assert g.get('secondary_line') and g['config']['policy_gain'] == 4.0 and 'depth_sensitivity' not in g
