#!/usr/bin/env python3
"""Copy the outputs of scripts/round_profile.sh <tag> from gpurun_out/ into profiles/ under the names profiles/README.md
lists, and derive profiles/traffic.json (per-move HBM bytes of the search kernels, tagged) from the PMC summary.
usage: collect_profiles.py <tag>"""
import glob, json, os, shutil, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
names = {
    'bench_%s.json': '%s_bench.json', 'bench_pong_%s.json': '%s_bench_pong_secondary.json',
    'bench_split_%s.json': '%s_bench_split_f16_secondary.json', 'bench_pong_split_%s.json': '%s_bench_pong_split_f16_secondary.json',
    'bench_breakout_%s.json': '%s_bench_breakout_secondary.json', 'bench_tictactoe_%s.json': '%s_bench_tictactoe_secondary.json',
    'bench_tree_%s.json': '%s_bench_tree_secondary.json', 'bench_rccl_world1_%s.json': '%s_bench_rccl_world1.json',
    'mfma_util_%s.json': '%s_mfma_util.json', 'ingest_bench_%s.json': '%s_ingest_bench.json',
    'one_replay_8ranks_4threads_%s.json': '%s_one_replay_8ranks_4threads.json',
    'one_replay_8ranks_8threads_%s.json': '%s_one_replay_8ranks_8threads.json',
    'tree_traffic_%s.json': '%s_tree_traffic.json', 'learner_speed_%s.json': '%s_learner_speed.json',
    'bench_selflaunch_2ranks_1gpu_%s.json': '%s_bench_selflaunch_2ranks_1gpu.json',
    'bench_selflaunch_8ranks_1gpu_%s.json': '%s_bench_selflaunch_8ranks_1gpu.json',
    'bench_learner_%s.json': '%s_bench_learner_secondary.json', 'learner_kernels_%s.csv': '%s_learner_kernels.csv',
    'bench_steps20_%s.json': '%s_bench_steps20.json', 'weight_sync_ab_%s.txt': '%s_weight_sync_ab.txt',
    'parity_full_grid_%s.txt': '%s_parity_full_grid.txt',
    'bench_8ranks_fullsize_1gpu_%s.json': '%s_bench_8ranks_fullsize_1gpu.json', 'host_threads_%s.txt': '%s_host_threads.txt',
    'one_replay_shapes_%s.txt': '%s_one_replay_shapes.txt', 'learner_step_sweep_%s.json': '%s_learner_step_sweep.json',
    'learner_step_kernels_256_%s.csv': '%s_learner_step_kernels_256.csv', 'learner_step_kernels_2048_%s.csv': '%s_learner_step_kernels_2048.csv',
    'learner_timeline_%s.txt': '%s_learner_timeline.txt',
}
# the product's own entry point (train --selfplay_only): its summary line as JSON
ts = os.path.join(G, 'train_selfplay_%s.txt' % tag)
if os.path.exists(ts):
  import re
  lines = open(ts).read().strip().splitlines()
  m = re.search(r'frames accepted by replay: (\d+) \((\d+) after (\d+) priming moves\), games: (\d+), ([0-9.]+) s -> (\d+) env-steps/s', lines[-1] if lines else '')
  if m:
    json.dump({'what': 'python -m model_based_rl_amd.train --selfplay_only --num_envs 4096 --num_simulations 30 --max_moves 4864 --prime_moves 768 '
                       '(LunarLander shapes): frames accepted by the replay after the priming moves / wall seconds of Actor.launch',
               'frames': int(m.group(1)), 'frames_timed': int(m.group(2)), 'prime_moves': int(m.group(3)), 'games': int(m.group(4)),
               'seconds': float(m.group(5)), 'env_steps_per_s': int(m.group(6)), 'line': lines[-1]},
              open(os.path.join(P, '%s_train_selfplay.json' % tag), 'w'), indent=1)
    print('->', '%s_train_selfplay.json' % tag)
for src, dst in names.items():
  s = os.path.join(G, src % tag)
  if os.path.exists(s):
    text = open(s).read().strip().splitlines()
    open(os.path.join(P, dst % tag), 'w').write((text[-1] if s.endswith('.json') and text and text[-1].startswith('{') and len(text) > 1 else '\n'.join(text)) + '\n')
    print('->', dst % tag)
ks = glob.glob(os.path.join(G, 'prof_%s' % tag, '**', '*kernel_stats.csv'), recursive=True)
if ks:      # (a tag run twice leaves two: the latest)
  shutil.copy(max(ks, key=os.path.getmtime), os.path.join(P, '%s_kernel_stats.csv' % tag)); print('->', '%s_kernel_stats.csv' % tag)
for f in glob.glob(os.path.join(G, 'phase_cycles_%s_*.json' % tag)):
  shutil.copy(f, P); print('->', os.path.basename(f))
t = os.path.join(G, 'traffic_%s.json' % tag)
if os.path.exists(t):
  tj = json.load(open(t))
  line = json.loads(open(os.path.join(G, 'bench_%s.json' % tag)).read().strip().splitlines()[-1])
  mpl = int(line['roofline'].get('moves_per_launch', 1))
  out = {k: v for k, v in tj.items() if k in ('k_search_fused', 'k_search_h2', 'k_pack_weights')}
  for k in ('k_search_fused', 'k_search_h2'):
    if k in out:
      out[k]['moves_per_launch'] = mpl
      out[k]['hbm_bytes_per_move'] = out[k]['hbm_bytes_per_launch'] / mpl
  out['tag'] = tag
  json.dump(out, open(os.path.join(P, 'traffic.json'), 'w'), indent=1)
  print('-> traffic.json (%s)' % tag)
