"""Secondary bench line (never the headline): the second roofline SURVEY.md s8(d)(ii) asks for -- the stand-alone tree
kernels k_tree_select (the descent of mcts.py:83-94) and k_tree_expand_backup (node.expand + backpropagate, mcts.py:97-99,
126-143), the pair behind mz_select / mz_expand_backup that every external network (BASELINE configs[4]) and the
Node front-end (mcts.MCTS.run) go through.  4096 trees, 4 actions, 50 simulations per move, network outputs replayed from
fixed device buffers (no network in the loop): gather / scatter graph work, priced against HBM (8 TB/s) and, since the
38 MB working set is cache resident, against the guide's L2 and Infinity-Cache figures.  Called by bench.py --workload tree.

Algorithmic bytes per simulation and tree (A actions, leaf depth d; node = N i32 + W f64 + P f64 + R f32 =
24 B, to_play i8, expansion index i32):
  select          d * (4 + 24 A)            read   (per level: the parent's expansion index + A children)
  expand_backup   29 (d + 1)                 RMW    (per path node: N, W, MinMax pair, R, to_play)
                  + 24 A                     write  (the new children)
                  + 4 A + 8                  read   (logits, value, reward)
(the hidden-state gather / scatter of s8(d)'s 0.94 KB figure, 400 B, is torch's index_select / copy_ on this path.)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_TBPS, HBM_MEASURED_TBPS, L2_TBPS, MALL_TBPS = 8.0, 6.29, 34.5, 8.6      # MI355X_MICROARCH.md


def main(args):
  if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    raise SystemExit('--workload tree is a one-GPU secondary line')
  device = torch.device('cuda', 0)
  torch.cuda.set_device(device)
  sys.path.insert(0, ROOT)
  from model_based_rl_amd.engine import Engine
  B = args.envs or 4096
  A, SIMS = 4, 50
  moves = args.steps if args.steps != 512 else 40
  warmup = args.warmup if args.warmup != 64 else 4
  eng = Engine(B, 1, A, SIMS, seed=1234, device=device)
  g = torch.Generator(device=device); g.manual_seed(7)
  # "recorded" network outputs: one fixed set per simulation, resident on the device
  val = [torch.randn(B, device=device, generator=g) * 0.3 for _ in range(SIMS + 1)]
  rew = [torch.rand(B, device=device, generator=g) * 2 - 1 for _ in range(SIMS)]
  lg = [torch.randn(B, A, device=device, generator=g) for _ in range(SIMS + 1)]
  depth_sum = torch.zeros(B, dtype=torch.int64, device=device)
  ev = lambda: torch.cuda.Event(enable_timing=True)

  def move(m, timing=None):
    eng.root_load(val[SIMS], lg[SIMS])
    eng.root_prepare(None, None, None, device_rng=True, move=m)
    sel = eng.select()
    for s in range(SIMS):
      if timing is not None:
        depth_sum.add_(sel[3].long())
      # expand + backup of this simulation and the descent of the next in one launch (mz_expand_backup_select)
      sel = eng.expand_backup_select(val[s], rew[s], lg[s], last=(s + 1 == SIMS))

  for m in range(warmup):
    move(m)
  torch.cuda.synchronize(device)
  t0 = time.perf_counter()
  for m in range(moves):
    move(warmup + m)
  torch.cuda.synchronize(device)
  dt = time.perf_counter() - t0
  # per-kernel clock: start / stop events on each kernel's own dispatch (mz_tree_pair_timed), every launch of 4 moves;
  # the leaf depths of the same simulations come from a replay of those moves through mz_select (same seeds, same trees)
  timing = []
  for m in range(4):
    eng.root_load(val[SIMS], lg[SIMS])
    eng.root_prepare(None, None, None, device_rng=True, move=warmup + moves + m)
    for s in range(SIMS):
      timing.append(eng.tree_pair_timed(val[s], rew[s], lg[s]))
  for m in range(4):
    move(warmup + moves + m, [])
  torch.cuda.synchronize(device)
  sel_us = 1e3 * float(np.mean([a for a, _ in timing if a >= 0]))      # (-1: the root kernel had already selected)
  exb_us = 1e3 * float(np.mean([b for _, b in timing]))
  d_mean = float(depth_sum.sum().item()) / (B * len(timing))
  bytes_sel = B * d_mean * (4 + 24 * A)
  bytes_exb = B * (29 * (d_mean + 1) + 24 * A + 4 * A + 8)
  ach_sel, ach_exb = bytes_sel / (sel_us * 1e-6) / 1e9, bytes_exb / (exb_us * 1e-6) / 1e9
  both = (bytes_sel + bytes_exb) / ((sel_us + exb_us) * 1e-6) / 1e9
  NN = 1 + (SIMS + 1) * A
  traffic = None
  tfile = os.path.join(ROOT, 'profiles', 'r04_b_tree_traffic.json')      # (FETCH_SIZE / WRITE_SIZE passes of this command, builder-run)
  if os.path.exists(tfile) and B == 4096:
    traffic = json.load(open(tfile))

  out = {
      'metric': 'MCTS sims/sec/GPU (tree kernels alone: mz_expand_backup_select, network outputs replayed)',
      'value': B * SIMS * moves / dt, 'unit': 'simulations/s', 'n_gpus': 1, 'steps': moves, 'warmup': warmup,
      'ms_per_step': 1e3 * dt / moves, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f64 tree arithmetic (i32 / f64 / f32 node fields)', 'data': 'synthetic', 'secondary_line': True,
      'config': {'workload': 'stand-alone tree kernels, %d trees x %d actions x %d simulations per move, %d nodes per tree, '
                             'replayed network outputs, Dirichlet noise on the device' % (B, A, SIMS, NN),
                 'envs_per_gpu': B, 'num_simulations': SIMS, 'mean_leaf_depth': d_mean,
                 'working_set_bytes': B * ((NN + 3 + 3) // 4 * 4) * 32 + B * (SIMS + 2) * 4,
                 'node_pool': '32-byte node records (W f64, P f64, N i32, E i32, R f32, to_play i8): the A children of a node are 32 A contiguous bytes',
                 'note': 'one step = one move = root + %d x k_tree_step_ext (expand + backup + the next descent in one launch, '
                         'mz_expand_backup_select); wall time includes the launch gaps of %d dependent launches per move; the per-kernel '
                         'figures below clock the two stand-alone kernels (mz_select, mz_expand_backup) on their own dispatches' % (SIMS, SIMS + 3),
                 'us_per_simulation_wall': 1e6 * dt / moves / SIMS},
      'env_steps_per_s': B * moves / dt,
      'roofline': {'bound': 'hbm', 'kernel': 'k_tree_select + k_tree_expand_backup (pair)', 'achieved': both, 'peak': HBM_TBPS * 1e3,
                   'unit': 'GB/s', 'frac': both / (HBM_TBPS * 1e3), 'traffic': traffic,
                   'per_kernel': {
                       'k_tree_select': {'us_per_launch': sel_us, 'algorithmic_bytes_per_launch': bytes_sel, 'achieved_GBps': ach_sel,
                                         'frac_of_hbm_peak': ach_sel / (HBM_TBPS * 1e3)},
                       'k_tree_expand_backup': {'us_per_launch': exb_us, 'algorithmic_bytes_per_launch': bytes_exb,
                                                'achieved_GBps': ach_exb, 'frac_of_hbm_peak': ach_exb / (HBM_TBPS * 1e3)}},
                   'against_the_caches': {'working set': 'cache resident (%.0f MB of node records)' % (B * NN * 32 / 1e6),
                                          'frac_of_infinity_cache_random_rows_8.6TBps': both / (MALL_TBPS * 1e3),
                                          'frac_of_l2_34.5TBps': both / (L2_TBPS * 1e3),
                                          'frac_of_measured_hbm_6.29TBps': both / (HBM_MEASURED_TBPS * 1e3)},
                   'reading': 'these launches are latency-, not bandwidth-bound: a descent is d dependent round trips of A children '
                              'each (one 16-lane group per tree), %.0f KB of algorithmic traffic per launch in %.1f us; what '
                              'bounds them is the chain of dependent cache round trips and the %.1f us launch floor, not bytes' %
                              (bytes_sel / 1e3, sel_us, 1.5),
                   'hbm_side_traffic': ({'source': 'profiles/r04_b_tree_traffic.json: FETCH_SIZE / WRITE_SIZE passes of this command',
                                         'k_tree_step_ext_fetched_over_algorithmic_of_the_pair': (traffic['k_tree_step_ext']['hbm_bytes_per_launch'] / (bytes_sel + bytes_exb)
                                                                                                     if 'k_tree_step_ext' in traffic else None),
                                         'k_tree_select_fetched_over_algorithmic': traffic['k_tree_select']['hbm_bytes_per_launch'] / bytes_sel,
                                         'k_tree_expand_backup_fetched_over_algorithmic': traffic['k_tree_expand_backup']['hbm_bytes_per_launch'] / bytes_exb}
                                        if traffic and 'k_tree_select' in traffic else None),
                   'clock': 'start / stop events on each kernel\'s own dispatch (hipExtLaunchKernelGGL, mz_tree_pair_timed), every '
                            'launch of 4 moves (%d launches of each kernel)' % len(timing)},
  }
  print(json.dumps(out), flush=True)
  eng.close()
