#!/usr/bin/env python3
"""Fidelity of the reference-shaped CPU baseline (oracle/ref_shaped.py): times it next to the IMPORTED, unmodified
reference (/root/reference, build container only) on the same search-only move loop -- the protocol of SURVEY.md s6:
initial inference + root expand + noise + MCTS.run + select_action, random-init FCNetwork, one thread -- alternating
the two so that clock drift hits both, and writes the ratio (accepted: 1.0 +- 0.1) to profiles/.

  python scripts/ref_shaped_ratio.py [--moves 150] [--out profiles/r02_ref_shaped_ratio.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import make_goldens as mg          # the reference import + config helpers of the golden generator
import ref_shaped


def reference_rate(ref, obs_dim, actions, sims, moves, seed):
  cfg = mg.make_ref_config(ref, ['--num_simulations', str(sims), '--seed', str(seed)], actions, (obs_dim,))
  torch.manual_seed(seed)
  np.random.seed(seed + 3)
  net = ref.networks.FCNetwork(obs_dim, actions, torch.device('cpu'), cfg).eval()
  mcts = ref.mcts.MCTS(cfg)
  rng = np.random.RandomState(seed)

  def play(n):
    with torch.inference_mode():
      for _ in range(n):
        obs = torch.from_numpy(np.float32(rng.standard_normal(obs_dim))).unsqueeze(0)
        init = net.initial_inference(obs)
        root = ref.mcts.Node(0)
        root.expand(init, 1, range(actions))
        root.add_exploration_noise(cfg.root_dirichlet_alpha, cfg.root_exploration_fraction)
        mcts.run(root, net)
        _ = root.value() - init.value.item()
        cfg.select_action(root, 1.0)

  play(3)
  t0 = time.perf_counter()
  play(moves)
  return moves / (time.perf_counter() - t0)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--moves', type=int, default=150)
  ap.add_argument('--rounds', type=int, default=3)
  ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r02_ref_shaped_ratio.json'))
  a = ap.parse_args()
  torch.set_num_threads(1)
  ref = mg._import_reference()
  rows = []
  for name, obs_dim, actions, sims in (('LunarLander-v2 shapes', 8, 4, 30), ('Pong-ram shapes', 128, 6, 50)):
    r_ref, r_own = [], []
    for k in range(a.rounds):
      r_ref.append(reference_rate(ref, obs_dim, actions, sims, a.moves, k))
      r_own.append(ref_shaped.measure(obs_dim, actions, sims, a.moves, k)['env_steps_per_s'])
    rows.append({'workload': name, 'sims': sims, 'reference_env_steps_per_s': float(np.median(r_ref)),
                 'ref_shaped_env_steps_per_s': float(np.median(r_own)),
                 'ratio': float(np.median(r_own) / np.median(r_ref)), 'rounds': a.rounds, 'moves_per_round': a.moves,
                 'all_reference': r_ref, 'all_ref_shaped': r_own})
    print(rows[-1], flush=True)
  out = {'what': 'oracle/ref_shaped.py timed beside the imported reference (search-only move loop, 1 thread)',
         'cpu': open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t'), 'rows': rows}
  json.dump(out, open(a.out, 'w'), indent=1)
  bad = [r for r in rows if not 0.9 <= r['ratio'] <= 1.1]
  if bad:
    raise SystemExit('ratio outside 1.0 +- 0.1: %s' % bad)


if __name__ == '__main__':
  main()
