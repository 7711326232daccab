"""Soak: a long run of the benchmarked whole-moves launch (4096 envs, 8 moves per launch) with moves sampled all along it
checked against the CPU oracle under the margin rule (tests/test_gpu_bench_parity.py): a rare corrupted tile -- the realistic
failure of hand-scheduled MFMA wait states -- would show as a wrong tree above the margin somewhere in the run.

  python scripts/soak_parity.py [--shape lunar|pong] [--moves 4000] [--checks 40] [--split]      -> one summary line (+ --out file)"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc
from model_based_rl_amd.engine import Engine, records_view
from tests.test_gpu_bench_parity import SHAPES, MARGIN, philox_action_uniform, G

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='lunar'); ap.add_argument('--moves', type=int, default=4000)
ap.add_argument('--checks', type=int, default=40); ap.add_argument('--split', action='store_true'); ap.add_argument('--out', default=None)
a = ap.parse_args()
sh = SHAPES[a.shape]
O, A, sims, B, seed, chunk = sh['O'], sh['A'], sh['sims'], 4096, 99, 8
w = orc.load_weights(np.load(os.path.join(G, sh['gold'] + '.npz')))
eng = Engine(B, O, A, sims, seed=seed, split_f16=a.split)
assert eng.selfplay_moves_per_launch() == 16
eng.set_weights(w)
if sh['u8']:
  eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
eng.selfplay_noise_log(True)
eng.selfplay_reset(200, 1.0, stagger=True)
cfg, net = orc.tree_cfg(A, sims), orc.FCNet(w, O, A)
rng = np.random.RandomState(1)
nchunks = a.moves // chunk
check_chunks = set(rng.choice(nchunks, size=min(a.checks, nchunks), replace=False).tolist())
tot_wide = tot_bad = tot_trees = tot_below_same = tot_below = 0
worst_rv = 0.0
t0 = time.time()
for c in range(nchunks):
  eng.selfplay_steps(chunk)
  buf, n = eng.selfplay_drain()
  if c not in check_chunks:
    continue
  torch.cuda.synchronize()
  rec = buf[:n].numpy().copy()
  rv = records_view(rec, O, A)
  k = int(rng.randint(chunk))                 # a move somewhere inside this launch
  m = c * chunk + k
  raw = rec[k, :, :O]
  obs = (raw - np.float32(0.0)) / np.float32(255.0) if sh['u8'] else raw
  ref = orc.search_fc_threads(cfg, net, obs, noise=eng.selfplay_noise(m), temperature=1.0,
                              uniform=philox_action_uniform(seed, np.arange(B), m), tree=False)
  wide = ref['margin'] > MARGIN
  same = np.all(rv['child_visits'][k] == ref['child_visits'].astype(np.float32), axis=1) & (rv['action'][k] == ref['action'])
  tot_trees += B; tot_wide += int(wide.sum()); tot_bad += int((wide & ~same).sum())
  tot_below += int((~wide).sum()); tot_below_same += int((same & ~wide).sum())
  worst_rv = max(worst_rv, float(np.abs(rv['root_value'][k] - ref['root_value'])[wide].max()))
eng.close()
out = {'shape': a.shape, 'split_f16': bool(a.split), 'moves_played': nchunks * chunk, 'moves_checked': len(check_chunks),
       'trees_checked': tot_trees, 'trees_above_margin': tot_wide, 'wrong_above_margin': tot_bad,
       'identical_below_margin': tot_below_same, 'trees_below_margin': tot_below, 'margin': MARGIN,
       'max_root_value_diff_above_margin': worst_rv, 'seconds': time.time() - t0}
print(json.dumps(out))
if a.out:
  with open(a.out, 'a') as f:
    f.write(json.dumps(out) + '\n')
assert tot_bad == 0
