"""Soak of the benchmarked whole-moves launch (4096 envs, 16 moves per launch) with NO margin: launches sampled all along a
long run are replayed -- all 16 of their moves, all 4096 trees -- through the oracle's tree (oracle/mz_oracle.c) on the
device's own logged network outputs (mz_sim_io), Dirichlet draws and select_action uniforms; every record (visit
distribution, action, root value and error as float64) must be identical.  A rare corrupted tile or a wrong tree-code corner
shows up as one differing tree somewhere in the run.

  python scripts/soak_exact.py [--shape lunar|pong|ttt] [--moves 8000] [--checks 40] [--split] [--out file.jsonl]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc
from model_based_rl_amd.engine import Engine, records_view
from tests.parity_util import philox_action_uniform, replay_move
from tests.test_gpu_fused_exact import LOG_SHAPES, G

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='lunar'); ap.add_argument('--moves', type=int, default=8000)
ap.add_argument('--checks', type=int, default=40); ap.add_argument('--split', action='store_true'); ap.add_argument('--out', default=None)
a = ap.parse_args()
sh = LOG_SHAPES[a.shape]
O, A, sims, B, seed, chunk = sh['O'], sh['A'], sh['sims'], 4096, 2025, 16
game = bool(sh.get('game'))
w = orc.load_weights(np.load(os.path.join(G, sh['gold'] + '.npz')))
eng = Engine(B, O, A, sims, seed=seed, split_f16=a.split, **(dict(two_players=True, known_bounds=(-1.0, 1.0), discount=1.0) if game else {}))
if game:
  eng.selfplay_set_env('tictactoe')
assert eng.selfplay_moves_per_launch() == 16
eng.set_weights(w)
if sh['u8']:
  eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
eng.selfplay_noise_log(True)
eng.selfplay_reset(200, 1.0, stagger=True)
log = eng.sim_io('log', keep_moves=chunk)
cfg = orc.tree_cfg(A, sims, two_players=game, known_bounds=(-1.0, 1.0) if game else (None, None), discount=1.0 if game else 0.997)
rng = np.random.RandomState(1)
nchunks = a.moves // chunk
check = set(rng.choice(nchunks, size=min(a.checks, nchunks), replace=False).tolist())
trees = below = longest = 0
t0 = time.time()
for c in range(nchunks):
  eng.selfplay_steps(chunk)
  buf, n = eng.selfplay_drain()
  if c not in check:
    continue
  torch.cuda.synchronize()
  rec = buf[:n].numpy().copy()
  rv = records_view(rec, O, A)
  io_all = log.cpu().numpy()
  for k in range(chunk):
    m = c * chunk + k
    io = io_all[m % chunk]
    noise = eng.selfplay_noise(m)
    u = philox_action_uniform(seed, np.arange(B), m)
    if game:
      legal, to_play = (rec[k, :, :O] == 0).astype(np.uint8), rv['to_play'][k].astype(np.int8)
    else:
      legal, to_play = None, np.ones(B, np.int8)
    ref = replay_move(cfg, B, A, sims, io, noise, 0.25, to_play, legal, 1.0, u, want_tree=(k == chunk - 1))
    ok = (np.all(rv['child_visits'][k] == ref['child_visits'].astype(np.float32), axis=1) & (rv['action'][k] == ref['action']) &
          (rv['root_value'][k] == ref['root_value']) & (rv['error'][k] == ref['root_value'] - ref['v0'].astype(np.float64)))
    assert ok.all(), ('move %d: trees %s differ from the oracle on their own logged outputs' % (m, np.flatnonzero(~ok)[:8].tolist()))
    trees += B
    below += int((ref['margin'] <= 1e-4).sum())
    if ref['tree'] is not None:
      longest = max(longest, int((ref['tree']['N'] > 0).sum(1).max()))
eng.sim_io('off')
eng.close()
out = {'shape': a.shape, 'split_f16': bool(a.split), 'moves_played': nchunks * chunk, 'launches_checked': len(check),
       'moves_checked': len(check) * chunk, 'trees_checked': trees, 'trees_differing': 0,
       'trees_with_a_decision_within_1e-4_of_a_tie': below, 'most_visited_nodes_in_a_checked_tree': longest,
       'rule': 'records identical to oracle/mz_oracle.c replaying the logged network outputs: no margin, no excluded tree',
       'seconds': time.time() - t0}
print(json.dumps(out))
if a.out:
  with open(a.out, 'a') as f:
    f.write(json.dumps(out) + '\n')
