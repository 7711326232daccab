"""The launch bench.py times, at the size bench.py times it, END TO END against the CPU oracle (network included).

BENCH_rNN's `value` is produced by the persistent HEAD instantiation of the search kernel (whole moves inside one launch,
mz_selfplay_steps) on a full grid: 4096 environments = 256 workgroups, 16 moves per launch.  This test runs exactly that --
three launches of 16 moves (bench.py's launch size), a weight update before the third -- with the per-move Dirichlet log
(mz_selfplay_noise_log) and the simulation log (mz_sim_io) on, takes the observation of a move from its experience record,
and replays single moves (the first, one in the middle of the second launch, one in the middle of the third, the last)
through oracle/mz_oracle.c -- the oracle's OWN float32 network evaluation and its tree -- on all 4096 trees.

What is proven where:
  * the TREE code of this launch is proven exactly, on every tree, by tests/test_gpu_fused_exact.py (the device's logged
    network outputs replayed through the oracle's tree: no margin, no excluded trees);
  * here the two float32 NETWORK evaluations are different summation orders, so a select_child decision (mcts.py:104-113)
    whose two best scores are closer than that noise may legitimately resolve the other way.  The oracle reports, per tree,
    the smallest top-2 score gap of the search (pinned against the reference's own number in tests/test_oracle_tree.py).
    Trees whose margin exceeds MARGIN are EXPECTED to have the oracle's visit vector and action; an exception is accepted
    only if it is explained: that tree, replayed through the oracle's tree on the device's own logged outputs, gives exactly
    the device's result (so the tree code is not the cause), and its logged root outputs lie within the network tolerance of
    the oracle's (value: one step of the reference's float32 inverse-transform staircase, config.py:27-33; logits 1e-5) --
    at most MAX_EXPLAINED per move (r03's soak met one such tree in 1.69 M).  Trees below the margin are counted and printed.
"""
import json
import os

import numpy as np
import pytest

from tests.parity_util import philox_action_uniform, replay_move

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
MARGIN = 1e-4
MAX_EXPLAINED = 2


SHAPES = {
    # BASELINE configs[1]: LunarLander-v2 shapes, 30 simulations  /  configs[3]: Pong-ram shapes, 50 simulations, bytes + norm_obs
    'lunar': dict(gold='g1_net_lunar', O=8, A=4, sims=30, u8=False),
    'pong': dict(gold='g1_net_pong', O=128, A=6, sims=50, u8=True),
    # BASELINE configs[0]'s game at throughput size: TicTacToe on the device, two players (the digest test below only)
    'ttt': dict(gold='g1_net_ttt', O=9, A=9, sims=30, u8=False, game=True),
}


def perturbed(w, seed):
  """a second set of weights ('the learner published an update'): every tensor nudged by ~1 %"""
  rng = np.random.RandomState(seed)
  return {k: (v * (1 + 0.01 * rng.standard_normal(v.shape))).astype(np.float32) for k, v in w.items()}


def value_error_stats(dev_v, ref_v):
  """|device value - oracle value| over the checked roots, as the distribution VERDICT r04 item 6 asks for: the fraction
  above north_star's 1e-5 and the largest deviation in steps of the reference's own float32 staircase
  (config.py:27-33 quantises its output in steps of ~1.2e-4 (1 + |v|))"""
  d = np.abs(np.asarray(dev_v, np.float64) - np.asarray(ref_v, np.float64))
  step = 1.2e-4 * (1.0 + np.abs(np.asarray(ref_v, np.float64)))
  q = np.quantile(d, [0.5, 0.9, 0.99, 0.999])
  return {'roots': int(d.size), 'frac_above_1e-5': float((d > 1e-5).mean()), 'max_abs': float(d.max()),
          'max_in_staircase_steps': float((d / step).max()), 'median': float(q[0]), 'p90': float(q[1]), 'p99': float(q[2]),
          'p999': float(q[3])}


@pytest.mark.parametrize('shape,split,ntt', [('lunar', False, False), ('lunar', True, False), ('pong', False, False), ('pong', True, False),
                                            ('lunar', False, True), ('pong', False, True)])
def test_persistent_launch_on_a_full_grid_vs_oracle(shape, split, ntt):
  """ntt: the same launch with --no_target_transform (config.py:30-33 skipped): the float32 staircase of the inverse
  transform is gone, and EVERY value the check touches -- all 4096 root values of every checked move, the root values and
  errors of the record -- must hold north_star's 1e-5 with no staircase allowance."""
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine, records_view
  sh = SHAPES[shape]
  O, A, sims = sh['O'], sh['A'], sh['sims']
  B, T, seed, chunk = 4096, 11, 1234, 16                 # bench.py: 4096 envs, 16 moves per launch
  w0 = orc.load_weights(np.load(os.path.join(G, sh['gold'] + '.npz')))
  w1 = perturbed(w0, 5)
  eng = Engine(B, O, A, sims, seed=seed, split_f16=split, no_target_transform=ntt)
  assert eng.selfplay_moves_per_launch() == 16           # the whole-moves (HEAD) launch is what runs
  assert eng.split_f16 == split
  eng.set_weights(w0)
  if sh['u8']:
    eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
  eng.selfplay_noise_log(True)
  eng.selfplay_reset(T, 1.0, stagger=True)
  log = eng.sim_io('log', keep_moves=3 * chunk)
  eng.selfplay_steps(chunk)
  eng.selfplay_steps(chunk)
  eng.set_weights(w1)
  eng.selfplay_export_trees(True)                        # (a run-time flag of the same instantiation: the last move's trees)
  eng.selfplay_steps(chunk)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  assert n == 3 * chunk
  rec = buf[:n].numpy().copy()
  rv = records_view(rec, O, A)
  tree = eng.export_tree()
  cfg = orc.tree_cfg(A, sims)
  report = []
  dev_v0, ref_v0 = [], []
  for m, w in ((0, w0), (chunk + 3, w0), (2 * chunk + 1, w1), (3 * chunk - 1, w1)):
    raw = rec[m, :, :O]
    for b in (0, 1777, B - 1):                           # the record's observation is the synthetic env's (env, episode, t)
      assert np.array_equal(raw[b], eng.synth_obs(b, int(rv['episode'][m, b]), int(rv['step'][m, b]))[0])
    obs = (raw - np.float32(0.0)) / np.float32(255.0) if sh['u8'] else raw       # actors.py:134-137 in float32
    noise = eng.selfplay_noise(m)
    assert np.abs(noise.sum(1) - 1).max() < 1e-12 and noise.min() >= 0
    u = philox_action_uniform(seed, np.arange(B), m)
    ref = orc.search_fc_threads(cfg, orc.FCNet(w, O, A, no_target_transform=ntt), obs, noise=noise, temperature=1.0, uniform=u,
                                tree=(m == 3 * chunk - 1))
    v0_dev = log[m].cpu().numpy()[:, 0, 0]               # the device's root value of every tree of this move (mz_sim_io log)
    dev_v0.append(v0_dev); ref_v0.append(ref['v0'])
    if ntt:                                              # no staircase: every root value within 1e-5, on ALL trees
      assert np.abs(v0_dev - ref['v0']).max() <= 1e-5, (m, np.abs(v0_dev - ref['v0']).max())
    wide = ref['margin'] > MARGIN
    cv = rv['child_visits'][m]
    same = np.all(cv == ref['child_visits'].astype(np.float32), axis=1)
    same_act = rv['action'][m] == ref['action']
    drv = np.abs(rv['root_value'][m] - ref['root_value'])
    derr = np.abs(rv['error'][m] - (ref['root_value'] - ref['v0'].astype(np.float64)))
    line = ('%s%s move %2d: %d of %d trees above margin %.0e; visit vectors identical in %d of them and in %d of the %d '
            'below; max |root value - oracle| %.2e' % (shape, ' split-f16' if split else '', m, wide.sum(), B, MARGIN,
                                                      (same & wide).sum(), (same & ~wide).sum(), (~wide).sum(),
                                                      drv[wide].max()))
    print(line)
    report.append(line)
    assert wide.mean() > 0.5                              # (the guard must not empty the test)
    bad = np.flatnonzero(wide & ~same)
    if bad.size:      # expected: none.  An exception must be explained by the network outputs, not by the tree code
      assert bad.size <= MAX_EXPLAINED, (m, bad[:8], ref['margin'][bad[:8]], cv[bad[:8]], ref['child_visits'][bad[:8]])
      io = log[m].cpu().numpy()[bad]
      rep = replay_move(cfg, bad.size, A, sims, io, noise[bad], 0.25, np.ones(bad.size, np.int8), None, 1.0, u[bad])
      assert np.array_equal(rep['child_visits'].astype(np.float32), cv[bad]) and np.array_equal(rep['action'], rv['action'][m][bad])
      dv = np.abs(io[:, 0, 0] - ref['v0'][bad])
      assert np.all(dv <= (1e-5 if ntt else 1.5e-4 * (1 + np.abs(ref['v0'][bad])))), (m, bad, dv)
      line += '; %d tree(s) above the margin differ, explained by their network outputs: %s' % (bad.size, bad.tolist())
      report[-1] = line
      print(line)
    ok = wide & same
    assert np.all(same_act[ok]), (m, np.flatnonzero(ok & ~same_act)[:8])
    # Scalars per TREE, derived (a constant 5e-4 until r05): root.value() is the mean of <= sims + 1 backed-up
    # values, each a discounted sum of network scalars that lie within ONE step of the reference's own float32 staircase
    # (Config.inverse_transform, config.py:27-33: steps of <= 1.5e-4 (1 + |x|)) -- so the mean moves by at most one step at the
    # tree's LARGEST scalar; the root error is a difference of two such numbers.  smax: from the device's logged outputs of this move.
    io_m = log[m].cpu().numpy()
    smax = np.abs(io_m[:, :, :2]).max(axis=(1, 2)).astype(np.float64)
    step = np.full(smax.shape, 1e-5) if ntt else 1.5e-4 * (1.0 + smax)       # (--no_target_transform: no staircase, 1e-5 on every scalar)
    assert np.all(drv[ok] <= step[ok]) and np.all(derr[ok] <= 2 * step[ok]), (m, (drv[ok] / step[ok]).max(), (derr[ok] / step[ok]).max())
    assert np.all(np.take_along_axis(cv, rv['action'][m][:, None], -1) > 0)
    if ref.get('tree') is not None:                       # the last move: every integer field of the exported trees
      for k in ('N', 'E'):
        eq = np.all(tree[k] == ref['tree'][k], axis=1)
        assert np.all(eq[wide & same]), (k, np.flatnonzero(wide & same & ~eq)[:8])
      whole = np.all(tree['N'] == ref['tree']['N'], axis=1) & np.all(tree['E'] == ref['tree']['E'], axis=1)
      # value sums of whole-identical trees: W of a node = the sum of its N backed-up values, each within one staircase step at
      # the tree's largest scalar (a constant 5e-3 until r05)
      dW = np.abs(tree['W'] - ref['tree']['W'])
      lim = tree['N'].astype(np.float64) * step[:, None] + 1e-12
      assert np.all(dW[whole] <= lim[whole]), float((dW[whole] / lim[whole]).max())
      assert np.array_equal(tree['noise'], noise)
  eng.sim_io('off')
  eng.close()
  stats = value_error_stats(np.concatenate(dev_v0), np.concatenate(ref_v0))
  stats.update(shape=shape, split_f16=split, no_target_transform=ntt, what='|root value of the device network - oracle| over the 4 checked moves x 4096 trees')
  line = 'value error distribution: %s' % json.dumps(stats)
  print(line)
  report.append(line)
  # the stated deviation as a measured number: with the target transform at most one staircase step; without it 1e-5 everywhere
  assert stats['max_in_staircase_steps'] <= 1.25 and (not ntt or stats['frac_above_1e-5'] == 0.0)
  if not ntt and not split:
    assert stats['frac_above_1e-5'] <= 0.03
  out = os.environ.get('MZ_PARITY_REPORT')
  if out:
    with open(out, 'a') as f:
      f.write('\n'.join(report) + '\n')


def _selfplay_digest(shape, persist, split=False):
  """records of 3 chunks x 16 moves at 4096 envs (bench.py's launch shape) as one sha256, in a child process (the launch
  structure is chosen at mz_create from MZ_NO_PERSIST)"""
  import subprocess, sys
  sh = SHAPES[shape]
  code = '''
import hashlib, os, sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import oracle as orc
from model_based_rl_amd.engine import Engine
w = orc.load_weights(np.load(%r))
game = %r
eng = Engine(4096, %d, %d, %d, seed=77, split_f16=%r, **(dict(two_players=True, known_bounds=(-1.0, 1.0), discount=1.0) if game else {}))
if game:
  eng.selfplay_set_env('tictactoe')
assert (eng.selfplay_moves_per_launch() > 0) == %r
eng.set_weights(w)
if %r:
  eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
eng.selfplay_reset(11, 1.0, stagger=True)
h = hashlib.sha256()
for k in range(3):
  eng.selfplay_steps(16)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  h.update(buf[:n].numpy().tobytes())
print('DIGEST', h.hexdigest())
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(G, sh['gold'] + '.npz'), bool(sh.get('game')),
       sh['O'], sh['A'], sh['sims'], split, persist, sh['u8'])
  env = dict(os.environ)
  env.pop('MZ_NO_PERSIST', None)
  if not persist:
    env['MZ_NO_PERSIST'] = '1'
  out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  return [l for l in out.stdout.splitlines() if l.startswith('DIGEST')][-1]


@pytest.mark.parametrize('shape,split', [('lunar', False), ('pong', False), ('lunar', True), ('ttt', False)])
def test_persistent_launch_equals_kernel_per_phase_on_a_full_grid(shape, split):
  """The whole-moves launch on 256 workgroups against the hipGraph of root + search kernels per move (MZ_NO_PERSIST=1):
  48 moves of 4096 environments, every byte of every record identical (same device functions, same keys).  'ttt': the
  two-player whole-moves launch of the device TicTacToe environment against its launch-per-step form (observe, initial
  inference, Dirichlet over the legal moves, root, search, env step as separate kernels)."""
  assert _selfplay_digest(shape, True, split) == _selfplay_digest(shape, False, split)
