"""would assigning a workgroup's 16 trees to its 4 waves BY DEPTH shorten the Pong-ram shapes' barrier
wait (1.57 k of 35 k cycles per simulation, profiles/phase_cycles_r04_a_pong.json: `bar`)?

The fused kernel searches 16 trees per workgroup in lock-step; a wave owns 4 trees (16 lanes each) and its tree step lasts
as long as its DEEPEST tree's descent (one instruction stream); the workgroup's barrier then waits for the slowest wave.  So
a simulation's tree step lasts max over all 16 trees, whichever wave owns which tree; the measured `bar` is the mean over
the waves of (slowest wave - own wave).  This script takes real per-simulation leaf depths (the CPU oracle searching the
Pong-ram shapes with the reference-initialised network, 50 simulations) and evaluates both assignments: consecutive trees
per wave (what the kernel does) and trees sorted by the depth they reached in the previous simulations (the proposal).
CPU only.  usage: barrier_wait_experiment.py [out.json]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc

O, A, sims, B = 128, 6, 50, 1024
w = orc.load_weights(np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_net_pong.npz')))
net = orc.FCNet(w, O, A)
rng = np.random.RandomState(0)
obs = (rng.randint(0, 256, size=(B, O)).astype(np.float32) - 0.0) / 255.0
noise = rng.dirichlet([0.25] * A, size=B)
t = orc.Trees(orc.tree_cfg(A, sims), B)
h0, v0, lg0 = net.initial(obs)
t.root_expand(np.ones(B, np.int8), lg0, None)
t.add_noise(noise, 0.25)
hpool = np.zeros((B, sims + 1, 50), np.float32)
hpool[:, 0] = h0
depth = np.zeros((sims, B), np.int32)
for s in range(sims):
  leaf, slot, act, d = t.select()
  depth[s] = d
  h, r, v, lg = net.recurrent(hpool[np.arange(B), slot], act)
  hpool[:, s + 1] = h
  t.expand_backup(v, r, lg)

LEVEL = 380.0        # cycles per level of the descent on the critical path (t_select / levels, phase tables)
def evaluate(order):
  """order[s]: permutation of the 16 trees of every workgroup for simulation s -> (sum over sims of the workgroup's
  critical depth, mean barrier wait in levels)"""
  crit = wait = 0.0
  for s in range(sims):
    d = depth[s].reshape(-1, 16)
    d = np.take_along_axis(d, order[s], 1).reshape(-1, 4, 4)          # [workgroup][wave][tree]
    wave = d.max(2)
    wg = wave.max(1)
    crit += wg.mean()
    wait += (wg[:, None] - wave).mean()
  return crit / sims, wait / sims

ident = [np.tile(np.arange(16), (B // 16, 1)) for _ in range(sims)]
run_mean = np.zeros((B // 16, 16))
sorted_order = []
for s in range(sims):
  sorted_order.append(np.argsort(-run_mean, 1, kind='stable'))       # deepest-so-far trees share a wave
  run_mean = (run_mean * s + depth[s].reshape(-1, 16)) / (s + 1)
oracle_order = [np.argsort(-depth[s].reshape(-1, 16), 1, kind='stable') for s in range(sims)]   # a perfect predictor
out = {'shapes': 'Pong-ram (obs 128, 6 actions, 50 simulations), %d trees = %d workgroups of 16' % (B, B // 16),
       'mean_leaf_depth': float(depth.mean()), 'cycles_per_level': LEVEL}
for name, order in (('consecutive trees per wave (the kernel)', ident), ('sorted by running mean depth', sorted_order),
                    ('sorted by the depth of THIS simulation (perfect predictor)', oracle_order)):
  crit, wait = evaluate(order)
  out[name] = {'critical_levels_per_simulation': crit, 'barrier_wait_levels': wait, 'barrier_wait_cycles': wait * LEVEL,
               'tree_step_critical_path_cycles': crit * LEVEL}
out['reading'] = ('the critical path (deepest of the 16 trees) is the same for every assignment; sorting only lets the OTHER waves '
                  'finish earlier, i.e. the measured barrier wait goes UP while the simulation takes exactly as long: nothing to gain '
                  'from the assignment -- the wait is the price of lock-step trees, and only a shallower deepest tree or a cheaper level '
                  'shortens it')
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
  json.dump(out, open(sys.argv[1], 'w'), indent=1)
