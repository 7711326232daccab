#!/usr/bin/env python3
"""One-off evidence for the clamp ReLU (csrc/mz_fused.hip.h, "ReLU as a clamp"): the SAME searches through two builds of
libmz_hip.so -- one whose search kernel applies nn.ReLU with v_max on an unscaled weight stream (the build before the
change, MZ_HIP_LIB=<old>), one with the power-of-two scaled stream and the [0, 1] clamp -- compared bit for bit: visit
counts, value sums (float64), priors, rewards, MinMaxStats, root values.
  usage: relu_clamp_check.py <old lib> <new lib> [envs]      (runs each half in a child process)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, types
sys.path.insert(0, %r)
import numpy as np, torch
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
B = int(sys.argv[1])
out = {}
for name, O, A, sims, gain in (('lunar', 8, 4, 30, 1.0), ('pong', 128, 6, 50, 1.0), ('lunar_x40', 8, 4, 30, 40.0)):
  torch.manual_seed(7)
  sd = FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).state_dict()
  if gain != 1.0:       # larger activations: another power of two gets chosen
    for k in sd:
      if k.endswith('fc1.weight') or k.endswith('fc1.bias'):
        sd[k] = sd[k] * gain
  rng = np.random.RandomState(11)
  obs = rng.standard_normal((B, O)).astype(np.float32)
  noise = rng.dirichlet([0.25] * A, size=B)
  eng = Engine(B, O, A, sims)
  eng.set_weights(flatten_weights(sd))
  out[name + '_scale'] = np.asarray(eng.weight_scale()) if hasattr(eng, 'weight_scale') else np.zeros(4, np.float32)
  eng.initial_inference(obs)
  eng.root_prepare(None, None, noise)
  eng.search()
  fin = eng.finalize(np.ones(B), rng.random_sample(B))
  for k, v in fin.items():
    out[name + '_fin_' + k] = v.cpu().numpy()
  t = eng.export_tree(hidden=True)
  for k, v in t.items():
    out[name + '_' + k] = np.asarray(v)
  eng.close()
# the whole-moves launch: 16 moves of the self-play loop, records as they are drained
for name, O, A, sims in (('lunar', 8, 4, 30), ('pong', 128, 6, 50)):
  torch.manual_seed(7)
  sd = FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).state_dict()
  eng = Engine(B, O, A, sims, seed=99)
  eng.set_weights(flatten_weights(sd))
  if O == 128:
    eng.selfplay_set_obs(uint8_obs=True, obs_min=np.zeros(O, np.float32), obs_range=np.full(O, 255, np.float32))
  eng.selfplay_reset(64, 1.0, stagger=True)
  eng.selfplay_steps(16)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  out[name + '_records'] = buf[:n].numpy().copy()
  eng.close()
np.savez(sys.argv[2], **out)
''' % ROOT


def main():
  old, new = sys.argv[1], sys.argv[2]
  B = sys.argv[3] if len(sys.argv) > 3 else '4096'
  res = {}
  for tag, lib in (('old', old), ('new', new)):
    subprocess.check_call([sys.executable, '-c', CHILD, B, '/tmp/relu_%s.npz' % tag], env=dict(os.environ, MZ_HIP_LIB=lib))
  import numpy as np
  a, b = np.load('/tmp/relu_old.npz'), np.load('/tmp/relu_new.npz')
  report = {'envs': int(B), 'old': old, 'new': new, 'arrays': {}}
  ok = True
  for k in a.files:
    if k.endswith('_scale'):
      report[k] = [float(x) for x in b[k]]
      continue
    same = bool(np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)))
    report['arrays'][k] = same
    ok &= same
  report['bit_identical'] = ok
  print(json.dumps(report, indent=1))
  return 0 if ok else 1


if __name__ == '__main__':
  sys.exit(main())
