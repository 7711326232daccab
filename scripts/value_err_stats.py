"""value / reward scalar error of the fused kernel's network against the oracle after one simulation (4096 rows)"""
import sys, numpy as np
sys.path.insert(0, '.')
from model_based_rl_amd.engine import Engine
from oracle import oracle as orc
from tests.test_gpu_net import _random_weights
O, A, B = 8, 4, 4096
w = _random_weights(O, A, 23)
eng = Engine(B, O, A, 4, seed=3); eng.set_weights(w)
obs = np.random.RandomState(9).standard_normal((B, O)).astype(np.float32) * 2
eng.initial_inference(obs); eng.root_prepare(None, None, None, device_rng=True, move=0); eng.search(1)
t = eng.export_tree(hidden=True)
child = np.array([int(np.flatnonzero(t['N'][b, 1:1 + A])[0]) for b in range(B)], np.int32)
h1o, r1o, v1o, _ = orc.FCNet(w, O, A).recurrent(t['hidden'][:, 0, :], child)
idx = np.arange(B)
for name, got, ref in (('hidden', t['hidden'][:, 1, :], h1o), ('reward', t['R'][idx, 1 + child], r1o), ('value', t['W'][idx, 1 + child].astype(np.float32), v1o)):
  d = np.abs(np.asarray(got, np.float64) - ref)
  d = d.reshape(B, -1).max(1)
  print('%-7s max %.2e  mean %.2e  rows <= 1e-5: %.2f %%  <= 1e-6: %.2f %%' % (name, d.max(), d.mean(), 100 * (d <= 1e-5).mean(), 100 * (d <= 1e-6).mean()))
