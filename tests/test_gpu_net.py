"""HIP fused FCNetwork inference (f32 MFMA) vs the reference's PyTorch-CPU goldens and the oracle.
Tolerances as stated in tests/test_oracle_net.py (1e-5 on hidden/logits; value/reward 1e-5 with the
reference's own float32 staircase caveat)."""
import os

import numpy as np
import pytest

from tests.test_oracle_net import TOL, scalar_close

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def engine_for(g, B, nt=False, sims=4, ns=False):
  from model_based_rl_amd.engine import Engine
  from oracle import oracle as orc
  w = orc.load_weights(g)
  eng = Engine(B, int(g['O']), int(g['A']), sims, no_target_transform=nt, no_support=ns)
  eng.set_weights({k: v for k, v in w.items()})
  return eng, w


@pytest.mark.parametrize('name', ['g1_net_ttt', 'g1_net_lunar', 'g1_net_pong', 'g1_net_lunar_notransform',
                                  'g1_net_lunar_nosupport'])
def test_net_vs_golden(name):
  g = np.load(os.path.join(G, name + '.npz'))
  ns = name.endswith('nosupport')
  nt = name.endswith('notransform') or ns
  eng, _ = engine_for(g, 64, nt and not ns, ns=ns)
  eng.initial_inference(g['obs'])
  v, lg, h = [x.cpu().numpy() for x in eng.root_outputs()]
  assert np.abs(h - g['init_hidden']).max() <= TOL
  assert np.abs(lg - g['init_logits']).max() <= TOL
  scalar_close(v, g['init_value'], not nt)
  h2, r2, v2, lg2 = [x.cpu().numpy() for x in eng.recurrent_inference(g['init_hidden'], g['actions'])]
  assert np.abs(h2 - g['rec_hidden']).max() <= TOL
  assert np.abs(lg2 - g['rec_logits']).max() <= TOL
  scalar_close(v2, g['rec_value'], not nt)
  scalar_close(r2, g['rec_reward'], not nt)
  eng.close()


@pytest.mark.parametrize('name,rows', [('g1_net_lunar', 4096), ('g1_net_pong', 1000), ('g1_net_ttt', 37)])
def test_net_vs_oracle_large(name, rows):
  """full batch (and ragged sizes that do not fill the 16-row tiles) against the oracle; inputs scaled up
  so the value/reward heads leave the near-zero regime."""
  from oracle import oracle as orc
  g = np.load(os.path.join(G, name + '.npz'))
  eng, w = engine_for(g, rows)
  O, A = int(g['O']), int(g['A'])
  rng = np.random.RandomState(3)
  wbig = {k: (v * (3.0 if k.endswith('value.weight') or k.endswith('reward.weight') else 1.0)).astype(np.float32)
          for k, v in w.items()}
  eng.set_weights(wbig)
  net = orc.FCNet(wbig, O, A)
  obs = rng.standard_normal((rows, O)).astype(np.float32) * 2
  eng.initial_inference(obs)
  v, lg, h = [x.cpu().numpy() for x in eng.root_outputs()]
  ho, vo, lgo = net.initial(obs)
  assert np.abs(h - ho).max() <= TOL and np.abs(lg - lgo).max() <= TOL
  scalar_close(v, vo)
  act = rng.randint(0, A, rows).astype(np.int32)
  h2, r2, v2, lg2 = [x.cpu().numpy() for x in eng.recurrent_inference(ho, act)]
  h2o, r2o, v2o, lg2o = net.recurrent(ho, act)
  assert np.abs(h2 - h2o).max() <= TOL and np.abs(lg2 - lg2o).max() <= TOL
  scalar_close(v2, v2o)
  scalar_close(r2, r2o)
  assert np.abs(v2o).max() > 0.05     # the heads were actually exercised away from zero
  eng.close()


def test_missing_weights_is_an_error():
  from model_based_rl_amd.engine import Engine
  eng = Engine(16, 8, 4, 4)
  with pytest.raises(RuntimeError, match='weights not set'):
    eng.initial_inference(np.zeros((16, 8), np.float32))
  with pytest.raises(ValueError):
    eng.set_weights(np.zeros(10, np.float32))
  eng.close()


def _random_weights(O, A, seed, scale_heads=3.0):
  import types
  import torch
  from model_based_rl_amd.networks import FCNetwork
  torch.manual_seed(seed)
  cfg = types.SimpleNamespace(value_support=(-15, 15), reward_support=(-15, 15), no_support=False,
                              no_target_transform=False)
  w = {k: v.numpy().copy() for k, v in FCNetwork(O, A, torch.device('cpu'), cfg).state_dict().items()}
  for k in w:
    if k.endswith('value.weight') or k.endswith('reward.weight'):
      w[k] = (w[k] * scale_heads).astype(np.float32)
  return w


@pytest.mark.parametrize('O,A,rows', [(300, 5, 50), (255, 3, 33), (256, 18, 16), (1, 2, 17)])
def test_root_kernel_wide_observations(O, A, rows):
  """the root kernel's run-time first stage: observation widths around and beyond one 256-column LDS chunk
  (obs_dim + 1 bias column = 256 -> exactly one chunk, 257 / 301 -> two), and the 1-column corner."""
  from model_based_rl_amd.engine import Engine
  from oracle import oracle as orc
  w = _random_weights(O, A, 11)
  eng = Engine(rows, O, A, 4)
  eng.set_weights(w)
  obs = np.random.RandomState(5).standard_normal((rows, O)).astype(np.float32)
  eng.initial_inference(obs)
  v, lg, h = [x.cpu().numpy() for x in eng.root_outputs()]
  ho, vo, lgo = orc.FCNet(w, O, A).initial(obs)
  assert np.abs(h - ho).max() <= TOL and np.abs(lg - lgo).max() <= TOL
  scalar_close(v, vo)
  eng.close()


@pytest.mark.parametrize('O,A,B', [(8, 4, 4096), (128, 6, 256), (9, 9, 100), (8, 18, 64)])
def test_fused_search_network_one_simulation(O, A, B):
  """The network inside the fused search kernel, without compounding: after ONE simulation the pool holds the
  hidden state of the expansion, the expanded child its reward, and its value_sum is the leaf value.  Against the
  oracle's recurrent inference on the device's own root hidden state and the action the descent took."""
  from model_based_rl_amd.engine import Engine
  from oracle import oracle as orc
  w = _random_weights(O, A, 23)
  eng = Engine(B, O, A, 4, seed=3)
  eng.set_weights(w)
  obs = np.random.RandomState(9).standard_normal((B, O)).astype(np.float32) * 2
  eng.initial_inference(obs)
  eng.root_prepare(None, None, None, device_rng=True, move=0)
  eng.search(1)
  t = eng.export_tree(hidden=True)
  child = np.array([int(np.flatnonzero(t['N'][b, 1:1 + A])[0]) for b in range(B)], np.int32)
  assert np.all(t['N'][:, 1:1 + A].sum(1) == 1)
  h1o, r1o, v1o, _ = orc.FCNet(w, O, A).recurrent(t['hidden'][:, 0, :], child)
  assert np.abs(t['hidden'][:, 1, :] - h1o).max() <= TOL
  idx = np.arange(B)
  scalar_close(t['R'][idx, 1 + child], r1o)
  scalar_close(t['W'][idx, 1 + child].astype(np.float32), v1o)
  assert np.abs(v1o).max() > 0.05 and np.abs(r1o).max() > 0.05
  eng.close()
