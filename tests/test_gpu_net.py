"""HIP fused FCNetwork inference (f32 MFMA) vs the reference's PyTorch-CPU goldens and the oracle.
Tolerances as stated in tests/test_oracle_net.py (1e-5 on hidden/logits; value/reward 1e-5 with the
reference's own float32 staircase caveat)."""
import os

import numpy as np
import pytest

from tests.test_oracle_net import TOL, scalar_close

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def engine_for(g, B, nt=False, sims=4):
  from model_based_rl_amd.engine import Engine
  from oracle import oracle as orc
  w = orc.load_weights(g)
  eng = Engine(B, int(g['O']), int(g['A']), sims, no_target_transform=nt)
  eng.set_weights({k: v for k, v in w.items()})
  return eng, w


@pytest.mark.parametrize('name', ['g1_net_ttt', 'g1_net_lunar', 'g1_net_pong', 'g1_net_lunar_notransform'])
def test_net_vs_golden(name):
  g = np.load(os.path.join(G, name + '.npz'))
  nt = name.endswith('notransform')
  eng, _ = engine_for(g, 64, nt)
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
