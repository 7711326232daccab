"""SURVEY.md s8 row f1 (replay sampling + n-step targets) inside the driver's GPU record: the host-side parity tests of
tests/test_replay_native.py need no GPU, so the `-m "not gpu"` run is where they normally execute; these wrappers run the
same checks under the gpu marker, so that GPUTEST_rNN re-confirms them on the GPU box's host (where the replay actually
runs beside the device loop) -- plus the parallel ingest against one thread there."""
import os

import pytest

from tests import test_replay_native as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('path', T.FILES, ids=[os.path.basename(f)[:-4] for f in T.FILES])
def test_sample_batch_matches_reference_on_the_gpu_host(path):
  T.test_sample_batch_matches_reference(path)


@pytest.mark.parametrize('path', T.FILES, ids=[os.path.basename(f)[:-4] for f in T.FILES])
def test_save_history_matches_reference_on_the_gpu_host(path):
  T.test_save_history_matches_reference(path)


def test_parallel_ingest_is_bit_identical_on_the_gpu_host():
  T.test_parallel_ingest_is_bit_identical_to_one_thread(40, 500, 3000)


def test_packed_byte_observations_in_ram_records():
  """The -ram- shapes with their 128 byte observations PACKED in the experience record (mz_selfplay_set_obs uint8_obs = 2: four
  bytes per float slot, 48 instead of 144 floats per Pong-ram record) against the unpacked records of the same run: every field
  equal, the observations byte for byte; and a replay with obs_u8 fed the packed chunk samples the batch a plain replay fed the
  unpacked chunk samples (game.py:93-96: History keeps the raw observation; the learner gets float32 back)."""
  import types
  import numpy as np
  import torch
  from oracle import oracle as orc
  from model_based_rl_amd.engine import Engine, records_view
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  B, O, A, sims, moves = 256, 128, 6, 50, 24
  w = orc.load_weights(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g1_net_pong.npz')))
  out = {}
  for packed in (False, True):
    eng = Engine(B, O, A, sims, seed=99)
    eng.set_weights(w)
    eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0], packed=packed)
    eng.selfplay_reset(7, 1.0, stagger=True)
    assert eng.rec_floats == (O // 4 if packed else O) + A + 10
    eng.selfplay_steps(moves)
    buf, n = eng.selfplay_drain()
    torch.cuda.synchronize()
    assert n == moves and buf.shape[-1] == eng.rec_floats
    out[packed] = buf[:n].numpy().copy()
    eng.close()
  a, b = records_view(out[False], O, A), records_view(out[True], O, A, obs_u8=True)
  assert b['obs'].dtype == np.uint8 and np.array_equal(a['obs'], b['obs'].astype(np.float32))
  for k in ('child_visits', 'root_value', 'error', 'reward', 'action', 'done', 'step', 'env_id', 'episode'):
    assert np.array_equal(a[k], b[k]), k
  cfg = dict(batch_size=32, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(O,), action_space=A, window_size=1 << 15, window_step=None,
             num_unroll_steps=5, td_steps=10, max_history_length=500, discount=0.997, seed=0)
  ra, rb = PrioritizedReplay(types.SimpleNamespace(**cfg)), PrioritizedReplay(types.SimpleNamespace(obs_u8=True, **cfg))
  ra.ingest_records(out[False], moves, B)
  rb.ingest_records(out[True], moves, B)
  assert ra.size() == rb.size() > 0 and ra.tree.total_priority == rb.tree.total_priority
  import random
  random.seed(5); np.random.seed(5)
  (oa, aa, (ta, va, pa)), ia, wa = ra.sample_batch()
  random.seed(5); np.random.seed(5)
  (ob, ab, (tb, vb, pb)), ib, wb = rb.sample_batch()
  assert ia == ib and np.array_equal(oa, ob) and aa == ab and np.array_equal(ta, tb) and np.array_equal(va, vb) and np.array_equal(pa, pb)
  assert np.array_equal(wa, wb) and oa.dtype == np.float32
