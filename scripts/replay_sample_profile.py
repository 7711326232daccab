"""development aid: where the time of PrioritizedReplay.sample_batch_arrays / update goes on this host (a 131 k-frame replay
of synthetic LunarLander-shaped records, batch 256): the native calls with fresh draws (cache-cold) and the Python around them"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import model_based_rl_amd
from model_based_rl_amd.config import make_config
from model_based_rl_amd.replay_buffer import PrioritizedReplay, _p

cfg = make_config(['--environment', 'LunarLander-v2', '--window_size', '200000', '--batch_size', '256', '--use_gpu_for', 'actors'])
rp = PrioritizedReplay(cfg)
O, A, B, n = 8, 4, 512, 64
rec = O + A + 10
rng = np.random.default_rng(0)
for it in range(4):
  r = np.zeros((n, B, rec), np.float32)
  r[..., :O] = rng.standard_normal((n, B, O))
  r[..., O:O + A] = rng.dirichlet([1] * A, size=(n, B)).astype(np.float32)
  r[..., O + A:O + A + 2] = np.ascontiguousarray(rng.standard_normal((n, B))).view(np.float32).reshape(n, B, 2)
  r[..., O + A + 2:O + A + 4] = np.ascontiguousarray(np.abs(rng.standard_normal((n, B))) + 0.1).view(np.float32).reshape(n, B, 2)
  r[..., O + A + 4] = rng.uniform(-1, 1, (n, B))
  ints = r[..., O + A + 5:].view(np.int32)
  ints[..., 0] = rng.integers(0, A, (n, B)); ints[-1, :, 1] = 1; ints[..., 2] = np.arange(n)[:, None]; ints[..., 3] = np.arange(B)[None, :]; ints[..., 4] = it
  rp.ingest_records(r, n, B)
bs, K = 256, 5
def timed(f, reps=500):
  for _ in range(50): f()
  t = time.perf_counter()
  for _ in range(reps): f()
  return (time.perf_counter() - t) / reps * 1e6
print('replay of %d frames' % rp.size())
print('sample_batch_arrays            %.1f us' % timed(rp.sample_batch_arrays))
print('sample_batch (reference form)  %.1f us' % timed(rp.sample_batch))
b, idxs = rp.sample_batch_arrays()
err = rng.standard_normal(bs)
print('update (256 leaves)            %.1f us' % timed(lambda: rp.update(idxs, err)))
total = rp.tree.total_priority; seg = total / bs; i = np.arange(bs, dtype=np.float64)
obs = np.zeros((bs, O), np.float32); actions = np.zeros((bs, K), np.int32)
t_rew = np.zeros((bs, K + 1), np.float32); t_val = np.zeros((bs, K + 1), np.float32); t_pol = np.zeros((bs, K + 1, A), np.float32)
ix = np.zeros(bs, np.int64); pri = np.zeros(bs, np.float64)
acc = 0.0
for _ in range(400):
  draws = seg * i + (seg * (i + 1) - seg * i) * np.random.random(bs)
  args = (rp._h, _p(draws), bs, _p(obs), _p(actions), _p(t_rew), _p(t_val), _p(t_pol), _p(ix), _p(pri))
  t = time.perf_counter(); rp.lib.mzr_sample_batch(*args); acc += time.perf_counter() - t
print('  native mzr_sample_batch, fresh draws   %.1f us' % (acc / 400 * 1e6))
acc = 0.0
for _ in range(400):
  ii = np.random.randint(rp.tree.max_capacity - 1, rp.tree.max_capacity - 1 + rp.tree.num_memories, size=bs).astype(np.int64) if hasattr(rp.tree, 'max_capacity') else idxs
  pp = np.random.random(bs) + 0.01
  t = time.perf_counter(); rp.tree.update(ii, pp); acc += time.perf_counter() - t
print('  SumTree.update, fresh leaves            %.1f us' % (acc / 400 * 1e6))
