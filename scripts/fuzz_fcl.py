"""Randomised sweep of the native learner step (mz_fcl_step, csrc/mz_fcl.hip.h) over its shape space: batch 16..160 (multiples
of 16), K 1..7, 1..14 actions, 1..300 observation features, supports of 3..64 bins, with / without the target transform,
float32 / float64 importance weights, int32 / int64 actions -- every parameter's gradient, the priority refresh and the loss
sums against PyTorch autograd on the same parameters and batch with the native run's ReLU patterns (tests/test_learner.py:
masked_reference -- every entry within 2e-5 of its tensor's scale).
usage: fuzz_fcl.py [configurations] [seed]"""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import model_based_rl_amd
from model_based_rl_amd.config import make_config
from model_based_rl_amd.learners import Learner, _NativeFC, _GraphedUpdate
from tests.test_learner import _random_batch, Sink, grads_close, masked_reference

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
tmp = tempfile.mkdtemp()
t0 = time.time()
worst_all = 0.0
for it in range(n_cfg):
  bs = 16 * int(rng.integers(1, 11)); K = int(rng.integers(1, 8)); A = int(rng.integers(1, 15)); O = int(rng.integers(1, 301))
  vlo, vhi = -int(rng.integers(1, 32)), int(rng.integers(1, 33)); rlo, rhi = -int(rng.integers(1, 32)), int(rng.integers(1, 33))
  ntt = bool(rng.integers(0, 2))
  cfg = make_config(['--environment', 'LunarLander-v2', '--seed', str(it), '--batch_size', str(bs), '--num_unroll_steps', str(K),
                     '--use_gpu_for', 'actors', 'learner', '--runs_dir', os.path.join(tmp, 'r%d' % it), '--run_tag', 'x', '--no_tune_gemms',
                     '--value_support', str(vlo), str(vhi), '--reward_support', str(rlo), str(rhi)] + (['--no_target_transform'] if ntt else []))
  cfg.obs_space, cfg.action_space = (O,), A
  sink = Sink()
  learner = Learner(cfg, sink, sink)
  net = learner.network
  with torch.no_grad():
    g = torch.Generator().manual_seed(100 + it)
    for p in net.parameters():
      p.add_((torch.randn(p.shape, generator=g) * 0.05).to(p.device))
  host = _random_batch(rng, bs, K, O, A, lo=vlo - 3, hi=vhi + 3)
  if rng.integers(0, 2): host['w'] = host['w'].astype(np.float32)
  if rng.integers(0, 2): host['act'] = host['act'].astype(np.int32)
  assert _NativeFC.eligible(learner, host), (bs, K, A, O)
  dev = [torch.from_numpy(host[k]).to(learner.device) for k in _GraphedUpdate.ORDER]
  nat = _NativeFC(learner, host)
  learner._loss_dev.zero_()
  got_errors = nat.step(*dev, no_update=True)
  got = nat.grad()
  got_l = learner._loss_dev.tolist()
  try:
    want, new_errors, losses = masked_reference(learner, nat, dev)
    worst = grads_close(got, want)
  except AssertionError as e:
    raise AssertionError((it, bs, K, A, O) + tuple(e.args))
  assert (got_errors - new_errors).abs().max().item() <= 2e-4 * (1 + new_errors.abs().max().item()), (it, bs, K, A, O)
  for a_, b_ in zip(got_l, losses):
    assert abs(a_ - b_) <= 1e-5 * max(1.0, abs(b_)), (it, bs, K, A, O)
  worst_all = max(worst_all, worst)
  nat.close()
print('%d configurations agree with autograd (worst relative gradient difference of any entry %.2g), %.0f s' % (n_cfg, worst_all, time.time() - t0))
