"""kernel development: GPU time of ONE native learner update (mz_fcl_step: csrc/mz_fcl.hip.h) on a random device-resident
batch, HIP events around N back-to-back steps, for a list of batch sizes -- no replay, no host loop, nothing but the step's
launches.  LunarLander shapes (obs 8, 4 actions, K = 5) unless MZ_SWEEP_ENV says otherwise.
usage: fcl_step_sweep.py [batches, comma separated] [steps per batch] [out.json]
(under `rocprofv3 --kernel-trace --stats` with ONE batch size it gives that batch's per-kernel durations)"""
import json, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from bench_learner import step_flop, F32_MFMA_TFLOPS

SHAPES = {'lunar': ('LunarLander-v2', 8, 4, []), 'pong': ('Pong-ramNoFrameskip-v4', 128, 6, []),
          'ttt': ('TicTacToe', 9, 9, ['--no_target_transform'])}


class Sink(object):
  def update(self, idxs, errors): pass
  def store_weights(self, w, step): pass
  def get_stats(self, key=None): return {0: 3}
  def add_initial_throughput(self, f, g): pass
  def get_throughput(self): return {'frames': 0, 'games': 0}


def batch_of(rng, bs, K, O, A):
  obs = rng.standard_normal((bs, O)).astype(np.float32)
  act = rng.integers(0, A, size=(bs, K)).astype(np.int64)
  t_rew = rng.uniform(-3, 3, size=(bs, K + 1)).astype(np.float32)
  t_val = rng.uniform(-18, 18, size=(bs, K + 1)).astype(np.float32)
  t_pol = rng.dirichlet([0.5] * A, size=(bs, K + 1)).astype(np.float32)
  w = rng.uniform(0.2, 1.0, size=bs)
  return {'obs': obs, 'act': act, 't_rew': t_rew, 't_val': t_val, 't_pol': t_pol, 'w': w}


def measure(bs, steps, shape='lunar', K=5, extra=()):
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.learners import Learner, _NativeFC, _GraphedUpdate
  env, O, A, flags = SHAPES[shape]
  cfg = make_config(['--environment', env, '--seed', '3', '--batch_size', str(bs), '--num_unroll_steps', str(K), '--use_gpu_for', 'actors', 'learner',
                     '--runs_dir', os.path.join(tempfile.gettempdir(), 'mz_sweep'), '--run_tag', 'x', '--no_tune_gemms'] + flags + list(extra))
  cfg.obs_space, cfg.action_space = (O,), A
  sink = Sink()
  learner = Learner(cfg, sink, sink)
  host = batch_of(np.random.default_rng(bs), bs, K, O, A)
  dev = [torch.from_numpy(host[k]).to(learner.device) for k in _GraphedUpdate.ORDER]
  if os.environ.get('MZ_SWEEP_PINNED', '0')[:1] == '1':      # the batch in pinned host memory, read by the kernels over PCIe (what mz_fcl_run does)
    dev = [torch.from_numpy(host[k]).pin_memory() for k in _GraphedUpdate.ORDER]
  nat = _NativeFC(learner, host)
  for _ in range(10):
    nat.step(*dev)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  best = None
  for _ in range(3):
    torch.cuda.synchronize(); e0.record()
    evs = [torch.cuda.Event() for _ in range(6)] if os.environ.get('MZ_SWEEP_EVENTS', '0')[:1] == '1' else None      # (an event record behind every step, as mz_fcl_run has)
    for i in range(steps):
      nat.step(*dev)
      if evs: evs[i % 6].record()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / steps
    best = us if best is None else min(best, us)
  Sv, Sr = cfg.value_support_max - cfg.value_support_min + 1, cfg.reward_support_max - cfg.reward_support_min + 1
  flop = step_flop(bs, K, O, A, Sv, Sr)
  nat.close()
  return {'batch': bs, 'us_per_update': best, 'samples_per_s_gpu_only': bs / (best * 1e-6), 'flop_per_update': flop,
          'frac_f32_mfma_peak': flop / (best * 1e-6) / 1e12 / F32_MFMA_TFLOPS}


if __name__ == '__main__':
  batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '256,512,1024,2048,4096').split(',')]
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
  shape = os.environ.get('MZ_SWEEP_ENV', 'lunar')
  extra = os.environ.get('MZ_SWEEP_FLAGS', '').split()
  out = [measure(b, steps, shape, extra=extra) for b in batches]
  for r in out:
    print('batch %5d  %8.1f us / update   %6.2f M samples/s   frac %.3f' % (r['batch'], r['us_per_update'], r['samples_per_s_gpu_only'] / 1e6, r['frac_f32_mfma_peak']))
  if len(sys.argv) > 3:
    json.dump({'shape': shape, 'what': 'mz_fcl_step on a device-resident batch, HIP events around %d steps (best of 3)' % steps, 'sweep': out}, open(sys.argv[3], 'w'), indent=1)
