"""Secondary bench line (never the headline): the learner step of SURVEY.md s8 row f2 (reference learners.py:164-230) -- FCNetwork,
LunarLander shapes, batch 256, K = 5 unroll, AdamW -- as `Learner.learn` runs it: batches from the native replay sampled ahead
(replay_buffer.sample_batch_arrays through learners._BatchSource), the update as the five HIP launches of mz_fcl_update
(csrc/mz_fcl.hip.h), priority refreshes one update behind.  Called by `bench.py --workload learner`.

`value` = updates per second of that loop (host-bound today); `roofline` prices the update's GPU time (HIP events around 50
back-to-back updates, copies included) against the f32 MFMA peak with the step's algorithmic FLOP: forward
bs x (2 x layer products over the unroll) and twice that for the backward pass.  `torch_graph` is the same loop with the PyTorch
step captured in one hipGraph (--no_native_learner)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense f32 matrix peak


def step_flop(bs, K, O, A, Sv, Sr, H=50, F=512):
  KD = H + A
  fwd = 2 * (O * F + F * H) + K * 2 * (KD * F + F * H) + (K + 1) * 2 * (H * F + F * Sv) + (K + 1) * 2 * (H * F + F * A) + K * 2 * (KD * F + F * Sr)
  return 3 * bs * fwd


def main(args):
  if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    raise SystemExit('--workload learner is a one-GPU secondary line')
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'scripts'))
  import learner_graph_speed as ls
  updates = args.steps if args.steps != 512 else 1000
  out = {}
  for name, extra in (('native', []), ('torch_graph', ['--no_native_learner'])):
    cfg, storage, replay, learner = ls.setup(extra)
    ls.loop(learner, replay, max(30, args.warmup if args.warmup != 64 else 30))
    runs = [ls.loop(learner, replay, updates) for _ in range(max(1, args.runs))]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if learner._native is not None:
      host = learner._host_batch(replay.sample_batch_arrays())[0]
      run = lambda: learner._native.launch(host)
    else:
      run = learner._graph.graph.replay
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    v = np.array([r['updates_per_second'] for r in runs])
    out[name] = {'updates_per_second': float(v.mean()), 'std': float(v.std()), 'runs': v.tolist(), 'gpu_ms_per_update': e0.elapsed_time(e1) / 50,
                 'native_step': learner._native is not None, 'replay_frames': replay.size(), 'cfg': cfg}
  cfg = out['native'].pop('cfg'); out['torch_graph'].pop('cfg')
  bs, K, A = cfg.batch_size, cfg.num_unroll_steps, cfg.action_space
  O = int(np.prod(cfg.obs_space))
  Sv, Sr = cfg.value_support_max - cfg.value_support_min + 1, cfg.reward_support_max - cfg.reward_support_min + 1
  flop = step_flop(bs, K, O, A, Sv, Sr)
  n = out['native']
  achieved = flop / (n['gpu_ms_per_update'] * 1e-3) / 1e12
  line = {'metric': 'learner_updates_per_second', 'value': n['updates_per_second'], 'unit': 'updates/s', 'n_gpus': 1, 'steps': updates, 'warmup': 30,
          'ms_per_step': 1e3 / n['updates_per_second'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
          'data': 'synthetic self-play records in the native replay (%d frames), random-init weights' % n['replay_frames'],
          'config': {'workload': 'learner step, FCNetwork LunarLander shapes (obs %d, actions %d), batch %d, K = %d unroll, AdamW; '
                                 'Learner.learn\'s loop: sample ahead, mz_fcl_update, refresh one update behind' % (O, A, bs, K),
                     'runs': '%d x %d updates: mean +- std' % (len(n['runs']), updates)},
          'runs': {'mean': n['updates_per_second'], 'std': n['std'], 'values': n['runs']},
          'roofline': {'bound': 'mfma', 'kernel': 'mz_fcl_update (k_fcl_chain_fwd4, k_fcl_heads, k_fcl_chain_bwd4, k_fcl_dw, k_fcl_adam + 2 copies)',
                       'achieved': achieved, 'peak': F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / F32_MFMA_TFLOPS, 'traffic': None,
                       'flop_per_update': flop, 'us_per_update': 1e3 * n['gpu_ms_per_update'],
                       'note': 'latency- / launch-bound: dependent phases of a few microseconds; the loop itself is host-bound (value < 1 / us_per_update)'},
          'torch_graph': {k: out['torch_graph'][k] for k in ('updates_per_second', 'std', 'gpu_ms_per_update')},
          'secondary': True}
  print(json.dumps(line), flush=True)
  return 0
