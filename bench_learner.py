"""Secondary bench line (never the headline): the learner step of SURVEY.md s8 row f2 (reference learners.py:164-230) -- FCNetwork,
LunarLander shapes, batch 256, K = 5 unroll, AdamW -- as `Learner.learn` runs it: batches from the native replay sampled ahead
(replay_buffer.sample_batch_arrays through learners._BatchSource), the update as the HIP launches of mz_fcl_update
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


def sweep_point(bs, updates, runs=3):
  """one batch size of the sweep, through the product's loop: Learner.launch(updates) on the handles train.launch builds (native step
  + native loop); GPU time per update from HIP events around 100+ updates of mz_fcl_run; host microseconds per update inside the call"""
  from model_based_rl_amd import rayshim as ray
  cfg, storage, replay, learner = setup(['--batch_size', str(bs)])      # (the replay's threads: the product's default, 8 from batch 1024 up)
  ray.get(learner.launch.remote(30))
  vals = []
  for _ in range(max(1, runs)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ray.get(learner.launch.remote(updates))
    torch.cuda.synchronize()
    vals.append(updates / (time.perf_counter() - t0))
  lrn = learner._obj
  assert lrn._native is not None and getattr(lrn, 'native_loop_updates', 0), 'the sweep measures the native step and loop'
  rep = getattr(replay, '_obj', replay)
  lrn.flush_priorities()
  n_ev = max(100, min(300, updates))
  f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); f0.record()
  lrn._native.run(rep, n_ev)
  lrn._native.flush()
  f1.record(); torch.cuda.synchronize()
  us = 1e3 * f0.elapsed_time(f1) / n_ev
  host = lrn._native.run_stats()
  K, A = cfg.num_unroll_steps, cfg.action_space
  O = int(np.prod(cfg.obs_space))
  Sv, Sr = cfg.value_support_max - cfg.value_support_min + 1, cfg.reward_support_max - cfg.reward_support_min + 1
  flop = step_flop(bs, K, O, A, Sv, Sr)
  v = np.array(vals)
  mid = float(np.median(v))                           # (median of the timed calls: one slow call on a shared box does not move the point)
  point = {'batch': bs, 'updates_per_s': mid, 'std': float(v.std()), 'calls': [float(x) for x in v], 'samples_per_s': mid * bs,
           'timed_updates': updates, 'us_per_update_gpu': us, 'frac': flop / (us * 1e-6) / 1e12 / F32_MFMA_TFLOPS,
           'frac_of_wall': flop * mid / 1e12 / F32_MFMA_TFLOPS, 'flop_per_update': flop,
           'host_us_per_update': {k: host[k] for k in ('sample_us', 'refresh_us', 'launch_us', 'wait_us', 'call_us')},
           'replay_sampling_threads': int(getattr(rep, 'ingest_threads', 0) or 0)}
  if lrn._native is not None:
    lrn._native.close()
  return point


def setup(extra, updates_hint=1000):
  """storage, replay and learner behind actor handles, wired as train.launch wires them (reference train.py:62-78); the replay is
  filled by the product's own Actor (1024 environments, 128 moves of the device loop) before the learner starts"""
  import tempfile
  from model_based_rl_amd import rayshim as ray
  from model_based_rl_amd import train
  from model_based_rl_amd.actors import Actor
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.learners import Learner
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  from model_based_rl_amd.shared_storage import SharedStorage
  cfg = make_config(['--environment', 'LunarLander-v2', '--num_simulations', '30', '--seed', '0', '--num_envs', '1024', '--episode_length', '32',
                     '--window_size', '200000', '--batch_size', '256', '--stored_before_train', '50000', '--use_gpu_for', 'actors', 'learner',
                     '--runs_dir', os.path.join(tempfile.gettempdir(), 'mz_bench_runs'), '--run_tag', 'learner_%d' % os.getpid()] + extra)
  storage = ray.remote(SharedStorage).remote(cfg)
  replay = ray.remote(PrioritizedReplay).remote(cfg)
  train.publish_initial_weights(cfg, storage)
  actor = Actor(0, cfg, storage, replay)
  actor.launch(128)
  actor.close()
  actor.engine.close()
  learner = ray.remote(Learner).remote(cfg, storage, replay)
  return cfg, storage, replay, learner


def main(args):
  if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    raise SystemExit('--workload learner is a one-GPU secondary line')
  import contextlib
  sys.path.insert(0, ROOT)
  from model_based_rl_amd import rayshim as ray
  updates = args.steps if args.steps != 512 else 1000
  warm = max(30, args.warmup if args.warmup != 64 else 30)
  out = {}
  with contextlib.redirect_stdout(sys.stderr):
    variants = (('native', []), ('torch_graph', ['--no_native_learner']))
    if os.environ.get('MZ_LEARNER_ONLY'):          # (profiling: one variant)
      variants = tuple(v for v in variants if v[0] in os.environ['MZ_LEARNER_ONLY'].split(','))
    for name, extra in variants:
      cfg, storage, replay, learner = setup(extra)
      ray.get(learner.launch.remote(warm))          # Learner.launch -> Learner.learn(max_steps) (learners.py:115-153)
      runs = []
      for _ in range(max(1, args.runs)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ray.get(learner.launch.remote(updates))     # the timed region: ONE call of the product's entry point
        torch.cuda.synchronize()
        runs.append(updates / (time.perf_counter() - t0))
      lrn = learner._obj                            # (the object behind the handle: GPU time of one update, for the roofline)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      if lrn._native is not None:
        host = lrn._host_batch(ray.get(replay.sample_batch_arrays.remote()))[0]
        run = lambda: lrn._native.launch(host)
      else:
        run = lrn._graph.graph.replay
      torch.cuda.synchronize(); e0.record()
      for _ in range(50): run()
      e1.record(); torch.cuda.synchronize()
      loop_ms = None
      if lrn._native is not None and getattr(lrn, 'native_loop_updates', 0):
        # GPU time per update of the native loop itself (mz_fcl_run: the kernels read the batch from pinned staging, no copies):
        # events around 300 updates of the loop the timed runs above went through
        rep = getattr(replay, '_obj', replay)
        lrn.flush_priorities()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); f0.record()
        lrn._native.run(rep, 300)
        lrn._native.flush()
        f1.record(); torch.cuda.synchronize()
        loop_ms = f0.elapsed_time(f1) / 300
      v = np.array(runs)
      out[name] = {'updates_per_second': float(v.mean()), 'std': float(v.std()), 'runs': v.tolist(), 'gpu_ms_per_update': e0.elapsed_time(e1) / 50,
                   'native_loop_gpu_ms_per_update': loop_ms,
                   'native_step': lrn._native is not None, 'native_loop': bool(getattr(lrn, 'native_loop_updates', 0)),
                   'replay_frames': ray.get(replay.size.remote()), 'training_step': lrn.training_step,
                   'last_throughput': lrn.get_last_throughput(), 'cfg': cfg,
                   'native_loop_host_us_per_update': lrn._native.run_stats() if lrn._native is not None else None}
  cfg = out['native'].pop('cfg')
  out.setdefault('torch_graph', {'updates_per_second': None, 'std': None, 'gpu_ms_per_update': None, 'cfg': None}).pop('cfg')
  bs, K, A = cfg.batch_size, cfg.num_unroll_steps, cfg.action_space
  O = int(np.prod(cfg.obs_space))
  Sv, Sr = cfg.value_support_max - cfg.value_support_min + 1, cfg.reward_support_max - cfg.reward_support_min + 1
  flop = step_flop(bs, K, O, A, Sv, Sr)
  n = out['native']
  us = 1e3 * (n['native_loop_gpu_ms_per_update'] or n['gpu_ms_per_update'])      # the loop the value went through
  achieved = flop / (us * 1e-6) / 1e12
  line = {'metric': 'learner_updates_per_second', 'value': n['updates_per_second'], 'unit': 'updates/s', 'n_gpus': 1, 'steps': updates, 'warmup': 30,
          'ms_per_step': 1e3 / n['updates_per_second'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
          'data': 'synthetic self-play records in the native replay (%d frames), random-init weights' % n['replay_frames'],
          'config': {'workload': 'learner step, FCNetwork LunarLander shapes (obs %d, actions %d), batch %d, K = %d unroll, AdamW; '
                                 'timed call: Learner.launch(%d) = Learner.learn on the replay / storage handles train.launch builds '
                                 '(send_weights every %d, save_state every %d, loss + throughput scalars every %d updates)'
                                 % (O, A, bs, K, updates, cfg.send_weights_frequency, cfg.save_state_frequency, cfg.learner_log_frequency),
                     'native_loop': n['native_loop'], 'native_loop_host_us_per_update': n['native_loop_host_us_per_update'],
                     'runs': '%d x %d updates: mean +- std' % (len(n['runs']), updates)},
          'runs': {'mean': n['updates_per_second'], 'std': n['std'], 'values': n['runs']},
          'roofline': {'bound': 'mfma', 'kernel': 'mz_fcl_run / mz_fcl_update (k_fcl_fb, k_fcl_dwa at batch <= 256)',
                       'achieved': achieved, 'peak': F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / F32_MFMA_TFLOPS, 'traffic': None,
                       'flop_per_update': flop, 'us_per_update': us,
                       'us_per_update_single_calls': 1e3 * n['gpu_ms_per_update'],
                       'note': 'latency-bound: dependent phases of a few microseconds (two launches per update at batch 256); us_per_update = HIP events around '
                               '300 updates of the native loop (mz_fcl_run), us_per_update_single_calls = the same step driven one mz_fcl_update call '
                               'at a time (pinned staging + two copies)'},
          'torch_graph': {k: out['torch_graph'][k] for k in ('updates_per_second', 'std', 'gpu_ms_per_update')},
          'secondary': True}
  # the batch sweep: the same loop at batch 256 ... 4096 -- updates/s, samples/s, roofline fraction (GPU time
  # per update on HIP events), host microseconds per update (sampling runs on the replay's thread pool from batch 512 up)
  if not os.environ.get('MZ_LEARNER_NO_SWEEP'):
    batches = [int(x) for x in (getattr(args, 'batch', None) or '256,512,1024,2048,4096').split(',')]
    with contextlib.redirect_stdout(sys.stderr):
      line['batch_sweep'] = [sweep_point(b, max(200, min(updates, updates * 512 // b)), runs=3) for b in batches]
    line['batch_sweep_what'] = ('Learner.launch(n) per point on the handles train.launch builds; frac = algorithmic FLOP / (HIP-event time per update '
                                'of mz_fcl_run) / %.1f TFLOP/s; frac_of_wall = the same with the wall-clock rate' % F32_MFMA_TFLOPS)
  print(json.dumps(line), flush=True)
  return 0
