"""Secondary bench line (never the headline): BASELINE.json configs[4] on one GPU -- Breakout image shapes
([4, 96, 96] uint8 frames, 4 actions), MuZeroNetwork (23.4 M parameters, reference networks.py:498-555) at
num_simulations=50, 4096 self-play environments per GPU (env-steps/s by env count, MI355X: 512: 1.49 k, 1024: 1.97 k,
2048: 2.34 k, 4096: 2.47 k -- MIOpen's convolutions want the rows).  The network runs through PyTorch-ROCm / MIOpen in float32
(no hand-written conv kernels); the search runs through the engine's external-inference entry
points (mz_select / mz_expand_backup) with the hidden states in a device-resident pool -- no host synchronisation in
the simulation loop (model-based-rl_amd/torch_search.py).  Called by bench.py --workload breakout.
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_F32_MFMA_TFLOPS = 157.3


def muzero_flops(A, S=31):
  """algorithmic FLOP of one recurrent_inference row / one initial_inference row of MuZeroNetwork"""
  conv = lambda cin, cout, hw: 2 * cin * cout * 9 * hw
  block = lambda c, hw: 2 * conv(c, c, hw)
  fc = lambda i, o: 2 * i * o
  recurrent = conv(129, 128, 36) + 16 * block(128, 36) + fc(4608, 512) + fc(512, S)          # dynamics
  prediction = 16 * block(128, 36) + 2 * fc(4608, 512) + fc(512, S) + fc(512, A)
  representation = (conv(4, 64, 48 * 48) + 2 * block(64, 48 * 48) + conv(64, 128, 24 * 24) + 3 * block(128, 24 * 24) +
                    3 * block(128, 12 * 12) + 16 * block(128, 36))
  return recurrent + prediction, representation + prediction


def main(args):
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world > 1:
    raise SystemExit('--workload breakout is a one-GPU secondary line')
  device = torch.device('cuda', 0)
  torch.cuda.set_device(device)
  sys.path.insert(0, ROOT)
  from model_based_rl_amd.networks import MuZeroNetwork
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  from model_based_rl_amd.torch_search import TorchSelfplay

  B = args.envs or 4096
  A, SIMS, T, OBS = 4, 50, 64, (4, 96, 96)
  steps = args.steps if args.steps != 512 else 20            # (bench.py's FC default would be ~14 minutes here)
  warmup = args.warmup if args.warmup != 64 else 2
  cfg = types.SimpleNamespace(action_space=A, num_simulations=SIMS, two_players=False, known_bounds=(None, None),
                              discount=0.997, pb_c_base=19652, pb_c_init=1.25, init_value_score=0.0,
                              root_dirichlet_alpha=0.25, root_exploration_fraction=0.25, obs_space=OBS, episode_length=T,
                              seed=0, batch_size=512, obs_u8=True, epsilon=0.01, alpha=1.0, beta=1.0, window_size=4096,
                              window_step=None, num_unroll_steps=5, td_steps=10, max_history_length=500)
  if os.environ.get('MZ_MIOPEN_FIND', '0')[:1] == '1':
    torch.backends.cudnn.benchmark = True
  torch.manual_seed(0)
  net = MuZeroNetwork(OBS[0], A, device, types.SimpleNamespace()).eval()
  norm = (torch.zeros(1, device=device), torch.full((1,), 255.0, device=device))      # --norm_obs --obs_range 0 255
  sp = TorchSelfplay(cfg, net, B, device, seed=1234, norm=norm)
  replay = PrioritizedReplay(cfg)
  dev = [torch.empty(B, sp.rec_floats, dtype=torch.float32, device=device) for _ in range(2)]
  pinned = [torch.empty(1, B, sp.rec_floats, dtype=torch.float32).pin_memory() for _ in range(2)]
  events = [torch.cuda.Event(), torch.cuda.Event()]
  copy_stream = torch.cuda.Stream(device)

  def run(moves):
    pending, k = None, 0
    for _ in range(moves):
      sp.play_move(dev[k & 1])
      copy_stream.wait_stream(torch.cuda.current_stream(device))
      with torch.cuda.stream(copy_stream):
        pinned[k & 1][0].copy_(dev[k & 1], non_blocking=True)
      events[k & 1].record(copy_stream)
      if pending is not None:
        pending[1].synchronize()
        replay.ingest_records(pending[0], 1, B)
      pending = (pinned[k & 1], events[k & 1])
      k += 1
    if pending is not None:
      pending[1].synchronize()
      replay.ingest_records(pending[0], 1, B)

  run(max(1, warmup))
  # one move with torch's synchronisation debugging armed: any host sync inside search / finalize / env step raises
  torch.cuda.synchronize(device)
  torch.cuda.set_sync_debug_mode('error')
  try:
    sp.play_move(dev[0])
    host_syncs = 0
  finally:
    torch.cuda.set_sync_debug_mode('default')
  torch.cuda.synchronize(device)
  pinned[0][0].copy_(dev[0])               # that move advanced the environments: its records belong to their histories too
  replay.ingest_records(pinned[0], 1, B)
  frames0 = replay.get_throughput()['frames']
  t0 = time.perf_counter()
  run(steps)
  torch.cuda.synchronize(device)
  dt = time.perf_counter() - t0
  frames = replay.get_throughput()['frames'] - frames0
  # the two network calls of a move on their own (events on the current stream, outside the timed region)
  def gpu_ms(fn, n):
    fn(); torch.cuda.synchronize(device)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
      fn()
    b.record(); torch.cuda.synchronize(device)
    return a.elapsed_time(b) / n
  with torch.inference_mode():
    obs_f = (sp.envs.obs.to(torch.float32) - norm[0]) / norm[1]
    hid = net.representation(obs_f)
    act = torch.zeros(B, dtype=torch.int32, device=device)
    init_ms = gpu_ms(lambda: net.initial_inference(obs_f), 2)
    rec_ms = gpu_ms(lambda: net.recurrent_inference(hid, act), 5)
    del obs_f, hid
  f_rec, f_init = muzero_flops(A)
  flop_per_move = B * (SIMS * f_rec + f_init)
  achieved = flop_per_move * steps / dt / 1e12
  out = {
      'metric': 'env-steps/sec (self-play, whole node) at num_simulations=%d' % SIMS,
      'value': B * steps / dt, 'unit': 'env-steps/s', 'n_gpus': 1, 'steps': steps, 'warmup': warmup,
      'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32', 'data': 'synthetic', 'secondary_line': True,
      'config': {'workload': 'Breakout image shapes (obs uint8 [4,96,96] + norm_obs 0 255, actions 4), MuZeroNetwork '
                             '(23.4 M parameters) through PyTorch-ROCm/MIOpen f32 behind mz_select/mz_expand_backup, '
                             'num_simulations=%d, %d parallel self-play envs, synthetic fixed-length episodes T=%d, '
                             'random-init weights (torch.manual_seed(0))' % (SIMS, B, T),
                 'envs_per_gpu': B, 'num_simulations': SIMS, 'episode_len': T,
                 'host_syncs_in_simulation_loop': host_syncs,
                 'note': 'env-steps executed per second; the replay accepted %d frames in the region (episodes of %d '
                         'moves are longer than the region)' % (frames, T)},
      'mcts_sims_per_s_per_gpu': B * steps * SIMS / dt,
      'roofline': {'bound': 'mfma', 'kernel': 'MIOpen f32 convolutions of recurrent_inference (achieved = whole-path FLOP / wall time)',
                   'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': None, 'flop_per_move': flop_per_move},
      'record_bytes_per_env_step': 4 * sp.rec_floats,
      'network_calls': {'initial_inference_ms': init_ms, 'recurrent_inference_ms': rec_ms,
                        'recurrent_tflops': B * f_rec / (rec_ms * 1e-3) / 1e12,
                        'share_of_move': (init_ms + SIMS * rec_ms) / (1e3 * dt / steps),
                        'clock': 'HIP events around back-to-back calls at %d rows, outside the timed region' % B},
  }
  # kernel-level view of this command: rocprofv3 --kernel-trace --stats summary committed under profiles/ (the builder's run;
  # scripts/breakout_shares.py): dominant kernel with its average duration and share, GPU-time share by category
  shares = os.path.join(ROOT, 'profiles', 'r03_breakout_kernel_shares.json')
  if os.path.exists(shares) and B == 4096:
    sj = json.load(open(shares))
    out['roofline']['dominant_kernel'] = sj['dominant_kernel']
    out['roofline']['gpu_time_share_by_category'] = {k: round(v['share'], 4) for k, v in sj['share_by_category'].items()}
    out['roofline']['kernel_trace'] = 'profiles/r03_breakout_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this command, builder-run; not re-measured here)'
  print(json.dumps(out), flush=True)
  sp.close()
