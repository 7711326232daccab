"""One learner configuration, a few hundred graphed updates: the program rocprofv3 --kernel-trace --stats wraps to see which
kernels an update consists of (kernel development).  usage: learner_profile.py [n_updates] [extra config flags...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_based_rl_amd.config import make_config
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.learners import Learner
from model_based_rl_amd.networks import get_network
from model_based_rl_amd.replay_buffer import PrioritizedReplay
from model_based_rl_amd.shared_storage import SharedStorage
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = make_config(['--environment', 'LunarLander-v2', '--num_simulations', '30', '--seed', '0', '--num_envs', '1024', '--window_size', '200000',
                   '--batch_size', '256', '--use_gpu_for', 'actors', 'learner', '--runs_dir', '/tmp/mz_lp', '--run_tag', 'x'] + sys.argv[2:])
storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
torch.manual_seed(0)
eng = Engine.from_config(cfg, 1024)
eng.set_weights(flatten_weights(get_network(cfg, torch.device('cpu')).state_dict()))
eng.selfplay_reset(32, 1.0, stagger=True)
for _ in range(4):
  eng.selfplay_steps(16); buf, k = eng.selfplay_drain(); torch.cuda.synchronize(); replay.ingest_records(buf[:k], k, 1024)
eng.close()
learner = Learner(cfg, storage, replay)
for _ in range(10):
  learner.update_weights(replay.sample_batch(), defer_priorities=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
  learner.update_weights(replay.sample_batch(), defer_priorities=True)
learner.flush_priorities(); torch.cuda.synchronize()
print('%.1f updates/s' % (n / (time.perf_counter() - t0)))
