"""Where do the ~1.5 host cores of one rank's self-play loop go?  CPU seconds per thread (/proc/self/task/*/stat) over one
Actor.launch of 3072 moves at the bench's size (4096 envs, LunarLander shapes), by thread: the launching thread, the record
pipe's worker (game statistics + the hand-over), the rayshim handles' threads (the replay's runs the native ingest call), the
native replay's ingest pool and inserter."""
import os, sys, time, threading, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import model_based_rl_amd
from model_based_rl_amd import rayshim as ray
from model_based_rl_amd.actors import Actor
from model_based_rl_amd.config import make_config
from model_based_rl_amd.networks import FCNetwork
from model_based_rl_amd.replay_buffer import PrioritizedReplay
from model_based_rl_amd.shared_storage import SharedStorage

def task_times():
  out = {}
  for tid in os.listdir('/proc/self/task'):
    try:
      f = open('/proc/self/task/%s/stat' % tid).read()
      rest = f[f.rindex(')') + 2:].split()
      out[int(tid)] = (int(rest[11]) + int(rest[12])) / os.sysconf('SC_CLK_TCK')
    except OSError:
      pass
  return out

THREADS = os.environ.get('MZ_PROBE_INGEST_THREADS')
PONG = os.environ.get('MZ_PROBE_WORKLOAD') == 'pong'      # Pong-ram shapes: obs 128 bytes, 6 actions, 50 simulations, --norm_obs
O, A = (128, 6) if PONG else (8, 4)
cfg = make_config((['--environment', 'Pong-ramNoFrameskip-v4', '--num_simulations', '50', '--norm_obs', '--obs_range', '0', '255', '--ingest_threads', THREADS or '4']
                   if PONG else ['--environment', 'LunarLander-v2', '--num_simulations', '30'] + (['--ingest_threads', THREADS] if THREADS else [])) +
                  ['--num_envs', '4096', '--seed', '1', '--window_size', str(1 << 21), '--weight_sync_frequency', '128', '--runs_dir', '/tmp/mz_runs',
                   '--run_tag', 'cpu_probe', '--fixed_temperatures', '1.0'])
storage, replay = ray.remote(SharedStorage).remote(cfg), ray.remote(PrioritizedReplay).remote(cfg)
torch.manual_seed(0)
storage.store_weights.remote(FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).get_weights(), 1).result()
actor = Actor(0, cfg, storage, replay)
actor.launch(768)
names = {t.native_id: t.name for t in threading.enumerate()}
names[threading.main_thread().native_id] = 'main (launches)'
names[actor._pipe.thread.native_id] = 'record pipe worker'
names[replay._t.native_id] = 'replay handle (ingest call)'
names[storage._t.native_id] = 'storage handle'
t0, w0 = task_times(), time.perf_counter()
MOVES = 1536 if PONG else 3072
actor.launch(MOVES)
dt = time.perf_counter() - w0
t1 = task_times()
def comm(k):
  try:
    return open('/proc/self/task/%d/comm' % k).read().strip()
  except OSError:
    return '?'
rows = sorted(((t1[k] - t0.get(k, 0.0)) / dt, names.get(k, 'native thread %d (%s)' % (k, comm(k)))) for k in t1)
for busy, name in reversed(rows):
  if busy > 0.005:
    print('%-34s %.3f cores' % (name, busy))
print('total %.3f cores over %.2f s; %.2f M env-steps/s' % (sum(b for b, _ in rows), dt, 4096 * MOVES / dt / 1e6))
