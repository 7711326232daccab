"""Learner updates per second on an idle GPU and beside a device self-play loop on the same GPU (same stream / own stream)."""
import sys, time, types, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_based_rl_amd.config import make_config
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.learners import Learner
from model_based_rl_amd.replay_buffer import PrioritizedReplay
from model_based_rl_amd.shared_storage import SharedStorage
from model_based_rl_amd.networks import get_network
cfg = make_config(['--environment', 'TicTacToe', '--two_players', '--td_steps', '10', '--discount', '1', '--known_bounds', '-1', '1',
                   '--num_simulations', '30', '--seed', '0', '--num_envs', '1024', '--window_size', '200000', '--use_gpu_for', 'actors', 'learner',
                   '--runs_dir', '/tmp/mz_ls', '--run_tag', 'x'])
storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
torch.manual_seed(0)
w = flatten_weights(get_network(cfg, torch.device('cpu')).state_dict())
eng = Engine.from_config(cfg, 1024)
eng.set_weights(w); eng.selfplay_set_env('tictactoe'); eng.selfplay_reset(9, 1.0)
for _ in range(8):
  eng.selfplay_steps(8); buf, n = eng.selfplay_drain(); torch.cuda.synchronize(); replay.ingest_records(buf[:n], n, 1024)
print('replay size', replay.size())
learner = Learner(cfg, storage, replay)
def loop(n):
  t0 = time.perf_counter()
  ts = [0, 0]
  for _ in range(n):
    a = time.perf_counter(); batch = replay.sample_batch(); b = time.perf_counter(); learner.update_weights(batch); c = time.perf_counter()
    ts[0] += b - a; ts[1] += c - b
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  return n / dt, ts[0] / n * 1e3, ts[1] / n * 1e3
loop(20)
print('learner alone: %.1f updates/s (sample %.2f ms, update %.2f ms)' % loop(100))
import threading
stop = False
def actor(own_stream):
  st = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
  ev = torch.cuda.Event()
  with torch.cuda.stream(st):
    while not stop:
      eng.selfplay_steps(8); buf, n = eng.selfplay_drain()
      if own_stream == 'poll':          # wait like bench.Pipeline does: query + sleep, no blocking runtime call
        ev.record(st)
        while not ev.query():
          time.sleep(0.0002)
      else:
        st.synchronize()
for own in (False, True, 'poll'):
  stop = False
  th = threading.Thread(target=actor, args=(own,)); th.start()
  time.sleep(0.2)
  print('learner beside the device loop (1024 envs, actor on %s stream): %.1f updates/s (sample %.2f ms, update %.2f ms)' %
        (('its own' if own else 'the default',) + loop(60)) + ('  [actor waits by event.query() + sleep]' if own == 'poll' else ''))
  stop = True; th.join()

# the same device loop in ANOTHER PROCESS on the same GPU (what a second rank sharing the GPU would be)
import subprocess
child = subprocess.Popen([sys.executable, '-c', '''
import sys, time, types, torch
sys.path.insert(0, %r)
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
torch.manual_seed(0)
eng = Engine(1024, 9, 9, 30, two_players=True, known_bounds=(-1.0, 1.0), discount=1.0)
eng.set_weights(flatten_weights(FCNetwork(9, 9, torch.device("cpu"), types.SimpleNamespace()).state_dict()))
eng.selfplay_set_env("tictactoe"); eng.selfplay_reset(9, 1.0)
print("child running", flush=True)
t0 = time.time()
while time.time() - t0 < 12:
  eng.selfplay_steps(8); eng.selfplay_drain(); torch.cuda.synchronize()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))], stdout=subprocess.PIPE, text=True)
child.stdout.readline()
time.sleep(1.0)
print('learner beside the device loop in ANOTHER PROCESS (1024 envs): %.1f updates/s (sample %.2f ms, update %.2f ms)' % loop(100))
child.wait()

# where does an update spend its time beside the device loop?  (torch.profiler-free: wall clock around the phases)
import torch.nn.functional as F
stop = False
th = threading.Thread(target=actor, args=('poll',)); th.start(); time.sleep(0.2)
batch = replay.sample_batch()
(observations, actions, (t_rew, t_val, t_pol)), idxs, isw = batch
net, opt, dev = learner.network, learner.optimizer, learner.device
tt = {}
def lap(name, t0):
  torch.cuda.current_stream().synchronize(); tt[name] = tt.get(name, 0) + time.perf_counter() - t0; return time.perf_counter()
for it in range(20):
  t0 = time.perf_counter()
  obs = torch.from_numpy(np.ascontiguousarray(observations)).to(dev); t0 = lap('h2d', t0)
  value, _, pl, hidden = net.initial_inference(obs); t0 = lap('initial fwd', t0)
  loss = value.float().sum() + pl.float().sum()
  for i, action in enumerate(zip(*actions), 1):
    value, reward, pl, hidden = net.recurrent_inference(hidden, action)
    loss = loss + value.float().sum() + reward.float().sum() + pl.float().sum()
  t0 = lap('5 recurrent fwd', t0)
  opt.zero_grad(); loss.backward(); t0 = lap('backward', t0)
  opt.step(); t0 = lap('optimizer', t0)
stop = True; th.join()
print('per update beside the actor (ms):', {k: round(v / 20 * 1e3, 2) for k, v in tt.items()})
