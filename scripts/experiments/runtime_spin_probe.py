import os, sys, time, types
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import model_based_rl_amd
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
def task_times():
  out = {}
  for tid in os.listdir('/proc/self/task'):
    try:
      f = open('/proc/self/task/%s/stat' % tid).read(); rest = f[f.rindex(')') + 2:].split()
      out[int(tid)] = (int(rest[11]) + int(rest[12])) / os.sysconf('SC_CLK_TCK')
    except OSError: pass
  return out
def measure(label, fn, secs=1.0):
  t0, w0 = task_times(), time.perf_counter()
  while time.perf_counter() - w0 < secs: fn()
  dt = time.perf_counter() - w0; t1 = task_times()
  top = sorted(((t1[k] - t0.get(k, 0.0)) / dt, k) for k in t1)[-3:]
  print(label, [(round(b, 2), k) for b, k in reversed(top)], 'threads', len(t1), flush=True)
main = os.getpid()
torch.manual_seed(0)
eng = Engine(4096, 8, 4, 30, seed=1)
eng.set_weights(flatten_weights(FCNetwork(8, 4, torch.device('cpu'), types.SimpleNamespace()).state_dict()))
eng.selfplay_reset(256, 1.0, stagger=True)
measure('idle after engine', lambda: time.sleep(0.01))
pinned = torch.empty(16, 4096, eng.rec_floats).pin_memory()
def steps_sync():
  eng.selfplay_steps(16); eng.selfplay_drain(pinned, 16); torch.cuda.synchronize()
measure('steps + drain + synchronize', steps_sync)
ev = torch.cuda.Event()
def steps_poll():
  eng.selfplay_steps(16); eng.selfplay_drain(pinned, 16); ev.record()
  while not ev.query(): time.sleep(0.0002)
measure('steps + drain + event poll', steps_poll)
cs = torch.cuda.Stream()
def steps_copy_stream():
  eng.selfplay_steps(16); eng.selfplay_drain(pinned, 16, copy_stream=cs); ev.record(cs)
  while not ev.query(): time.sleep(0.0002)
measure('steps + drain on copy stream + poll', steps_copy_stream)
def steps_only():
  eng.selfplay_steps(16); eng.selfplay_drain(pinned, 16); ev.record()
  time.sleep(0.006)
measure('steps + drain, sleep instead of waiting', steps_only)
torch.cuda.synchronize()
measure('idle at the end', lambda: time.sleep(0.01))
print('main tid', main)
