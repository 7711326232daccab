"""Learner updates per second on an idle GPU: the native step (mz_fcl_update: five HIP launches per update, no graph;
the default) against the PyTorch step captured in one hipGraph (--no_native_learner) and against eager PyTorch
launches (--no_graph_learner), FCNetwork, batch 256, K = 5 (reference learners.py:164-230).  The loop is
Learner.learn's: sample_batch from the native replay, update_weights with the priority refresh one update behind.
usage: learner_graph_speed.py [out.json]
environment: MZ_LS_ONLY=native,graph,eager (variants to run); MZ_LS_HANDLE=1 (the replay behind a rayshim handle, as under
train.py: sampled on its own thread); MZ_LS_NO_PREFETCH=1 (no batch source: one sample_batch_arrays call per update);
MZ_LS_DEPTH (batches in flight on a handle, default 4); MZ_SWITCH_INTERVAL (the interpreter's thread switch interval)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_based_rl_amd.config import make_config
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.learners import Learner
from model_based_rl_amd.networks import get_network
from model_based_rl_amd.replay_buffer import PrioritizedReplay
from model_based_rl_amd.shared_storage import SharedStorage

def setup(extra):
  cfg = make_config(['--environment', 'LunarLander-v2', '--num_simulations', '30', '--seed', '0', '--num_envs', '1024',
                     '--window_size', '200000', '--batch_size', '256', '--use_gpu_for', 'actors', 'learner',
                     '--runs_dir', '/tmp/mz_ls', '--run_tag', 'x'] + extra)
  storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
  torch.manual_seed(0)
  eng = Engine.from_config(cfg, 1024)
  eng.set_weights(flatten_weights(get_network(cfg, torch.device('cpu')).state_dict()))
  eng.selfplay_reset(32, 1.0, stagger=True)
  for _ in range(8):
    eng.selfplay_steps(16); buf, n = eng.selfplay_drain(); torch.cuda.synchronize(); replay.ingest_records(buf[:n], n, 1024)
  eng.close()
  return cfg, storage, replay, Learner(cfg, storage, replay)

def loop(learner, replay, n):
  """Learner._learn_loop's body: batches through _BatchSource (a plain replay: sampled in this thread, two per native call;
  MZ_LS_HANDLE=1: through a rayshim handle, i.e. on the replay's own thread as under train.py), priority refreshes one update behind"""
  from model_based_rl_amd.learners import _BatchSource
  from model_based_rl_amd import rayshim
  target = rayshim._Handle(replay) if os.environ.get('MZ_LS_HANDLE') else replay
  src = _BatchSource(target, int(os.environ.get('MZ_LS_DEPTH', '4'))) if not os.environ.get('MZ_LS_NO_PREFETCH') else None
  learner._source = src
  ts = [0.0, 0.0]
  t0 = time.perf_counter()
  for _ in range(n):
    a = time.perf_counter(); batch = src.get() if src else replay.sample_batch_arrays(); b = time.perf_counter()
    learner.update_weights(batch, defer_priorities=True); c = time.perf_counter()
    ts[0] += b - a; ts[1] += c - b
  learner.flush_priorities()
  torch.cuda.synchronize()
  if src: src.close()
  if target is not replay:
    target._q.put((None, (), {}, None)); target._t.join(timeout=5)
  learner._source = None
  dt = time.perf_counter() - t0
  return {'updates_per_second': n / dt, 'batch_wait_ms': 1e3 * ts[0] / n, 'update_call_ms': 1e3 * ts[1] / n}

def main():
  if os.environ.get('MZ_SWITCH_INTERVAL'):          # experiment: the interpreter's thread switch interval (default 5 ms)
    sys.setswitchinterval(float(os.environ['MZ_SWITCH_INTERVAL']))
  out = {}
  variants = (('native', []), ('graph', ['--no_native_learner']), ('eager', ['--no_graph_learner']))
  if os.environ.get('MZ_LS_ONLY'):
    variants = tuple(v for v in variants if v[0] in os.environ['MZ_LS_ONLY'].split(','))
  for name, extra in variants:
    cfg, storage, replay, learner = setup(extra)
    loop(learner, replay, 30)
    res = [loop(learner, replay, 1000 if name != 'eager' else 300) for _ in range(3)]
    t0 = time.perf_counter()
    for _ in range(200): replay.sample_batch_arrays()
    sample_ms = (time.perf_counter() - t0) / 200 * 1e3
    out[name] = max(res, key=lambda r: r['updates_per_second'])
    out[name]['runs_updates_per_second'] = [r['updates_per_second'] for r in res]
    out[name]['sample_batch_arrays_ms'] = sample_ms
    if name in ('graph', 'native'):
      assert (learner._native is not None) == (name == 'native')
      # GPU time of one update (graph replay, or the native step's launches with their two copies), on the event clock
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      if name == 'native':
        host = learner._host_batch(replay.sample_batch_arrays())[0]
        run = lambda: learner._native.launch(host)
      else:
        run = learner._graph.graph.replay
      torch.cuda.synchronize(); e0.record()
      for _ in range(50): run()
      e1.record(); torch.cuda.synchronize()
      out[name]['graph_replay_gpu_ms'] = e0.elapsed_time(e1) / 50
      out[name]['replay_size'] = replay.size()
  first = out[variants[0][0]]
  out['what'] = 'FCNetwork (LunarLander shapes: obs 8, 4 actions), batch 256, K = 5 unroll, AdamW; idle MI355X; native replay of %d frames' % first.get('replay_size', 0)
  if 'eager' in out and 'graph' in out:
    out['speedup_graph_over_eager'] = out['graph']['updates_per_second'] / out['eager']['updates_per_second']
  if 'native' in out and 'graph' in out:
    out['speedup_native_over_graph'] = out['native']['updates_per_second'] / out['graph']['updates_per_second']
  print(json.dumps(out, indent=1))
  if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)

if __name__ == '__main__':
  main()
