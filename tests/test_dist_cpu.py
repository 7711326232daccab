"""N > 1 host logic on CPU (gloo, world_size 2): the single exchange of the path -- the flattened weight
broadcast from the learner/storage rank (distributed.RankStorage.get_weights; RCCL on the GPU box) -- and the
rank-wise reductions bench.py reports (max time, summed frames); plus the shard map env_id_offset."""
import os

import numpy as np
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from model_based_rl_amd import distributed as D
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
# the path's one exchange through the product's own class: RankStorage.get_weights (collective; over gloo here, RCCL on the GPU box)
class Store(object):
  def get_weights(self, games, key): return torch.arange(198410, dtype=torch.float32) * 0.5, 7
  def is_ready(self): return True
rs = D.RankStorage(rank, world, 'cpu', 198410, storage=Store() if rank == 0 else None, storage_call=lambda o, n, *a: getattr(o, n)(*a),
                   backend='gloo', flatten=lambda w: w)
flat, step = rs.get_weights(3, rank)
assert step == 7 and torch.equal(flat, torch.arange(198410, dtype=torch.float32) * 0.5)
cs = rs.collective_stats()          # the keys the N > 1 bench line carries (VERDICT r05 item 2)
assert cs['broadcasts'] == 1 and cs['fallback_reason'] == 'backend gloo' and cs['ranks_in_comm'] is None and not cs['native_rccl_broadcast']
assert cs['broadcast_us']['n'] == 1 and cs['broadcast_us']['max'] >= cs['broadcast_us']['mean'] > 0
t = torch.tensor([1.0 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
f = torch.tensor([100.0 * (rank + 1)], dtype=torch.float64); dist.all_reduce(f, op=dist.ReduceOp.SUM)
assert t.item() == world and f.item() == 100.0 * world * (world + 1) / 2
# shard map: rank r owns global env ids [r*B, (r+1)*B)
B = 16
ids = torch.arange(rank * B, (rank + 1) * B)
allids = [torch.zeros(B, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allids, ids)
assert torch.equal(torch.cat(allids), torch.arange(world * B))
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
''' % ROOT


def test_weight_broadcast_and_reductions_world2(tmp_path):
  script = tmp_path / 'w.py'
  script.write_text(WORKER)
  env = dict(os.environ, MASTER_ADDR='127.0.0.1')
  out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)], env=env,
                       capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  assert out.stdout.count('ok') == 2


RANK_WORKER = r'''
import os, sys, threading, types
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from model_based_rl_amd import distributed as D
from model_based_rl_amd.engine import flatten_weights
from model_based_rl_amd.networks import FCNetwork
from model_based_rl_amd.replay_buffer import PrioritizedReplay
from model_based_rl_amd.shared_storage import SharedStorage

os.environ['MZ_DIST_BACKEND'] = 'gloo'
rank, world, device, backend = D.init_process_group()
assert device is None and backend == 'gloo'          # no GPU in this container: host logic only
O, A, B, T, chunk, nchunks = 3, 2, 5, 7, 8, 4
rec = O + A + 10


def records(r):
  """deterministic experience records of rank r: [moves][B][rec], episodes of T steps, staggered"""
  rng = np.random.RandomState(100 + r)
  moves = chunk * nchunks
  x = np.zeros((moves, B, rec), np.float32)
  x[..., :O] = rng.standard_normal((moves, B, O))
  x[..., O + A + 2:O + A + 4] = rng.standard_normal((moves, B))[..., None].view(np.float32)     # error (float64)
  ints = x[..., O + A + 5:].view(np.int32)
  for b in range(B):
    t = (r * B + b) %% T
    for m in range(moves):
      ints[m, b, 1] = int(t + 1 >= T); ints[m, b, 2] = t; ints[m, b, 3] = r * B + b
      t = 0 if t + 1 >= T else t + 1
  return x


cfg = types.SimpleNamespace(batch_size=4, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(O,), action_space=A,
                            window_size=4096, window_step=None, num_unroll_steps=5, td_steps=10, max_history_length=16,
                            discount=0.997, seed=0, num_actors=world)
torch.manual_seed(0)
net = FCNetwork(8, 4, torch.device('cpu'), types.SimpleNamespace())
w0 = net.get_weights()
w1 = {k: v + 1.0 for k, v in w0.items()}
call = lambda obj, name, *a: getattr(obj, name)(*a)
storage = replay = None
rings = {}
if rank == 0:
  storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
  storage.store_weights(w0, 0)
  rings = {r: D.ShmRing('mzt_%%s_%%d' %% (os.environ['MASTER_PORT'], r), chunk, B, rec, slots=2, create=True) for r in range(1, world)}
dist.barrier()
stop = threading.Event()
lock = threading.Lock()
if rank == 0:
  def replay_call(name, *a):
    with lock:
      return getattr(replay, name)(*a)
  server = threading.Thread(target=D.serve_rings, args=(rings, replay_call, B, stop))
  server.start()
  sink = types.SimpleNamespace(ingest_records=lambda buf, n, B_, base=0: replay_call('ingest_records', buf, n, B_, base))
else:
  ring = D.ShmRing('mzt_%%s_%%d' %% (os.environ['MASTER_PORT'], rank))
  sink = D.RingReplay(ring)
rs = D.RankStorage(rank, world, 'cpu', flatten_weights(w0).numel(), storage=storage, storage_call=call, backend='gloo')
assert rs.is_ready()
mine = records(rank)
games = 0
for c in range(nchunks):
  flat, step = rs.get_weights(games, rank)                # collective: every rank, same point of the loop
  want_w, want_step = (w0, 0) if c < 2 else (w1, 7)
  assert torch.equal(flat, flatten_weights(want_w)) and step == want_step, (rank, c, step)
  sink.ingest_records(mine[c * chunk:(c + 1) * chunk], chunk, B, rank * B)
  games += int(mine[c * chunk:(c + 1) * chunk, :, O + A + 6].view(np.int32).sum())
  if rank == 0 and c == 1:
    storage.store_weights(w1, 7)                          # the learner publishes (learners.py:85-86)
rs.get_weights(games, rank)
if rank > 0:
  ring.close_producer()
if rank == 0:
  server.join(timeout=60)
  assert not server.is_alive()
  # the one replay now holds what a single process ingesting every rank's records holds
  ref = PrioritizedReplay(cfg)
  for r in range(world):
    x = records(r)
    for c in range(nchunks):
      ref.ingest_records(np.ascontiguousarray(x[c * chunk:(c + 1) * chunk]), chunk, B, r * B)
  assert replay.get_throughput() == ref.get_throughput() and replay.get_throughput()['frames'] > 0
  n = replay.size()
  assert n == ref.size() and np.array_equal(np.sort(replay.tree.leaves(n)), np.sort(ref.tree.leaves(n)))
  tot = sum(int(records(r)[..., O + A + 6].view(np.int32).sum()) for r in range(world))
  assert replay.get_throughput()['games'] == tot
  assert storage.get_stats('actor_games') == {r: int(records(r)[..., O + A + 6].view(np.int32).sum()) for r in range(world)}
dist.barrier()
stop.set()
for x in list(rings.values()) + ([ring] if rank > 0 else []):
  x.release()
dist.destroy_process_group()
print('rank', rank, 'ok')
''' % ROOT


def test_rank_storage_and_experience_rings_world2(tmp_path):
  """The multi-rank product wiring on CPU (gloo, world 2; distributed.py): the collective weight pull -- flat buffer +
  training step from rank 0's SharedStorage, per-actor game counts back -- and the merge of every rank's experience
  chunks into the ONE replay through shared-memory rings (train.py:62-78, actors.py:81-85,169 of the reference)."""
  script = tmp_path / 'r.py'
  script.write_text(RANK_WORKER)
  env = dict(os.environ, MASTER_ADDR='127.0.0.1')
  out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29541', str(script)], env=env,
                       capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
  assert out.stdout.count('ok') == 2


def test_collective_only_rank_pulls_at_the_actors_cadence():
  """train --dedicated_learner_rank: the rank without an actor enters the collective weight pull exactly where an actor's
  device loop does (actors.py of this package, run_selfplay: once up front, whenever the move count crosses a multiple of
  weight_sync_frequency, once at the end) and stops on the same broadcast training step."""
  import types
  from model_based_rl_amd.train import _CollectiveOnly

  class Storage(object):
    def __init__(self, step_at):
      self.pulls, self.step_at = 0, step_at
    def is_ready(self):
      return True
    def get_weights(self, games, key):
      self.pulls += 1
      return None, self.step_at(self.pulls)

  cfg = types.SimpleNamespace(weight_sync_frequency=16, training_steps=100)
  st = Storage(lambda n: 0)
  c = _CollectiveOnly(0, cfg, st, 8)
  c.launch(max_moves=64)
  assert st.pulls == 1 + 4 + 1 and c.move_counter == 64
  st = Storage(lambda n: 100 if n >= 4 else n)              # the learner finishes: seen at the 4th pull, the loop ends there
  c = _CollectiveOnly(0, cfg, st, 8)
  c.launch(max_moves=None)
  assert st.pulls == 4 + 1 and c.training_step == 100 and c.move_counter == 48
  st = Storage(lambda n: 0)                                 # one move per iteration (the torch-network actors)
  c = _CollectiveOnly(0, types.SimpleNamespace(weight_sync_frequency=5, training_steps=9), st, 1)
  c.launch(max_moves=12)
  assert st.pulls == 1 + 2 + 1


def test_shm_ring_orders_payload_before_head(tmp_path):
  """distributed.ShmRing: chunks come out in order and complete; head / tail travel through the release / acquire helpers of
  libmz_replay.so (a producer thread against a consumer thread of this process)"""
  import threading
  import time
  from model_based_rl_amd import distributed as D
  name = 'mzt_ring_%d' % os.getpid()
  prod = D.ShmRing(name, chunk=4, B=32, rec=22, slots=3, create=True)
  cons = D.ShmRing(name)
  seen = []

  def consume():
    while not cons.finished():
      got = cons.poll()
      if got is None:
        time.sleep(0.0002)          # (as distributed.serve_rings does: a spinning consumer would hold the interpreter lock)
        continue
      data, n, packed = got
      assert not packed
      seen.append((n, data[:n].copy()))
      cons.done()

  th = threading.Thread(target=consume)
  th.start()
  rng = np.random.RandomState(0)
  sent = []
  for i in range(40):
    n = 1 + i % 4
    rec = rng.standard_normal((4, 32, 22)).astype(np.float32)
    sent.append((n, rec[:n].copy()))
    prod.put(rec, n, pack=False)
  prod.close_producer()
  th.join(timeout=30)
  assert not th.is_alive() and len(seen) == 40 and cons.pending() == 0
  for (n0, r0), (n1, r1) in zip(sent, seen):
    assert n0 == n1 and np.array_equal(r0, r1)
  cons.release(); prod.release()


def test_preflight_refuses_before_any_gpu_call(tmp_path):
  """distributed.preflight (bench.py / train.py call it before a rank touches a GPU): too few devices, too little shared memory
  for the experience rings, too few cores -> SystemExit with one sentence; and bench.py --gpus 2 with an artificially large
  /dev/shm need exits non-zero, prints that sentence and starts no rank."""
  import pytest
  from model_based_rl_amd import distributed as D
  ok = D.preflight(8, shm_need=D.ring_bytes(8, 16, 4096, 22), ingest_threads=4, devices=8, shm_free=1 << 30, cores=16)
  assert ok['ranks'] == 8 and ok['shm_need_bytes'] == 7 * (64 + 4 * (8 + 16 * 4096 * 22 * 4)) and ok['cores_wanted'] == 8.0
  with pytest.raises(SystemExit, match='8 ranks asked for but 1 GPU'):
    D.preflight(8, devices=1, shm_free=1 << 30, cores=16)
  assert D.preflight(8, devices=1, shm_free=1 << 30, cores=16, shared_gpu_ok=True)['visible_devices'] == 1
  with pytest.raises(SystemExit, match='experience rings of 8 ranks need'):
    D.preflight(8, shm_need=D.ring_bytes(8, 16, 4096, 22), devices=8, shm_free=64 << 20, cores=16)      # a container's 64 MB default
  with pytest.raises(SystemExit, match='host cores'):
    D.preflight(8, ingest_threads=4, devices=8, shm_free=1 << 30, cores=4)
  assert D.preflight(1, devices=0, shm_free=0, cores=1)['ranks'] == 1          # one rank: nothing to refuse
  env = dict(os.environ, MZ_SHARED_GPU_OK='1', MZ_PREFLIGHT_SHM_NEED=str(1 << 50))
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--no-cpu-baseline'], env=env,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
  assert out.returncode != 0 and 'pre-flight: the experience rings of 2 ranks need' in out.stderr
  assert not [l for l in out.stdout.splitlines() if l.startswith('{')] and 'torch.distributed.run' not in out.stderr
