"""One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" for the CPU / one-GPU tests):
the reference's process wiring (train.py:62-78 -- one SharedStorage, one PrioritizedReplay, N actors, one learner, all
talking through Ray's object store) mapped onto ranks.

  rank 0        learner + SharedStorage + THE PrioritizedReplay (native, host) + actor 0
  rank r > 0    actor r  (environments r*B .. (r+1)*B - 1, RNG keyed by global env id)

What is exchanged, and where:
  * weights, learner -> actors (learners.py:85-86, 132-133; actors.py:81-85, 157-158): the learner publishes into rank
    0's SharedStorage as in the reference; at every weight-sync boundary of the actors (every `weight_sync_frequency`
    moves -- the same move count on every rank, the actors run in lock-step chunks) ALL ranks enter `RankStorage.
    get_weights`: ONE broadcast of the flat float32 buffer (engine.WEIGHT_ORDER; 0.79 MB for the LunarLander FCNetwork)
    from rank 0, straight into device memory, followed by the device-side repack (mz_set_weights(on_device=1)); the
    training step and the per-actor game counts (shared_storage.py:12-14) ride in one small int64 all-gather.
  * experiences, actors -> the one replay (actors.py:169; replay_buffer.py:113-122): NOT a collective.  Every rank
    r > 0 owns a single-producer / single-consumer ring of record chunks in POSIX shared memory (one node); a thread on
    rank 0 drains the rings into the one replay with env_base = r * B (mzr_ingest_records_from), rank 0's own actor
    ingests directly.  The learner therefore samples from the union of all ranks' experience, as the reference's does.
"""
import os
import threading
import time
from multiprocessing import shared_memory

import numpy as np
import torch

from .engine import flatten_weights


def _dist():
  import torch.distributed as dist
  return dist


# ------------------------------------------------------------------------------------------------ experience rings
class ShmRing(object):
  """SPSC ring of record chunks in shared memory: header int64 [head, tail, closed, chunk, B, rec] + `slots` chunks of
  [chunk][B][rec] float32, each preceded by its move count.  The producer (an actor rank) only writes `head` and the
  slot it owns, the consumer (rank 0) only writes `tail`."""
  HDR = 8

  def __init__(self, name, chunk=0, B=0, rec=0, slots=4, create=False):
    self.name = name
    if create:
      size = self.HDR * 8 + slots * (8 + chunk * B * rec * 4)
      try:
        old = shared_memory.SharedMemory(name=name)
        old.close(); old.unlink()
      except FileNotFoundError:
        pass
      self.shm = shared_memory.SharedMemory(name=name, create=True, size=size)
      self.hdr = np.ndarray((self.HDR,), np.int64, self.shm.buf, 0)
      self.hdr[:] = [0, 0, 0, chunk, B, rec, slots, 0]
    else:
      self.shm = shared_memory.SharedMemory(name=name)
      self.hdr = np.ndarray((self.HDR,), np.int64, self.shm.buf, 0)
    self.chunk, self.B, self.rec, self.slots = [int(x) for x in self.hdr[3:7]]
    self.slot_bytes = 8 + self.chunk * self.B * self.rec * 4
    self.owner = create
    # head / tail / closed are read with acquire and written with release ordering (mzr_load_acquire_i64 /
    # mzr_store_release_i64, include/mz_replay.h): a chunk's payload is visible to the consumer before the head that
    # announces it, on any host -- not a property borrowed from x86-64's store order
    from . import _abi
    lib = _abi.load_replay()
    base = self.hdr.ctypes.data
    self._load = lambda i: int(lib.mzr_load_acquire_i64(base + 8 * i))
    self._store = lambda i, v: lib.mzr_store_release_i64(base + 8 * i, int(v))

  def _slot(self, i):
    off = self.HDR * 8 + (i % self.slots) * self.slot_bytes
    n = np.ndarray((1,), np.int64, self.shm.buf, off)
    data = np.ndarray((self.chunk, self.B, self.rec), np.float32, self.shm.buf, off + 8)
    return n, data

  # producer side
  def put(self, records, n_moves):
    """records: host float32 [>= n_moves][B][rec] (numpy or pinned torch tensor)."""
    head = self._load(0)                    # (only this side writes it)
    while head - self._load(1) >= self.slots:
      time.sleep(0.0002)
    n, data = self._slot(head)
    src = records.numpy() if torch.is_tensor(records) else np.asarray(records)
    data[:n_moves] = src[:n_moves]          # (a synchronous copy: complete when the statement returns)
    n[0] = n_moves
    self._store(0, head + 1)                # release: published after the payload

  def close_producer(self):
    self._store(2, 1)

  # consumer side
  def poll(self):
    """-> (view [n][B][rec], n) of the oldest unconsumed chunk, or None; call done() when it has been ingested."""
    tail = self._load(1)
    if tail >= self._load(0):               # acquire: the chunk behind a head we have seen is complete
      return None
    n, data = self._slot(tail)
    return data, int(n[0])

  def done(self):
    self._store(1, self._load(1) + 1)       # release: the slot is free once the ingest has read it

  def pending(self):
    """chunks put and not yet done"""
    return self._load(0) - self._load(1)

  def finished(self):
    return bool(self._load(2)) and self._load(1) >= self._load(0)

  def release(self):
    self.hdr = self._load = self._store = None
    try:
      self.shm.close()
      if self.owner:
        self.shm.unlink()
    except (FileNotFoundError, BufferError):
      pass


class RingReplay(object):
  """What an actor on rank r > 0 holds in place of the replay buffer: `ingest_records` ships the chunk to rank 0."""

  def __init__(self, ring):
    self.ring = ring
    self.frames = 0

  def ingest_records(self, records, n_moves, B, env_base=0):
    self.ring.put(records, int(n_moves))
    self.frames += int(n_moves) * int(B)

  def save_history(self, *a, **k):
    raise NotImplementedError('host-environment actors run on the replay rank')

  def get_throughput(self):
    return {'frames': self.frames, 'games': 0}


def serve_rings(rings, replay_call, B, stop):
  """rank 0 thread: drain every rank's ring into the one replay (env_base = rank * B) until all producers closed."""
  live = dict(rings)
  while live and not stop.is_set():
    idle = True
    for r, ring in list(live.items()):
      got = ring.poll()
      if got is not None:
        data, n = got
        replay_call('ingest_records', data, n, B, r * B)
        ring.done()
        idle = False
      elif ring.finished():
        del live[r]
    if idle:
      time.sleep(0.0005)


# ------------------------------------------------------------------------------------------------ weights
class RankStorage(object):
  """SharedStorage surface (shared_storage.py:4-25) as every actor rank sees it.  get_weights is COLLECTIVE: all ranks
  call it at the same point of their loop; rank 0 reads its real SharedStorage, everybody receives the flat weights by
  one broadcast and the training step / game counts by one small all-gather."""

  def __init__(self, rank, world, device, num_weights, storage=None, storage_call=None, backend='nccl',
               flatten=flatten_weights):
    self.rank, self.world, self.device = rank, world, torch.device(device)
    self.storage, self.call, self.flatten = storage, storage_call, flatten
    self.cdev = self.device if backend != 'gloo' else torch.device('cpu')
    self.flat = torch.zeros(int(num_weights), dtype=torch.float32, device=self.cdev)
    self.training_step = 0
    self.broadcasts = 0

  def is_ready(self):
    if self.rank == 0:
      return self.call(self.storage, 'is_ready')
    return True                     # the first collective get_weights delivers them

  def get_weights(self, games, actor_key):
    dist = _dist()
    step = 0
    if self.rank == 0:
      weights, step = self.call(self.storage, 'get_weights', games, actor_key)
      flat = weights if torch.is_tensor(weights) else self.flatten(weights)
      self.flat.copy_(flat.to(self.cdev))
    dist.broadcast(self.flat, src=0)                                   # the path's one exchange: RCCL over xGMI
    meta = torch.tensor([step, games], dtype=torch.int64, device=self.cdev)
    gathered = [torch.zeros_like(meta) for _ in range(self.world)]
    dist.all_gather(gathered, meta)
    self.training_step = int(gathered[0][0])
    if self.rank == 0:
      for r in range(1, self.world):                                   # shared_storage.py:13: per-actor game counts
        self.call(self.storage, 'get_weights', int(gathered[r][1]), r)
    self.broadcasts += 1
    return (self.flat if self.flat.device == self.device else self.flat.to(self.device)), self.training_step


def rccl_mapped():
  """is librccl mapped into this process? (the evidence that backend 'nccl' really is RCCL here)"""
  try:
    return any('librccl' in line for line in open('/proc/self/maps'))
  except OSError:
    return False


def init_process_group():
  """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher (python -m torch.distributed.run)."""
  dist = _dist()
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  backend = os.environ.get('MZ_DIST_BACKEND', 'nccl')
  device = None
  if torch.cuda.is_available():
    local = local % max(1, torch.cuda.device_count())      # several ranks on one GPU only in the gloo self-tests
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
  if backend == 'nccl':
    dist.init_process_group('nccl', device_id=device)
  else:
    dist.init_process_group(backend)
  return rank, world, device, backend
