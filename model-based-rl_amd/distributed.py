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
  BLOB = 1 << 40            # slot counts from here on: a blob of finished history slices of (count - BLOB) bytes

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
    lib = self._lib = _abi.load_replay()
    base = self.hdr.ctypes.data
    self._load = lambda i: int(lib.mzr_load_acquire_i64(base + 8 * i))
    self._store = lambda i, v: lib.mzr_store_release_i64(base + 8 * i, int(v))

  def _slot(self, i):
    off = self.HDR * 8 + (i % self.slots) * self.slot_bytes
    n = np.ndarray((1,), np.int64, self.shm.buf, off)
    data = np.ndarray((self.chunk, self.B, self.rec), np.float32, self.shm.buf, off + 8)
    return n, data

  # producer side
  def put(self, records, n_moves, pack=True):
    """records: host float32 [>= n_moves][B][rec] (numpy or pinned torch tensor).  pack: the copy into the slot is the transposing
    one (mzr_pack_env_major: [B][n_moves][rec]) -- the per-record work the ONE replay's host would otherwise do with strided
    reads is done here, by the producing rank; the slot's count is stored negative to say so."""
    head = self._load(0)                    # (only this side writes it)
    while head - self._load(1) >= self.slots:
      time.sleep(0.0002)
    n, data = self._slot(head)
    src = records.numpy() if torch.is_tensor(records) else np.asarray(records)
    if pack and n_moves > 0 and os.environ.get('MZ_RING_NO_PACK', '0')[:1] != '1':      # (MZ_RING_NO_PACK=1: the r05 hand-off, A/B runs)
      src = np.ascontiguousarray(src[:n_moves], np.float32)
      self._lib.mzr_pack_env_major(src.ctypes.data, data.ctypes.data, int(n_moves), self.B, self.rec)
      n[0] = -n_moves
    else:
      data[:n_moves] = src[:n_moves]        # (a synchronous copy: complete when the statement returns)
      n[0] = n_moves
    self._store(0, head + 1)                # release: published after the payload

  def put_slices(self, assembler):
    """the finished history slices a producing rank's assembler holds (mz_assembler, include/mz_replay.h), oldest first, as blobs of at
    most one slot each; -> slots published"""
    lib, put = self._lib, 0
    while lib.mzr_asm_pending(assembler) > 0:
      head = self._load(0)
      while head - self._load(1) >= self.slots:
        time.sleep(0.0002)
      n, data = self._slot(head)
      nbytes = int(lib.mzr_asm_take(assembler, data.ctypes.data, self.slot_bytes - 8))
      if nbytes < 0:
        from . import _abi
        _abi.check_replay(-1, 'mzr_asm_take')
      if nbytes == 0:
        break
      n[0] = self.BLOB + nbytes
      self._store(0, head + 1)
      put += 1
    return put

  def close_producer(self):
    self._store(2, 1)

  # consumer side
  def poll(self):
    """-> (view of the slot, n, packed) of the oldest unconsumed chunk, or None; call done() when it has been ingested.  packed:
    the slot holds [B][n][rec] (the producer's mzr_pack_env_major), else [n][B][rec]."""
    tail = self._load(1)
    if tail >= self._load(0):               # acquire: the chunk behind a head we have seen is complete
      return None
    n, data = self._slot(tail)
    k = int(n[0])
    if k >= self.BLOB:                      # a blob of finished slices (put_slices): k - BLOB bytes
      return data, k - self.BLOB, 'slices'
    return data, abs(k), k < 0

  def done(self):
    self._store(1, self._load(1) + 1)       # release: the slot is free once the ingest has read it

  def pending(self):
    """chunks put and not yet done"""
    return self._load(0) - self._load(1)

  def finished(self):
    return bool(self._load(2)) and self._load(1) >= self._load(0)

  def release(self):
    self.hdr = self._load = self._store = self._lib = None
    try:
      self.shm.close()
      if self.owner:
        self.shm.unlink()
    except (FileNotFoundError, BufferError):
      pass


class RingReplay(object):
  """What an actor on rank r > 0 holds in place of the replay buffer.  With the run's config: the per-ENVIRONMENT half of the replay's
  ingest (open game buffers, flush rules, history slicing, priorities: actors.py:160-173, replay_buffer.py:110-111) runs HERE, in a
  native assembler of this rank (mz_assembler), and `ingest_records` ships the finished slices to rank 0, whose one replay only
  copies them and inserts their leaves.  Without a config (and with MZ_RING_RAW=1): the record chunk itself travels."""

  def __init__(self, ring, config=None):
    self.ring = ring
    self.frames = 0
    self.asm = None
    if config is not None and os.environ.get('MZ_RING_RAW', '0')[:1] != '1':
      import ctypes as C
      from . import _abi
      from .replay_buffer import native_config
      self._lib = _abi.load_replay()
      self.asm = C.c_void_p()
      _abi.check_replay(self._lib.mzr_asm_create(C.byref(native_config(config)), int(ring.B), C.byref(self.asm)), 'mzr_asm_create')

  def ingest_records(self, records, n_moves, B, env_base=0, env_major=False):
    if self.asm is None:
      self.ring.put(records, int(n_moves))    # (packed environment-major on the way into the ring)
    else:
      from . import _abi
      src = records.numpy() if torch.is_tensor(records) else np.ascontiguousarray(records, np.float32)
      _abi.check_replay(self._lib.mzr_asm_feed(self.asm, src.ctypes.data, int(n_moves), int(B), int(src.shape[-1])), 'mzr_asm_feed')
      self.ring.put_slices(self.asm)
    self.frames += int(n_moves) * int(B)

  def close(self):
    if self.asm is not None:
      self._lib.mzr_asm_destroy(self.asm)
      self.asm = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def save_history(self, *a, **k):
    raise NotImplementedError('host-environment actors run on the replay rank')

  def get_throughput(self):
    return {'frames': self.frames, 'games': 0}


def serve_rings(rings, replay_call, B, stop, threads=1, direct=None):
  """rank 0: drain every rank's ring into the one replay (env_base = rank * B) until all producers closed.  threads > 1: the
  rings are dealt out to that many drain threads -- a blob of slices is copied by the thread that polled it, outside the replay's
  lock (mzr_ingest_slices), so the copies of several rings overlap and only the hand-over to the inserter is serial.  direct: the
  replay OBJECT for the slice blobs (a rayshim handle would queue every call on its one thread; the native handle takes calls from
  any thread)."""
  if threads > 1 and len(rings) > 1:
    keys = sorted(rings)
    parts = [dict((k, rings[k]) for k in keys[i::threads]) for i in range(min(threads, len(keys)))]
    ths = [threading.Thread(target=serve_rings, args=(p, replay_call, B, stop, 1, direct), daemon=True) for p in parts[1:]]
    for t in ths:
      t.start()
    serve_rings(parts[0], replay_call, B, stop, 1, direct)
    for t in ths:
      t.join()
    return
  live = dict(rings)
  while live and not stop.is_set():
    idle = True
    for r, ring in list(live.items()):
      got = ring.poll()
      if got is not None:
        data, n, packed = got
        if packed == 'slices':             # assembled by the producing rank: n bytes of finished slices
          if direct is not None:
            direct.ingest_slices(data, n, r * B)
          else:
            replay_call('ingest_slices', data, n, r * B)
        else:
          replay_call('ingest_records', data, n, B, r * B, packed)
        ring.done()
        idle = False
      elif ring.finished():
        del live[r]
    if idle:
      time.sleep(0.0005)


# ------------------------------------------------------------------------------------------------ pre-flight
def usable_cores():
  """cores this process may use: the affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256 logical CPUs under a
  16-CPU quota; threads beyond the quota only time-slice)"""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
    if quota != 'max':
      n = min(n, max(1, int(int(quota) / int(period))))
  except (OSError, ValueError):
    pass
  return n


def ring_bytes(world, chunk, B, rec, slots=4):
  """shared memory the one-replay layout takes: a ShmRing per rank r > 0 (ShmRing.__init__'s size)"""
  return max(0, world - 1) * (ShmRing.HDR * 8 + slots * (8 + chunk * B * rec * 4))


def preflight(world, shm_need=0, ingest_threads=0, shared_gpu_ok=False, devices=None, shm_free=None, cores=None, shm_dir='/dev/shm'):
  """Checks a multi-rank run needs to pass BEFORE any rank touches a GPU (a failure after that point leaves N - 1 ranks inside
  a collective): visible devices >= ranks (unless the ranks are meant to share GPUs: the one-GPU self-tests), free shared memory
  >= the experience rings' bytes (a container's 64 MB /dev/shm default ends `train --ranks 8` with SIGBUS at the first chunk),
  usable cores >= half a core per rank (the measured host share of a rank, profiles/r05_host_threads.txt) + the replay's ingest
  threads.  -> a dict of what was found; raises SystemExit with ONE sentence otherwise.  devices / shm_free / cores: overrides (tests)."""
  if devices is None:
    devices = torch.cuda.device_count()                # (counting devices does not initialise the GPU)
  if shm_free is None:
    try:
      st = os.statvfs(shm_dir)
      shm_free = st.f_bavail * st.f_frsize
    except OSError:
      shm_free = None
  if cores is None:
    cores = usable_cores()
  found = {'ranks': int(world), 'visible_devices': int(devices), 'shm_need_bytes': int(shm_need), 'shm_free_bytes': shm_free,
           'usable_cores': int(cores), 'cores_wanted': 0.5 * world + ingest_threads}
  if world > 1 and not shared_gpu_ok and devices < world:
    raise SystemExit('pre-flight: %d ranks asked for but %d GPU(s) visible (one process per GPU; set MZ_SHARED_GPU_OK=1 for the '
                     'several-ranks-on-one-GPU self-test).' % (world, devices))
  if shm_need and shm_free is not None and shm_free < shm_need:
    raise SystemExit('pre-flight: the experience rings of %d ranks need %.1f MB of %s and %.1f MB are free (raise the '
                     'container\'s shm size or lower --num_envs / the chunk).' % (world, shm_need / 1e6, shm_dir, shm_free / 1e6))
  if world > 1 and cores < 0.5 * world + ingest_threads:
    raise SystemExit('pre-flight: %d ranks + %d ingest threads want %.1f host cores and %d are usable (affinity mask / cgroup quota).'
                     % (world, ingest_threads, 0.5 * world + ingest_threads, cores))
  return found


# ------------------------------------------------------------------------------------------------ weights
class FlatWeights(object):
  """What RankStorage.get_weights hands an actor over RCCL: the device buffer the broadcast fills (`tensor`), the event
  that says it has (`event`, on the storage's side stream -- the consumer's stream waits for it, not the host), and the
  learner rank's mz_weights_scale_ok for this weight set (`scale_ok`, rode along with the training step).  consumed():
  called by the consumer once its stream has the repack queued -- the storage's next write into this buffer waits for it."""

  def __init__(self, storage, slot, tensor, scale_ok, event):
    self.storage, self.slot, self.tensor, self.scale_ok, self.event = storage, slot, tensor, scale_ok, event

  def consumed(self):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(self.tensor.device))
    self.storage._consumed[self.slot] = ev


class RankStorage(object):
  """SharedStorage surface (shared_storage.py:4-25) as every actor rank sees it.  get_weights is COLLECTIVE: all ranks
  call it at the same point of their loop; rank 0 reads its real SharedStorage, everybody receives the flat weights by
  one broadcast and the training step / game counts by one small all-gather.

  backend "nccl" (the GPUs of a node, RCCL over xGMI): the broadcast is `mz_broadcast_weights` -- ncclBroadcast called from
  libmz_hip.so on a communicator of its own (mz_comm_create; torch.distributed only carries rank 0's unique id) -- issued on
  a SIDE stream into one of two device buffers, so that it neither waits for the moves queued on an actor's stream nor
  makes that stream wait longer than the broadcast itself takes; the training step, the game counts and rank 0's
  mz_weights_scale_ok travel over a host-side (gloo) group: nothing on the host waits for the GPU, a weight pull never
  drains the launch-ahead pipeline.  MZ_TORCH_COLLECTIVES=1 or backend "gloo" (CPU tensors; the one-GPU self-tests):
  torch.distributed's own broadcast / all_gather."""

  def __init__(self, rank, world, device, num_weights, storage=None, storage_call=None, backend='nccl',
               flatten=flatten_weights, scale_check=None):
    self.rank, self.world, self.device = rank, world, torch.device(device)
    self.storage, self.call, self.flatten, self.scale_check = storage, storage_call, flatten, scale_check
    self.cdev = self.device if backend != 'gloo' else torch.device('cpu')
    self.training_step = 0
    self.broadcasts = 0
    self.native = backend == 'nccl' and os.environ.get('MZ_TORCH_COLLECTIVES', '0')[:1] != '1'
    # what a multi-rank record needs to be read without the logs (collective_stats): why the library's communicator is not in use
    # (None: it is), how many ranks RCCL says it spans, HIP-event pairs around every broadcast on the side stream
    self.fallback_reason = None if self.native else ('backend %s' % backend if backend != 'nccl' else 'MZ_TORCH_COLLECTIVES=1')
    self.ranks_in_comm = None
    self._bcast_events, self._bcast_us = [], []
    self._done_last = None
    n = int(num_weights)
    self.ctrl_group = None          # (None: the default group)
    if not self.native:
      self._flats = [torch.zeros(n, dtype=torch.float32, device=self.cdev)]
      self._last = 0
      return
    import ctypes as C
    from . import _abi
    dist = _dist()
    # host-side control group (gloo): the unique id's bootstrap, then per pull the training step, game counts and scale_ok.
    # Everything a caller needs besides the weights (barriers, timing reductions) can go through it too (`ctrl_group`): the
    # process group's own RCCL communicator is then never instantiated -- one communicator (and one proxy thread: ~0.6 of a
    # host core each, scripts/experiments/rank_storage_cpu_probe.py) per rank instead of two
    self.meta_group = self.ctrl_group = dist.new_group(backend='gloo')
    self.comm, err = C.c_void_p(), None
    try:
      self.lib = _abi.load()
      if os.environ.get('MZ_COMM_FAIL', '0')[:1] == '1':      # (tests: the fallback below)
        raise RuntimeError('MZ_COMM_FAIL=1')
      rccl = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')      # PyTorch-ROCm's own copy: ONE HIP runtime per process
      _abi.check(self.lib.mz_comm_load(rccl.encode() if os.path.exists(rccl) else None), 'mz_comm_load')
      uid = (C.c_char * 128)()
      if rank == 0:
        _abi.check(self.lib.mz_comm_unique_id(uid), 'mz_comm_unique_id')
    except Exception as exc:            # (no librccl / no symbol: every rank must still enter the collectives below)
      err, uid = exc, (C.c_char * 128)()
    box = [bytes(uid.raw)]
    dist.broadcast_object_list(box, src=0, group=self.meta_group)      # bootstrap only
    # 'ready to create' (ADVICE r05): a rank whose library / librccl / unique id failed never starts the blocking collective
    # init below, and no other rank does either -- they would sit in ncclCommInitRank for the whole MZ_COMM_TIMEOUT
    ready = torch.tensor([0 if err is not None else 1], dtype=torch.int64)
    dist.all_reduce(ready, op=dist.ReduceOp.MIN, group=self.meta_group)
    if err is None and int(ready[0]) == 0:
      err = RuntimeError('another rank could not load librccl / the library')
    if err is None:
      # ncclCommInitRank is collective and blocking: it runs on a thread of its own, and a rank that has waited MZ_COMM_TIMEOUT
      # seconds (default 120) gives up on it -- every rank then does (the wait is symmetric) and the fallback below takes over
      import threading
      uid = (C.c_char * 128).from_buffer_copy(box[0])
      result = []

      def create():
        try:
          with torch.cuda.device(self.device):
            _abi.check(self.lib.mz_comm_create(rank, world, uid, C.byref(self.comm)), 'mz_comm_create')
          result.append(None)
        except Exception as exc:
          result.append(exc)
      th = threading.Thread(target=create, daemon=True)
      th.start()
      th.join(float(os.environ.get('MZ_COMM_TIMEOUT', '120')))
      if not result:
        # the thread is still inside ncclCommInitRank: creating ANOTHER communicator on this device beside it (the fallback) would
        # race it, and a handle completed later would leak.  The process ends here, non-zero; the launcher ends the other ranks.
        raise SystemExit('RankStorage (rank %d of %d): mz_comm_create (ncclCommInitRank) did not return within MZ_COMM_TIMEOUT = %s s: '
                         'giving up (no fallback beside a half-initialised communicator)' % (rank, world, os.environ.get('MZ_COMM_TIMEOUT', '120')))
      err = result[0]
    # all ranks or none: a rank that could not build its communicator sends every rank back to torch.distributed's collectives
    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.meta_group)
    if int(ok[0]) == 0:
      import sys
      print('RankStorage: the library\'s RCCL communicator is not available on every rank (%s): falling back to '
            'torch.distributed\'s broadcast' % (err if err is not None else 'another rank failed'), file=sys.stderr, flush=True)
      if self.comm:
        self.lib.mz_comm_destroy(self.comm)
        self.comm = None
      self.native, self.ctrl_group = False, None
      self.fallback_reason = str(err) if err is not None else 'another rank failed to create its communicator'
      self._flats = [torch.zeros(n, dtype=torch.float32, device=self.cdev)]
      self._last = 0
      return
    cnt = C.c_int(0)
    if self.lib.mz_comm_count(self.comm, C.byref(cnt)) == 0:
      self.ranks_in_comm = int(cnt.value)
    self.side = torch.cuda.Stream(self.device)
    self._flats = [torch.zeros(n, dtype=torch.float32, device=self.device) for _ in range(2)]
    self._pinned = [torch.zeros(n, dtype=torch.float32).pin_memory() for _ in range(2)] if rank == 0 else None
    self._pinned_np = [t.numpy() for t in self._pinned] if rank == 0 else None
    self._staged = [None, None]                              # event behind the host-to-device copy out of _pinned[k]
    self._consumed = [None, None]                            # event behind the last repack that read _flats[k]
    self._last = 0

  @property
  def flat(self):
    """the buffer the last broadcast filled; the caller's current stream is ordered behind that broadcast (it ran on the side
    stream: a reader that never went through FlatWeights.event -- train.py's rank_weight_sums on a collective-only rank -- would
    otherwise race it)"""
    if self._done_last is not None:
      torch.cuda.current_stream(self.device).wait_event(self._done_last)
    return self._flats[self._last]

  def collective_stats(self):
    """the N > 1 bench line's `collectives` block: which path the weights travel, ranks RCCL reports for the communicator,
    HIP-event time of mz_broadcast_weights on the side stream (mean / max over the pulls whose events have completed)"""
    keep = []
    for e0, e1 in self._bcast_events:
      if e1.query():
        self._bcast_us.append(1e3 * e0.elapsed_time(e1))
      else:
        keep.append((e0, e1))
    self._bcast_events = keep
    us = self._bcast_us
    return {'native_rccl_broadcast': bool(self.native), 'ranks_in_comm': self.ranks_in_comm, 'fallback_reason': self.fallback_reason,
            'broadcasts': int(self.broadcasts), 'broadcast_us': {'mean': float(np.mean(us)), 'max': float(np.max(us)), 'n': len(us)} if us else None}

  def is_ready(self):
    if self.rank == 0:
      return self.call(self.storage, 'is_ready')
    return True                     # the first collective get_weights delivers them

  def _gather_meta(self, meta, group=None):
    dist = _dist()
    gathered = [torch.zeros_like(meta) for _ in range(self.world)]
    dist.all_gather(gathered, meta, group=group)
    self.training_step = int(gathered[0][0])
    if self.rank == 0:
      for r in range(1, self.world):                                   # shared_storage.py:13: per-actor game counts
        self.call(self.storage, 'get_weights', int(gathered[r][1]), r)
    return gathered

  def get_weights(self, games, actor_key):
    dist = _dist()
    step, host = 0, None
    if self.rank == 0:
      weights, step = self.call(self.storage, 'get_weights', games, actor_key)
      host = weights if torch.is_tensor(weights) else self.flatten(weights)
    if not self.native:
      if self.rank == 0:
        self._flats[0].copy_(host.to(self.cdev))
      t0 = time.perf_counter()
      dist.broadcast(self._flats[0], src=0)
      self._bcast_us.append(1e6 * (time.perf_counter() - t0))      # (the fallback paths: host wall time of the call)
      self._gather_meta(torch.tensor([step, games], dtype=torch.int64, device=self.cdev))
      self.broadcasts += 1
      flat = self._flats[0]
      return (flat if flat.device == self.device else flat.to(self.device)), self.training_step
    k = self.broadcasts & 1
    flat, ok = self._flats[k], 1
    if self.rank == 0:
      host = host.detach().to('cpu', torch.float32).reshape(-1)
      ok = int(self.scale_check(host)) if self.scale_check is not None else 1
      if self._staged[k] is not None:
        self._staged[k].synchronize()                # the copy out of this staging slot, two pulls ago
      # (numpy's single-threaded copy: torch's CPU copy_ of 200 k floats wakes its whole intra-op thread pool -- 16 cores busy and
      # 4 ms per pull on a 256-CPU box under a 16-CPU quota, scripts/experiments/rank_storage_cpu_probe.py)
      np.copyto(self._pinned_np[k], host.numpy())
    import ctypes as C
    from . import _abi
    with torch.cuda.stream(self.side):
      if self._consumed[k] is not None:
        self.side.wait_event(self._consumed[k])      # the repack that read this buffer two pulls ago (stream order, no host wait)
      if self.rank == 0:
        flat.copy_(self._pinned[k], non_blocking=True)
        self._staged[k] = torch.cuda.Event()
        self._staged[k].record(self.side)
      # the path's one exchange: RCCL over xGMI, on this side stream (a pair of HIP events around it: collective_stats)
      t0 = torch.cuda.Event(enable_timing=True)
      t0.record(self.side)
      _abi.check(self.lib.mz_broadcast_weights(self.comm, C.c_void_p(flat.data_ptr()), flat.numel(), 0, C.c_void_p(self.side.cuda_stream)),
                 'mz_broadcast_weights')
      done = torch.cuda.Event(enable_timing=True)
      done.record(self.side)
      self._bcast_events.append((t0, done))
      if len(self._bcast_events) > 64:
        self.collective_stats()
    gathered = self._gather_meta(torch.tensor([step, games, ok], dtype=torch.int64), group=self.meta_group)
    self.broadcasts += 1
    self._last, self._done_last = k, done
    return FlatWeights(self, k, flat, int(gathered[0][2]), done), self.training_step

  def close(self):
    if getattr(self, 'native', False) and getattr(self, 'comm', None):
      torch.cuda.synchronize(self.device)
      self.lib.mz_comm_destroy(self.comm)
      self.comm = None


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
    # (no device_id: the group's own RCCL communicator is created lazily -- never, when RankStorage's own communicator carries the
    # weights and its host-side group everything else)
    dist.init_process_group('nccl')
  else:
    dist.init_process_group(backend)
  return rank, world, device, backend
