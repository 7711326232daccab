"""train.py-shaped driver (reference train.py:62-78): storage + replay + actors + learner wired together through
the thread-backed ray shim, no Ray.  With --selfplay_only the learner is left out and `publish_initial_weights`
stands in for Learner.send_weights at start-up (learners.py:85-86,116).

  python -m model_based_rl_amd.train --environment LunarLander-v2 --num_envs 4096 --num_simulations 30 --seed 0 \
      --max_moves 64
"""
import os
import sys
import time
import types

import numpy as np
import torch

from . import rayshim as ray
from .actors import Actor
from .config import build_parser, Config, env_shapes
from .learners import Learner
from .networks import FCNetwork
from .replay_buffer import PrioritizedReplay
from .shared_storage import SharedStorage


def publish_initial_weights(config, storage):
  from .networks import get_network
  torch.manual_seed(config.seed or 0)
  net = get_network(config, torch.device('cpu'))
  storage.store_weights.remote(net.get_weights(), 0).result()


def share_gpu_in_turns(config, actor_keys):
  """Decided from the DEVICES, not from a flag or the device count: where an Actor and the Learner of this process
  resolve to the same torch.device (actors.py / learners.py pick theirs from --actors_gpu_device_ids / --learner_gpu_device_id,
  else the process's current device) they take turns on it (gpu_turns.py) -- two busy queues do not share an MI355X gracefully,
  both lose ~8 x, and under --ranks N the slowed rank holds every other rank back at each collective weight pull.  Sets
  config.gpu_turns and says so once; --no_gpu_turns keeps them apart-less (measurements).  -> True where turns were switched on."""
  if getattr(config, 'gpu_turns', False) or getattr(config, 'no_gpu_turns', False):
    return False
  if 'learner' not in getattr(config, 'use_gpu_for', []) or not torch.cuda.is_available():
    return False
  cur = torch.cuda.current_device()
  lid = getattr(config, 'learner_gpu_device_id', None)
  learner_dev = lid if lid is not None else cur
  ids = getattr(config, 'actors_gpu_device_ids', None)
  actor_devs = {(ids[k] if ids else cur) for k in actor_keys}
  if learner_dev not in actor_devs:
    return False
  config.gpu_turns = True
  print('train: an actor and the learner of this process both run on cuda:%d: they take turns on it (gpu_turns.py; two busy queues '
        'on one MI355X lose ~8 x each).  Give each its own GPU (--ranks N --dedicated_learner_rank, or --learner_gpu_device_id) to '
        'run them side by side.' % learner_dev, file=sys.stderr)
  return True


def launch(config, max_moves, selfplay_only=False, learner_steps=None, state=None):
  """train.launch (train.py:62-78); state: a checkpoint written by Learner.save_state (--load_state, train.py:130-134): the
  learner resumes weights, optimiser, training step and throughput totals (learners.py:62-70), every actor its weights,
  training step and game count (actors.py:75-79)"""
  ray.init()
  if not selfplay_only:
    share_gpu_in_turns(config, range(config.num_actors))
  storage = ray.remote(SharedStorage).remote(config)
  replay = ray.remote(PrioritizedReplay).remote(config)
  actors = [ray.remote(Actor).remote(k, config, storage, replay, state) for k in range(config.num_actors)]
  prime = int(getattr(config, 'prime_moves', 0) or 0)
  if selfplay_only:
    publish_initial_weights(config, storage)
  else:
    learner = ray.remote(Learner).remote(config, storage, replay, state)
  if prime > 0 and max_moves is not None and max_moves > prime:
    # --prime_moves: the first (staggered, partial) episodes of every environment and the one-time costs (kernel load, pinned
    # buffers) are played before the clock starts, so that the printed rate is the steady state of the loop -- B / episode_length
    # games end per move -- the state the reference's frames_per_second is quoted in (learners.py:94-109)
    ray.get([a.launch.remote(prime) for a in actors])
    max_moves -= prime
  frames0 = ray.get(replay.get_throughput.remote())['frames']
  t0 = time.time()
  workers = [a.launch.remote(max_moves) for a in actors]
  if not selfplay_only:
    workers.append(learner.launch.remote(learner_steps))
  ray.get(workers)
  dt = time.time() - t0
  thr = ray.get(replay.get_throughput.remote())
  thr['env_steps_per_s'] = (thr['frames'] - frames0) / dt
  print('frames accepted by replay: %d (%d after %d priming moves), games: %d, %.2f s -> %.0f env-steps/s' %
        (thr['frames'], thr['frames'] - frames0, prime, thr['games'], dt, thr['env_steps_per_s']))
  if not selfplay_only:            # the reference's own throughput scalars (learners.py:88-113)
    lt = ray.get(learner.get_last_throughput.remote())
    if lt:
      print('learner: %.0f frames/s, %.2f updates/s, replay_ratio %.3g, sample_ratio %.3g' %
            (lt['frames_per_second'], lt['updates_per_second'], lt['replay_ratio'], lt['sample_ratio']))
    thr = dict(thr, learner=lt)
  ray.shutdown()
  return thr


def launch_ranks(config, max_moves, selfplay_only=False, learner_steps=None, state=None):
  """One process per GPU (this function runs in every rank; see distributed.py for what is exchanged where):
  rank 0 = learner + storage + the one replay + actor 0, rank r = actor r.  train.py:62-78 on ranks instead of Ray."""
  import json
  import threading
  from . import distributed as D
  from .networks import flat_size, flatten_state, get_network
  import torch.distributed as dist
  _preflight_ranks(int(os.environ.get('WORLD_SIZE', '1')), config)      # (before this rank touches its GPU)
  rank, world, device, backend = D.init_process_group()
  config.num_actors = world
  config.actors_gpu_device_ids = None                 # every rank's actor runs on the rank's own current device
  B = int(config.num_envs)
  torch_net = config.architecture != 'FCNetwork'
  O, A = int(np.prod(config.obs_space)), int(config.action_space)
  from .actors import selfplay_chunk
  rec, chunk = ((O + 3) // 4 if getattr(config, 'obs_u8', False) else O) + A + 10, selfplay_chunk(config)
  torch.manual_seed(config.seed or 0)
  probe = get_network(config, torch.device('cpu'))
  n_flat = flat_size(probe) if torch_net else sum(v.numel() for v in probe.state_dict().values())
  run_id = 'mz_%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'run'))
  ray.init()
  rings, storage, replay, learner, stop = {}, None, None, None, threading.Event()
  if rank == 0 and not selfplay_only and not (bool(getattr(config, 'dedicated_learner_rank', False)) and world > 1):
    # rank 0 runs actor 0 AND the learner on one GPU: they take turns (share_gpu_in_turns, decided before either exists); without
    # turns rank 0 would run at a fraction of the other ranks' pace and every rank would wait for it at each collective weight pull
    share_gpu_in_turns(config, [0])
  if rank == 0:
    storage = ray.remote(SharedStorage).remote(config)
    if not getattr(config, 'ingest_threads', None) and world > 4:
      # the ONE replay takes every rank's records: 4 ingest threads accept 72 M LunarLander-shaped records/s (7.2 GPUs' worth), 8
      # accept 103 M (scripts/one_replay_bench.py); bounded by the CPUs this process may use
      try:
        cpus = len(os.sched_getaffinity(0))
      except AttributeError:
        cpus = os.cpu_count() or 1
      config.ingest_threads = max(1, min(8, world, cpus - 1))
    replay = ray.remote(PrioritizedReplay).remote(config)
    rings = {r: D.ShmRing('%s_%d' % (run_id, r), chunk, B, rec, slots=4, create=True) for r in range(1, world)}
  from .actors import _call
  from .engine import config_scale_check, flatten_weights
  rstorage = D.RankStorage(rank, world, device, n_flat, storage=storage, storage_call=_call, backend=backend,
                           flatten=flatten_state if torch_net else flatten_weights, scale_check=config_scale_check(config))
  ctrl = rstorage.ctrl_group          # (the storage's host-side group where the weights travel on the library's own communicator)
  dist.barrier(group=ctrl) if ctrl is not None else dist.barrier()      # the rings exist
  if rank == 0:
    actor_replay = replay
    # (the rings' slice blobs go to the replay OBJECT from up to four drain threads: the native handle takes calls from any thread)
    server = threading.Thread(target=D.serve_rings, args=(rings, lambda name, *a: _call(replay, name, *a), B, stop,
                                                          min(4, max(1, len(rings))), getattr(replay, '_obj', replay)), daemon=True)
    server.start()
    workers = []
    if selfplay_only:
      storage.store_weights.remote(probe.get_weights(), 0).result()
    else:
      learner = ray.remote(Learner).remote(config, storage, replay, state)
      workers.append(learner.launch.remote(learner_steps))
  else:
    ring = D.ShmRing('%s_%d' % (run_id, rank))
    actor_replay = D.RingReplay(ring, config)      # (this rank assembles its environments' history slices itself)
  del probe
  dedicated = bool(getattr(config, 'dedicated_learner_rank', False)) and world > 1 and not selfplay_only

  if dedicated and config.environment == 'TicTacToe' and (getattr(config, 'parity_rng', False) or B == 1):
    raise SystemExit('--dedicated_learner_rank: host-environment actors pull weights per game, not per move count')
  if dedicated and rank == 0:
    # rank 0's GPU belongs to the learner alone (an actor beside it takes turns with it -- share_gpu_in_turns -- and so runs at
    # about half the other ranks' pace, holding every rank back at each collective weight pull): rank 0 only joins the
    # collectives, at the cadence the actors' loops enter them (Actor.run_selfplay)
    actor = _CollectiveOnly(rank, config, rstorage, chunk)
  else:
    actor = Actor(rank, config, rstorage, actor_replay, state)
  t0 = time.time()
  actor.launch(max_moves)
  if rank > 0:
    ring.close_producer()
  # what every rank's actor ended up with: the same broadcast buffer, the same training step
  mine = torch.tensor([float(rstorage.flat.double().sum()), float(actor.training_step), float(actor.games_played)],
                      dtype=torch.float64, device=torch.device('cpu') if ctrl is not None else rstorage.cdev)
  every = [torch.zeros_like(mine) for _ in range(world)]
  dist.all_gather(every, mine, group=ctrl)
  summary = None
  if rank == 0:
    server.join(timeout=120)        # (`drained` in the summary says whether the rings were emptied in time)
    ray.get(workers)
    dt = time.time() - t0
    thr = ray.get(replay.get_throughput.remote())
    stats = ray.get(storage.get_stats.remote())
    summary = {'ranks': world, 'frames': thr['frames'], 'games': thr['games'], 'seconds': dt,
               'env_steps_per_s': thr['frames'] / dt, 'training_step': stats['training_step'],
               'actor_games': {int(k): int(v) for k, v in stats['actor_games'].items()},
               'weight_broadcasts': rstorage.broadcasts, 'actor_training_step': actor.training_step,
               'rank_weight_sums': [float(x[0]) for x in every], 'rank_training_steps': [int(x[1]) for x in every],
               'rank_games': [int(x[2]) for x in every],
               'replay_size': ray.get(replay.size.remote()), 'backend': backend, 'rccl_mapped': D.rccl_mapped(),
               'weights_on_device': bool(rstorage.flat.is_cuda), 'ingest_threads': ray.get(replay.get_ingest_threads.remote()),
               'drained': not server.is_alive(), 'dedicated_learner_rank': bool(dedicated)}
    if not selfplay_only:          # the reference's own throughput scalars (learners.py:88-113)
      lt = ray.get(learner.get_last_throughput.remote())
      summary.update({k: lt[k] for k in ('updates_per_second', 'replay_ratio', 'sample_ratio', 'frames_per_second') if k in lt})
    print('MZ_TRAIN_SUMMARY ' + json.dumps(summary), flush=True)
  dist.barrier(group=ctrl) if ctrl is not None else dist.barrier()
  stop.set()
  for ring_ in list(rings.values()) + ([ring] if rank > 0 else []):
    ring_.release()
  if hasattr(actor, 'selfplay'):
    actor.selfplay.close()
  rstorage.close()
  dist.destroy_process_group()
  ray.shutdown()
  return summary


class _CollectiveOnly(object):
  """What a rank without an actor does in `train --ranks N --dedicated_learner_rank`: the weight pulls of
  Actor.run_selfplay (initial, one per weight_sync_frequency moves, final) and nothing else, so that the collective
  RankStorage.get_weights stays in step on every rank.  Each pull blocks until the actor ranks arrive."""

  def __init__(self, rank, config, storage, chunk):
    self.rank, self.config, self.storage, self.chunk = rank, config, storage, chunk
    self.training_step, self.games_played, self.move_counter = 0, 0, 0

  def _sync(self):
    from .actors import _call
    _, self.training_step = _call(self.storage, 'get_weights', self.games_played, self.rank)

  def launch(self, max_moves=None):
    cfg = self.config
    from .actors import _call, chunk_schedule
    while not _call(self.storage, 'is_ready'):
      time.sleep(0.05)
    self._sync()
    sync_every = max(1, cfg.weight_sync_frequency)
    for m in chunk_schedule(max_moves, self.chunk):      # the chunks of Actor.run_selfplay's device loop
      if self.training_step >= cfg.training_steps:
        break
      self.move_counter += m
      if (self.move_counter // sync_every) != ((self.move_counter - m) // sync_every):
        self._sync()
    self._sync()


def _preflight_ranks(n, config):
  """distributed.preflight for `train --ranks n`: devices, /dev/shm for the n - 1 experience rings, host cores -- before any GPU call"""
  from . import distributed as D
  from .actors import selfplay_chunk
  try:
    O, A = int(np.prod(config.obs_space)), int(config.action_space)
    rec = ((O + 3) // 4 if getattr(config, 'obs_u8', False) else O) + A + 10
    need = D.ring_bytes(n, selfplay_chunk(config), int(config.num_envs), rec)
  except (AttributeError, TypeError, ValueError):
    need = 0
  need = int(os.environ.get('MZ_PREFLIGHT_SHM_NEED', '0')) or need
  shared = os.environ.get('MZ_DIST_BACKEND', 'nccl') == 'gloo' or os.environ.get('MZ_SHARED_GPU_OK', '0')[:1] == '1'
  threads = int(getattr(config, 'ingest_threads', 0) or 0) or (min(8, n) if n > 4 else 1)
  return D.preflight(n, shm_need=need, ingest_threads=min(threads, 4), shared_gpu_ok=shared)


def _spawn_ranks(n, argv):
  """`train --ranks N` without a launcher: start `python -m torch.distributed.run` as a child (before anything in
  this process touches a GPU) and hand its exit code on."""
  import socket
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), MASTER_ADDR='127.0.0.1')
  env.setdefault('MZ_RUN_TAG', time.strftime('%Y-%m-%d_%H-%M-%S'))      # one run directory for all ranks
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
         '127.0.0.1', '--master-port', str(port), '-m', 'model_based_rl_amd.train'] + list(argv)
  return subprocess.call(cmd, env=env)


def main(argv=None):
  argv = list(sys.argv[1:] if argv is None else argv)
  p = build_parser()
  p.add_argument('--max_moves', type=int, default=None,
                 help='moves per environment (default 64; with --selfplay_only and no --max_moves: 768 priming + 4096 timed moves, so that '
                      'the printed env-steps/s is the steady state of the loop); negative: until the learner has reached --training_steps')
  p.add_argument('--selfplay_only', action='store_true')
  p.add_argument('--learner_steps', type=int, default=None)
  p.add_argument('--ranks', type=int, default=0,
                 help='one process per GPU over torch.distributed: rank 0 = learner + storage + replay + actor 0')
  p.add_argument('--dedicated_learner_rank', action='store_true',
                 help='with --ranks N: rank 0 runs no actor -- its GPU is the learner\'s alone, ranks 1..N-1 play')
  p.add_argument('--prime_moves', type=int, default=0,
                 help='moves played before the clock of the printed env-steps/s starts (the staggered first episodes); part of --max_moves')
  p.add_argument('--load_state', type=str, default=None,
                 help='resume from a checkpoint written by Learner.save_state (train.py:130-134): its config is the run\'s config; '
                      '--training_steps / --max_moves / --runs_dir given here override it')
  args = vars(p.parse_args(argv))
  max_moves, selfplay_only, learner_steps = args.pop('max_moves'), args.pop('selfplay_only'), args.pop('learner_steps')
  if max_moves is None:
    max_moves = 64
    if selfplay_only:              # `train --selfplay_only --num_envs 4096`: the actors' steady-state rate (what bench.py reports)
      max_moves = 768 + 4096
      if not args.get('prime_moves'):
        args['prime_moves'] = 768
  ranks = args.pop('ranks')
  spawn = bool(ranks) and 'RANK' not in os.environ
  if max_moves is not None and max_moves < 0:
    max_moves = None              # run until the learner has reached --training_steps (actors.py:93)
  load_state = args.pop('load_state')
  state = None
  cfg = Config(args)
  if load_state:                    # train.py:130-134: launch(state['config'], date, state=state)
    state = torch.load(load_state, map_location='cpu', weights_only=False)
    given = {a.lstrip('-').split('=', 1)[0] for a in argv if a.startswith('--')}      # (both `--flag value` and `--flag=value`)
    saved = state['config']
    overridable = ('training_steps', 'runs_dir', 'num_envs', 'stored_before_train', 'save_state_frequency', 'use_gpu_for')
    for k in overridable:
      if k in given:
        setattr(saved, k, getattr(cfg, k))
    ignored = sorted(k for k in given if k not in overridable and k not in ('load_state', 'max_moves', 'selfplay_only', 'learner_steps', 'ranks',
                                                                           'dedicated_learner_rank', 'prime_moves')
                     and getattr(saved, k, None) != getattr(cfg, k, None))
    if ignored:
      print('train --load_state: the checkpoint\'s config is the run\'s config (train.py:130-134); ignored on the command line: %s'
            % ', '.join('--' + k for k in ignored), file=sys.stderr)
    cfg = saved
  cfg.action_space, cfg.obs_space = env_shapes(cfg)
  from .config import obs_are_bytes
  cfg.obs_u8 = obs_are_bytes(cfg)                    # image frames and -ram- observations travel as bytes (records, replay)
  if cfg.seed is None:
    cfg.seed = 0
  if cfg.run_tag is None:           # train.py:83-90: a date-stamped run directory (the launcher's start time under --ranks)
    cfg.run_tag = os.environ.get('MZ_RUN_TAG') or time.strftime('%Y-%m-%d_%H-%M-%S')
  if spawn:                         # `train --ranks N` without a launcher: pre-flight here (one sentence instead of N tracebacks), then the ranks
    _preflight_ranks(ranks, cfg)
    raise SystemExit(_spawn_ranks(ranks, argv))
  if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) >= 1 and ranks:
    return launch_ranks(cfg, max_moves, selfplay_only, learner_steps, state)
  return launch(cfg, max_moves, selfplay_only, learner_steps, state)


if __name__ == '__main__':
  main()
