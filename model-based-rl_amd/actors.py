"""Self-play actor with the reference's surface (actors.py:16-182): Actor(actor_key, config, storage,
replay_buffer, state=None) with launch / run_selfplay / play_game / sync_weights / load_state.

One Actor = one GPU = `config.num_envs` environments searched in lock-step (the reference runs one
environment per Ray actor process).  Two paths:
  * synthetic on-device environments (shape-faithful stand-ins for the gym envs that are not installed):
    the whole move loop of actors.py:131-173 runs on the device (Engine.selfplay_steps), experience records
    come back through pinned memory and are ingested by the native replay (replay.ingest_records);
  * host environments (TicTacToe, or any gym-0.x style env): the per-move body is driven from Python,
    batched over the actor's games; with `config.parity_rng` the Dirichlet noise and the action samples
    come from numpy's global stream in the reference's order, which makes a one-environment actor
    reproduce the reference's games move for move (tests/test_gpu_actor.py::test_actor_reproduces_reference_games).
"""
import random
import time
from copy import deepcopy

import numpy as np
import torch

import os

from .engine import Engine, flatten_weights, records_view
from .envs import get_environment
from .logger import Logger
from .networks import get_network


def set_all_seeds(seed):
  """utils.py:136-144: torch s, cuda s+1, random s+2, numpy s+3."""
  if seed is None:
    seed = random.randint(0, 1000)
  torch.manual_seed(seed)
  if torch.cuda.is_available():
    torch.cuda.manual_seed_all(seed + 1)
  random.seed(seed + 2)
  np.random.seed(seed + 3)


def _call(obj, name, *args, **kwargs):
  """obj.name(...) for plain objects, obj.name.remote(...) + get for rayshim/ray handles."""
  fn = getattr(obj, name)
  if hasattr(fn, 'remote'):
    return fn.remote(*args, **kwargs).result()
  return fn(*args, **kwargs)


def selfplay_chunk(config):
  """Moves per launch / drain / ingest chunk of the device loop: what one launch of the persistent search kernel plays
  (mz_selfplay_moves_per_launch: 16 whole moves; 8 until r03_i).  A torch network plays one move per iteration.  Every
  rank -- and the rank without an actor, train._CollectiveOnly -- derives its weight-pull cadence from this number."""
  if getattr(config, 'architecture', 'FCNetwork') != 'FCNetwork':
    return 1
  return max(1, int(getattr(config, 'selfplay_chunk', None) or 16))


def chunk_schedule(total, chunk):
  """`total` moves in equal chunks of at most `chunk` (20 -> 10 + 10, not 16 + 4: every launch of the persistent
  self-play kernel pays its start-up once); total None: `chunk` for ever."""
  if total is None:
    while True:
      yield chunk
  left = int(total)
  if left <= 0:
    return
  even = -(-left // -(-left // chunk))
  while left > 0:
    m = min(even, left)
    yield m
    left -= m


class _RecordPipe(object):
  """Host side of the device self-play loop (actors.py:126-173 for `num_envs` environments at a time): the actor's thread
  only launches -- a chunk of moves whose records the kernels store straight into one of NBUF pinned buffers
  (Engine.selfplay_steps_into; MZ_RECORD_COPY=1: into the device ring, then a D2H copy on a copy stream, the path until
  r05 -- its cross-stream dependency kept a runtime thread spinning, profiles/r05_host_threads.txt); a worker thread waits
  for each chunk's event and hands it over (game statistics of actors.py:99-117, then replay_buffer.ingest_records: the
  native replay releases the GIL and splits the environments of a chunk over its ingest threads) -- up to NBUF - 1 chunks
  behind the GPU, so that one slow ingest does not idle it.  join() returns when every chunk has been handed over."""
  NBUF = 4

  def __init__(self, actor, chunk):
    import queue
    import threading
    eng = actor.engine
    self.actor, self.chunk, self.rec_floats = actor, chunk, eng.rec_floats
    self.pinned = [torch.empty(chunk, eng.B, eng.rec_floats, dtype=torch.float32).pin_memory() for _ in range(self.NBUF)]
    self.events = [torch.cuda.Event() for _ in range(self.NBUF)]
    self.copy_stream = torch.cuda.Stream(actor.device) if os.environ.get('MZ_RECORD_COPY') == '1' else None
    self.free, self.work, self.failed = queue.Queue(), queue.Queue(), []
    for i in range(self.NBUF):
      self.free.put(i)
    self.thread = threading.Thread(target=self._worker, daemon=True)
    self.thread.start()

  @staticmethod
  def _wait(ev):
    # the chunk's records are a few milliseconds out: sleep-poll instead of hipEventSynchronize, which spins a core with
    # or without hipEventBlockingSync (one process per GPU shares the host's cores with seven others)
    while not ev.query():
      time.sleep(0.0002)

  def _worker(self):
    torch.cuda.set_device(self.actor.device)
    while True:
      item = self.work.get()
      if item is None:
        return
      try:
        i, n = item
        self._wait(self.events[i])
        if not self.failed:
          self.actor._hand_over(self.pinned[i], n)
      except BaseException as exc:      # surfaced by submit() / join()
        self.failed.append(exc)
      finally:
        self.free.put(item[0])
        self.work.task_done()

  def submit(self, moves, wait=False):
    """launch `moves` moves into a free pinned buffer; wait: until their records have arrived (a GPU shared in turns)"""
    eng = self.actor.engine
    if self.failed:
      raise self.failed[0]
    i = self.free.get()               # (blocks while the worker is NBUF chunks behind)
    if self.copy_stream is None:
      eng.selfplay_steps_into(self.pinned[i], moves)
      n = moves
      self.events[i].record()
    else:
      eng.selfplay_steps(moves)
      _, n = eng.selfplay_drain(self.pinned[i], moves, copy_stream=self.copy_stream)      # overlaps the next chunk's moves
      self.events[i].record(self.copy_stream)
    if wait:
      self.events[i].synchronize()
    self.work.put((i, n))

  def join(self):
    self.work.join()
    if self.failed:
      raise self.failed[0]

  def close(self):
    self.work.put(None)


class Actor(Logger):

  def __init__(self, actor_key, config, storage, replay_buffer, state=None):
    set_all_seeds(config.seed + actor_key if config.seed is not None else None)
    self.actor_key = actor_key
    self.config = deepcopy(config)
    self.run_tag, self.group_tag = getattr(config, 'run_tag', None) or 'run', getattr(config, 'group_tag', None)
    self.storage, self.replay_buffer = storage, replay_buffer
    if not torch.cuda.is_available():
      raise RuntimeError('GPU was requested but torch.cuda.is_available() is False.')   # actors.py:41
    ids = getattr(config, 'actors_gpu_device_ids', None)
    self.device = torch.device('cuda', ids[actor_key] if ids else torch.cuda.current_device())
    self.num_envs = int(getattr(config, 'num_envs', 1))
    # TicTacToe: the reference's environment object on the host for parity runs (numpy's global stream) and single games;
    # a pool of games runs the same rules on the device (mz_selfplay_set_env, csrc/mz_selfplay.hip.h)
    self.host_env = config.environment == 'TicTacToe' and (bool(getattr(config, 'parity_rng', False)) or self.num_envs == 1)
    # FCNetwork: the engine's own fused HIP kernels.  Any other architecture (MuZeroNetwork / TinyNetwork): the torch
    # network stays in the loop behind the batched external-inference path (actors.py:45-47 is network-agnostic)
    self.torch_net = getattr(config, 'architecture', 'FCNetwork') != 'FCNetwork'
    if self.torch_net:
      from .torch_search import TorchSelfplay
      if self.host_env:
        raise NotImplementedError('host environments are served with FCNetwork; the conv networks take image observations')
      self.network = get_network(config, self.device).eval()
      norm = None
      if getattr(config, 'norm_obs', False):
        lo = torch.tensor(config.obs_range[::2], dtype=torch.float32, device=self.device)
        hi = torch.tensor(config.obs_range[1::2], dtype=torch.float32, device=self.device)
        norm = (lo, hi - lo)
      self.selfplay = TorchSelfplay(config, self.network, self.num_envs, self.device, seed=config.seed or 0,
                                    env_id_offset=actor_key * self.num_envs, norm=norm)
      self.engine = self.selfplay.search.engine
    else:
      # (the device RNG is keyed by (seed, GLOBAL environment id, episode, move): one seed for every actor, so that what an
      # environment plays does not depend on how many actors share the environments -- SURVEY.md s8e)
      self.engine = Engine.from_config(config, self.num_envs, device=self.device,
                                       seed=config.seed or 0, env_id_offset=actor_key * self.num_envs)
    if self.host_env:
      self.environments = [get_environment(config) for _ in range(self.num_envs)]
      for env in self.environments:
        env.seed(config.seed)
    if config.fixed_temperatures:                      # actors.py:49-53
      self.temperature = config.fixed_temperatures[actor_key]
      self.worker_id = 'actors/temp={}'.format(round(self.temperature, 1))
    else:
      self.worker_id = 'actor-{}'.format(actor_key)
    if getattr(config, 'norm_obs', False):
      self.obs_min = np.array(config.obs_range[::2], dtype=np.float32)
      self.obs_range = np.array(config.obs_range[1::2], dtype=np.float32) - self.obs_min
    # index of this actor's first environment among the replay's environments (one replay for all actors, train.py:71-72)
    self.env_base = actor_key * self.num_envs
    self.experiences_collected = 0
    self.training_step = 0
    self.games_played = 0
    self.move_counter = 0
    self.weight_pulls = 0        # weight sets adopted (actors.py:83-85)
    self._game_stats = None
    if state is not None:
      self.load_state(state)
    self._pipe = None           # _RecordPipe of the device loop, kept across launch() calls
    self._stream = None
    self._selfplay_started = False
    self._launched = False
    self.record_tap = None      # (tests: callable(records [n, B, rec] numpy view) on every chunk before it is ingested)
    self.last_run = {}
    self._turns = None
    if getattr(config, 'gpu_turns', False):      # --gpu_turns: an actor and a learner of this process share ONE GPU (gpu_turns.py)
      from . import gpu_turns
      gpu_turns.register(self.device, 'actor')
      self._turns = gpu_turns
    Logger.__init__(self)

  # actors.py:75-79
  def _set_weights(self, weights):
    from .distributed import FlatWeights
    if isinstance(weights, FlatWeights):            # the device buffer an RCCL broadcast fills on the storage's side stream
      torch.cuda.current_stream(self.device).wait_event(weights.event)      # (the stream waits, not the host)
      if self.torch_net:
        from .networks import load_flat
        load_flat(self.network, weights.tensor)
      else:
        self.engine.set_weights(weights.tensor, scale_ok=weights.scale_ok)
      weights.consumed()
      return
    if self.torch_net:
      if torch.is_tensor(weights):                  # the flat buffer of the weight broadcast (distributed.RankStorage)
        from .networks import load_flat
        load_flat(self.network, weights)
      else:
        self.network.load_weights(weights)          # networks.py:36-37
    else:
      self.engine.set_weights(weights)

  def load_state(self, state):
    self.run_tag = os.path.join(str(self.run_tag), 'resumed', '{}'.format(state['training_step']))
    self._set_weights(state['weights'])
    self.training_step = state['training_step']
    self.games_played = state['actor_games'][self.actor_key]

  # actors.py:81-85
  def sync_weights(self, force=False):
    weights, training_step = _call(self.storage, 'get_weights', self.games_played, self.actor_key)
    if training_step != self.training_step or force:
      self._set_weights(weights)
      self.training_step = training_step
      self.weight_pulls += 1

  def _log_games(self, rv):
    """games/{return,length,avg_value,max_value} (actors.py:99-117) from a chunk of experience records [moves, B]: one
    running game per environment; the games that ended in a move are logged as one averaged point at i = games_played.
    Vectorised over the chunk: per environment the chunk is cut into games at its `done` records (segment sums / maxima by
    np.*.reduceat over an environment-major layout with one sentinel column, so that the game still running at the end of
    the chunk is a non-empty segment), the game running at the start continues with the carried-over statistics."""
    done = rv['done'] != 0
    M, B = done.shape
    if self._game_stats is None:
      self._game_stats = {'ret': np.zeros(B), 'len': np.zeros(B), 'sumv': np.zeros(B), 'maxv': np.full(B, -np.inf)}
    st = self._game_stats
    if not done.any():
      st['ret'] += rv['reward'].sum(0, dtype=np.float64); st['len'] += M; st['sumv'] += rv['root_value'].sum(0)
      st['maxv'] = np.maximum(st['maxv'], rv['root_value'].max(0))
      return
    rew = np.zeros((B, M + 1)); val = np.zeros((B, M + 1)); vmx = np.full((B, M + 1), -np.inf)
    rew[:, :M] = rv['reward'].T; val[:, :M] = rv['root_value'].T; vmx[:, :M] = val[:, :M]
    eb, em = np.nonzero(done.T)                       # (environment, move) of every game end, environment-major
    env0 = np.arange(B) * (M + 1)
    starts = np.concatenate((env0, eb * (M + 1) + em + 1))
    starts.sort()
    flat = lambda a: a.reshape(-1)
    ret, sumv, maxv = np.add.reduceat(flat(rew), starts), np.add.reduceat(flat(val), starts), np.maximum.reduceat(flat(vmx), starts)
    length = np.diff(np.append(starts, B * (M + 1))).astype(np.float64)
    first = np.searchsorted(starts, env0)             # every environment's first segment: the game carried in
    last = np.append(first[1:], starts.size) - 1      # ... and its last: the game still running (the sentinel counts as a step)
    ret[first] += st['ret']; sumv[first] += st['sumv']; length[first] += st['len']; maxv[first] = np.maximum(maxv[first], st['maxv'])
    length[last] -= 1
    st['ret'], st['len'], st['sumv'], st['maxv'] = ret[last], length[last], sumv[last], maxv[last]
    closed = np.ones(starts.size, bool)
    closed[last] = False                              # the closed segments come in (environment, move) order, like (eb, em)
    ret, length, sumv, maxv = ret[closed], length[closed], sumv[closed], maxv[closed]
    per_move = np.bincount(em, minlength=M)
    played = self.games_played + np.cumsum(per_move)
    f = max(1, self.config.actor_log_frequency)
    logged = np.flatnonzero((per_move > 0) & (played // f != (played - per_move) // f))
    if logged.size:                                   # one averaged point per move in which games ended
      mean = lambda x: np.bincount(em, weights=x, minlength=M) / np.maximum(per_move, 1)
      stats = (('games/return', mean(ret)), ('games/length', mean(length)), ('games/avg_value', mean(sumv / length)),
               ('games/max_value', mean(maxv)))
      self.log_points([(tag, int(played[m]), v[m]) for m in logged for tag, v in stats])
    self.games_played = int(played[-1])

  def _temperature(self):
    if self.config.fixed_temperatures:
      return self.temperature
    return self.config.visit_softmax_temperature(self.training_step)

  # ---------------------------------------------------------------- host environments
  def play_game(self, game):
    """actors.py:126-176.  `game`: one Game, as the reference calls it, or a list of up to `num_envs` Games searched in
    lock-step (one per environment of this actor); returns when every game is terminal.  Rows of the engine beyond the
    given games repeat the first game's inputs and are ignored."""
    games = list(game) if isinstance(game, (list, tuple)) else [game]
    assert 1 <= len(games) <= self.num_envs, (len(games), self.num_envs)
    cfg, eng, A = self.config, self.engine, self.config.action_space
    temperature = self._temperature()
    live = [True] * len(games)
    pad = [0] * (self.num_envs - len(games))            # engine rows without a game of their own
    while any(live):
      obs = np.stack([np.float32(g.get_observation(-1)).reshape(-1) for g in games])
      if getattr(cfg, 'norm_obs', False):
        obs = (obs - self.obs_min) / self.obs_range
      legal = np.zeros((len(games), A), np.uint8)
      noise = np.zeros((len(games), A)) if cfg.parity_rng else None
      for i, g in enumerate(games):
        acts = np.asarray(g.environment.legal_actions())
        legal[i, acts] = 1
        if cfg.parity_rng:       # mcts.py:59, one draw per move of length #legal
          noise[i, acts] = np.random.dirichlet([cfg.root_dirichlet_alpha] * len(acts))
      to_play = np.array([g.to_play for g in games], np.int8)
      if pad:
        obs, legal, to_play = np.concatenate([obs, obs[pad]]), np.concatenate([legal, legal[pad]]), np.concatenate([to_play, to_play[pad]])
        noise = None if noise is None else np.concatenate([noise, noise[pad]])
      eng.initial_inference(obs)
      eng.root_prepare(to_play, legal, noise, device_rng=not cfg.parity_rng, move=self.move_counter)
      eng.search()
      if cfg.parity_rng and temperature:
        uniform = np.random.random_sample(len(games))   # the draw np.random.choice(n, p=...) consumes (config.py:77)
      else:
        uniform = None if not cfg.parity_rng else np.zeros(len(games))
      if pad and uniform is not None:
        uniform = np.concatenate([uniform, uniform[pad]])
      out = {k: v.cpu().numpy() for k, v in eng.finalize(temperature, uniform, move=self.move_counter).items()}
      self.move_counter += 1
      for i, g in enumerate(games):
        if not live[i]:
          continue
        action = int(out['action'][i])
        if cfg.parity_rng and not temperature:             # config.py:79: numpy's own tie draw
          vc = out['visit_counts'][i][np.flatnonzero(legal[i])]
          action = int(np.flatnonzero(legal[i])[np.random.choice(np.where(vc == vc.max())[0])])
        g.history.errors.append(float(out['error'][i]))
        g.apply(action)
        g.store_search_statistics(out['child_visits'][i], float(out['root_value'][i]))
        self.experiences_collected += 1
        if self.experiences_collected % cfg.weight_sync_frequency == 0:
          self.sync_weights()
        # actors.py:160-169
        if (g.history_idx - g.previous_collect_to) == cfg.max_history_length or g.done or g.terminal:
          overlap = cfg.num_unroll_steps + cfg.td_steps
          if not g.history.dones[g.previous_collect_to - 1]:
            collect_from = max(0, g.previous_collect_to - overlap)
          else:
            collect_from = g.previous_collect_to
          history = g.get_history_sequence(collect_from)
          _call(self.replay_buffer, 'save_history', history, ignore=None if g.done else overlap, terminal=g.terminal)
        if g.step >= cfg.max_steps:
          g.environment.was_real_done = True
          live[i] = False
        if g.terminal:
          live[i] = False

  # ---------------------------------------------------------------- run loop
  def run_selfplay(self, max_moves=None, chunk=None):
    """actors.py:87-124.  max_moves: play this many MORE moves per environment (None: until the learner has reached
    training_steps); the device loop continues where the previous call stopped (episodes, record ring, pipeline)."""
    while not _call(self.storage, 'is_ready'):
      time.sleep(0.05)
    self.sync_weights(force=not self._launched)      # (a later call continues: its predecessor ended with a forced pull)
    self._launched = True
    cfg = self.config
    if self.host_env:
      stop_at = None if max_moves is None else self.move_counter + max_moves
      while self.training_step < cfg.training_steps and (stop_at is None or self.move_counter < stop_at):
        games = [cfg.new_game(env) for env in self.environments]
        self.play_game(games)
        for g in games:                                  # actors.py:99-117
          self.games_played += 1
          if self.games_played % max(1, cfg.actor_log_frequency) == 0:
            self.log_scalar(tag='games/return', value=g.sum_rewards, i=self.games_played)
            self.log_scalar(tag='games/length', value=g.step, i=self.games_played)
            self.log_scalar(tag='games/avg_value', value=g.sum_values / max(1, g.history_idx), i=self.games_played)
            self.log_scalar(tag='games/max_value', value=g.max_value, i=self.games_played)
      return
    if self.torch_net:
      return self._run_selfplay_torch(max_moves)
    # synthetic on-device environments: the launch-ahead pipeline of _RecordPipe
    eng = self.engine
    chunk = selfplay_chunk(cfg) if chunk is None else int(chunk)
    temperature = self._temperature()
    if not self._selfplay_started:
      if getattr(cfg, 'norm_obs', False) or '-ram' in str(cfg.environment):
        # the -ram- environments emit bytes; --norm_obs is applied inside the root kernel (actors.py:134-137)
        norm = getattr(cfg, 'norm_obs', False)
        ram = '-ram' in str(cfg.environment)
        eng.selfplay_set_obs(uint8_obs=ram, obs_min=self.obs_min if norm else None, obs_range=self.obs_range if norm else None,
                             packed=ram and bool(getattr(cfg, 'obs_u8', False)))      # (bytes in the records: a replay with obs_u8)
      if cfg.environment == 'TicTacToe':
        eng.selfplay_set_env('tictactoe')
      eng.selfplay_reset(cfg.episode_length, temperature, stagger=True)
      self._selfplay_temperature = temperature
      self._selfplay_started = True
    elif temperature != self._selfplay_temperature:
      eng.selfplay_set_temperature(temperature)
      self._selfplay_temperature = temperature
    if self._pipe is None or self._pipe.chunk < chunk or self._pipe.rec_floats != eng.rec_floats:
      if self._pipe is not None:
        self._pipe.close()
      self._pipe = _RecordPipe(self, chunk)
    pipe = self._pipe
    sync_every = max(1, cfg.weight_sync_frequency)      # experiences per environment between weight pulls
    t0, moves0, pulls0 = time.perf_counter(), self.move_counter, self.weight_pulls
    for m in chunk_schedule(max_moves, chunk):
      if self.training_step >= cfg.training_steps:
        break
      if self._turns is not None:
        turn = self._turns.turn(self.device)
        with turn:      # (a learner on the same GPU: one chunk of moves per turn, the GPU to ourselves for it)
          pipe.submit(m, wait=turn is not self._turns.NO_TURNS)
      else:
        pipe.submit(m)
      self.move_counter += m
      self.experiences_collected += m * eng.B
      if (self.move_counter // sync_every) != ((self.move_counter - m) // sync_every):
        self.sync_weights()
        # actors.py:128-129: the temperature of the schedule is evaluated at the start of every game; on the device the
        # new value reaches each environment at its next episode start (games in progress keep theirs)
        if self._temperature() != self._selfplay_temperature:
          self._selfplay_temperature = self._temperature()
          eng.selfplay_set_temperature(self._selfplay_temperature)
    pipe.join()                   # every chunk has been logged and ingested: games_played is final
    self.last_run = {'moves': self.move_counter - moves0, 'seconds': time.perf_counter() - t0, 'chunk': chunk,
                     'weight_pulls': self.weight_pulls - pulls0}
    self.sync_weights(force=True)

  def _hand_over(self, buf, n):
    """one chunk of records, on the pipeline's worker thread: the games this actor finished (actors.py:94-99 counts one per
    play_game return; reported to the storage with the next weight pull, actors.py:82, shared_storage.py:12-14) and their
    logged statistics, then the replay (actors.py:169)"""
    eng = self.engine
    view = buf[:n].numpy()
    if self.record_tap is not None:
      self.record_tap(view)
    rv = records_view(view, eng.O, eng.A, obs_u8=eng.obs_packed)
    # a weight set whose clamp-ReLU scale the device could not confirm is POISONED with NaN by the repack (k_relu_scale; the host
    # decided scale_ok on its copy of the weights): such records must not reach the replay unnoticed.  One move's
    # root values: 4096 doubles per chunk
    if n > 0 and not np.isfinite(rv['root_value'][n - 1]).all():
      raise RuntimeError('Actor-%s: non-finite root values in the experience records (move %d): the search ran on a weight set the '
                         'device-side scale check rejected (mz_weights_scale_ok on the host disagreed), or the weights themselves are '
                         'not finite' % (self.worker_id if hasattr(self, 'worker_id') else '?', self.move_counter))
    self._log_games(rv)
    _call(self.replay_buffer, 'ingest_records', buf, n, eng.B, self.env_base)

  def _run_selfplay_torch(self, max_moves=None):
    """The same loop for a torch network (MuZeroNetwork / TinyNetwork, config 5): one move of all environments per
    iteration through torch_search.TorchSelfplay; the move's records go D2H behind it on the same stream (2 ms of a
    1.5-s move at Breakout shapes: a copy stream's cross-stream dependency keeps a runtime thread spinning for the whole
    move, profiles/r05_host_threads.txt) and reach the replay through the same bulk ingest while the next move is searched."""
    cfg, sp = self.config, self.selfplay
    temperature = self._temperature()
    sp.temperature.fill_(temperature)
    sp.set_temperature(temperature)
    dev = [torch.empty(sp.B, sp.rec_floats, dtype=torch.float32, device=self.device) for _ in range(2)]
    pinned = [torch.empty(1, sp.B, sp.rec_floats, dtype=torch.float32).pin_memory() for _ in range(2)]
    events = [torch.cuda.Event(), torch.cuda.Event()]
    sync_every = max(1, cfg.weight_sync_frequency)
    pending, k = None, 0

    def hand_over(p):
      buf, ev = p
      while not ev.query():       # (sleep-poll: hipEventSynchronize spins a core)
        time.sleep(0.0005)
      self._log_games(records_view(buf.numpy(), sp.O, sp.A, obs_u8=sp.obs_u8))
      _call(self.replay_buffer, 'ingest_records', buf, 1, sp.B, self.env_base)

    import contextlib
    stop_at = None if max_moves is None else self.move_counter + max_moves
    while self.training_step < cfg.training_steps and (stop_at is None or self.move_counter < stop_at):
      turn = self._turns.turn(self.device) if self._turns is not None else contextlib.nullcontext()
      with turn:
        sp.play_move(dev[k & 1])
        pinned[k & 1][0].copy_(dev[k & 1], non_blocking=True)
        events[k & 1].record()
        if self._turns is not None and turn is not self._turns.NO_TURNS:
          events[k & 1].synchronize()
      if pending is not None:
        hand_over(pending)
      pending = (pinned[k & 1], events[k & 1])
      k += 1
      self.move_counter += 1
      self.experiences_collected += sp.B
      if self.move_counter % sync_every == 0:
        self.sync_weights()
        if self._temperature() != temperature:
          temperature = self._temperature()
          sp.set_temperature(temperature)
    if pending is not None:
      hand_over(pending)
    self.sync_weights(force=True)

  def launch(self, max_moves=None):
    print('Actor-{} is online on {}.'.format(self.actor_key, self.device))
    # the device loop gets a HIP stream of its own: on the default stream every kernel of a learner sharing the GPU
    # (train.py, --use_gpu_for actors learner) would queue behind whole-moves launches of several milliseconds each
    # (measured: 124 updates/s alone, 3.7 beside an actor on the same stream; scripts/experiments/learner_speed.py)
    # (one stream per actor, kept across launch() calls: the record ring orders its slots behind the drains of this stream)
    if self._stream is None:
      self._stream = torch.cuda.Stream(self.device)
    stream = self._stream
    stream.wait_stream(torch.cuda.current_stream(self.device))
    with torch.inference_mode(), torch.cuda.stream(stream):
      self.run_selfplay(max_moves=max_moves)
    torch.cuda.current_stream(self.device).wait_stream(stream)

  def close(self):
    if self._pipe is not None:
      self._pipe.close()
      self._pipe = None
