"""The one-replay topology of `train --ranks N` at scale, without N GPUs: N - 1 producer PROCESSES push record chunks (the
device loop's shape: 16 moves x 4096 envs; LunarLander records, or --shape pong: the 192-byte Pong-ram records) through their ShmRing to rank 0's drain thread
(distributed.serve_rings), which feeds THE one native replay with the parallel ingest -- exactly the host path of
train.launch_ranks, with the GPUs replaced by pre-generated chunks.  Reports the records/s the replay accepts when the
producers push as fast as the rings take them (the cap of the layout on this host), and how long a producer waited for
a free slot when it paces itself at one GPU's rate (--pace_ms per chunk).

  python scripts/one_replay_bench.py [--ranks 8] [--chunks 150] [--threads 4] [--pace_ms 0] [--shape lunar|pong]"""
import argparse, json, multiprocessing as mp, os, sys, threading, time, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, O, A, T, CH = 4096, 8, 4, 256, 16
U8 = False
if '--shape' in sys.argv and sys.argv[sys.argv.index('--shape') + 1] == 'pong':      # Pong-ram records: 128 observation bytes packed
  O, A, U8 = 128, 6, True                                                             # into 32 float slots, 6 actions (192 bytes)
OS = (O + 3) // 4 if U8 else O
REC = OS + A + 10


def make_chunks(n, seed, env_base):
  rng = np.random.RandomState(seed)
  t = ((np.arange(env_base, env_base + B, dtype=np.uint32) * np.uint32(2654435761)) >> 8) % T
  ep = np.zeros(B, np.int32)
  out = []
  for _ in range(n):
    rec = rng.standard_normal((CH, B, REC)).astype(np.float32)
    ints = rec[..., OS + A + 5:].view(np.int32)
    for m in range(CH):
      ints[m, :, 0] = 1; ints[m, :, 1] = (t + 1 >= T); ints[m, :, 2] = t; ints[m, :, 3] = np.arange(B); ints[m, :, 4] = ep
      done = t + 1 >= T
      ep += done; t = np.where(done, 0, t + 1)
    out.append(rec)
  return out


def replay_cfg(threads):
  return types.SimpleNamespace(batch_size=256, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(O,), action_space=A, window_size=1 << 21,
                               window_step=None, num_unroll_steps=5, td_steps=10, max_history_length=500, discount=0.997, seed=0,
                               ingest_threads=threads, obs_u8=U8)


def producer(name, rank, nchunks, pace_ms, go, result):
  from model_based_rl_amd import distributed as D
  data = make_chunks(16, rank, rank * B)            # cycled (the replay only needs well-formed games)
  ring = D.ShmRing(name)
  sink = D.RingReplay(ring, replay_cfg(1))          # (as train.launch_ranks: this rank assembles its history slices; MZ_RING_RAW=1: the chunks travel)
  go.wait()
  waited, t_next = 0.0, time.perf_counter()
  for i in range(nchunks):
    if pace_ms:
      t_next += pace_ms * 1e-3
      dt = t_next - time.perf_counter()
      if dt > 0:
        time.sleep(dt)
    t0 = time.perf_counter()
    sink.ingest_records(data[i % len(data)], CH, B)
    waited += time.perf_counter() - t0
  ring.close_producer()
  result.put((rank, waited))
  sink.close()
  ring.release()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--ranks', type=int, default=8)
  ap.add_argument('--chunks', type=int, default=150)
  ap.add_argument('--threads', type=int, default=4)
  ap.add_argument('--pace_ms', type=float, default=0.0)
  ap.add_argument('--json', default=None)
  ap.add_argument('--shape', default='lunar', choices=['lunar', 'pong'])
  ap.add_argument('--gpu_rate', type=float, default=None, help='records/s of one GPU at this shape (default: 10.0 M lunar, 5.4 M pong)')
  a = ap.parse_args()
  from model_based_rl_amd import distributed as D
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  cfg = replay_cfg(a.threads)
  replay = PrioritizedReplay(cfg)
  run = 'mz_onereplay_%d' % os.getpid()
  rings = {r: D.ShmRing('%s_%d' % (run, r), CH, B, REC, slots=4, create=True) for r in range(1, a.ranks)}
  ctx = mp.get_context('spawn')
  go, result = ctx.Event(), ctx.Queue()
  procs = [ctx.Process(target=producer, args=('%s_%d' % (run, r), r, a.chunks, a.pace_ms, go, result)) for r in range(1, a.ranks)]
  for p in procs:
    p.start()
  own = make_chunks(16, 0, 0)
  time.sleep(2.0 + 0.2 * a.ranks)                  # producers generate their data
  stop = threading.Event()
  call = lambda name, *args: getattr(replay, name)(*args)
  lock = threading.Lock()
  def locked_call(name, *args):
    with lock:                                       # (rayshim serialises calls on the replay handle; here: a lock)
      return call(name, *args)
  server = threading.Thread(target=D.serve_rings, args=(rings, locked_call, B, stop, min(a.threads, 4), replay), daemon=True)
  f0 = replay.get_throughput()['frames']
  t0 = time.perf_counter()
  go.set()
  server.start()
  t_next = time.perf_counter()
  for i in range(a.chunks):                          # rank 0's own actor ingests directly
    if a.pace_ms:
      t_next += a.pace_ms * 1e-3
      dt = t_next - time.perf_counter()
      if dt > 0:
        time.sleep(dt)
    locked_call('ingest_records', own[i % len(own)], CH, B, 0)
  server.join(timeout=600)
  replay.size()                                      # (waits for the deferred insertions: the region ends when the tree has them)
  dt = time.perf_counter() - t0
  frames = replay.get_throughput()['frames'] - f0
  waits = dict(result.get(timeout=60) for _ in procs)
  for p in procs:
    p.join()
  for r in rings.values():
    r.release()
  gpu_rate = a.gpu_rate or (5.4e6 if a.shape == 'pong' else 10.0e6)
  out = {'ranks': a.ranks, 'ingest_threads': replay.ingest_threads, 'chunks_per_rank': a.chunks, 'pace_ms': a.pace_ms,
         'records_in': a.ranks * a.chunks * CH * B, 'frames_accepted': frames, 'seconds': dt,
         'records_per_s': a.ranks * a.chunks * CH * B / dt, 'one_gpu_rate_records_per_s': gpu_rate,
         'gpus_worth': a.ranks * a.chunks * CH * B / dt / gpu_rate, 'shape': a.shape, 'record_bytes': REC * 4, 'moves_per_chunk': CH,
         'producer_seconds_in_put': {int(k): round(v, 3) for k, v in sorted(waits.items())},
         'host_cpus_usable': len(os.sched_getaffinity(0))}
  print(json.dumps(out))
  if a.json:
    json.dump(out, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
  main()
