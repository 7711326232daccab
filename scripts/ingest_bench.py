"""CPU-only timing of the native replay ingest on synthetic records shaped like the device loop's (bench config: 8 moves x
4096 envs per chunk, LunarLander shapes), for 1 / 2 / 4 / 8 ingest threads; every thread count must leave the same tree.

  python scripts/ingest_bench.py [--gap_ms 3.3] [--json out.json]      (--gap_ms: idle time between calls, as behind one GPU)"""
import json, sys, time, types, numpy as np
sys.path.insert(0, '.')
from model_based_rl_amd.replay_buffer import PrioritizedReplay
B, O, A, T = 4096, 8, 4, 256
CHUNKS, WARM = 200, 60
cfg = types.SimpleNamespace(batch_size=256, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(O,), action_space=A, window_size=1 << 21,
                            window_step=None, num_unroll_steps=5, td_steps=10, max_history_length=500, discount=0.997, seed=0)


def chunks():
  rng = np.random.RandomState(0)
  t = ((np.arange(B, dtype=np.uint32) * np.uint32(2654435761)) >> 8) % T
  ep = np.zeros(B, np.int32)
  for it in range(CHUNKS):
    rec = rng.standard_normal((8, B, O + A + 10)).astype(np.float32)
    ints = rec[..., O + A + 5:].view(np.int32)
    for m in range(8):
      ints[m, :, 0] = 1; ints[m, :, 1] = (t + 1 >= T); ints[m, :, 2] = t; ints[m, :, 3] = np.arange(B); ints[m, :, 4] = ep
      done = t + 1 >= T
      ep += done; t = np.where(done, 0, t + 1)
    yield rec


out, ref = {}, None
data = list(chunks())          # generated up front: the timed calls arrive like the device's chunks do, not 20 ms apart
gap = float(sys.argv[sys.argv.index('--gap_ms') + 1]) * 1e-3 if '--gap_ms' in sys.argv else 0.0
for threads in (1, 2, 4, 8):
  rp = PrioritizedReplay(cfg)
  rp.set_ingest_threads(threads)
  ts = []
  for i, rec in enumerate(data):
    if i == WARM:
      rp.size(); t_all = time.perf_counter()          # (size() waits for the deferred insertions)
    if gap:
      time.sleep(gap)
    t0 = time.perf_counter(); rp.ingest_records(rec, 8, B); ts.append(time.perf_counter() - t0)
  rp.size()
  sustained = (CHUNKS - WARM) * 8 * B / (time.perf_counter() - t_all - gap * (CHUNKS - WARM))
  ts = np.array(ts[WARM:]) * 1e3
  sig = (rp.tree.total_priority, rp.size(), rp.get_throughput()['frames'], float(rp.tree.leaves(1 << 16).sum()))
  ref = ref or sig
  assert sig == ref, (threads, sig, ref)
  out[threads] = dict(median_ms=float(np.median(ts)), p90_ms=float(np.percentile(ts, 90)), max_ms=float(ts.max()),
                      records_per_s=float(sustained))
  print('%d ingest thread(s): a call of 8 moves x 4096 envs returns in median %.2f ms, p90 %.2f, max %.2f ms; sustained incl. the '
        'deferred tree insertion: %.1f M records/s' %
        (threads, out[threads]['median_ms'], out[threads]['p90_ms'], out[threads]['max_ms'], out[threads]['records_per_s'] / 1e6))
print('same tree total / size / frames / leaf sum for every thread count:', ref)
if '--json' in sys.argv:
  json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)
