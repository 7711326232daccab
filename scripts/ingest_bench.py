"""CPU-only timing of the native replay ingest on synthetic records shaped like the device loop's (bench config)."""
import sys, time, types, numpy as np
sys.path.insert(0, '.')
from model_based_rl_amd.replay_buffer import PrioritizedReplay
B, O, A, T = 4096, 8, 4, 256
cfg = types.SimpleNamespace(batch_size=256, epsilon=0.01, alpha=1.0, beta=1.0, obs_space=(O,), action_space=A, window_size=1 << 21,
                            window_step=None, num_unroll_steps=5, td_steps=10, max_history_length=500, discount=0.997, seed=0)
rp = PrioritizedReplay(cfg)
rng = np.random.RandomState(0)
t = ((np.arange(B, dtype=np.uint32) * np.uint32(2654435761)) >> 8) % T
ep = np.zeros(B, np.int32)
ts = []
for it in range(200):
  rec = rng.standard_normal((8, B, O + A + 10)).astype(np.float32)
  ints = rec[..., O + A + 5:].view(np.int32)
  for m in range(8):
    ints[m, :, 0] = 1; ints[m, :, 1] = (t + 1 >= T); ints[m, :, 2] = t; ints[m, :, 3] = np.arange(B); ints[m, :, 4] = ep
    done = t + 1 >= T
    ep += done; t = np.where(done, 0, t + 1)
  t0 = time.perf_counter(); rp.ingest_records(rec, 8, B); ts.append(time.perf_counter() - t0)
full = np.array(ts) * 1e3
print('slowest calls (index: ms):', ', '.join('%d: %.1f' % (i, full[i]) for i in np.argsort(full)[-6:][::-1]))
ts = np.array(ts[60:]) * 1e3
print('ingest of 8 moves x 4096 envs: median %.2f ms, p90 %.2f, max %.2f ms; frames %d' % (np.median(ts), np.percentile(ts, 90), ts.max(), rp.get_throughput()['frames']))
