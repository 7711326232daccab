"""Does a CU mask give a colocated learner its speed back?  The device loop runs on a stream restricted to the first N_ACTOR
CUs (hipExtStreamCreateWithCUMask), the learner step on a stream restricted to the remaining ones."""
import ctypes as C, os, sys, time, threading, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.init()
hip = None
for name in ('libamdhip64.so', os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')):
  try:
    hip = C.CDLL(name); break
  except OSError:
    pass
assert hip is not None


def masked_stream(lo, hi, total=256):
  words = (total + 31) // 32
  mask = (C.c_uint32 * words)()
  for cu in range(lo, hi):
    mask[cu // 32] |= 1 << (cu % 32)
  st = C.c_void_p()
  rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
  assert rc == 0, rc
  return torch.cuda.ExternalStream(st.value)


from model_based_rl_amd.config import make_config
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.learners import Learner
from model_based_rl_amd.replay_buffer import PrioritizedReplay
from model_based_rl_amd.shared_storage import SharedStorage
from model_based_rl_amd.networks import get_network
ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
N_ACTOR = (ENVS + 15) // 16
cfg = make_config(['--environment', 'LunarLander-v2', '--num_simulations', '30', '--seed', '0', '--num_envs', str(ENVS), '--window_size', '400000',
                   '--use_gpu_for', 'actors', 'learner', '--runs_dir', '/tmp/mz_cm', '--run_tag', 'x'])
storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
torch.manual_seed(0)
w = flatten_weights(get_network(cfg, torch.device('cpu')).state_dict())
eng = Engine.from_config(cfg, ENVS)
eng.set_weights(w); eng.selfplay_reset(64, 1.0, stagger=True)
for _ in range(10):
  eng.selfplay_steps(8); buf, n = eng.selfplay_drain(); torch.cuda.synchronize(); replay.ingest_records(buf[:n], n, ENVS)
learner = Learner(cfg, storage, replay)


def loop(n, stream=None):
  ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
  with ctx:
    for _ in range(5):
      learner.update_weights(replay.sample_batch())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      learner.update_weights(replay.sample_batch())
    torch.cuda.synchronize()
  return n / (time.perf_counter() - t0)


print('learner alone: %.1f updates/s; on a stream masked to %d CUs: %.1f' % (loop(60), 256 - N_ACTOR, loop(60, masked_stream(N_ACTOR, 256))))
for masked in (False, True):
  stop = False
  moves = [0]
  def actor():
    st = masked_stream(0, N_ACTOR) if masked else torch.cuda.Stream()
    with torch.cuda.stream(st):
      while not stop:
        eng.selfplay_steps(8); eng.selfplay_drain(); st.synchronize(); moves[0] += 8
  th = threading.Thread(target=actor); th.start(); time.sleep(0.3)
  m0, t0 = moves[0], time.perf_counter()
  ups = loop(40, masked_stream(N_ACTOR, 256) if masked else None)
  rate = (moves[0] - m0) * ENVS / (time.perf_counter() - t0)
  print('%s: learner %.1f updates/s beside an actor with %d envs (%d workgroups) doing %.2f M env-steps/s' %
        ('CU masks (actor: CUs 0..%d, learner: the other %d)' % (N_ACTOR - 1, 256 - N_ACTOR) if masked else 'no masks', ups, ENVS, N_ACTOR, rate / 1e6))
  stop = True; th.join()
