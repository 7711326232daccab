"""SHA-1 of the experience records of K self-play moves (env: K moves, B envs, O obs dim, A actions, SIMS; MZ_HIP_LIB selects
the library build, MZ_NO_PERSIST / MZ_SPLIT_F16 the kernel path): two builds or paths that must agree bit for bit print
the same digest.  Long runs (K in the thousands) are the soak test of a kernel change.  INTO=1: the records are stored by
the kernels straight into pinned memory (Engine.selfplay_steps_into, what the Actor uses) instead of ring + drain."""
import hashlib, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import model_based_rl_amd  # noqa: F401
from model_based_rl_amd.engine import Engine
from model_based_rl_amd.networks import FCNetwork
torch.manual_seed(0)
g = lambda k, d: int(os.environ.get(k, d))
B, O, A, sims, K = g('B', 64), g('O', 8), g('A', 4), g('SIMS', 30), g('K', 1)
net = FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace())
eng = Engine(B, O, A, sims, discount=0.997, seed=1, device='cuda')
eng.set_weights(net.get_weights())
if O > 64:
  eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0])
eng.selfplay_reset(16, 1.0, True)
h = hashlib.sha1()
done = 0
into = torch.empty(32, B, eng.rec_floats).pin_memory() if g('INTO', 0) else None
while done < K:
  k = min(32, K - done)
  if into is not None:
    eng.selfplay_steps_into(into, k)
    rec, n = into, k
  else:
    eng.selfplay_steps(k)
    rec, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  h.update(rec[:n].numpy().tobytes())
  done += k
print('moves', done, 'moves_per_launch', eng.selfplay_moves_per_launch(), 'digest', h.hexdigest())
