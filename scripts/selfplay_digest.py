"""SHA-1 of the experience records of K self-play moves (env: K, B; MZ_HIP_LIB selects the library build, MZ_NO_PERSIST /
MZ_SPLIT_F16 the kernel path): two builds or paths that must agree bit for bit print the same digest."""
import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import model_based_rl_amd
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
import types
torch.manual_seed(0)
B, O, A, sims = int(os.environ.get("B", "64")), 8, 4, 30
net = FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace())
eng = Engine(B, O, A, sims, discount=0.997, seed=1, device='cuda')
eng.set_weights(net.get_weights())
eng.selfplay_reset(16, 1.0, True)
print('reset ok', flush=True)
eng.selfplay_steps(int(os.environ.get('K', '1')))
torch.cuda.synchronize()
print('steps ok', flush=True)
rec, n = eng.selfplay_drain()
torch.cuda.synchronize()
r = rec[:n].numpy()
print(n, r[0, 0, :12], r[0, 0, -5:].view(np.int32))
import hashlib
print('digest', hashlib.sha1(r.tobytes()).hexdigest())
