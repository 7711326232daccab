"""Cycles per phase of a MOVE inside the persistent self-play launch (mz_selfplay_phase_profile), mean of four 16-move launches at 4096
environments; MZ_HIP_LIB selects the library (A/B of two builds: profiles/r05_kernel_experiments.txt).
usage: move_phases.py <obs_dim> <actions> <simulations>"""
import sys, os, json, types
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import model_based_rl_amd
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
O, A, sims = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
eng = Engine(4096, O, A, sims, seed=1234)
eng.set_weights(flatten_weights(FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).state_dict()))
eng.selfplay_reset(256, 1.0, stagger=True)
eng.selfplay_steps(64); eng.selfplay_drain(); torch.cuda.synchronize()
acc = None
for _ in range(4):
    d = eng.selfplay_phase_profile(16); eng.selfplay_drain(); torch.cuda.synchronize()
    acc = d if acc is None else {k: acc[k] + d[k] for k in d}
print(os.environ.get('MZ_HIP_LIB'), {k: round(v / 4) for k, v in acc.items()}, 'total', round(sum(acc.values()) / 4))
