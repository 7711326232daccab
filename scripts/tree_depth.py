"""mean / max depth of the leaves expanded by one search (random-init weights), per shape"""
import sys, types, numpy as np, torch
sys.path.insert(0, '.')
from model_based_rl_amd.engine import Engine
from model_based_rl_amd.networks import FCNetwork
for O, A, SIMS, two in ((8, 4, 30, False), (128, 6, 50, False), (9, 9, 30, False), (9, 9, 30, True)):
  torch.manual_seed(0)
  net = FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).eval()
  B = 1024
  eng = Engine(B, O, A, SIMS, seed=1, **(dict(two_players=True, known_bounds=(-1.0, 1.0), discount=1.0) if two else {}))
  eng.set_weights(net.state_dict())
  eng.initial_inference(torch.randn(B, O, device='cuda'))
  eng.root_prepare(None, None, None, device_rng=True, move=0)
  eng.search()
  E = eng.export_tree()['E']            # [B, NN] expansion index of every node (-1 = leaf)
  NN = E.shape[1]
  depth_of_e = np.zeros((B, SIMS + 2), np.int32)     # depth of the node with expansion index e
  d = []
  for b in range(B):
    # node n = 1 + e_parent * A + a  ->  parent's expansion index = (n - 1) // A
    for n in np.flatnonzero(E[b] >= 0):
      if n == 0: continue
      depth_of_e[b, E[b, n]] = depth_of_e[b, (n - 1) // A] + 1 if False else 0
    # expansion indices are assigned in order, so parents precede children: one ordered pass
    order = sorted((int(E[b, n]), int(n)) for n in np.flatnonzero(E[b] >= 0) if n != 0)
    dep = {0: 0}
    for e, n in order:
      dep[e] = dep[(n - 1) // A] + 1
    d += [v for k, v in dep.items() if k != 0]
  d = np.array(d)
  print('obs %d actions %d sims %d: mean leaf depth %.2f, max %d, levels descended per simulation (mean) %.2f' % (O, A, SIMS, d.mean(), d.max(), d.mean()))
  eng.close()
