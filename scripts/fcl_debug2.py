"""development aid: locate the first tape of mz_fcl_step that differs from PyTorch on the batch where the gradients differ"""
import os, sys, tempfile, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import model_based_rl_amd
from model_based_rl_amd import _abi
from model_based_rl_amd.config import make_config
from model_based_rl_amd.learners import Learner, _GraphedUpdate
from tests.test_learner import _random_batch, Sink

tmp = tempfile.mkdtemp()
cfg = make_config(['--environment', 'LunarLander-v2', '--seed', '1', '--batch_size', '64', '--use_gpu_for', 'actors', 'learner',
                   '--runs_dir', os.path.join(tmp, 'n'), '--run_tag', 'x'])
cfg.obs_space, cfg.action_space = (8,), 4
s = Sink()
a = Learner(cfg, s, s)
rng = np.random.default_rng(11)
K, bs, A = 5, 64, 4
def tape(which):
  lib = _abi.load()
  n = lib.mz_fcl_read_tape(a._native.h, which, None, 0)
  out = np.empty(n, np.float32)
  assert lib.mz_fcl_read_tape(a._native.h, which, out.ctypes.data_as(C.c_void_p), n) == n
  return out
for step in range(3):
  h = _random_batch(rng, 64, 5, 8, 4)
  batch = ((h['obs'], h['act'], (h['t_rew'], h['t_val'], h['t_pol'])), list(range(64)), h['w'])
  if step == 2:
    dev = [torch.from_numpy(h[k]).to(a.device) for k in _GraphedUpdate.ORDER]
    obs, act, t_rew, t_val, t_pol, w = dev
    net = a.network
    with torch.no_grad():
      onehot = torch.nn.functional.one_hot(act, A).float()
      hh = net.representation(obs); hs = [hh]
      for i in range(K):
        hh = torch.relu(net.LN(net.transition_head(torch.cat((hh, onehot[:, i]), 1)))); hs.append(hh)
      keep = a._loss_dev.clone()
      a._native.step(*dev, no_update=True)
      a._loss_dev.copy_(keep)
      T_h = tape(4).reshape(K + 1, 64, bs); T_a1 = tape(7).reshape(3, K + 1, 512, bs); T_d2 = tape(8).reshape(3, K + 1, 64, bs)
      T_d1 = tape(9).reshape(3, K + 1, 512, bs); T_dH = tape(10).reshape(3, K + 1, 64, bs)
      g = ((1.0 / K) / bs * w).float()
      for p in range(K + 1):
        hp = hs[p]
        print('p', p, 'h diff %.2g' % np.abs(T_h[p, :50].T - hp.cpu().numpy()).max())
        a1 = torch.relu(net.policy_head.fc1(hp))
        lg = net.policy_head.policy(a1)
        sm = torch.softmax(lg, 1)
        t = t_pol[:, p]
        d2 = g[:, None] * (sm * t.sum(1, keepdim=True) - t)
        d1 = (d2 @ net.policy_head.policy.weight) * (a1 > 0)
        dh = d1 @ net.policy_head.fc1.weight
        for nm, T, ref in (('a1', T_a1[1, p].T, a1), ('d2', T_d2[1, p, :A].T, d2), ('d1', T_d1[1, p].T, d1), ('dH', T_dH[1, p, :50].T, dh)):
          ref = ref.cpu().numpy()
          d = np.abs(T - ref)
          r, c = np.unravel_index(d.argmax(), d.shape)
          print('   %s max diff %.3g (ref max %.3g) at row %d col %d: got %.6g want %.6g; rows differing > 1e-6*max: %s' %
                (nm, d.max(), np.abs(ref).max(), r, c, T[r, c], ref[r, c], np.where(d.max(1) > 1e-5 * np.abs(ref).max())[0][:10]))
  a.update_weights(batch)
