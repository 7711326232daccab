"""development aid: where a k_fcl_heads workgroup spends its time (mz_fcl_heads_profile: s_memtime stamps at the phase
boundaries, workgroup 0 of every head at unroll position 1, and of k_fcl_chain_fwd4's position 2; LunarLander shapes, batch
256, K = 5).  Shader-clock cycles / 100."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import learner_graph_speed as ls
from model_based_rl_amd import _abi
cfg, storage, replay, learner = ls.setup([])
ls.loop(learner, replay, 50)
lib, h = _abi.load(), learner._native.h
_abi.check(lib.mz_fcl_heads_profile(h, 1, None), 'arm')
names = ['inputs -> LDS', 'fc1 products', 'fc1 epilogue + barrier', 'fc2 partials + barrier', 'reduce + barrier', 'loss + barrier',
         'd2 tape + fc2^T products', 'mask + barrier', 'fc1^T partials + barrier', 'reduce + barrier', 'd hidden stored']
acc = np.zeros((3, 11)); cacc = np.zeros(5); kacc = np.zeros(4)
N = 20
for _ in range(N):
  ls.loop(learner, replay, 3)
  torch.cuda.synchronize()
  buf = np.zeros(96, np.uint64)
  _abi.check(lib.mz_fcl_heads_profile(h, 0, buf.ctypes.data_as(C.c_void_p)), 'read')
  st = buf[:48].reshape(3, 16).astype(np.float64)
  acc += np.diff(st[:, :12], axis=1) / 100.0
  cacc += np.diff(buf[48:54].astype(np.float64)) / 100.0
  kacc += np.diff(buf[54:59].astype(np.float64)) / 100.0
tl = buf[59:64].astype(np.float64)
if tl[0] > 0:      # the fused forward launch's timeline (constant 100 MHz clock): microseconds from chain workgroup 0's start
  print('fused forward launch (last sample of %d): chain workgroup 0 ends at %.2f us; the last position\'s value unit of group 0 starts at %.2f, '
        'is past its wait at %.2f, ends at %.2f us' % (N, (tl[1] - tl[0]) / 100.0, (tl[2] - tl[0]) / 100.0, (tl[3] - tl[0]) / 100.0, (tl[4] - tl[0]) / 100.0))
if buf[12] > 0 and tl[0] > 0:      # k_fcl_fb: the backward pass of chain workgroup 0 on the same clock
  print('one-launch forward + backward: chain workgroup 0 has the last position\'s d hidden at %.2f us, ends its backward pass at %.2f us; '
        'later positions it had to poll for (all samples): %d' % ((float(buf[12]) - tl[0]) / 100.0, (float(buf[13]) - tl[0]) / 100.0, int(buf[14])))
  print('  the value unit of group 0 at position K - 1: starts at %.2f, is past its wait at %.2f, ends at %.2f us' % tuple((float(buf[k]) - tl[0]) / 100.0 for k in (45, 46, 47)))
  print('  last heads unit ends at %.2f us, last weight-gradient job at %.2f, last chain workgroup at %.2f' % tuple((float(buf[k]) - tl[0]) / 100.0 for k in (65, 64, 66)))
  arr = [float(buf[k]) for k in (15, 28, 29, 30, 31, 44)]
  print('  positions\' d hidden in hand at (us): ' + ', '.join('p%d %.2f' % (q, (arr[q] - tl[0]) / 100.0) for q in range(5, -1, -1) if arr[q] > 0))
acc /= N
for i, nme in enumerate(names):
  print('%-28s value %6.2f  policy %6.2f  reward %6.2f' % (nme, acc[0, i], acc[1, i], acc[2, i]))
print('%-28s value %6.2f  policy %6.2f  reward %6.2f (x 100 cycles)' % ('total', acc[0].sum(), acc[1].sum(), acc[2].sum()))
if buf[74] > 0:      # backward position 2 of chain workgroup 0 (last sample)
  bst = np.diff(buf[67:75].astype(np.float64)) / 100.0
  for nme, x in zip(['wave 0: LayerNorm backwards', 'requests of the next position', 'barrier', 'fc2^T products + mask + tapes', 'fc1^T partials', 'barrier', 'wave 0: reduce'], bst):
    print('chain bwd position 2: %-36s %6.2f' % (nme, x))
  print('chain bwd position 2: total %6.2f (x 100 cycles)' % bst.sum())
cacc /= N
for nme, x in zip(['fc1 products', 'epilogue + barrier', 'fc2 partials + barrier', 'wave 0: reduce + LayerNorm + tapes', 'barrier'], cacc):
  print('chain fwd position 2: %-36s %6.2f' % (nme, x))
print('chain fwd position 2: total %6.2f (x 100 cycles)' % cacc.sum())
kacc /= N
for nme, x in zip(['start -> requests out, observations in LDS', 'position 0', 'wait for the resident weights', 'positions 1..K'], kacc):
  print('chain fwd kernel: %-44s %7.2f' % (nme, x))
