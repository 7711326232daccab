"""Parity statistics of the two search kernels against the CPU oracle on the headline shape (4096 trees, 30 simulations,
golden FCNetwork weights): share of trees with identical visit vectors, identical whole trees, worst hidden-state and
root-value deviation -- exact float32 kernel and the opt-in split-f16 kernel side by side."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_search as T
for split in ('0', '1'):
  os.environ['MZ_SPLIT_F16'] = split
  for name, B, sims in (('g1_net_lunar', 4096, 30), ('g1_net_pong', 1024, 50)):
    out, ex, ref = T.run_both(name, B, sims)
    same = np.all(out['visit_counts'] == ref['visit_counts'], axis=1)
    whole = np.all(ex['N'] == ref['tree']['N'], axis=1) & np.all(ex['E'] == ref['tree']['E'], axis=1)
    print('split_f16=%s %-12s identical visit vectors %.2f %%  identical trees %.2f %%  max |hidden - oracle| on those %.2e  '
          'max |root value - oracle| %.2e' % (split, name, 100 * same.mean(), 100 * whole.mean(),
          np.abs(ex['hidden'][whole] - ref['hpool'][whole]).max(), np.abs(out['root_value'] - ref['root_value'])[same].max()))
