"""Oracle replay ingest (priorities + SumTree) vs the reference's PrioritizedReplay goldens
(replay_buffer.py:19-40, 110-122), bit-exact."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as orc

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'g3_game_*.npz')))


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_ingest(path):
  g = np.load(path)
  cap = int(g['max_capacity'])
  tree = orc.SumTree(cap, cap)
  frames = games = 0
  for k in range(int(g['n_flushes'])):
    meta = g['flush_meta'][k]
    ignore, terminal = int(meta[3]), int(meta[4])
    errors = g['flush%d_errors' % k]
    if ignore >= 0:
      errors = errors[:-ignore] if ignore > 0 else errors[:0]   # python's errors[:-0] == []
    pri = orc.priorities(errors, float(g['epsilon']), float(g['per_alpha']))
    tree.add(pri)
    frames += len(pri); games += terminal
    assert tree.total == float(g['flush%d_total_priority' % k])
    assert tree.num_memories == int(meta[5]) and frames == int(meta[6]) and games == int(meta[7])
  assert np.array_equal(tree.leaves(tree.num_memories), g['replay_leaves'])
  # stratified draws -> leaves (replay_buffer.py:136-142)
  for draw, idx in zip(g['sample_draws'], g['sample_idxs']):
    assert tree.get_leaf(float(draw)) == int(idx)
