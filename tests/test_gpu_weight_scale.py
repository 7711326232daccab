"""The power-of-two scaling of the search kernel's weight stream (csrc/mz_fused.hip.h "ReLU as a clamp", mz_engine.hip
k_relu_scale): nn.ReLU of the four 512-wide hidden layers (networks.py:70-93, 96-119) runs as a [0, 1] clamp on
activations scaled by 2^-k, with the consuming layers scaled by 2^k.  What is proven here, through the C ABI:
the scale is chosen from the weights and is a power of two; results do not depend on it bit for bit (the same network
written with its hidden layers times 2^j and its out layers times 2^-j -- the same function, nn.ReLU being positively
homogeneous -- gets another k and gives identical trees); weight sets that admit no scale, and roots whose hidden state
comes from outside, run the stand-alone kernels and still agree with the CPU restatement of the reference."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MARGIN = 1e-4


def _state(O, A, seed):
  from model_based_rl_amd.networks import FCNetwork
  torch.manual_seed(seed)
  return {k: v.clone() for k, v in FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).state_dict().items()}


def _rescaled(sd, j):
  """the same function: hidden layers of the four heads the search runs times 2^j, the layers reading them times 2^-j"""
  out = {k: v.clone() for k, v in sd.items()}
  f = float(2.0 ** j)
  for head, last in (('value_head', 'value'), ('policy_head', 'policy'), ('reward_head', 'reward'), ('transition_head', 'out')):
    out[head + '.fc1.weight'] *= f
    out[head + '.fc1.bias'] *= f
    out[head + '.' + last + '.weight'] /= f
  return out


def _search(sd, O, A, sims, B, seed=5):
  from model_based_rl_amd.engine import Engine, flatten_weights
  rng = np.random.RandomState(seed)
  obs = rng.standard_normal((B, O)).astype(np.float32)
  noise = rng.dirichlet([0.25] * A, size=B)
  eng = Engine(B, O, A, sims)
  eng.set_weights(flatten_weights(sd))
  scale = eng.weight_scale()
  eng.initial_inference(obs)
  eng.root_prepare(None, None, noise)
  eng.search()
  fin = {k: v.cpu().numpy() for k, v in eng.finalize(np.ones(B), rng.uniform(size=B)).items()}
  tree = eng.export_tree(hidden=True)
  eng.close()
  return scale, fin, tree, (obs, noise)


@pytest.mark.parametrize('O,A,sims,gain', [(8, 4, 30, 1.0), (128, 6, 50, 1.0), (8, 4, 30, 37.0)])
def test_no_activation_reaches_the_chosen_power_of_two(O, A, sims, gain):
  """The premise of the clamp, checked on what a search actually computes: over every hidden state of every tree (root
  and all expansions) and every action, the largest pre-activation of the four 512-wide hidden layers stays below the
  2^k the engine chose from the weights alone -- with room (the bound is an L1 bound: it cannot be tight on real data)."""
  sd = _state(O, A, 3)
  if gain != 1.0:
    for k in sd:
      if k.endswith('fc1.weight') or k.endswith('fc1.bias') or k.startswith('LN.'):
        sd[k] = sd[k] * (gain if not k.startswith('LN.') else 3.0)
  scale, _, tree, _ = _search(sd, O, A, sims, 256)
  assert scale[3] == 1.0
  two_k = float(scale[2])
  h = tree['hidden'].reshape(-1, 50).astype(np.float64)
  h = h[np.abs(h).sum(1) > 0]                                # (unused pool slots are zero)
  w = {k: v.numpy().astype(np.float64) for k, v in sd.items()}
  worst = 0.0
  for head in ('value_head', 'policy_head'):
    worst = max(worst, np.abs(h @ w[head + '.fc1.weight'].T + w[head + '.fc1.bias']).max())
  for head in ('reward_head', 'transition_head'):
    W = w[head + '.fc1.weight']
    base = h @ W[:, :50].T + w[head + '.fc1.bias']
    for a in range(A):
      worst = max(worst, np.abs(base + W[:, 50 + a]).max())
  print('largest hidden pre-activation %.3f, chosen 2^k = %g' % (worst, two_k))
  assert worst < 0.9 * two_k and two_k <= 64 * max(worst, 1.0)


@pytest.mark.parametrize('O,A,sims', [(8, 4, 30), (128, 6, 50)])
def test_results_do_not_depend_on_the_scale(O, A, sims):
  sd = _state(O, A, 3)
  s0, f0, t0, _ = _search(sd, O, A, sims, 512)
  assert s0[3] == 1.0 and s0[1] * s0[2] == 1.0 and s0[2] >= 1.0 and np.log2(s0[2]) == np.round(np.log2(s0[2]))
  for j in (3, -2, 9):
    s1, f1, t1, _ = _search(_rescaled(sd, j), O, A, sims, 512)
    assert s1[3] == 1.0
    assert s1[2] == s0[2] * 2.0 ** j or j < 0                # the bound follows the activations (never below 2^0)
    for k in ('N', 'E', 'W', 'P', 'R', 'minmax', 'hidden'):
      assert np.array_equal(t0[k].view(np.uint8), t1[k].view(np.uint8)), (j, k)
    for k in f0:
      assert np.array_equal(f0[k].view(np.uint8), f1[k].view(np.uint8)), (j, k)


def _oracle(sd, O, A, sims, obs, noise):
  from oracle import oracle as orc
  from model_based_rl_amd.engine import flatten_weights
  B = obs.shape[0]
  t = orc.Trees(orc.tree_cfg(A, sims, False, (None, None), 0.997), B)
  net = orc.FCNet({k: v.numpy() for k, v in sd.items()}, O, A)
  t.search_fc(net, obs, np.ones(B, np.int8), np.ones((B, A), np.uint8), noise, 0.25)
  return t.export(), t.margin()


def test_weights_that_admit_no_scale_run_the_stand_alone_kernels():
  """One absurd weight (1e30 in a hidden layer's unused corner: a column no input ever excites is still part of the
  bound): no power of two below 2^40 covers the bound, the engine says so and searches with the stand-alone kernels --
  same trees as the CPU restatement wherever its decisions were not near-ties; the fused-only entry points refuse."""
  from model_based_rl_amd.engine import Engine, flatten_weights
  O, A, sims, B = 8, 4, 30, 256
  sd = _state(O, A, 3)
  sd['reward_head.fc1.weight'][5, 2] = 1e30
  sd['reward_head.reward.weight'][:, 5] = 0.0               # (its activation reaches nothing: the function stays tame)
  scale, fin, tree, (obs, noise) = _search(sd, O, A, sims, B)
  assert scale[3] == 0.0 and scale[1] == 1.0 and scale[2] == 1.0
  ref, margin = _oracle(sd, O, A, sims, obs, noise)
  wide = margin > MARGIN
  same = np.all(tree['N'] == ref['N'], axis=1)
  assert wide.sum() > 0.8 * B and np.all(same[wide]), (wide.sum(), (wide & ~same).sum())
  eng = Engine(B, O, A, sims)
  eng.set_weights(flatten_weights(sd))
  eng.initial_inference(obs)
  eng.root_prepare(None, None, noise)
  with pytest.raises(RuntimeError, match='fused kernel not in use'):
    eng.search_timed()
  # ... and the next tame weight set switches back
  eng.set_weights(flatten_weights(_state(O, A, 3)))
  assert eng.weight_scale()[3] == 1.0
  eng.initial_inference(obs)
  eng.root_prepare(None, None, noise)
  eng.search_timed()
  eng.close()


def test_self_play_loop_with_weights_that_admit_no_scale():
  """the device self-play loop on such a weight set: per-step launches of the stand-alone kernels, same records as the
  CPU restatement's search on the loop's own observations and noise (margin rule)"""
  from model_based_rl_amd.engine import Engine, flatten_weights, records_view
  from oracle import oracle as orc
  O, A, sims, B = 8, 4, 30, 64
  sd = _state(O, A, 3)
  sd['transition_head.fc1.weight'][7, 1] = -3e29
  sd['transition_head.out.weight'][:, 7] = 0.0
  eng = Engine(B, O, A, sims, seed=77)
  eng.set_weights(flatten_weights(sd))
  assert eng.weight_scale()[3] == 0.0
  eng.selfplay_noise_log(True)
  eng.selfplay_reset(32, 1.0, stagger=True)
  eng.selfplay_steps(3)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  rv = records_view(buf[:n].numpy(), O, A)
  noise = eng.selfplay_noise(2)
  eng.close()
  assert n == 3
  t = orc.Trees(orc.tree_cfg(A, sims, False, (None, None), 0.997), B)
  net = orc.FCNet({k: v.numpy() for k, v in sd.items()}, O, A)
  t.search_fc(net, rv['obs'][2], np.ones(B, np.int8), np.ones((B, A), np.uint8), noise, 0.25)
  _, cv, _, _ = t.finalize(np.ones(B), np.zeros(B))
  wide = t.margin() > MARGIN
  same = np.all(rv['child_visits'][2] == cv.astype(np.float32), axis=1)
  assert wide.sum() > 0.7 * B and np.all(same[wide]), (wide.sum(), (wide & ~same).sum())


def test_root_hidden_from_outside_runs_the_stand_alone_kernels():
  """mz_root_load with a hidden state that is no LayerNorm output -- built so that one hidden unit of the reward head
  exceeds the 2^k the weight set's scale was chosen for (a clamp at 2^k would cut it off): the bound does not cover such
  a root, so mz_search must not run the clamp kernel on it.  Compared with the CPU restatement run from the same root."""
  from model_based_rl_amd.engine import Engine, flatten_weights
  from oracle import oracle as orc
  O, A, sims, B = 8, 4, 30, 128
  sd = _state(O, A, 3)
  probe = Engine(B, O, A, sims)
  probe.set_weights(flatten_weights(sd))
  two_k = float(probe.weight_scale()[2])
  probe.close()
  rng = np.random.RandomState(9)
  row = sd['reward_head.fc1.weight'][0, :50].numpy()
  c = 1.5 * two_k / row[row > 0].sum()
  hidden = (c * (row > 0)[None, :] * rng.uniform(0.9, 1.1, (B, 50))).astype(np.float32)
  slack = float(sd['reward_head.fc1.weight'][0, 50:].abs().max()) + abs(float(sd['reward_head.fc1.bias'][0]))
  assert (hidden @ row).min() - slack > 1.2 * two_k         # that unit's activation, whatever the action
  noise = rng.dirichlet([0.25] * A, size=B)
  net = orc.FCNet({k: v.numpy() for k, v in sd.items()}, O, A)
  w = {k: v.numpy() for k, v in sd.items()}
  hid1 = np.maximum(hidden @ w['policy_head.fc1.weight'].T + w['policy_head.fc1.bias'], 0)
  logits = (hid1 @ w['policy_head.policy.weight'].T + w['policy_head.policy.bias']).astype(np.float32)
  value = np.zeros(B, np.float32)                           # (the root's value enters no tree decision)
  eng = Engine(B, O, A, sims)
  eng.set_weights(flatten_weights(sd))
  assert eng.weight_scale()[3] == 1.0
  eng.root_load(value, logits, hidden)
  eng.root_prepare(None, None, noise)
  eng.search()
  tree = eng.export_tree()
  eng.root_load(value, logits, hidden)
  eng.root_prepare(None, None, noise)
  with pytest.raises(RuntimeError, match='fused kernel not in use'):
    eng.search_timed()
  eng.close()
  # MCTS.run from that root, step by step (mcts.py:78-102)
  t = orc.Trees(orc.tree_cfg(A, sims, False, (None, None), 0.997), B)
  t.root_expand(np.ones(B, np.int8), logits, np.ones((B, A), np.uint8))
  t.add_noise(noise, 0.25)
  hp = np.zeros((B, sims + 1, 50), np.float32)
  hp[:, 0] = hidden
  for s_ in range(sims):
    _, pslot, act, _ = t.select()
    h, r, v, lg = net.recurrent(hp[np.arange(B), pslot], act)
    hp[:, s_ + 1] = h
    t.expand_backup(v, r, lg)
  ref, margin = t.export(), t.margin()
  wide = margin > MARGIN
  same = np.all(tree['N'] == ref['N'], axis=1)
  assert wide.sum() > 0.7 * B and np.all(same[wide]), (wide.sum(), (wide & ~same).sum())


def test_async_weight_pull_equals_the_synchronous_one_and_does_not_wait():
  """mz_set_weights_async (what Engine.set_weights does with host weights; reference actors.py:81-85): the same packed stream
  and scale as mz_set_weights, bit for bit -- and its host side does not wait for the moves queued before it: with ~20 ms of
  self-play launches in the queue the call returns in a fraction of that, the moves queued AFTER it play with the new weights."""
  import time
  from model_based_rl_amd.engine import Engine, flatten_weights
  O, A, sims, B = 8, 4, 30, 4096
  sd0, sd1 = _state(O, A, 0), _state(O, A, 1)
  obs = np.random.RandomState(0).standard_normal((B, O)).astype(np.float32)

  def outputs(eng):
    eng.initial_inference(obs)
    return [x.cpu().numpy() for x in eng.root_outputs()]

  a, b = Engine(B, O, A, sims, seed=3), Engine(B, O, A, sims, seed=3)
  a.set_weights(flatten_weights(sd1), sync=True)
  b.set_weights(sd0)
  b.selfplay_reset(64, 1.0, stagger=True)
  pinned = torch.empty(48, B, b.rec_floats, dtype=torch.float32).pin_memory()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    b.selfplay_steps(16)                # ~6.5 ms of GPU work each
  t1 = time.perf_counter()
  b.set_weights(sd1)                    # host weights: flatten, mz_weights_scale_ok, mz_set_weights_async
  t2 = time.perf_counter()
  torch.cuda.synchronize()
  t3 = time.perf_counter()
  queued_ms, call_ms, gpu_ms = 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0)
  assert gpu_ms > 12.0, gpu_ms                           # the launches really were in flight ...
  assert call_ms < 0.35 * gpu_ms, (call_ms, gpu_ms)      # ... and the pull did not wait for them
  b.selfplay_drain(pinned, 48)
  torch.cuda.synchronize()
  for x, y in zip(outputs(a), outputs(b)):
    assert np.array_equal(x, y)
  assert np.array_equal(a.weight_scale(), b.weight_scale()) and b.weight_scale()[3] == 1.0
  # a device source with the host's decision riding along (the RCCL path, distributed.RankStorage): same result
  flat0 = flatten_weights(sd0)
  dev = flat0.cuda()
  b.set_weights(dev, scale_ok=b.weights_scale_ok(flat0))
  a.set_weights(flat0, sync=True)
  for x, y in zip(outputs(a), outputs(b)):
    assert np.array_equal(x, y)
  # ... and the call with a device source is capturable: replaying the graph after the buffer changed repacks the new weights
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    b.set_weights(dev, scale_ok=1)      # (warm-up on the capture stream)
  torch.cuda.current_stream().wait_stream(side)
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    b.set_weights(dev, scale_ok=1)
  dev.copy_(flatten_weights(sd1))
  g.replay()
  a.set_weights(flatten_weights(sd1), sync=True)
  for x, y in zip(outputs(a), outputs(b)):
    assert np.array_equal(x, y)
  a.close(); b.close()


def test_async_pull_of_weights_that_admit_no_scale():
  """the host's decision (mz_weights_scale_ok = 0) routes the engine to the stand-alone kernels exactly like the synchronous
  pull's read back does"""
  from model_based_rl_amd.engine import Engine, flatten_weights
  O, A, sims, B = 8, 4, 12, 64
  sd = _state(O, A, 2)
  bad = {k: v.clone() for k, v in sd.items()}
  bad['reward_head.fc1.weight'][5, 2] = 1e30
  eng = Engine(B, O, A, sims)
  eng.set_weights(sd)
  assert eng.search_kernel_info()['kind'] == 'fused' and eng.weight_scale()[3] == 1.0
  eng.set_weights(bad)
  assert eng.search_kernel_info()['kind'] == 'standalone' and eng.weight_scale()[3] == 0.0
  eng.set_weights(sd)
  assert eng.search_kernel_info()['kind'] == 'fused' and eng.weight_scale()[3] == 1.0
  eng.close()
