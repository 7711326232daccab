"""BASELINE.json configs[4]: the batched external-inference actor path for torch networks (MuZeroNetwork /
TinyNetwork through PyTorch-ROCm behind mz_select / mz_expand_backup; reference actors.py:139-145, mcts.py:82-99)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_convnets import G, NAMES, build, forward

pytestmark = pytest.mark.gpu


def make_cfg(A, sims, obs_space=(4, 96, 96), **kw):
  d = dict(action_space=A, num_simulations=sims, two_players=False, known_bounds=(None, None), discount=0.997,
           pb_c_base=19652, pb_c_init=1.25, init_value_score=0.0, root_dirichlet_alpha=0.25,
           root_exploration_fraction=0.25, obs_space=obs_space, episode_length=4, seed=3)
  d.update(kw)
  return types.SimpleNamespace(**d)


def ulp_diff(a, b):
  a = np.ascontiguousarray(a, np.float64).view(np.int64); b = np.ascontiguousarray(b, np.float64).view(np.int64)
  return np.abs(a - b)


# Tolerance of the GPU forward passes (MIOpen float32 convolutions) against the reference's CPU outputs: north_star's 1e-5
# on hidden states and logits (measured on MI355X through all 38 convolution layers + the min-max rescaling: <= 1.4e-6
# hidden, <= 2.7e-7 logits); the value / reward scalars are allowed one step of Config.inverse_transform's float32
# staircase (tests/test_oracle_net.py; measured <= 3.6e-7).
TOL = {'hidden': 1e-5, 'logits': 1e-5, 'scalar': 2e-4}


@pytest.mark.parametrize('name', NAMES)
def test_forward_matches_reference_gpu(name):
  g = np.load(os.path.join(G, name + '.npz'))
  got = forward(build(g, 'cuda'), g, 'cuda')
  worst = {}
  for k, v in got.items():
    kind = 'hidden' if 'hidden' in k else ('logits' if 'logits' in k else 'scalar')
    worst[k] = float(np.abs(v - g[k]).max())
  print(name, {k: '%.2g' % v for k, v in worst.items()})
  for k, v in worst.items():
    kind = 'hidden' if 'hidden' in k else ('logits' if 'logits' in k else 'scalar')
    assert v <= TOL[kind], (k, v)


@pytest.mark.parametrize('arch,B,A,sims', [('TinyNetwork', 256, 4, 50), ('MuZeroNetwork', 24, 4, 12), ('TinyNetwork', 40, 6, 20),
                                             ('MuZeroNetwork', 512, 4, 50)])       # BASELINE configs[4]'s network and simulation count
def test_batched_search_tree_is_bit_exact_given_the_torch_outputs(arch, B, A, sims):
  """(i) the tree the device builds around a torch network equals, field for field, the tree the CPU oracle builds
  when it is fed the very same network outputs: descents (leaf, parent slot, action, depth) of every simulation, N, E,
  to_play, W, R, MinMaxStats bit-exact, priors within a few ulp (device exp vs glibc), visit distributions and root
  values bit-exact; and the loop makes no host synchronisation (torch's sync debug mode raises on any)."""
  from oracle import oracle as orc
  from model_based_rl_amd import networks
  from model_based_rl_amd.torch_search import BatchedSearch
  C = 4 if A == 4 else 2
  torch.manual_seed(5)
  net = getattr(networks, arch)(C, A, torch.device('cuda'), types.SimpleNamespace()).eval()
  cfg = make_cfg(A, sims, obs_space=(C, 96, 96))
  bs = BatchedSearch(cfg, net, B, 'cuda', seed=9)
  trace = []
  bs.on_simulation = lambda s, leaf, slot, act, depth, out: trace.append(
      (leaf.clone(), slot.clone(), act.clone(), depth.clone(), out.value.reshape(B).clone(), out.reward.reshape(B).clone(),
       out.policy_logits.reshape(B, A).clone()))
  rng = np.random.RandomState(1)
  obs = torch.from_numpy(rng.randint(0, 256, size=(B, C, 96, 96)).astype(np.float32) / np.float32(255)).cuda()
  noise = rng.dirichlet([0.25] * A, size=B)
  noise_d = torch.from_numpy(noise).cuda()
  bs.run(obs, noise=noise_d, device_rng=False)          # warm-up: MIOpen picks its kernels, the pool is allocated
  trace.clear()
  torch.cuda.synchronize()
  torch.cuda.set_sync_debug_mode('error')
  try:
    init = bs.run(obs, noise=noise_d, device_rng=False)
  finally:
    torch.cuda.set_sync_debug_mode('default')
  fin = {k: v.cpu().numpy() for k, v in bs.finalize(1.0, np.full(B, 0.5)).items()}
  ex = bs.engine.export_tree()
  t = orc.Trees(orc.tree_cfg(A, sims), B)
  t.root_expand(np.ones(B, np.int8), init.policy_logits.cpu().numpy().reshape(B, A))
  t.add_noise(noise, 0.25)
  assert len(trace) == sims
  for s, rec in enumerate(trace):
    leaf, slot, act, depth, v, r, lg = [x.cpu().numpy() for x in rec]
    ol, osl, oa, od = t.select()
    assert np.array_equal(ol, leaf) and np.array_equal(osl, slot) and np.array_equal(oa, act) and np.array_equal(od, depth), s
    t.expand_backup(v, r, lg)
  eo = t.export()
  EX = eo['EX'].astype(bool)
  for k in ('N', 'E', 'TP'):
    assert np.array_equal(ex[k][EX], eo[k][EX]), k
  assert np.array_equal(ex['W'][EX], eo['W'][EX]) and np.array_equal(ex['R'].astype(np.float64)[EX], eo['R'][EX])
  assert np.array_equal(ex['minmax'], eo['minmax'])
  assert ulp_diff(ex['P'][EX], eo['P'][EX]).max() <= 8
  act_o, cv_o, rv_o, vc_o = t.finalize(1.0, np.full(B, 0.5))
  assert np.array_equal(vc_o, fin['visit_counts']) and np.array_equal(cv_o, fin['child_visits'])
  assert np.array_equal(rv_o, fin['root_value']) and np.array_equal(act_o, fin['action'])
  assert (ex['N'][:, 0] == sims).all()
  # the hidden states the network saw: the pool row of the parent the descent named
  assert np.isfinite(bs.pool.cpu().numpy()).all()
  bs.close()


def test_torch_selfplay_records_and_actor_loop():
  """TorchSelfplay / Actor with --architecture TinyNetwork: record layout, episode bookkeeping, per-env temperature at
  episode starts, ingestion by the native replay, games accounting through the storage."""
  from model_based_rl_amd.actors import Actor
  from model_based_rl_amd.config import make_config
  from model_based_rl_amd.engine import records_view
  from model_based_rl_amd.networks import get_network
  from model_based_rl_amd.replay_buffer import PrioritizedReplay
  from model_based_rl_amd.shared_storage import SharedStorage
  cfg = make_config(['--architecture', 'TinyNetwork', '--environment', 'BreakoutNoFrameskip-v4', '--stack_obs', '4', '--num_envs', '24',
                     '--num_simulations', '6', '--episode_length', '5', '--seed', '2', '--window_size', '2048',
                     '--weight_sync_frequency', '4', '--training_steps', '100'])
  storage, replay = SharedStorage(cfg), PrioritizedReplay(cfg)
  torch.manual_seed(0)
  storage.store_weights(get_network(cfg, torch.device('cpu')).get_weights(), 0)
  actor = Actor(0, cfg, storage, replay)
  seen = []
  orig = replay.ingest_records
  replay.ingest_records = lambda buf, n, B, base=0: (seen.append(buf[:n].numpy().copy()), orig(buf, n, B, base))[1]
  actor.launch(max_moves=12)
  rec = np.concatenate(seen, 0)
  O, A = 4 * 96 * 96, 4
  assert cfg.obs_space == (4, 96, 96) and cfg.obs_u8          # channels = --stack_obs (utils.py:27-35); frames travel as bytes
  assert rec.shape == (12, 24, O // 4 + A + 10)               # ... 4 per float slot: a quarter of the float32 record
  rv = records_view(rec, O, A, obs_u8=True)
  assert rv['obs'].dtype == np.uint8 and rv['obs'].shape == (12, 24, O) and rv['obs'].max() == 255 and rv['obs'].min() == 0
  assert abs(float(rv['obs'].mean()) - 127.5) < 1.0
  # what the learner gets back out of the replay: the same frames as float32 (np.float32(obs), learners.py:167-168 normalises)
  replay.batch_size = 4
  (bobs, _, _), _, _ = replay.sample_batch()
  assert bobs.shape == (4, 4, 96, 96) and bobs.dtype == np.float32
  frames = {rv['obs'][m, b].tobytes() for m in range(12) for b in range(24)}
  assert all(bobs[i].astype(np.uint8).tobytes() in frames and np.array_equal(bobs[i], np.round(bobs[i])) for i in range(4))
  assert np.allclose(rv['child_visits'].sum(-1), 1.0, atol=1e-6)
  assert np.all(np.take_along_axis(rv['child_visits'], rv['action'][..., None], -1) > 0)
  assert np.array_equal(rv['env_id'][0], np.arange(24)) and set(np.unique(rv['done'])) == {0, 1}
  for b in range(24):                      # fixed-length episodes of 5 steps, staggered starts
    st = rv['step'][:, b]
    assert np.all((st[1:] == st[:-1] + 1) | ((st[1:] == 0) & (st[:-1] == 4)))
    assert np.array_equal(rv['done'][:, b], (st == 4).astype(np.int32))
  assert np.isfinite(rv['root_value']).all() and np.abs(rv['reward']).max() <= 1.0
  games = int(rv['done'].sum())
  assert actor.games_played == games and storage.get_stats('actor_games')[0] == games
  thr = replay.get_throughput()
  assert thr['games'] == games and thr['frames'] > 0
  actor.selfplay.close()


def test_train_driver_with_conv_network_and_learner():
  """train.py wiring (storage + the native replay + actor + learner, reference train.py:62-78) with a conv network:
  the actor searches with TinyNetwork behind the batched external-inference path, its records (image observations)
  reach the replay, the learner trains on sampled batches on the GPU and publishes weights the actor adopts."""
  from model_based_rl_amd import train
  thr = train.main(['--architecture', 'TinyNetwork', '--environment', 'BreakoutNoFrameskip-v4', '--stack_obs', '2', '--stack_actions',
                    '--num_envs', '16',
                    '--num_simulations', '4', '--seed', '1', '--episode_length', '6', '--max_moves', '40', '--window_size',
                    '1024', '--stored_before_train', '128', '--batch_size', '8', '--learner_steps', '3',
                    '--send_weights_frequency', '1', '--weight_sync_frequency', '4', '--norm_obs', '--obs_range', '0', '255',
                    '--use_gpu_for', 'actors', 'learner', '--run_tag', 'conv_test'])
  assert thr['frames'] >= 16 * 20 and thr['games'] >= 16 * 3
  assert thr['learner']['updates_per_second'] > 0


def test_affine_relu_epilogue_kernel():
  """mz_affine_relu (the BatchNorm + skip + ReLU epilogue of a residual block, one in-place pass) against the three
  PyTorch operations it replaces, on the shapes the conv networks use; cached scale / shift follow a weight reload."""
  from model_based_rl_amd import networks
  torch.manual_seed(4)
  dev = torch.device('cuda')
  for shape in ((37, 128, 6, 6), (5, 64, 48, 48), (3, 128, 12, 12)):
    bn = torch.nn.BatchNorm2d(shape[1]).to(dev).eval()
    with torch.no_grad():
      bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
    y = torch.randn(shape, device=dev)
    res = torch.randn(shape, device=dev)
    with torch.inference_mode():
      want1 = torch.relu(bn(y))
      want2 = torch.relu(bn(y) + res)
      got1 = networks._affine_relu_(y.clone(), *networks._bn_affine(bn))
      got2 = networks._affine_relu_(y.clone(), *networks._bn_affine(bn), residual=res)
    assert (got1 - want1).abs().max().item() <= 2e-6 and (got2 - want2).abs().max().item() <= 2e-6
    assert (got1 == 0).any() and (got1 > 0).any()
    with torch.no_grad():
      bn.load_state_dict({k: v * 0.5 if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    with torch.inference_mode():
      got3 = networks._affine_relu_(y.clone(), *networks._bn_affine(bn))
      assert (got3 - torch.relu(bn(y))).abs().max().item() <= 2e-6 and not torch.equal(got3, got1)
  with pytest.raises(RuntimeError, match='hw'):
    networks._affine_relu_(torch.zeros(2, 4, 3, 3, device=dev), torch.ones(4, device=dev), torch.zeros(4, device=dev))
