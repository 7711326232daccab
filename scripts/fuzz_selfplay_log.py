"""Randomised sweep of the device self-play loop over the dispatch table: random shapes (1..32 actions, observation sizes 1..200,
float / byte / packed-byte observations, 2..61 simulations, tree counts that are not multiples of 16, episode lengths, temperatures)
through mz_selfplay_steps -- whole moves inside one launch where the shape allows it, root + search kernels per move otherwise,
trees in LDS or in the pool, exact and split-f16 -- with the simulation log on: every move of every tree is replayed through the
oracle's TREE on the device's own logged network outputs, Dirichlet draws and uniforms, and must give the record's visit
distribution, action, root value and error exactly; the record's observation must be the synthetic env's.
usage: fuzz_selfplay_log.py [n_configs] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from model_based_rl_amd.engine import Engine, records_view
from tests.parity_util import env_switches, philox_action_uniform, random_weights, replay_move

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
done = whole = 0
t0 = time.time()
for it in range(n_cfg):
  A = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 16, 18, 21, 25, 32]))
  O = int(rng.choice([1, 3, 8, 9, 33, 64, 128, 200]))
  sims = int(rng.choice([2, 5, 17, 30, 50, 61]))
  B = int(rng.choice([5, 16, 17, 100, 250]))
  T = int(rng.choice([1, 3, 7]))
  temp = float(rng.choice([1.0, 1.0, 0.0]))
  obs_mode = ['float', 'u8', 'packed'][rng.randint(3)]
  variant = ['lds', 'pool', 'nopersist', 'split'][rng.randint(4)]
  split = variant == 'split' and A <= 13
  seed = int(rng.randint(1, 1 << 30))
  sw = {'pool': {'MZ_NO_LDS_TREES': '1'}, 'nopersist': {'MZ_NO_PERSIST': '1'}}.get(variant, {})
  with env_switches(**sw):
    eng = Engine(B, O, A, sims, seed=seed, split_f16=split, env_id_offset=int(rng.choice([0, 4096])))
  w = random_weights(O, A, seed=it)
  eng.set_weights(w)
  if obs_mode != 'float':
    eng.selfplay_set_obs(uint8_obs=True, obs_min=[0.0], obs_range=[255.0], packed=(obs_mode == 'packed'))
  info = eng.search_kernel_info()
  if info['kind'] == 'standalone':
    eng.close(); continue
  eng.selfplay_noise_log(True)
  eng.selfplay_reset(T, temp, stagger=True)
  moves = int(rng.choice([1, 3, 16, 19]))
  log = eng.sim_io('log', keep_moves=moves)
  eng.selfplay_steps(moves)
  buf, n = eng.selfplay_drain()
  torch.cuda.synchronize()
  assert n == moves
  rec = buf[:n].numpy().copy()
  rv = records_view(rec, O, A, obs_u8=(obs_mode == 'packed'))
  io_all = log.cpu().numpy()
  cfg = orc.tree_cfg(A, sims)
  env0 = eng.cfg.env_id_offset
  tag = (it, 'A', A, 'O', O, 'sims', sims, 'B', B, 'T', T, 'temp', temp, obs_mode, variant, info, 'moves/launch', eng.selfplay_moves_per_launch())
  for m in range(moves):
    for b in (0, B - 1):
      want = eng.synth_obs(env0 + b, int(rv['episode'][m, b]), int(rv['step'][m, b]))[0]
      assert np.array_equal(np.asarray(rv['obs'][m, b], np.float32), want), ('obs',) + tag
    assert np.array_equal(rv['env_id'][m], env0 + np.arange(B)), tag
    u = philox_action_uniform(seed, env0 + np.arange(B), m)
    ref = replay_move(cfg, B, A, sims, io_all[m], eng.selfplay_noise(m), 0.25, np.ones(B, np.int8), None, temp, u)
    assert np.array_equal(rv['child_visits'][m], ref['child_visits'].astype(np.float32)), ('visits', m) + tag
    if temp != 0.0:
      assert np.array_equal(rv['action'][m], ref['action']), ('action', m) + tag
    else:             # T = 0: uniform among the arg-max set (config.py:79) -- the device draws its own tie-break
      vc = ref['visit_counts']
      assert np.all(vc[np.arange(B), rv['action'][m]] == vc.max(1)), ('argmax action', m) + tag
    assert np.array_equal(rv['root_value'][m], ref['root_value']), ('root value', m) + tag
    assert np.array_equal(rv['error'][m], ref['root_value'] - ref['v0'].astype(np.float64)), ('error', m) + tag
  eng.sim_io('off'); eng.close()
  done += 1
  whole += int(tag[-1] > 0)
  print('ok', *tag, flush=True)
print('%d configurations identical (%d of them whole moves inside one launch; %d skipped: stand-alone kernels), %.0f s' %
      (done, whole, n_cfg - done, time.time() - t0))
