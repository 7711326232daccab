"""train.py-shaped driver (reference train.py:62-78): storage + replay + actors + learner wired together through
the thread-backed ray shim, no Ray.  With --selfplay_only the learner is left out and `publish_initial_weights`
stands in for Learner.send_weights at start-up (learners.py:85-86,116).

  python -m model_based_rl_amd.train --environment LunarLander-v2 --num_envs 4096 --num_simulations 30 --seed 0 \
      --max_moves 64
"""
import sys
import time
import types

import numpy as np
import torch

from . import rayshim as ray
from .actors import Actor
from .config import build_parser, Config, ENV_SHAPES
from .learners import Learner
from .networks import FCNetwork
from .replay_buffer import PrioritizedReplay
from .shared_storage import SharedStorage


def publish_initial_weights(config, storage):
  torch.manual_seed(config.seed or 0)
  net = FCNetwork(int(np.prod(config.obs_space)), config.action_space, torch.device('cpu'), config)
  storage.store_weights.remote(net.get_weights(), 0).result()


def launch(config, max_moves, selfplay_only=False, learner_steps=None):
  ray.init()
  storage = ray.remote(SharedStorage).remote(config)
  replay = ray.remote(PrioritizedReplay).remote(config)
  actors = [ray.remote(Actor).remote(k, config, storage, replay) for k in range(config.num_actors)]
  workers = [a.launch.remote(max_moves) for a in actors]
  if selfplay_only:
    publish_initial_weights(config, storage)
  else:
    learner = ray.remote(Learner).remote(config, storage, replay)
    workers.append(learner.launch.remote(learner_steps))
  t0 = time.time()
  ray.get(workers)
  dt = time.time() - t0
  thr = ray.get(replay.get_throughput.remote())
  print('frames accepted by replay: %d, games: %d, %.1f s -> %.0f env-steps/s' % (thr['frames'], thr['games'], dt,
                                                                                 thr['frames'] / dt))
  ray.shutdown()
  return thr


def main(argv=None):
  p = build_parser()
  p.add_argument('--max_moves', type=int, default=64)
  p.add_argument('--selfplay_only', action='store_true')
  p.add_argument('--learner_steps', type=int, default=None)
  args = vars(p.parse_args(argv))
  max_moves, selfplay_only, learner_steps = args.pop('max_moves'), args.pop('selfplay_only'), args.pop('learner_steps')
  cfg = Config(args)
  cfg.action_space, cfg.obs_space = ENV_SHAPES[cfg.environment]
  if cfg.seed is None:
    cfg.seed = 0
  return launch(cfg, max_moves, selfplay_only, learner_steps)


if __name__ == '__main__':
  main()
