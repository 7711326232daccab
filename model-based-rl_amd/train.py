"""train.py-shaped driver (reference train.py:62-78): storage + replay + actors wired together, no Ray.
Self-play side only this round: the learner (learners.py in the reference) is the next row of the scope
table; `publish_initial_weights` stands in for Learner.send_weights at start-up (learners.py:85-86,116) so the
actors have a network to search with.

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
from .networks import FCNetwork
from .replay_buffer import PrioritizedReplay
from .shared_storage import SharedStorage


def publish_initial_weights(config, storage):
  torch.manual_seed(config.seed or 0)
  net = FCNetwork(int(np.prod(config.obs_space)), config.action_space, torch.device('cpu'), config)
  storage.store_weights.remote(net.get_weights(), 0).result()


def launch(config, max_moves):
  ray.init()
  storage = ray.remote(SharedStorage).remote(config)
  replay = ray.remote(PrioritizedReplay).remote(config)
  actors = [ray.remote(Actor).remote(k, config, storage, replay) for k in range(config.num_actors)]
  publish_initial_weights(config, storage)
  t0 = time.time()
  ray.get([a.launch.remote(max_moves) for a in actors])
  dt = time.time() - t0
  thr = ray.get(replay.get_throughput.remote())
  print('frames accepted by replay: %d, games: %d, %.1f s -> %.0f env-steps/s' % (thr['frames'], thr['games'], dt,
                                                                                 thr['frames'] / dt))
  ray.shutdown()
  return thr


def main(argv=None):
  p = build_parser()
  p.add_argument('--max_moves', type=int, default=64)
  args = vars(p.parse_args(argv))
  max_moves = args.pop('max_moves')
  cfg = Config(args)
  cfg.action_space, cfg.obs_space = ENV_SHAPES[cfg.environment]
  if cfg.seed is None:
    cfg.seed = 0
  return launch(cfg, max_moves)


if __name__ == '__main__':
  main()
