"""Configuration for the self-play path: the flags of the reference's config.py:87-231 that the search /
actor / replay-ingest path reads, with the reference's names and defaults, plus the pool size of the GPU
actor (`--num_envs`).  `Config` carries the helper methods callers use on it (config.py:21-84).
Flags outside the path (optimizer, lr schedule, Atari wrappers, ...) are accepted where they are cheap to
carry and otherwise not re-created."""
import argparse

import numpy as np


class Config(object):

  def __init__(self, args):
    self.__dict__.update(args)
    lo, hi = self.value_support
    self.value_support_min, self.value_support_max = lo, hi
    self.value_support_range = list(range(lo, hi + 1))
    self.value_support_size = hi - lo + 1
    lo, hi = self.reward_support
    self.reward_support_min, self.reward_support_max = lo, hi
    self.reward_support_range = list(range(lo, hi + 1))
    self.reward_support_size = hi - lo + 1

  # config.py:41-49
  def visit_softmax_temperature(self, training_step):
    steps, temps = self.visit_softmax_steps, self.visit_softmax_temperatures
    for boundary, temp in zip(steps, temps):
      if training_step <= boundary:
        return temp
    return temps[len(steps)]

  # config.py:70-81 (for Node objects; the batched engine does this in mz_finalize)
  @staticmethod
  def select_action(node, temperature=0.):
    actions = list(node.children.keys())
    counts = np.array([child.visit_count for child in node.children.values()])
    if temperature:
      dist = counts ** (1 / temperature)
      dist = dist / dist.sum()
      idx = np.random.choice(len(actions), p=dist)
    else:
      idx = np.random.choice(np.where(counts == counts.max())[0])
    return actions[idx]

  def new_game(self, environment):
    from .game import Game
    return Game(environment, self)


def build_parser():
  p = argparse.ArgumentParser(description='MI355X-native MuZero self-play (reference flag names)')
  a = p.add_argument
  a('--architecture', type=str, default='FCNetwork', choices=['FCNetwork', 'MuZeroNetwork', 'TinyNetwork'])
  a('--stack_obs', type=int, default=1)
  a('--stack_actions', action='store_true')
  a('--value_support', nargs=2, type=int, default=[-15, 15])
  a('--reward_support', nargs=2, type=int, default=[-15, 15])
  a('--no_support', action='store_true')
  a('--no_target_transform', action='store_true')
  a('--seed', type=int, default=None)
  a('--environment', type=str, default='LunarLander-v2')
  a('--two_players', action='store_true')
  a('--obs_range', nargs='+', type=float, default=None)
  a('--norm_obs', action='store_true')
  a('--clip_rewards', action='store_true')
  a('--episode_life', action='store_true')
  a('--sticky_actions', type=int, default=1)
  a('--num_actors', type=int, default=1)
  a('--num_envs', type=int, default=4096, help='environments searched in lock-step by one GPU actor')
  a('--episode_length', type=int, default=256, help='synthetic fixed-length episodes (gym is not installed)')
  a('--ingest_threads', type=int, default=None,
    help='threads of the native replay: environments of a record chunk, the samples of a batch, a large priority refresh (default: 4, 8 from batch size 1024 up, bounded by the CPUs)')
  a('--max_steps', type=int, default=40000)
  a('--num_simulations', type=int, default=30)
  a('--max_history_length', type=int, default=500)
  a('--visit_softmax_temperatures', nargs=3, type=float, default=[1.0, 0.5, 0.25])
  a('--visit_softmax_steps', nargs=2, type=int, default=[15000, 30000])
  a('--fixed_temperatures', nargs='+', type=float, default=[])
  a('--root_dirichlet_alpha', type=float, default=0.25)
  a('--root_exploration_fraction', type=float, default=0.25)
  a('--init_value_score', type=float, default=0.0)
  a('--known_bounds', nargs=2, type=float, default=[None, None])
  a('--pb_c_base', type=int, default=19652)
  a('--pb_c_init', type=float, default=1.25)
  a('--window_size', type=int, default=100000)
  a('--window_step', type=int, default=None)
  a('--epsilon', type=float, default=0.01)
  a('--alpha', type=float, default=1.)
  a('--beta', type=float, default=1.)
  a('--beta_increment_per_sampling', type=float, default=0.001)
  a('--training_steps', type=int, default=100000000)
  a('--num_unroll_steps', type=int, default=5)
  a('--td_steps', type=int, default=10)
  a('--batch_size', type=int, default=256)
  a('--stored_before_train', type=int, default=50000)
  a('--send_weights_frequency', type=int, default=500)
  a('--weight_sync_frequency', type=int, default=1000)
  a('--discount', type=float, default=0.997)
  a('--optimizer', type=str, default='AdamW', choices=['RMSprop', 'Adam', 'AdamW', 'SGD'])
  a('--lr_init', type=float, default=0.0008)
  a('--weight_decay', type=float, default=1e-4)
  a('--momentum', type=float, default=0.9)
  a('--clip_grad', type=int, default=0)
  a('--scalar_loss', type=str, default='MSE', choices=['MSE', 'Huber'])
  a('--lr_scheduler', type=str, default=None, choices=['ExponentialLR', 'MuZeroLR', 'WarmUpLR'])
  a('--lr_decay_rate', type=float, default=0.1)
  a('--lr_decay_steps', type=int, default=100000)
  a('--learner_gpu_device_id', type=int, default=None)
  a('--unbatched_learner', action='store_true',
    help='FCNetwork learner step position by position as the reference writes it, instead of the three heads batched over '
         'all K + 1 unroll positions (learners.py)')
  a('--no_tune_gemms', action='store_true',
    help='graphed learner: keep the BLAS libraries\' default kernel choice instead of PyTorch TunableOp picking the fastest per shape')
  a('--no_hip_learner_ops', action='store_true',
    help='learner targets and categorical losses as PyTorch elementwise kernels instead of the single HIP launches of csrc/mz_learner.hip.h')
  a('--batches_per_fetch', type=int, default=15,
    help='reference config.py:173: batches sampled ahead of the updates that consume them; here a background thread keeps '
         'min(this, 4) batches sampled ahead (1: sample in the learner\'s own thread, just before each update)')
  a('--no_native_learner', action='store_true',
    help='FCNetwork learner step through PyTorch operators (GEMM library + autograd) instead of the HIP launches (three at batch 256) of '
         'csrc/mz_fcl.hip.h (mz_fcl_step)')
  a('--no_native_loop', action='store_true',
    help='drive the native learner step from Python, one update per call (learners.py), instead of mz_fcl_run taking the loop body '
         '-- sampling, update, priority refresh -- for a whole stretch of updates')
  a('--no_graph_learner', action='store_true',
    help='run the learner step as eager PyTorch launches instead of one captured hipGraph per update (learners.py)')
  a('--learner_log_frequency', type=int, default=100)
  a('--frames_before_fps_log', type=int, default=10000)
  a('--runs_dir', type=str, default='runs', help='root of the run directories (the reference writes ./runs)')
  a('--save_state_frequency', type=int, default=1000)
  a('--use_gpu_for', nargs='+', type=str, default=['actors'], choices=['actors', 'learner'])
  a('--actors_gpu_device_ids', nargs='+', type=int, default=None)
  a('--group_tag', type=str, default=None)
  a('--run_tag', type=str, default=None)
  a('--actor_log_frequency', type=int, default=1)
  a('--selfplay_chunk', type=int, default=None,
    help='moves per launch / drain / ingest chunk of the device self-play loop (default 16: one launch of the persistent search kernel)')
  a('--no_gpu_turns', action='store_true',
    help='do NOT let an actor and the learner of one process take turns on a GPU they share (train.share_gpu_in_turns switches turns on '
         'whenever both resolve to the same device; this flag is for measuring what the turns are worth)')
  a('--gpu_turns', action='store_true',
    help='an actor and a learner of this process share ONE GPU: they take turns, one chunk of moves / one update at a time (gpu_turns.py)')
  a('--gpu_turn_updates', type=int, default=8,
    help='with --gpu_turns: updates the learner runs per turn (the actor holds the GPU for one chunk of moves per turn)')
  a('--split_f16', action='store_true',
    help='FCNetwork GEMMs of the search as float16 high/low splits on the f16 matrix pipe (float32-level accuracy, not '
         'bit-identical to the exact-float32 default; include/mz_engine.h mz_config.split_f16)')
  a('--parity_rng', action='store_true',
    help="draw Dirichlet noise / action samples from numpy's global stream in the reference's order")
  return p


ENV_SHAPES = {   # (action_space, obs_space) of the environments the reference's README runs
    'TicTacToe': (9, (9,)), 'LunarLander-v2': (4, (8,)), 'Pong-ramNoFrameskip-v4': (6, (128,)),
    'Breakout-ramNoFrameskip-v4': (4, (128,)),
    # image observations of the Atari wrappers (wrappers.py:422-444: 96x96 uint8 frames) -- what MuZeroNetwork /
    # TinyNetwork take (SURVEY.md s7 on BASELINE configs[4]).  The channel count is NOT a property of the environment:
    # utils.py:27-35 builds the network with input_channels = stack_obs, doubled with --stack_actions (None below;
    # env_shapes() fills it in from the flags, so the same flags build the same conv1 as the reference's)
    'BreakoutNoFrameskip-v4': (4, (None, 96, 96)), 'PongNoFrameskip-v4': (6, (None, 96, 96)),
}


def env_shapes(config):
  """(action_space, obs_space) of config.environment (train.py:66-68 probes the environment for these).  Image
  environments: channels = stack_obs * (2 if stack_actions else 1), as utils.get_network does (utils.py:27-35);
  their frames are bytes (config.obs_u8: the experience records and the replay keep them as bytes, game.py:93-96)."""
  A, obs = ENV_SHAPES[config.environment]
  if obs[0] is None:
    ch = int(getattr(config, 'stack_obs', 1)) * (2 if getattr(config, 'stack_actions', False) else 1)
    obs = (ch,) + tuple(obs[1:])
  return A, tuple(obs)


def obs_are_bytes(cfg):
  """image frames and the 128 bytes of console RAM of the -ram- environments travel as bytes in the experience records and
  the replay (game.py:93-96 keeps the raw uint8 observation)"""
  return len(tuple(getattr(cfg, 'obs_space', ()))) == 3 or '-ram' in str(getattr(cfg, 'environment', ''))


def make_config(argv=None, **overrides):
  args = vars(build_parser().parse_args(argv))
  args.update(overrides)
  cfg = Config(args)
  if cfg.environment in ENV_SHAPES and not hasattr(cfg, 'action_space'):
    cfg.action_space, cfg.obs_space = env_shapes(cfg)   # train.py:66-68 probes the env for these
  if not hasattr(cfg, 'obs_u8'):
    cfg.obs_u8 = obs_are_bytes(cfg)
  return cfg
