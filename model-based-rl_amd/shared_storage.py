"""Parameter server with the reference's surface (shared_storage.py:4-25): latest weights + training step,
per-actor game counts.  In the multi-GPU layout rank 0's storage is the weight source and
distributed.RankStorage ships one flattened float32 buffer to every actor rank with an RCCL broadcast
(mz_broadcast_weights) instead of Ray's pickled state_dict."""


class SharedStorage(object):

  def __init__(self, config):
    self.stats = {'training_step': 0, 'actor_games': {k: 0 for k in range(config.num_actors)}}
    self.weights = None

  def get_weights(self, games, actor_key):
    self.stats['actor_games'][actor_key] = games
    return self.weights, self.stats['training_step']

  def store_weights(self, weights, step):
    self.stats['training_step'] = step
    self.weights = weights

  def get_stats(self, key=None):
    return self.stats if key is None else self.stats[key]

  def is_ready(self):
    return self.weights is not None

