"""Run-directory layout and metrics of the reference's Logger (logger.py:8-51) without TensorBoard (not installed):
`runs/<environment>[/<group_tag>]/<run_tag>/{<worker_id>, saves, config}`, `config/config.json` written once by the first
worker, and every scalar the reference sends to its SummaryWriter (actors.py:105-122, learners.py:88-113,138-153) appended
to `<worker_id>/metrics.csv` as `tag,step,value,wall_time`."""
import json
import os
import time


class Logger(object):
  """Mixin: expects self.config, self.run_tag, self.group_tag, self.worker_id."""

  def __init__(self):
    self.dirs = self.make_dirs()
    self.save_config()
    self._metrics_path = os.path.join(self.dirs['worker'], 'metrics.csv')
    if not os.path.isfile(self._metrics_path):
      with open(self._metrics_path, 'w') as f:
        f.write('tag,step,value,wall_time\n')
    self._metrics = open(self._metrics_path, 'a')

  # logger.py:34-50
  def make_dirs(self):
    base = os.path.join(getattr(self.config, 'runs_dir', 'runs'), str(self.config.environment))
    if self.group_tag is not None:
      base = os.path.join(base, str(self.group_tag))
    base = os.path.join(base, str(self.run_tag))
    dirs = {'base': base, 'worker': os.path.join(base, str(self.worker_id)), 'saves': os.path.join(base, 'saves'),
            'config': os.path.join(base, 'config')}
    for k in ('saves', 'config', 'worker'):
      os.makedirs(dirs[k], exist_ok=True)
    return dirs

  # logger.py:28-32
  def save_config(self):
    path = os.path.join(self.dirs['config'], 'config.json')
    if not os.path.isfile(path):
      with open(path, 'w') as f:
        json.dump(self.config.__dict__, f, indent=2, default=str)

  def log_scalar(self, value, tag, i):
    self._metrics.write('%s,%d,%.9g,%.3f\n' % (tag, int(i), float(value), time.time()))
    self._metrics.flush()

  def log_points(self, points):
    """[(tag, step, value), ...] with one flush"""
    now = time.time()
    self._metrics.write(''.join('%s,%d,%.9g,%.3f\n' % (tag, int(i), float(v), now) for tag, i, v in points))
    self._metrics.flush()

  def log_scalars(self, value_dict, group_tag, i):
    for k, v in value_dict.items():
      self.log_scalar(v, '%s/%s' % (group_tag, k), i)


def read_metrics(path):
  """-> {tag: [(step, value), ...]} of a metrics.csv"""
  out = {}
  with open(path) as f:
    next(f)
    for line in f:
      tag, step, value, _ = line.rstrip('\n').rsplit(',', 3)
      out.setdefault(tag, []).append((int(step), float(value)))
  return out
