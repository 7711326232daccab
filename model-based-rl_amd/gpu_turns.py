"""Turn-taking for an actor and a learner that share ONE GPU in one process (train.py with --use_gpu_for actors learner).

On MI355X two queues with work do not share the GPU gracefully when one of them runs the whole-moves self-play launch:
measured (scripts/experiments/learner_speed.py, scripts/experiments/cu_mask_probe.py) both sides lose ~8 x -- the learner 122 -> 15-19 updates/s,
the actor 8.7 -> 0.9 M env-steps/s -- with the loop in another thread, another process, on its own stream or behind CU masks
alike.  Mutual exclusion at chunk granularity costs nothing like that: the actor holds the GPU for one chunk of moves
(submit + wait), the learner for one update (until its stream is idle), strictly alternating when both want it (FIFO).

The reference has no such problem to solve: its actors are CPU processes (train.py:62-78, actors.py:36-47).
"""
import collections
import threading


class FairTurns(object):
  """FIFO lock: whoever asked first goes first, so neither a tight actor loop nor a tight learner loop starves the other."""

  def __init__(self):
    self._cv = threading.Condition()
    self._busy = False
    self._waiting = collections.deque()
    self.parties = set()

  def __enter__(self):
    me = object()
    with self._cv:
      self._waiting.append(me)
      while self._busy or self._waiting[0] is not me:
        self._cv.wait()
      self._waiting.popleft()
      self._busy = True
    return self

  def __exit__(self, *exc):
    with self._cv:
      self._busy = False
      self._cv.notify_all()
    return False


class _NoTurns(object):
  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False


_turns = {}
_lock = threading.Lock()
NO_TURNS = _NoTurns()


def register(device, party):
  """An actor ('actor') or a learner ('learner') announces that it computes on `device` in this process; returns the
  device's FairTurns.  `contended(device)` says whether both kinds are present."""
  key = str(device)
  with _lock:
    t = _turns.setdefault(key, FairTurns())
    t.parties.add(party)
    return t


def turn(device):
  """context manager: the device's turn lock if an actor AND a learner share it in this process, else a no-op"""
  t = _turns.get(str(device))
  if t is not None and {'actor', 'learner'} <= t.parties:
    return t
  return NO_TURNS
