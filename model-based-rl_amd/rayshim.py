"""A few lines of Ray's actor API (`@remote`, `.remote()`, `get`, `wait`, `init`, `shutdown`) over threads,
so a train.py-shaped driver runs without Ray (not installed; one process per GPU is the deployment model).
Each remote object owns one worker thread and runs its calls in submission order, like a Ray actor."""
import queue
import threading
from concurrent.futures import Future


class _Method(object):
  def __init__(self, handle, name):
    self._handle, self._name = handle, name

  def remote(self, *args, **kwargs):
    fut = Future()
    self._handle._q.put((self._name, args, kwargs, fut))
    return fut


class _Handle(object):
  def __init__(self, obj):
    self._obj = obj
    self._q = queue.Queue()
    self._t = threading.Thread(target=self._loop, daemon=True)
    self._t.start()

  def _loop(self):
    while True:
      name, args, kwargs, fut = self._q.get()
      if name is None:
        return
      try:
        fut.set_result(getattr(self._obj, name)(*args, **kwargs))
      except BaseException as e:   # propagate through get(), like ray.get
        fut.set_exception(e)

  def __getattr__(self, name):
    return _Method(self, name)


class _Remote(object):
  def __init__(self, cls):
    self._cls = cls

  def remote(self, *args, **kwargs):
    return _Handle(self._cls(*args, **kwargs))

  def __call__(self, *args, **kwargs):
    return self._cls(*args, **kwargs)


def remote(cls):
  return _Remote(cls)


def get(refs):
  if isinstance(refs, (list, tuple)):
    return [r.result() for r in refs]
  return refs.result()


def wait(refs, num_returns=1, timeout=None):
  import concurrent.futures as cf
  done, pending = cf.wait(refs, timeout=timeout, return_when=cf.FIRST_COMPLETED if num_returns < len(refs) else cf.ALL_COMPLETED)
  done = [r for r in refs if r in done][:num_returns]
  return done, [r for r in refs if r not in done]


def init(*args, **kwargs):
  return None


def shutdown():
  return None
