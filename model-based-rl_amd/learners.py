"""Learner with the reference's surface (learners.py:14-235): K-step unrolled training step on batches from the
prioritized replay, priority refresh, weight publication, checkpoints.  Semantics of the step: initial inference + K
recurrent steps, 0.5 gradient scale on the hidden state per step (learners.py:200), importance-sampling weighted
cross-entropy losses on the categorical supports, 1/K gradient scale on the total loss (214), AdamW with eps 1.5e-4
(utils.py:85-97).

On a GPU, for FCNetwork with Adam / AdamW and categorical losses, the whole update is `mz_fcl_update` (_NativeFC): two
hand-written HIP launches at the reference's batch 256 (csrc/mz_fcl.hip.h) straight from the host batch -- no PyTorch operator, no autograd tape, no graph
to capture; the parameters and the optimiser's state are views of the flat vectors those kernels update.  Every other case
(MuZeroNetwork / TinyNetwork, scalar losses, other optimisers, `--no_native_learner`) runs the same step as PyTorch
operators, captured in ONE hipGraph per update (stock `torch.cuda.CUDAGraph`: static batch tensors, capturable optimiser);
`--no_graph_learner` and CPU learners run that tensor code eagerly.  The loop (`learn`) takes its batches through
_BatchSource: sampled a few updates ahead, priority refreshes one update behind and not waited for (the reference's learner
prefetches `batches_per_fetch` batches and sends its refresh fire-and-forget, learners.py:124,182)."""
import os
import time
from copy import deepcopy

import numpy as np
import torch

from .actors import _call, set_all_seeds
from .logger import Logger
from .networks import get_network, support_to_scalar


def scalar_transform(x):
  """h(x) = sign(x)(sqrt(|x|+1)-1) + 0.001x  (config.py:51-54)."""
  return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + 0.001 * x


def scalar_to_support(x, lo, hi):
  """two-hot projection of scalars onto the integer support [lo, hi] (config.py:56-68)."""
  x = x.clamp(lo, hi)
  low, high = x.floor(), x.ceil()
  p_high = x - low
  out = torch.zeros(*x.shape, hi - lo + 1, device=x.device)
  out.scatter_(2, (high - lo).long().unsqueeze(-1), p_high.unsqueeze(-1))
  out.scatter_(2, (low - lo).long().unsqueeze(-1), (1 - p_high).unsqueeze(-1))
  return out


def soft_cross_entropy(logits, target):
  return (-target * torch.log_softmax(logits, dim=1)).sum(1)


def make_optimizer(config, params, capturable=False):
  """utils.get_optimizer (utils.py:73-83).  capturable: the step runs inside a captured graph -- step counters and the
  learning rate live in device tensors (the schedulers below write the rate in place)."""
  name = getattr(config, 'optimizer', 'AdamW')
  lr, wd = config.lr_init, getattr(config, 'weight_decay', 1e-4)
  if capturable and name in ('AdamW', 'Adam'):
    params = list(params)
    lr = torch.tensor(float(lr), dtype=torch.float32, device=params[0].device)
    cls = torch.optim.AdamW if name == 'AdamW' else torch.optim.Adam
    # fused: ONE multi-tensor kernel for all 22 parameters (the foreach form spends ~150 launches per step on the
    # per-parameter step counters and bias corrections)
    return cls(params, lr=lr, weight_decay=wd, eps=0.00015, capturable=True, fused=True)
  if name == 'AdamW':
    return torch.optim.AdamW(params, lr=lr, weight_decay=wd, eps=0.00015)
  if name == 'Adam':
    return torch.optim.Adam(params, lr=lr, weight_decay=wd, eps=0.00015)
  if name == 'RMSprop':
    return torch.optim.RMSprop(params, lr=lr, momentum=getattr(config, 'momentum', 0.9), eps=0.01, weight_decay=wd)
  if name == 'SGD':
    return torch.optim.SGD(params, lr=lr, momentum=getattr(config, 'momentum', 0.9), weight_decay=wd)
  raise NotImplementedError(name)


def _set_lr(optimizer, lr):
  """a capturable optimiser keeps its rate in a device tensor a captured graph reads: written in place (no host sync)"""
  for g in optimizer.param_groups:
    if torch.is_tensor(g['lr']):
      g['lr'].fill_(lr)
    else:
      g['lr'] = lr


class MuZeroLR(object):
  """lr_init * decay_rate ** (step / decay_steps) (utils.py:85-99)"""

  def __init__(self, optimizer, config):
    self.optimizer, self.lr_init = optimizer, config.lr_init
    self.rate, self.steps, self.lr_step, self.lr = config.lr_decay_rate, config.lr_decay_steps, 0, config.lr_init

  def step(self):
    self.lr_step += 1
    self.lr = self.lr_init * self.rate ** (self.lr_step / self.steps)
    _set_lr(self.optimizer, self.lr)


class ExponentialLR(object):
  """torch.optim.lr_scheduler.ExponentialLR(optimizer, lr_decay_rate) (utils.py:124-125): lr_init * rate ** step, as a closed
  form on the host (the torch class reads the rate back from the optimiser every step: a device sync when it is a tensor)"""

  def __init__(self, optimizer, config):
    self.optimizer, self.lr_init, self.rate, self.lr_step, self.lr = optimizer, config.lr_init, config.lr_decay_rate, 0, config.lr_init

  def step(self):
    self.lr_step += 1
    self.lr = self.lr_init * self.rate ** self.lr_step
    _set_lr(self.optimizer, self.lr)


class WarmUpLR(object):
  """linear ramp to lr_init over 5000 steps, then constant (utils.py:102-118)"""

  def __init__(self, optimizer, config, warm_up_steps=5000):
    self.optimizer, self.max_lr, self.n, self.lr_step = optimizer, config.lr_init, warm_up_steps, 0
    self._set(1 / self.n * self.max_lr)

  def _set(self, lr):
    self.lr = lr
    _set_lr(self.optimizer, lr)

  def step(self):
    self.lr_step += 1
    if self.lr_step <= self.n:
      self._set(self.lr_step / self.n * self.max_lr)


def make_lr_scheduler(config, optimizer):
  """utils.get_lr_scheduler (utils.py:121-132)"""
  name = getattr(config, 'lr_scheduler', None)
  if name is None:
    return None
  if name == 'ExponentialLR':
    return ExponentialLR(optimizer, config)
  if name == 'MuZeroLR':
    return MuZeroLR(optimizer, config)
  if name == 'WarmUpLR':
    return WarmUpLR(optimizer, config)
  raise NotImplementedError(name)


def _stream_ptr(t):
  import ctypes as C
  return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _SoftCE(torch.autograd.Function):
  """sum over the unroll positions of (-target * log_softmax(logits)).sum(-1)  (utils.py:53-60, learners.py:191-203) as one
  HIP launch forward and one backward (mz_soft_ce_forward / _backward) instead of ~11 elementwise PyTorch kernels.
  logits [P, bs, S] contiguous; target: any float32 tensor whose element (p, b, s) sits at p * tp + b * tb + s."""

  @staticmethod
  def forward(ctx, logits, target, tp, tb):
    import ctypes as C
    from . import _abi
    P, bs, S = logits.shape
    loss = torch.empty(bs, dtype=torch.float32, device=logits.device)
    _abi.check(_abi.load().mz_soft_ce_forward(C.c_void_p(logits.data_ptr()), C.c_void_p(target.data_ptr()), P, bs, S, tp, tb,
                                              C.c_void_p(loss.data_ptr()), _stream_ptr(logits)), 'mz_soft_ce_forward')
    ctx.save_for_backward(logits, target)
    ctx.strides = (tp, tb)
    return loss

  @staticmethod
  def backward(ctx, grad_loss):
    import ctypes as C
    from . import _abi
    logits, target = ctx.saved_tensors
    P, bs, S = logits.shape
    g = grad_loss.to(torch.float32).contiguous()
    out = torch.empty_like(logits)
    _abi.check(_abi.load().mz_soft_ce_backward(C.c_void_p(logits.data_ptr()), C.c_void_p(target.data_ptr()), C.c_void_p(g.data_ptr()),
                                               P, bs, S, ctx.strides[0], ctx.strides[1], C.c_void_p(out.data_ptr()),
                                               _stream_ptr(logits)), 'mz_soft_ce_backward')
    return out, None, None, None


class _NativeFC(object):
  """The FCNetwork update as the HIP launches of csrc/mz_fcl.hip.h (mz_fcl_step, include/mz_engine.h; two at batch 256): forward
  chain + heads (losses, their backward) + backward chain + the heads' weight gradients handing over inside one launch, the chain's weight gradients -- each strip
  followed by Adam / AdamW on its weights in the same workgroup -- no GEMM library, no
  autograd tape, nothing PyTorch launches.  The network's parameters and the optimiser's exp_avg / exp_avg_sq / step
  tensors become VIEWS of three flat device vectors (engine.WEIGHT_ORDER) the kernels update in place, so state_dicts,
  checkpoints, get_weights and the PyTorch step itself keep working on the same storage.  The kernels read the weights
  from packed fragment-order copies they rewrite at every step; a write from PyTorch's side (load_state_dict, the
  graph capture's restore) is noticed through the parameters' version counters and repacked (sync)."""

  @staticmethod
  def eligible(learner, host):
    from .networks import FCNetwork
    cfg, net = learner.config, learner.network
    if learner.device.type != 'cuda' or not isinstance(net, FCNetwork) or getattr(cfg, 'no_native_learner', False):
      return False
    if getattr(cfg, 'no_support', False) or getattr(cfg, 'optimizer', 'AdamW') not in ('AdamW', 'Adam') or not learner.use_graph:
      return False
    g = learner.optimizer.param_groups[0]
    if len(learner.optimizer.param_groups) != 1 or g.get('amsgrad') or g.get('maximize') or not torch.is_tensor(g['lr']):
      return False
    bs, K, A = host['obs'].shape[0], host['act'].shape[1], net.action_space
    Sv = cfg.value_support_max - cfg.value_support_min + 1
    Sr = cfg.reward_support_max - cfg.reward_support_min + 1
    return (bs % 16 == 0 and 1 <= K <= 7 and A <= 14 and Sv <= 64 and Sr <= 64 and host['obs'].ndim == 2 and host['obs'].shape[1] <= 1024 and
            host['act'].dtype in (np.int64, np.int32) and host['w'].dtype in (np.float64, np.float32) and
            all(host[k].dtype == np.float32 for k in ('obs', 't_rew', 't_val', 't_pol')))

  def __init__(self, learner, host):
    import ctypes as C
    from . import _abi
    from .engine import WEIGHT_ORDER
    cfg, net, dev, opt = learner.config, learner.network, learner.device, learner.optimizer
    self.lib, self.learner, self.dev = _abi.load(), learner, dev
    self.bs, self.K = host['obs'].shape[0], host['act'].shape[1]
    self.shape = {k: host[k].shape for k in _GraphedUpdate.ORDER}
    self.h = C.c_void_p()
    with torch.cuda.device(dev):
      _abi.check(self.lib.mz_fcl_create(self.bs, self.K, host['obs'].shape[1], net.action_space, int(cfg.value_support_min),
                                        int(cfg.value_support_max), int(cfg.reward_support_min), int(cfg.reward_support_max),
                                        int(bool(cfg.no_target_transform)), C.byref(self.h)), 'mz_fcl_create')
    named = dict(net.named_parameters())
    self.params = [named[k] for k in WEIGHT_ORDER]
    n = sum(p.numel() for p in self.params)
    assert n == self.lib.mz_fcl_num_params(self.h) and len(self.params) == len(list(net.parameters())), 'FCNetwork parameter layout'
    self.flat = torch.empty(n, dtype=torch.float32, device=dev)
    self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
    self.steps = torch.zeros(len(self.params), dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
      for i, p in enumerate(self.params):
        k = p.numel()
        st = opt.state[p]
        self.flat[off:off + k].copy_(p.reshape(-1))
        if 'exp_avg' in st:
          self.m[off:off + k].copy_(st['exp_avg'].reshape(-1))
          self.v[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
          self.steps[i] = float(st['step'])
        p.data = self.flat[off:off + k].view(p.shape)
        st['step'] = self.steps[i]
        st['exp_avg'] = self.m[off:off + k].view(p.shape)
        st['exp_avg_sq'] = self.v[off:off + k].view(p.shape)
        off += k
    g = opt.param_groups[0]
    self.lr = g['lr']
    self.versions = None
    self._source = None           # (replay object, its mz_fcl_source table) of run()
    self.in_flight = False        # run() returned with updates in flight whose refreshes are owed
    self._rng = None              # generator states taken over by run() (release_rng hands them back)
    self._rng_exclusive = True    # nobody else has been seen drawing from the global generators while run() held them
    self.sync(force=True)

  def fits(self, host):
    return all(host[k].shape == self.shape[k] for k in _GraphedUpdate.ORDER)

  def sync(self, force=False):
    """the packed weight copies follow the parameters: (re)bind after a write that did not come from mz_fcl_step"""
    import ctypes as C
    from . import _abi
    ver = tuple(p._version for p in self.params)
    if force or ver != self.versions:
      ptr = lambda t: C.c_void_p(t.data_ptr())
      _abi.check(self.lib.mz_fcl_bind(self.h, ptr(self.flat), ptr(self.m), ptr(self.v), ptr(self.steps), len(self.params),
                                      ptr(self.lr), _stream_ptr(self.flat)), 'mz_fcl_bind')
      self.versions = ver

  def step(self, obs, act, t_rew, t_val, t_pol, w, no_update=False):
    import ctypes as C
    from . import _abi
    cfg, g = self.learner.config, self.learner.optimizer.param_groups[0]
    ptr = lambda t: C.c_void_p(t.data_ptr())
    new_errors = torch.empty(self.bs, dtype=torch.float32, device=self.dev)
    b1, b2 = g['betas']
    _abi.check(self.lib.mz_fcl_step(self.h, ptr(obs), ptr(act), int(act.dtype == torch.int32), ptr(t_rew), ptr(t_val), ptr(t_pol), ptr(w), int(w.dtype == torch.float64),
                                    float(b1), float(b2), float(g['eps']), float(g['weight_decay']), float(getattr(cfg, 'clip_grad', 0) or 0),
                                    int(isinstance(self.learner.optimizer, torch.optim.AdamW)), int(bool(no_update)), ptr(new_errors),
                                    ptr(self.learner._loss_dev), _stream_ptr(self.flat)), 'mz_fcl_step')
    return new_errors

  def launch(self, host):
    """one update from the host batch (mz_fcl_update: pinned staging, one copy in, the step, the new errors on their way
    back); returns the slot errors() hands them over from"""
    import ctypes as C
    from . import _abi
    self.flush()
    cfg, g = self.learner.config, self.learner.optimizer.param_groups[0]
    ptr = lambda a: C.c_void_p(a.__array_interface__['data'][0])
    b1, b2 = g['betas']
    slot = C.c_int(0)
    _abi.check(self.lib.mz_fcl_update(self.h, ptr(host['obs']), ptr(host['act']), int(host['act'].dtype == np.int32), ptr(host['t_rew']),
                                      ptr(host['t_val']), ptr(host['t_pol']), ptr(host['w']), int(host['w'].dtype == np.float64),
                                      float(b1), float(b2), float(g['eps']), float(g['weight_decay']), float(getattr(cfg, 'clip_grad', 0) or 0),
                                      int(isinstance(self.learner.optimizer, torch.optim.AdamW)), C.c_void_p(self.learner._loss_dev.data_ptr()),
                                      _stream_ptr(self.flat), C.byref(slot)), 'mz_fcl_update')
    return slot.value

  def run(self, replay, n, lrs=None):
    """n updates of Learner.learn's loop body in ONE native call (mz_fcl_run): batches sampled straight into pinned staging by
    the native replay, the step's launches per update, priority refreshes handed to the replay as their errors arrive.
    replay: the PrioritizedReplay OBJECT (its native handle is called from this thread; the handle's own lock serialises it
    with the actors' ingest).  The states of Python's `random` generator (the stratified draws) and of numpy's legacy one (the
    padded actions, replay_buffer.py:150-151) and the replay's beta travel in and out."""
    import ctypes as C
    import random
    from . import _abi
    cfg, g, lib = self.learner.config, self.learner.optimizer.param_groups[0], self.lib
    if self._source is None or self._source[0] is not replay:
      rlib = _abi.load_replay()
      src = _abi.MzFclSource(replay._h.value if hasattr(replay._h, 'value') else replay._h,
                             C.cast(rlib.mzr_sample_batches_full, C.c_void_p), C.cast(rlib.mzr_update_errors_f32, C.c_void_p),
                             C.cast(rlib.mzr_last_error, C.c_void_p))
      self._source = (replay, src)
    src = self._source[1]
    # the two generators the reference's sample_batch draws from -- Python's `random` (the stratified draws) and numpy's legacy
    # global one (the padded actions) -- are MT19937 states: taken over once (random.getstate / np.random.get_state), advanced in
    # place by the native calls, handed back by release_rng() (flush, the end of learn()): no 1.6-Mbit integer per call
    if self._rng is not None and self._rng_exclusive:
      # the private copy is only right while NOBODY else draws from the process-global generators (the reference's replay
      # is a process of its own; here a host-environment actor or the Python sample_batch path may share them).  The globals must
      # still be what they were at the take-over; if not, this learner stops holding them across calls: from now on the states
      # go back after every segment (concurrent consumers then interleave per segment, as two threads would anyway)
      st, ns = self._rng[4], self._rng[5]
      now = np.random.get_state()
      if random.getstate() != st or int(now[2]) != int(ns[2]) or not np.array_equal(now[1], ns[1]):
        import sys
        print('native learner loop: another consumer drew from the global `random` / numpy generators while the loop held their states; '
              'handing them back after every segment from here on', file=sys.stderr)
        self._rng_exclusive = False
        self.release_rng()
    if self._rng is None:
      st, ns = random.getstate(), np.random.get_state()
      self._rng = (np.array(st[1][:624], np.uint32), C.c_int32(int(st[1][624])), np.array(ns[1], np.uint32), C.c_int32(int(ns[2])), st, ns)
    py_key, py_pos, key, pos = self._rng[:4]
    beta, pads = C.c_double(float(replay.beta)), C.c_int64(0)
    norm = getattr(cfg, 'norm_obs', False)
    O = int(np.prod(self.shape['obs'][1:]))
    mn = np.ascontiguousarray(np.broadcast_to(self.learner.obs_min.reshape(-1), (O,)), np.float32) if norm else None
    rg = np.ascontiguousarray(np.broadcast_to(self.learner.obs_range.reshape(-1), (O,)), np.float32) if norm else None
    lr = None if lrs is None else np.ascontiguousarray(lrs, np.float32)
    ptr = lambda a: None if a is None else C.c_void_p(a.__array_interface__['data'][0])
    b1, b2 = g['betas']
    _abi.check(lib.mz_fcl_run(self.h, C.byref(src), int(n), None, ptr(key), C.byref(pos), C.byref(beta), ptr(mn), ptr(rg),
                              float(b1), float(b2), float(g['eps']), float(g['weight_decay']), float(getattr(cfg, 'clip_grad', 0) or 0),
                              int(isinstance(self.learner.optimizer, torch.optim.AdamW)), ptr(lr),
                              C.c_void_p(self.learner._loss_dev.data_ptr()), _stream_ptr(self.flat), C.byref(pads), ptr(py_key),
                              C.byref(py_pos)), 'mz_fcl_run')
    replay.beta = np.float64(beta.value) if float(replay.beta) < 1 else replay.beta
    self.in_flight = n > 0 or (self.in_flight and n != 0)
    if not self._rng_exclusive:
      self.release_rng()
    return int(pads.value)

  def flush(self):
    """the updates mz_fcl_run left in flight: wait for them, hand their priority refreshes to the replay (mz_fcl_run with 0 updates)"""
    if self.in_flight and self._source is not None:
      self.run(self._source[0], 0)
      self.in_flight = False
    self.release_rng()

  def release_rng(self):
    """hand the generator states run() took over back to `random` and numpy (advanced by what the native calls drew)"""
    if self._rng is not None:
      import random
      py_key, py_pos, key, pos, st, ns = self._rng
      self._rng = None
      random.setstate((st[0], tuple(int(x) for x in py_key) + (int(py_pos.value),), st[2]))
      np.random.set_state((ns[0], key, int(pos.value), ns[3], ns[4]))

  def run_stats(self, reset=False):
    """where mz_fcl_run's host time went (development hook): microseconds per update"""
    import ctypes as C
    out = (C.c_double * 6)()
    self.lib.mz_fcl_run_stats(self.h, out, int(bool(reset)))
    n = max(1.0, out[5])
    return {'updates': int(out[5]), 'wait_us': 1e6 * out[0] / n, 'refresh_us': 1e6 * out[1] / n, 'sample_us': 1e6 * out[2] / n,
            'launch_us': 1e6 * out[3] / n, 'call_us': 1e6 * out[4] / n}

  def errors(self, slot):
    import ctypes as C
    from . import _abi
    out = np.empty(self.bs, np.float32)
    _abi.check(self.lib.mz_fcl_errors(self.h, int(slot), C.c_void_p(out.__array_interface__['data'][0])), 'mz_fcl_errors')
    return out

  def grad(self):
    """the last step's gradient as {parameter name: tensor} (tests)"""
    import ctypes as C
    from . import _abi
    from .engine import WEIGHT_ORDER
    out = np.empty(self.flat.numel(), np.float32)
    _abi.check(self.lib.mz_fcl_read_grad(self.h, out.ctypes.data_as(C.c_void_p), out.size), 'mz_fcl_read_grad')
    res, off = {}, 0
    for k, p in zip(WEIGHT_ORDER, self.params):
      res[k] = torch.from_numpy(out[off:off + p.numel()].reshape(tuple(p.shape)).copy())
      off += p.numel()
    return res

  def close(self):
    self.release_rng()
    if self.h:
      self.lib.mz_fcl_destroy(self.h)
      self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


class _GraphedUpdate(object):
  """Learner._device_step as one captured graph over static tensors (torch.cuda.CUDAGraph = a hipGraph on ROCm).
  launch(host arrays): one pinned staging buffer -> ONE host-to-device copy -> graph replay -> the new errors back into
  pinned memory, all on the current stream, nothing waited for; errors(slot) waits for that copy.  Two staging slots, so
  that batch i + 1 is staged while update i runs."""
  ORDER = ('obs', 'act', 't_rew', 't_val', 't_pol', 'w')

  def __init__(self, learner, host):
    dev = learner.device
    self.learner = learner
    self.meta = {k: (host[k].shape, host[k].dtype) for k in self.ORDER}
    # every input in one byte buffer (8-byte aligned pieces): one copy per update instead of six
    off, self.slices = 0, {}
    for k in self.ORDER:
      n = host[k].nbytes
      self.slices[k] = (off, n)
      off += (n + 7) & ~7
    self.stage = [torch.empty(off, dtype=torch.uint8).pin_memory() for _ in range(2)]
    self.stage_np = [st.numpy() for st in self.stage]
    # typed views of every input's piece of a staging slot: filling a slot is one np.copyto per input
    self.stage_views = [{k: buf[self.slices[k][0]:self.slices[k][0] + self.slices[k][1]].view(self.meta[k][1]).reshape(self.meta[k][0])
                         for k in self.ORDER} for buf in self.stage_np]
    self.dev_bytes = torch.empty(off, dtype=torch.uint8, device=dev)
    self.static = {}
    for k in self.ORDER:
      o, n = self.slices[k]
      self.static[k] = self.dev_bytes[o:o + n].view(torch.from_numpy(np.empty(0, self.meta[k][1])).dtype).view(self.meta[k][0])
    bs = host['obs'].shape[0]
    self.err_host = [torch.empty(bs, dtype=torch.float32).pin_memory() for _ in range(2)]
    self.events = [torch.cuda.Event(), torch.cuda.Event()]
    self.slot = 0
    self._fill(0, host)
    self.dev_bytes.copy_(self.stage[0], non_blocking=True)
    # capture must not train: remember parameters, optimiser state and loss sums, warm up + capture on a side stream, put
    # everything back IN PLACE (the graph holds the addresses)
    net, opt = learner.network, learner.optimizer
    params = [p for g in opt.param_groups for p in g['params']]
    saved_p = [p.detach().clone() for p in params]
    saved_s = [{k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in opt.state.get(p, {}).items()} for p in params]
    # (module buffers too: BatchNorm's running_mean / running_var / num_batches_tracked take a momentum update in every
    # training-mode forward -- four of them here -- and travel to the actors with get_weights)
    buffers = list(net.buffers())
    saved_b = [b.detach().clone() for b in buffers]
    saved_l = learner._loss_dev.clone()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
      for _ in range(3):
        learner._device_step(*[self.static[k] for k in self.ORDER])
    torch.cuda.current_stream(dev).wait_stream(side)
    self.graph = torch.cuda.CUDAGraph()
    # (thread_local: an actor's thread of the same process keeps launching and copying while this thread captures)
    with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
      self.new_errors = learner._device_step(*[self.static[k] for k in self.ORDER])
    with torch.no_grad():
      for p, sp, ss in zip(params, saved_p, saved_s):
        p.copy_(sp)
        for k, v in opt.state[p].items():
          if torch.is_tensor(v):
            v.copy_(ss[k]) if k in ss else v.zero_()
      for b, sb in zip(buffers, saved_b):
        b.copy_(sb)
      learner._loss_dev.copy_(saved_l)

  def fits(self, host):
    return all(host[k].shape == self.meta[k][0] and host[k].dtype == self.meta[k][1] for k in self.ORDER)

  def _fill(self, slot, host):
    views = self.stage_views[slot]
    for k in self.ORDER:
      np.copyto(views[k], host[k])

  def launch(self, host):
    slot = self.slot
    self.slot ^= 1
    self.events[slot].synchronize()          # (the update that used this slot two launches ago has read it)
    self._fill(slot, host)
    self.dev_bytes.copy_(self.stage[slot], non_blocking=True)
    self.graph.replay()
    self.err_host[slot].copy_(self.new_errors, non_blocking=True)
    self.events[slot].record()
    return slot

  def errors(self, slot):
    self.events[slot].synchronize()
    return self.err_host[slot].numpy().copy()


class _BatchSource(object):
  """The learner's side of the replay during learn(): batches sampled ahead of the updates that consume them and priority
  refreshes sent without waiting -- the reference's own pattern (learners.py:124-127: `batches_per_fetch`
  sample_batch.remote() calls in flight; learners.py:182: replay_buffer.update.remote fire-and-forget).
    * a rayshim / ray handle (train.py: the replay is shared with the actors): up to `depth` batches in flight on the handle's
      worker thread, which runs sampling, refreshes and the actors' ingest in submission order;
    * a plain in-process replay (nothing else touches it): sampled in the learner's own thread, two batches per native call --
      the GPU works on the previous updates meanwhile (their launches are asynchronous); measured steadier and faster than a
      private worker thread (5.5-5.9 k against 4.0-6.2 k updates/s: two Python threads share one interpreter lock).
  Batches come as arrays (replay_buffer.sample_batch_arrays / sample_batches_arrays) where the replay offers them.
  depth <= batches_per_fetch: a batch's priorities are at least as fresh as in the reference."""

  def __init__(self, replay, depth):
    from collections import deque
    self.replay, self.depth, self.inflight, self.sent, self.ready = replay, max(1, int(depth)), deque(), deque(), deque()
    self.remote = hasattr(getattr(replay, 'sample_batch'), 'remote')
    obj = getattr(replay, '_obj', replay)
    # the native replay samples two batches per call (sample_batches_arrays: half the per-call host overhead per batch)
    self.multi = 2 if callable(getattr(obj, 'sample_batches_arrays', None)) and self.depth >= 2 else 1
    arrays = callable(getattr(obj, 'sample_batch_arrays', None))
    self.method = getattr(replay, 'sample_batches_arrays' if self.multi > 1 else ('sample_batch_arrays' if arrays else 'sample_batch'))

  def _result(self, fut):
    return fut.result() if hasattr(fut, 'result') else __import__('ray').get(fut)

  def _submit(self):
    self.inflight.append(self.method.remote(self.multi) if self.multi > 1 else self.method.remote())

  def get(self):
    if not self.ready:
      if self.remote:
        while len(self.inflight) * self.multi < self.depth:
          self._submit()
        got = self._result(self.inflight.popleft())
        self._submit()
      else:
        got = self.method(self.multi) if self.multi > 1 else self.method()
      self.ready.extend(got if self.multi > 1 else [got])
    return self.ready.popleft()

  def settle(self):
    """collect what was submitted: the batches in flight become `ready`, the refreshes sent have been applied"""
    while self.inflight:
      got = self._result(self.inflight.popleft())
      self.ready.extend(got if self.multi > 1 else [got])
    while self.sent:
      self._result(self.sent.popleft())

  def update(self, idxs, errors):
    """fire-and-forget on a handle (a refresh that failed is reported at the next one); a direct call on a plain replay"""
    if not self.remote:
      self.replay.update(idxs, errors)
      return
    while self.sent and (not hasattr(self.sent[0], 'done') or self.sent[0].done()):
      self._result(self.sent.popleft())
    self.sent.append(self.replay.update.remote(idxs, errors))

  def close(self):
    for fut in list(self.inflight) + list(self.sent):      # (the handle's thread finishes what was submitted)
      try:
        self._result(fut)
      except Exception:
        pass
    self.inflight.clear(); self.sent.clear()


class Learner(Logger):

  def __init__(self, config, storage, replay_buffer, state=None):
    set_all_seeds(config.seed)
    self.config = deepcopy(config)
    self.run_tag, self.group_tag = getattr(config, 'run_tag', None) or 'run', getattr(config, 'group_tag', None)
    self.worker_id = 'learner'
    self.storage, self.replay_buffer = storage, replay_buffer
    if 'learner' in getattr(config, 'use_gpu_for', []):          # learners.py:27-37
      if not torch.cuda.is_available():
        raise RuntimeError('GPU was requested but torch.cuda.is_available() is False.')
      dev_id = getattr(config, 'learner_gpu_device_id', None)
      self.device = torch.device('cuda', dev_id if dev_id is not None else torch.cuda.current_device())
    else:
      self.device = torch.device('cpu')
    self.network = get_network(config, self.device)          # utils.get_network (utils.py:21-37)
    self.network.train()
    # one captured hipGraph per update on a GPU (module docstring); the optimisers with a capturable step only
    self.use_graph = (self.device.type == 'cuda' and not getattr(config, 'no_graph_learner', False) and
                      getattr(config, 'optimizer', 'AdamW') in ('AdamW', 'Adam'))
    self.optimizer = make_optimizer(config, self.network.parameters(), capturable=self.use_graph)
    self.lr_scheduler = make_lr_scheduler(config, self.optimizer)
    if getattr(config, 'scalar_loss', 'MSE') not in ('MSE', 'Huber'):
      raise NotImplementedError(config.scalar_loss)
    self.training_step = 0
    self._losses = {'reward': 0., 'value': 0., 'policy': 0.}
    self._loss_dev = torch.zeros(3, dtype=torch.float64, device=self.device)      # summed on the device, read when logged
    # the elementwise ends of the FCNetwork step as single HIP launches (csrc/mz_learner.hip.h); PyTorch's own elementwise
    # kernels on a CPU learner and with --no_hip_learner_ops
    self.hip_ops = self.device.type == 'cuda' and not getattr(config, 'no_hip_learner_ops', False)
    self._graph = None          # _GraphedUpdate, built from the first batch
    self._native = None         # _NativeFC (FCNetwork, Adam / AdamW, categorical losses), built from the first batch
    self._source = None         # _BatchSource while learn() runs
    self._pending = None        # (idxs, slot) of the update whose priority refresh has not reached the replay yet
    self.native_loop_updates = 0      # updates taken by mz_fcl_run (the loop body in native code)
    self.throughput = {'total_frames': 0, 'total_games': 0, 'training_step': 0, 'time': {'ups': 0, 'fps': 0}}
    self.last_throughput = {}
    if getattr(config, 'norm_obs', False):
      self.obs_min = np.array(config.obs_range[::2], dtype=np.float32)
      self.obs_range = np.array(config.obs_range[1::2], dtype=np.float32) - self.obs_min
    if state is not None:
      self.load_state(state)
    Logger.__init__(self)
    self.saves_dir = self.dirs['saves']

  def _tune_gemms(self):
    """PyTorch's TunableOp for the captured PyTorch step (the native step has no use for it: switched on when the first
    PyTorch graph is built)"""
    if getattr(self, '_tuned', False) or getattr(self.config, 'no_tune_gemms', False):
      return
    self._tuned = True
    # PyTorch's TunableOp: every GEMM shape of the step (2 x 512-wide layers on 256..1536 rows) is timed once against the
    # rocBLAS / hipBLASLt solutions during the graph's warm-up and the fastest one is what gets captured -- the default
    # heuristic picks 64 x 256 tiles for the skinny weight-gradient GEMMs ([512 x 54] over K = 256: 35 us each, a
    # quarter of the update; profiles/r04_learner_kernel_stats_untuned.csv).  Summation order may differ between
    # solutions: float32-rounding-level, inside every parity bound of tests/test_learner.py.
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(20)
    import tempfile
    tunable.set_filename(os.path.join(tempfile.gettempdir(), 'mz_tunableop_%d.csv' % os.getpid()))      # (its exit-time dump: not into the cwd)

  # learners.py:62-70
  def load_state(self, state):
    self.run_tag = os.path.join(str(self.run_tag), 'resumed', '{}'.format(state['training_step']))
    if getattr(self, '_native', None) is not None:      # (the optimiser's loaded state replaces the flat views: rebuilt at the next update)
      self._native.close()
    # a captured graph holds the OLD exp_avg / exp_avg_sq / step / lr tensors: load_state_dict below replaces them
    self._native, self._graph = None, None
    self.network.load_state_dict(state['weights'])
    # the optimiser's state comes from the checkpoint, HOW it steps (capturable / fused / foreach, the learning rate as a
    # device tensor the captured graph reads) stays this learner's: a checkpoint of an eager learner resumes under a graphed
    # one and vice versa
    keep = [{k: g.get(k) for k in ('capturable', 'fused', 'foreach', 'differentiable', 'maximize')} for g in self.optimizer.param_groups]
    self.optimizer.load_state_dict(state['optimizer'])
    for g, kept in zip(self.optimizer.param_groups, keep):
      g.update({k: v for k, v in kept.items() if k in g or v is not None})
      lr = float(g['lr'])
      g['lr'] = torch.tensor(lr, dtype=torch.float32, device=self.device) if self.use_graph else lr
    if self.use_graph:
      # a checkpoint of an EAGER learner leaves `step` a CPU tensor (Optimizer.load_state_dict casts it by the SAVED group's
      # capturable / fused flags); the capturable fused step of this learner wants it beside the parameter
      for p_, st in self.optimizer.state.items():
        if torch.is_tensor(st.get('step')):
          st['step'] = st['step'].to(device=p_.device, dtype=torch.float32)
    _call(self.replay_buffer, 'add_initial_throughput', state['total_frames'], state['total_games'])
    self.throughput['total_frames'] = state['total_frames']
    self.throughput['training_step'] = state['training_step']
    self.training_step = state['training_step']

  def _host_weights(self):
    """network.get_weights() (networks.py:39-40: the state_dict on the host).  With the native step the parameters are views of
    ONE flat device vector: one copy to the host and views of it, instead of one synchronising copy per tensor"""
    nat = self._native
    if nat is None or len(list(self.network.buffers())):
      return self.network.get_weights()
    from .engine import WEIGHT_ORDER
    host = nat.flat.detach().cpu()
    out, off = {}, 0
    for k, p in zip(WEIGHT_ORDER, nat.params):
      out[k] = host[off:off + p.numel()].view(p.shape)
      off += p.numel()
    return {k: out[k] for k in self.network.state_dict().keys()}

  # learners.py:72-83 (same dictionary keys)
  def save_state(self, path=None, wait=True):
    """wait=False (the loop's own periodic checkpoints): the state is copied to the host here, the FILE is written by a
    background thread -- the previous one is joined first, learn() joins the last"""
    thr = _call(self.replay_buffer, 'get_throughput')        # (the reference refreshes these in log_throughput)
    self.throughput['total_games'] = thr['games']
    self.throughput['total_frames'] = max(self.throughput['total_frames'], thr['frames'])
    self.flush_priorities()
    opt = self.optimizer.state_dict()
    for g in opt['param_groups']:               # (a plain float in the file, whatever this learner keeps it in)
      g['lr'] = float(g['lr'])
    # (views of the native step's flat vectors would each be saved with the whole vector's storage; and the writer thread
    # must not see tensors the next updates change: copies on the host)
    opt['state'] = {i: {k: (v.detach().to('cpu', copy=True) if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in opt['state'].items()}
    state = {'dirs': self.dirs, 'config': self.config, 'weights': self._host_weights(),
             'optimizer': opt, 'training_step': self.training_step,
             'total_games': self.throughput['total_games'], 'total_frames': self.throughput['total_frames'],
             'actor_games': _call(self.storage, 'get_stats', 'actor_games')}
    path = path or os.path.join(self.saves_dir, str(self.training_step))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    self._join_writer()
    if wait:
      torch.save(state, path)
    else:
      import threading
      self._writer = threading.Thread(target=torch.save, args=(state, path))
      self._writer.start()
    return path

  def _join_writer(self):
    w = getattr(self, '_writer', None)
    if w is not None:
      w.join()
      self._writer = None

  # learners.py:85-86
  def send_weights(self):
    _call(self.storage, 'store_weights', self._host_weights(), self.training_step)

  @property
  def losses_to_log(self):
    """{'reward', 'value', 'policy'}: loss sums since the last log (learners.py:228-230).  The steps add them up on the
    device; they come to the host when somebody looks."""
    acc = self._loss_dev.tolist()
    self._loss_dev.zero_()
    for i, k in enumerate(('reward', 'value', 'policy')):
      self._losses[k] += acc[i]
    return self._losses

  def _host_batch(self, batch):
    """the batch as the six arrays the step consumes (learners.py:165-180), normalised observations included.  Either the
    reference's tuple (sample_batch) or the array form (replay_buffer.sample_batch_arrays: no lists to convert)."""
    if isinstance(batch[0], dict):
      host, idxs = batch
      if getattr(self.config, 'norm_obs', False):
        host = dict(host, obs=np.ascontiguousarray((host['obs'] - self.obs_min) / self.obs_range, np.float32))
      return host, idxs
    (observations, actions, (target_rewards, target_values, target_policies)), idxs, is_weights = batch
    if getattr(self.config, 'norm_obs', False):
      observations = (observations - self.obs_min) / self.obs_range
    return {'obs': np.ascontiguousarray(observations, np.float32), 'act': np.ascontiguousarray(actions, np.int64),
            't_rew': np.ascontiguousarray(target_rewards), 't_val': np.ascontiguousarray(target_values),
            't_pol': np.ascontiguousarray(target_policies), 'w': np.ascontiguousarray(is_weights)}, idxs

  def _targets(self, value0, t_rew, t_val):
    """learners.py:176-189: the priority refresh (initial value against the first value target) and the categorical targets"""
    cfg = self.config
    no_support = getattr(cfg, 'no_support', False)
    init_value = value0 if no_support else support_to_scalar(value0, cfg.value_support_min, cfg.no_target_transform)
    new_errors = init_value.squeeze() - t_val[:, 0]
    if not cfg.no_target_transform:
      t_val, t_rew = scalar_transform(t_val), scalar_transform(t_rew)
    if not no_support:
      t_val = scalar_to_support(t_val, cfg.value_support_min, cfg.value_support_max)
      t_rew = scalar_to_support(t_rew, cfg.reward_support_min, cfg.reward_support_max)
    return new_errors, t_rew, t_val

  # learners.py:164-230 for FCNetwork with the three heads evaluated ONCE over all K + 1 unroll positions: only the
  # transition chain h_0 -> h_1 -> ... -> h_K is sequential (networks.py:158-165); value and policy of every position
  # (networks.py:151-156) and the rewards of the K transitions (networks.py:160-162) are batched GEMMs over (K + 1) bs
  # resp. K bs rows, the losses one expression over [K + 1, bs].  Same arithmetic per row, a third of the launches of the
  # position-by-position form below -- and this step is launch-bound (a captured update is ~3 ms of ~4 us launches).
  def _device_step_fc(self, obs, act, t_rew, t_val, t_pol, w):
    cfg, net = self.config, self.network
    K, bs, A = act.shape[1], obs.shape[0], net.action_space
    onehot = torch.nn.functional.one_hot(act, A).to(torch.float32)              # [bs, K, A]  (networks.py:167-174)
    h = net.representation(obs)
    hs, xs = [h], []
    for i in range(K):
      x = torch.cat((h, onehot[:, i]), dim=1)
      xs.append(x)
      h = torch.relu(net.LN(net.transition_head(x)))
      h.register_hook(lambda grad: grad * 0.5)                                  # learners.py:200
      hs.append(h)
    H = torch.cat(hs, dim=0)                                                    # [(K + 1) bs, 50], position-major
    value = net.value_head(H).view(K + 1, bs, -1)
    policy = net.policy_head(H).view(K + 1, bs, A)
    reward = net.reward_head(torch.cat(xs, dim=0)).view(K, bs, -1)
    no_support = getattr(cfg, 'no_support', False)
    if self.hip_ops and not no_support and K + 1 <= 8:
      # the elementwise ends as single HIP launches (csrc/mz_learner.hip.h): targets + priority refresh, and one fused
      # soft cross-entropy per head over all its positions
      import ctypes as C
      from . import _abi
      Sv, Sr = value.shape[2], reward.shape[2]
      sup_val = torch.empty(K + 1, bs, Sv, dtype=torch.float32, device=obs.device)
      sup_rew = torch.empty(K + 1, bs, Sr, dtype=torch.float32, device=obs.device)
      new_errors = torch.empty(bs, dtype=torch.float32, device=obs.device)
      v0 = value[0].detach()
      ptr = lambda t: C.c_void_p(t.data_ptr())
      _abi.check(_abi.load().mz_learner_targets(ptr(t_val), ptr(t_rew), ptr(v0), bs, K + 1, Sv, int(cfg.value_support_min), Sr,
                                                int(cfg.reward_support_min), int(bool(cfg.no_target_transform)), ptr(sup_val),
                                                ptr(sup_rew), ptr(new_errors), _stream_ptr(obs)), 'mz_learner_targets')
      value_loss = _SoftCE.apply(value, sup_val, bs * Sv, Sv)
      reward_loss = _SoftCE.apply(reward, sup_rew[1:], bs * Sr, Sr)
      policy_loss = _SoftCE.apply(policy, t_pol, A, (K + 1) * A)
      return new_errors, reward_loss, value_loss, policy_loss
    with torch.no_grad():
      new_errors, t_rew, t_val = self._targets(value[0], t_rew, t_val)
    t_pol, t_val, t_rew = t_pol.transpose(0, 1), t_val.transpose(0, 1), t_rew.transpose(0, 1)      # [K + 1, bs, ...]
    policy_loss = (-t_pol * torch.log_softmax(policy, dim=2)).sum(2).sum(0)
    if not no_support:
      value_loss = (-t_val * torch.log_softmax(value, dim=2)).sum(2).sum(0)
      reward_loss = (-t_rew[1:] * torch.log_softmax(reward, dim=2)).sum(2).sum(0)
    else:
      fn = torch.nn.SmoothL1Loss(reduction='none') if getattr(cfg, 'scalar_loss', 'MSE') == 'Huber' else torch.nn.MSELoss(reduction='none')
      value_loss = fn(value.squeeze(2), t_val).sum(0)
      reward_loss = fn(reward.squeeze(2), t_rew[1:]).sum(0)
    return new_errors, reward_loss, value_loss, policy_loss

  # learners.py:164-230 on device tensors: no host round trip inside, so it can be captured
  def _device_step(self, obs, act, t_rew, t_val, t_pol, w):
    cfg = self.config
    from .networks import FCNetwork
    if self._native is not None:          # the whole update as HIP launches (csrc/mz_fcl.hip.h)
      return self._native.step(obs, act, t_rew, t_val, t_pol, w)
    act = act.to(torch.int64)             # (sample_batch_arrays hands int32 actions over)
    if isinstance(self.network, FCNetwork) and not getattr(cfg, 'unbatched_learner', False):
      new_errors, reward_loss, value_loss, policy_loss = self._device_step_fc(obs, act, t_rew, t_val, t_pol, w)
      return self._finish_step(new_errors, reward_loss, value_loss, policy_loss, w)
    # any network (MuZeroNetwork / TinyNetwork; --unbatched_learner): position by position, as the reference writes it
    value, _, policy_logits, hidden = self.network.initial_inference(obs)
    no_support = getattr(cfg, 'no_support', False)
    with torch.no_grad():
      new_errors, t_rew, t_val = self._targets(value, t_rew, t_val)
    if not no_support:
      scalar_loss = soft_cross_entropy
    elif getattr(cfg, 'scalar_loss', 'MSE') == 'Huber':        # utils.py:62-70
      scalar_loss = torch.nn.SmoothL1Loss(reduction='none')
    else:
      scalar_loss = torch.nn.MSELoss(reduction='none')
    reward_loss = 0
    value_loss = scalar_loss(value.squeeze(), t_val[:, 0])
    policy_loss = soft_cross_entropy(policy_logits.squeeze(), t_pol[:, 0])
    # (the K action columns are on the device already: the reference hands recurrent_inference a Python tuple per unroll
    # step, learners.py:196-197 -- one pageable host-to-device copy per step)
    for i in range(1, act.shape[1] + 1):
      value, reward, policy_logits, hidden = self.network.recurrent_inference(hidden, act[:, i - 1])
      hidden.register_hook(lambda grad: grad * 0.5)
      reward_loss = reward_loss + scalar_loss(reward.squeeze(), t_rew[:, i])
      value_loss = value_loss + scalar_loss(value.squeeze(), t_val[:, i])
      policy_loss = policy_loss + soft_cross_entropy(policy_logits.squeeze(), t_pol[:, i])
    return self._finish_step(new_errors, reward_loss, value_loss, policy_loss, w)

  def _finish_step(self, new_errors, reward_loss, value_loss, policy_loss, w):
    """learners.py:208-230: importance weights, 1 / K gradient scale, backward, clipping, optimiser step, loss sums"""
    cfg = self.config
    reward_loss, value_loss, policy_loss = (w * reward_loss).mean(), (w * value_loss).mean(), (w * policy_loss).mean()
    total = reward_loss + value_loss + policy_loss
    total.register_hook(lambda grad: grad * (1 / cfg.num_unroll_steps))
    self.optimizer.zero_grad(set_to_none=True)
    total.backward()
    if getattr(cfg, 'clip_grad', 0):
      torch.nn.utils.clip_grad_norm_(self.network.parameters(), cfg.clip_grad)
    self.optimizer.step()
    self._loss_dev += torch.stack((reward_loss.detach(), value_loss.detach(), policy_loss.detach())).to(torch.float64)
    return new_errors

  def flush_priorities(self):
    """hand the last update's priority refresh to the replay (learners.py:182), waiting for its copy to arrive"""
    if self._native is not None:
      self._native.flush()          # (the updates the native loop left in flight)
    if self._pending is not None:
      idxs, slot, getter = self._pending
      self._pending = None
      if getattr(self, '_source', None) is not None:
        self._source.update(idxs, getter(slot))
      else:
        _call(self.replay_buffer, 'update', idxs, getter(slot))

  def update_weights(self, batch, defer_priorities=False):
    """One training step (learners.py:164-230).  defer_priorities (learn()'s loop): this batch's new errors go to the replay
    at the NEXT call, i.e. while the following update is already running on the GPU."""
    host, idxs = self._host_batch(batch)
    if self.use_graph:
      if self._native is not None and not self._native.fits(host):
        self.flush_priorities()
        self._native.close()
        self._native = None
      if self._native is None and (self._graph is None or not self._graph.fits(host)) and _NativeFC.eligible(self, host):
        self.flush_priorities()
        self._native = _NativeFC(self, host)
      if self._native is not None:
        # FCNetwork: the step's HIP launches from the host batch (mz_fcl_update), no PyTorch operator, no graph to capture
        self._native.sync()
        slot = self._native.launch(host)
        getter = self._native.errors
      else:
        if self._graph is None or not self._graph.fits(host):
          self.flush_priorities()
          self._tune_gemms()
          self._graph = _GraphedUpdate(self, host)
        slot = self._graph.launch(host)
        getter = self._graph.errors
      self.flush_priorities()                    # the previous batch's, whose copy has had a whole update to arrive
      self._pending = (idxs, slot, getter)
      if not defer_priorities:
        self.flush_priorities()
    else:
      dev = self.device
      new_errors = self._device_step(*[torch.from_numpy(host[k]).to(dev) for k in _GraphedUpdate.ORDER])
      if self._source is not None:
        self._source.update(idxs, new_errors.detach().cpu().numpy())
      else:
        _call(self.replay_buffer, 'update', idxs, new_errors.detach().cpu().numpy())
    if self.lr_scheduler is not None:             # learners.py:225-226
      self.lr_scheduler.step()

  # learners.py:88-113: the reference's own throughput scalars -- frames_per_second is its env-steps/sec metric
  def log_throughput(self, force=False):
    data = _call(self.replay_buffer, 'get_throughput')
    self.throughput['total_games'] = data['games']
    self.log_scalar(tag='games/finished', value=data['games'], i=self.training_step)
    new_frames = data['frames'] - self.throughput['total_frames']
    if new_frames > getattr(self.config, 'frames_before_fps_log', 10000) or (force and new_frames > 0):
      now = time.time()
      new_updates = self.training_step - self.throughput['training_step']
      ups = new_updates / max(1e-9, now - self.throughput['time']['ups'])
      fps = new_frames / max(1e-9, now - self.throughput['time']['fps'])
      replay_ratio = ups / fps
      self.throughput['total_frames'] = data['frames']
      self.throughput['training_step'] = self.training_step
      self.throughput['time']['ups'] = self.throughput['time']['fps'] = now
      self.last_throughput = {'frames_per_second': fps, 'updates_per_second': ups, 'replay_ratio': replay_ratio,
                              'sample_ratio': self.config.batch_size * replay_ratio, 'total_frames': data['frames']}
      for k, v in self.last_throughput.items():
        self.log_scalar(tag='throughput/' + k, value=v, i=self.training_step)

  # learners.py:115-153
  def learn(self, max_steps=None):
    cfg = self.config
    self.send_weights()
    self.throughput['time']['fps'] = time.time()
    while _call(self.replay_buffer, 'size') < cfg.stored_before_train:
      time.sleep(0.05)
    self.throughput['time']['ups'] = time.time()
    last = cfg.training_steps if max_steps is None else min(cfg.training_steps, self.training_step + max_steps)
    log_every = max(1, getattr(cfg, 'learner_log_frequency', 100))
    gpu_turns = None
    if self.device.type == 'cuda' and getattr(cfg, 'gpu_turns', False):      # --gpu_turns: this GPU is shared with an actor of this process
      from . import gpu_turns
      gpu_turns.register(self.device, 'learner')
    # batches sampled a few updates ahead, priority refreshes fire-and-forget (_BatchSource; the reference's learners.py:124,182)
    depth = min(4, int(getattr(cfg, 'batches_per_fetch', 15)))
    # (where mz_fcl_run will take the loop body, nothing is sampled ahead from Python: the first update's batch -- it tells the
    # native step its shapes -- is sampled on its own, every later one inside the native call, the draws in one order)
    self._source = _BatchSource(self.replay_buffer, depth) if depth > 1 and not self._native_loop_possible() else None
    try:
      self._learn_loop(cfg, last, log_every, self._source, gpu_turns)
      self.flush_priorities()
      self._log_losses_behind(log_every, final=True)
      self._join_writer()
    finally:
      if self._source is not None:
        self._source.close()
        self._source = None
    self.log_throughput(force=True)
    self.send_weights()

  def _native_loop_possible(self):
    """what _native_segment will need, as far as it is known before the first batch"""
    from .networks import FCNetwork
    cfg = self.config
    replay = getattr(self.replay_buffer, '_obj', self.replay_buffer)
    return (self.device.type == 'cuda' and self.use_graph and isinstance(self.network, FCNetwork) and hasattr(replay, '_h') and
            hasattr(replay, 'sample_batches_arrays') and not getattr(cfg, 'no_native_learner', False) and
            not getattr(cfg, 'no_native_loop', False) and not getattr(cfg, 'no_support', False))

  def _native_segment(self, cfg, last, log_every):
    """-> how many updates mz_fcl_run may take from here: up to the next step at which the loop does something in Python
    (send_weights, save_state, logging; learners.py:132-153), on a GPU shared in turns at most --gpu_turn_updates; 0 where the
    native loop does not apply (no native step yet, a replay that is not the native one of this process, --no_native_loop)"""
    replay = getattr(self.replay_buffer, '_obj', self.replay_buffer)
    if (self._native is None or getattr(cfg, 'no_native_loop', False) or
        not hasattr(replay, 'sample_batches_arrays') or not hasattr(replay, '_h') or
        int(replay.batch_size) != self._native.bs or int(cfg.num_unroll_steps) != self._native.K):
      return 0, None
    step = self.training_step
    n = last - step
    for f in (cfg.send_weights_frequency, getattr(cfg, 'save_state_frequency', 1000), log_every):
      n = min(n, f - step % f)
    if getattr(cfg, 'gpu_turns', False):
      n = min(n, max(1, int(getattr(cfg, 'gpu_turn_updates', 8))))
    return int(n), replay

  def _scheduled_lrs(self, n):
    """the learning rates of the next n updates (update i runs with the rate set after update i - 1, learners.py:225-226) and
    the scheduler advanced by n steps"""
    sch = self.lr_scheduler
    if sch is None:
      return None
    lrs = np.empty(n, np.float32)
    for i in range(n):
      lrs[i] = sch.lr
      sch.step()
    return lrs

  def _learn_loop(self, cfg, last, log_every, prefetch, gpu_turns):
    while self.training_step < last:
      n, replay = self._native_segment(cfg, last, log_every)
      if n > 0:
        # the loop body in native code (mz_fcl_run): n updates per call, Python only at the boundaries of _after_update; the
        # call returns with its last updates in flight -- what _after_update does overlaps them
        if self._pending is not None:
          self.flush_priorities()
        self._native.sync()
        turn = gpu_turns.turn(self.device) if gpu_turns is not None else None
        if turn is not None and turn is not gpu_turns.NO_TURNS:
          with turn:      # (an actor on the same GPU: n <= --gpu_turn_updates updates per turn, the GPU idle again at its end)
            self._native.run(replay, n, self._scheduled_lrs(n))
            self._native.flush()
        else:
          self._native.run(replay, n, self._scheduled_lrs(n))
        self.native_loop_updates += n
        self.training_step += n
        self._after_update(cfg, log_every)
        continue
      if prefetch is not None:
        batch = prefetch.get()
      else:
        arrays = callable(getattr(getattr(self.replay_buffer, '_obj', self.replay_buffer), 'sample_batch_arrays', None))
        batch = _call(self.replay_buffer, 'sample_batch_arrays' if arrays else 'sample_batch')
      if gpu_turns is not None:
        turn = gpu_turns.turn(self.device)
        with turn:       # (an actor on the same GPU: one update per turn, see gpu_turns.py)
          self.update_weights(batch, defer_priorities=True)
          if turn is not gpu_turns.NO_TURNS:
            torch.cuda.current_stream(self.device).synchronize()
      else:
        self.update_weights(batch, defer_priorities=True)
      self.training_step += 1
      self._after_update(cfg, log_every)

  def _log_losses_behind(self, log_every, final=False):
    """loss/{reward,value,policy} (learners.py:138-141) without waiting for the GPU: the sums of this interval are copied to
    pinned memory in stream order (behind the updates in flight) and zeroed; what is WRITTEN now is the previous interval's
    line, with its own step.  final: write what is pending (the end of learn())."""
    prev = getattr(self, '_loss_behind', None)
    if prev is not None:
      step, buf, ev, n = prev
      ev.synchronize()                           # (recorded a whole interval ago)
      vals = buf.tolist()
      self.log_points([('loss/' + k, step, vals[i] / n) for i, k in enumerate(('reward', 'value', 'policy'))])
      self._loss_behind = None
    if final:
      return
    bufs = getattr(self, '_loss_bufs', None)
    if bufs is None:
      bufs = self._loss_bufs = [torch.empty(3, dtype=torch.float64).pin_memory() for _ in range(2)]
      self._loss_flip = 0
    buf = bufs[self._loss_flip]
    self._loss_flip ^= 1
    buf.copy_(self._loss_dev, non_blocking=True)
    self._loss_dev.zero_()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(self.device))
    self._loss_behind = (self.training_step, buf, ev, log_every)

  def _after_update(self, cfg, log_every):
    """learners.py:132-153: what the loop does besides training, at the steps where it does it"""
    if self.training_step % cfg.send_weights_frequency == 0:
      self.send_weights()
    if self.training_step % getattr(cfg, 'save_state_frequency', 1000) == 0:
      self.save_state(wait=False)
    if self.training_step % log_every == 0:
      if self.device.type == 'cuda' and self._native is not None:
        self._log_losses_behind(log_every)
      else:
        for k in ('reward', 'value', 'policy'):
          self.log_scalar(tag='loss/' + k, value=self.losses_to_log[k] / log_every, i=self.training_step)
          self.losses_to_log[k] = 0
      self.log_throughput()
      if self.lr_scheduler is not None:
        self.log_scalar(tag='loss/learning_rate', value=self.optimizer.param_groups[0]['lr'], i=self.training_step)      # (what the optimizer really uses, every scheduler)

  def get_last_throughput(self):
    return dict(self.last_throughput)

  def launch(self, max_steps=None):
    print('Learner is online on {}.'.format(self.device))
    self.learn(max_steps)
