"""PyTorch definitions of the networks (training side + weight source).

FCNetwork: only its *definition* lives here; its inference inside the search is the fused HIP kernel
(csrc/mz_fused.hip.h, mz_root.hip.h).  MuZeroNetwork / TinyNetwork (the conv networks of BASELINE.json configs[4],
reference networks.py:393-718): definitions AND inference are PyTorch-ROCm / MIOpen (SURVEY.md s2 row 11 -- no
hand-written conv kernels); they serve the search through the batched external-inference path (torch_search.py).
Parameter names and the order in which modules are constructed equal the reference's, so state_dicts, checkpoints and
the flat weight order (engine.WEIGHT_ORDER) are interchangeable and a seeded default initialisation draws the same
weights (tests/golden/g6_*).  Surface: initial_inference / recurrent_inference -> NetworkOutput(value, reward,
policy_logits, hidden_state), load_weights / get_weights (networks.py:9-52); `action` may be a list (reference call
style), a numpy array or a device tensor (batched search: no host round trip).
"""
from collections import namedtuple

import torch
from torch import nn

NetworkOutput = namedtuple('network_output', ('value', 'reward', 'policy_logits', 'hidden_state'))

HIDDEN = 50
WIDTH = 512


class _TwoLayer(nn.Module):
  """Linear(in, 512) -> ReLU -> Linear(512, out); the second layer's attribute name varies per head."""

  def __init__(self, n_in, n_out, out_name):
    super().__init__()
    self.fc1 = nn.Linear(n_in, WIDTH)
    setattr(self, out_name, nn.Linear(WIDTH, n_out))
    self._out_name = out_name

  def forward(self, x):
    return getattr(self, self._out_name)(torch.relu(self.fc1(x.flatten(1))))


def support_to_scalar(logits, support_min, no_target_transform=False):
  """softmax expectation over the integer support + inverse of h(x)=sign(x)(sqrt(|x|+1)-1)+0.001x
  (reference config.py:27-33), float32."""
  p = torch.softmax(logits, dim=1)
  support = torch.arange(support_min, support_min + logits.shape[1], dtype=torch.float32, device=logits.device)
  x = (p * support).sum(1, keepdim=True)
  if no_target_transform:
    return x
  return torch.sign(x) * (((torch.sqrt(1 + 4 * 0.001 * (torch.abs(x) + 1 + 0.001)) - 1) / (2 * 0.001)) ** 2 - 1)


class FCNetwork(nn.Module):

  def __init__(self, input_dim, action_space, device, config):
    super().__init__()
    self.device = device
    self.action_space = int(action_space)
    self.no_support = bool(getattr(config, 'no_support', False))
    self.no_target_transform = bool(getattr(config, 'no_target_transform', False))
    vs = tuple(getattr(config, 'value_support', (-15, 15)))
    rs = tuple(getattr(config, 'reward_support', (-15, 15)))
    self.value_support_min, self.reward_support_min = vs[0], rs[0]
    v_out = 1 if self.no_support else vs[1] - vs[0] + 1
    r_out = 1 if self.no_support else rs[1] - rs[0] + 1
    self.representation_head = _TwoLayer(int(input_dim), HIDDEN, 'out')
    self.value_head = _TwoLayer(HIDDEN, v_out, 'value')
    self.policy_head = _TwoLayer(HIDDEN, self.action_space, 'policy')
    self.reward_head = _TwoLayer(HIDDEN + self.action_space, r_out, 'reward')
    self.transition_head = _TwoLayer(HIDDEN + self.action_space, HIDDEN, 'out')
    self.LN = nn.LayerNorm([HIDDEN], elementwise_affine=True)
    self.to(device)

  def representation(self, observation):
    return torch.relu(self.LN(self.representation_head(observation)))

  def prediction(self, hidden_state):
    value = self.value_head(hidden_state)
    if not self.training and not self.no_support:
      value = support_to_scalar(value, self.value_support_min, self.no_target_transform)
    return self.policy_head(hidden_state), value

  def dynamics(self, hidden_state, action):
    a = torch.as_tensor(action, dtype=torch.int64, device=hidden_state.device).reshape(-1)
    x = torch.cat((hidden_state, torch.nn.functional.one_hot(a, self.action_space).to(hidden_state.dtype)), dim=1)
    reward = self.reward_head(x)
    if not self.training and not self.no_support:
      reward = support_to_scalar(reward, self.reward_support_min, self.no_target_transform)
    return torch.relu(self.LN(self.transition_head(x))), reward

  def initial_inference(self, observation):
    hidden_state = self.representation(observation)
    policy_logits, value = self.prediction(hidden_state)
    return NetworkOutput(value, 0, policy_logits, hidden_state)

  def recurrent_inference(self, hidden_state, action):
    hidden_state, reward = self.dynamics(hidden_state, action)
    policy_logits, value = self.prediction(hidden_state)
    return NetworkOutput(value, reward, policy_logits, hidden_state)

  def load_weights(self, weights):
    self.load_state_dict(weights)

  def get_weights(self):
    return {k: v.cpu() for k, v in self.state_dict().items()}


# ---------------------------------------------------------------------------------------------------------------
# Conv networks (reference networks.py:393-718): residual tower MuZeroNetwork and TinyNetwork.

def _bn_affine(bn):
  """BatchNorm2d in inference mode is x * scale + shift per channel; the pair is cached on the module and recomputed when
  any of its four tensors has been written (load_state_dict / load_flat copy in place: the version counters move)."""
  key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.device)
  cache = getattr(bn, '_mz_affine', None)
  if cache is None or cache[0] != key:
    with torch.no_grad():
      scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
      shift = (bn.bias - bn.running_mean * scale).contiguous()
    cache = (key, scale, shift)
    bn._mz_affine = cache
  return cache[1], cache[2]


def _fused_ok(y):
  """may `y` go through the in-place HIP epilogue (mz_affine_relu)?  Only where autograd is off -- the kernel is invisible to
  it: eval() with gradients enabled (saliency, reanalyse with gradients) must take the PyTorch path -- and only for float32
  NCHW-contiguous tensors (a channels_last convolution output would be read with the wrong layout)."""
  return (not torch.is_grad_enabled() and y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and y.is_contiguous() and
          (y.shape[2] * y.shape[3]) % 4 == 0)


def _affine_relu_(y, scale, shift, residual=None):
  """y <- relu(y * scale[c] + shift[c] (+ residual)) in place, one HIP kernel (mz_affine_relu, include/mz_engine.h): the
  BatchNorm + skip + ReLU epilogue of a residual block.  PyTorch's own three elementwise kernels (addcmul, add, clamp) were
  15.7 % of the GPU time of `bench.py --workload breakout` (profiles/r03_breakout_kernel_stats.csv, before)."""
  from . import _abi
  import ctypes as C
  lib = _abi.load()
  _abi.check(lib.mz_affine_relu(C.c_void_p(y.data_ptr()), C.c_void_p(scale.data_ptr()), C.c_void_p(shift.data_ptr()),
                                None if residual is None else C.c_void_p(residual.data_ptr()), y.numel(), int(y.shape[1]),
                                int(y.shape[2] * y.shape[3]), C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)),
             'mz_affine_relu')
  return y


def _bn_infer(bn, x):
  """BatchNorm2d in inference mode as one fused multiply-add per element on the GPU: MIOpen's inference kernel
  (MIOpenBatchNormFwdInferSpatialEst) takes 144 us for a [512, 128, 6, 6] tensor on MI355X -- twice the 3x3 convolution
  in front of it -- against ~10 us for x * scale + shift.  Same arithmetic up to the rounding of scale / shift (~1e-7);
  CPU tensors and training mode go through the module itself."""
  if bn.training or not x.is_cuda:
    return bn(x)
  scale, shift = _bn_affine(bn)
  return torch.addcmul(shift.view(1, -1, 1, 1), x, scale.view(1, -1, 1, 1))


class _Block(nn.Module):
  """Two 3x3 convolutions around a skip connection; with BatchNorm it is the reference's ResidualBlock
  (networks.py:393-410), without it the TinyBlock (networks.py:557-567): relu(n2(conv2(relu(n1(conv1(x))))) + x).
  GPU inference with BatchNorm: each convolution's output gets its whole epilogue (BatchNorm, skip, ReLU) in one
  in-place pass (_affine_relu_)."""

  def __init__(self, channels, norm):
    super().__init__()
    self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, stride=1, padding=1, bias=False)
    if norm:
      self.bn1 = nn.BatchNorm2d(channels)
    self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, stride=1, padding=1, bias=False)
    if norm:
      self.bn2 = nn.BatchNorm2d(channels)
    self.norm = norm

  def forward(self, x):
    fused = self.norm and not self.training and _fused_ok(x)
    y = self.conv1(x)
    # (every tensor handed to the kernel is checked, the convolutions' outputs too)
    y = _affine_relu_(y, *_bn_affine(self.bn1)) if fused and _fused_ok(y) else torch.relu(_bn_infer(self.bn1, y) if self.norm else y)
    y = self.conv2(y)
    if fused and _fused_ok(y):
      return _affine_relu_(y, *_bn_affine(self.bn2), residual=x)
    y = _bn_infer(self.bn2, y) if self.norm else y
    return torch.relu(y + x)


def _tower(n, channels, norm=True):
  return nn.ModuleList([_Block(channels, norm) for _ in range(n)])


def _through(blocks, x):
  for b in blocks:
    x = b(x)
  return x


class _MuZeroRepresentation(nn.Module):
  """[B, C, 96, 96] -> [B, 128, 6, 6] (networks.py:413-446): stride-2 conv, 2 blocks @64, stride-2 conv, 3 blocks @128,
  avg-pool, 3 blocks, avg-pool, 16 blocks."""

  def __init__(self, input_channels):
    super().__init__()
    self.conv1 = nn.Conv2d(input_channels, 64, kernel_size=3, stride=2, padding=1)
    self.resblocks1 = _tower(2, 64)
    self.conv2 = nn.Conv2d(64, 128, kernel_size=3, stride=2, padding=1)
    self.resblocks2 = _tower(3, 128)
    self.avg_pool1 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
    self.resblocks3 = _tower(3, 128)
    self.avg_pool2 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
    self.resblocks = _tower(16, 128)

  def forward(self, x):
    x = _through(self.resblocks1, self.conv1(x))
    x = _through(self.resblocks2, self.conv2(x))
    x = _through(self.resblocks3, self.avg_pool1(x))
    return _through(self.resblocks, self.avg_pool2(x))


class _MuZeroDynamics(nn.Module):
  """[B, 128 + 1, 6, 6] -> (next state [B, 128, 6, 6], reward logits) (networks.py:449-471)."""

  def __init__(self, reward_out):
    super().__init__()
    self.conv = nn.Conv2d(128 + 1, 128, kernel_size=3, stride=1, padding=1)
    self.bn = nn.BatchNorm2d(128)
    self.resblocks = _tower(16, 128)
    self.fc1 = nn.Linear(6 * 6 * 128, 512)
    self.fc2 = nn.Linear(512, reward_out)

  def forward(self, x):
    return self._tail(self.conv(x))

  def forward_action(self, hidden_state, plane_value):
    """GPU inference: the 129th input channel is the action plane, constant over the 6x6 positions of a sample.  MIOpen
    has no fast kernel for 129 input channels (it falls back to a naive one: 3.4 ms per call on MI355X against 73 us for
    the 128-channel convolution), and a constant plane contributes its value times the layer's response to a plane of
    ones: conv(cat(hidden, plane)) = conv_128(hidden) + plane value * conv_1(ones) + bias.  plane_value: [B, 1, 1, 1]."""
    w = self.conv.weight
    # the plane of ones' response and the 128-channel slice of the kernel depend on the weights only: cached until the
    # weights are written (load_state_dict / load_flat copy in place: the version counter moves) -- the one-channel
    # convolution is a MIOpen naive kernel + transposes, ~6 launches per recurrent inference otherwise
    key = (w._version, w.device, tuple(hidden_state.shape[2:]), hidden_state.dtype)
    cache = getattr(self, '_mz_plane', None)
    if cache is None or cache[0] != key or torch.is_grad_enabled():
      ones = torch.ones((1, 1) + tuple(hidden_state.shape[2:]), dtype=hidden_state.dtype, device=hidden_state.device)
      response = nn.functional.conv2d(ones, w[:, 128:129].contiguous(), None, 1, 1)
      w128 = w[:, :128].contiguous()
      if not torch.is_grad_enabled():
        self._mz_plane = (key, response, w128)
    else:
      response, w128 = cache[1], cache[2]
    y = nn.functional.conv2d(hidden_state, w128, self.conv.bias, 1, 1)
    return self._tail(torch.addcmul(y, plane_value, response))

  def _tail(self, y):
    if not self.training and _fused_ok(y):
      state = _through(self.resblocks, _affine_relu_(y, *_bn_affine(self.bn)))
      return state, self.fc2(torch.relu(self.fc1(state.flatten(1))))
    state = _through(self.resblocks, torch.relu(_bn_infer(self.bn, y)))
    return state, self.fc2(torch.relu(self.fc1(state.flatten(1))))


class _MuZeroPrediction(nn.Module):
  """[B, 128, 6, 6] -> (policy logits, value logits) (networks.py:474-495)."""

  def __init__(self, action_space, value_out):
    super().__init__()
    self.resblocks = _tower(16, 128)
    self.fc_value = nn.Linear(6 * 6 * 128, 512)
    self.fc_value_o = nn.Linear(512, value_out)
    self.fc_policy = nn.Linear(6 * 6 * 128, 512)
    self.fc_policy_o = nn.Linear(512, action_space)

  def forward(self, x):
    y = _through(self.resblocks, x).flatten(1)
    return self.fc_policy_o(torch.relu(self.fc_policy(y))), self.fc_value_o(torch.relu(self.fc_value(y)))


class _ConvNetBase(nn.Module):
  """What MuZeroNetwork and TinyNetwork share (networks.py:498-555, 657-718): min-max scaling of hidden states over
  the channel dimension, the action as one extra plane of value action / action_space, eval-mode support -> scalar."""

  def _setup(self, action_space, device, config):
    self.device = device
    self.action_space = int(action_space)
    self.no_support = bool(getattr(config, 'no_support', False))
    self.no_target_transform = bool(getattr(config, 'no_target_transform', False))
    vs = tuple(getattr(config, 'value_support', (-15, 15)))
    rs = tuple(getattr(config, 'reward_support', (-15, 15)))
    self.value_support_min, self.reward_support_min = vs[0], rs[0]
    return (1 if self.no_support else vs[1] - vs[0] + 1), (1 if self.no_support else rs[1] - rs[0] + 1)

  @staticmethod
  def scale_state(state):
    lo = state.min(dim=1, keepdim=True)[0]
    hi = state.max(dim=1, keepdim=True)[0]
    return (state - lo) / (hi - lo)

  def attach_action(self, hidden_state, action):
    n, _, h, w = hidden_state.shape
    a = torch.as_tensor(action, device=hidden_state.device).reshape(n, 1, 1, 1).to(torch.float32)
    plane = a * torch.ones((n, 1, h, w), dtype=torch.float32, device=hidden_state.device) / self.action_space
    return torch.cat((hidden_state, plane), dim=1)

  def _scalar(self, logits, support_min):
    if self.training or self.no_support:
      return logits
    return support_to_scalar(logits, support_min, self.no_target_transform)

  def initial_inference(self, observation):
    hidden_state = self.representation(observation)
    policy_logits, value = self.prediction(hidden_state)
    return NetworkOutput(value, 0, policy_logits, hidden_state)

  def recurrent_inference(self, hidden_state, action):
    hidden_state, reward = self.dynamics(hidden_state, action)
    policy_logits, value = self.prediction(hidden_state)
    return NetworkOutput(value, reward, policy_logits, hidden_state)

  def load_weights(self, weights):
    self.load_state_dict(weights)

  def get_weights(self):
    return {k: v.cpu() for k, v in self.state_dict().items()}


class MuZeroNetwork(_ConvNetBase):
  """networks.py:498-555: 23.4 M parameters, hidden state [B, 128, 6, 6]."""

  def __init__(self, input_channels, action_space, device, config):
    super().__init__()
    v_out, r_out = self._setup(action_space, device, config)
    self.representation_head = _MuZeroRepresentation(int(input_channels))
    self.prediction_head = _MuZeroPrediction(self.action_space, v_out)
    self.dynamics_head = _MuZeroDynamics(r_out)
    self.to(device)

  def representation(self, observation):
    return self.scale_state(self.representation_head(observation))

  def prediction(self, hidden_state):
    policy, value = self.prediction_head(hidden_state)
    return policy, self._scalar(value, self.value_support_min)

  def dynamics(self, hidden_state, action):
    if self.training or not hidden_state.is_cuda:
      state, reward = self.dynamics_head(self.attach_action(hidden_state, action))
    else:
      n = hidden_state.shape[0]
      a = torch.as_tensor(action, device=hidden_state.device).reshape(n, 1, 1, 1).to(torch.float32)
      state, reward = self.dynamics_head.forward_action(hidden_state, a * 1.0 / self.action_space)
    return self.scale_state(state), self._scalar(reward, self.reward_support_min)


class _TinyRepresentation(nn.Module):
  """[B, C, 96, 96] -> [B, 64, 6, 6] (networks.py:570-592)."""

  def __init__(self, input_channels):
    super().__init__()
    self.conv1 = nn.Conv2d(input_channels, 32, kernel_size=3, stride=2, padding=1)
    self.max_pool1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    self.conv2 = nn.Conv2d(32, 64, kernel_size=3, stride=2, padding=1)
    self.max_pool2 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    self.block2 = _Block(64, norm=False)
    self.conv3 = nn.Conv2d(64, 64, kernel_size=3, stride=1, padding=1)

  def forward(self, x):
    x = self.max_pool1(torch.relu(self.conv1(x)))
    x = self.max_pool2(torch.relu(self.conv2(x)))
    return torch.tanh(self.conv3(self.block2(x)))


class _TinyHead(nn.Module):
  """block -> flatten -> Linear(512) -> ReLU -> Linear (networks.py:595-639); the two Linear layers carry the
  reference's attribute names."""

  def __init__(self, channels, n_out, fc_names):
    super().__init__()
    self.block1 = _Block(channels, norm=False)
    setattr(self, fc_names[0], nn.Linear(6 * 6 * channels, 512))
    setattr(self, fc_names[1], nn.Linear(512, n_out))
    self._fc = fc_names

  def forward(self, x):
    y = self.block1(x).flatten(1)
    return getattr(self, self._fc[1])(torch.relu(getattr(self, self._fc[0])(y)))


class _TinyHidden(nn.Module):
  """networks.py:642-654."""

  def __init__(self):
    super().__init__()
    self.block1 = _Block(65, norm=False)
    self.conv1 = nn.Conv2d(65, 64, kernel_size=3, stride=1, padding=1)

  def forward(self, x):
    return torch.tanh(self.conv1(self.block1(x)))


class TinyNetwork(_ConvNetBase):
  """networks.py:657-718: 4.06 M parameters, hidden state [B, 64, 6, 6]."""

  def __init__(self, input_channels, action_space, device, config):
    super().__init__()
    v_out, r_out = self._setup(action_space, device, config)
    self.representation_head = _TinyRepresentation(int(input_channels))
    self.value_head = _TinyHead(64, v_out, ('fc_value', 'fc_value_o'))
    self.reward_head = _TinyHead(64 + 1, r_out, ('fc1', 'fc2'))
    self.policy_head = _TinyHead(64, self.action_space, ('fc_policy', 'fc_policy_o'))
    self.transition_head = _TinyHidden()
    self.to(device)

  def representation(self, observation):
    return self.scale_state(self.representation_head(observation))

  def prediction(self, hidden_state):
    return self.policy_head(hidden_state), self._scalar(self.value_head(hidden_state), self.value_support_min)

  def dynamics(self, hidden_state, action):
    x = self.attach_action(hidden_state, action)
    reward = self._scalar(self.reward_head(x), self.reward_support_min)
    return self.scale_state(self.transition_head(x)), reward


def get_network(config, device=None):
  """utils.get_network (utils.py:21-37) without the environment probe: action_space / obs_space come from the config
  (train.py:66-68 injects them)."""
  if device is None:
    device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
  arch = getattr(config, 'architecture', 'FCNetwork')
  if arch == 'FCNetwork':
    import numpy as np
    return FCNetwork(int(np.prod(config.obs_space)), config.action_space, device, config)
  if arch in ('MuZeroNetwork', 'TinyNetwork'):
    # utils.py:27-35: input_channels = stack_obs, doubled with stack_actions; config.env_shapes() derives the image
    # environments' obs_space[0] from the same two flags, and a hand-built config may give obs_space directly
    channels = int(config.obs_space[0]) if len(tuple(config.obs_space)) == 3 else int(getattr(config, 'stack_obs', 1)) * (
        2 if getattr(config, 'stack_actions', False) else 1)
    return (MuZeroNetwork if arch == 'MuZeroNetwork' else TinyNetwork)(channels, config.action_space, device, config)
  raise NotImplementedError('%s (the reference\'s AttentionNetwork / HopfieldNetwork do not run at HEAD, SURVEY.md s2 row 11)' % arch)


# ---- one flat float32 buffer per network: what the weight broadcast ships (distributed.RankStorage)
def _float_items(sd):
  return [(k, v) for k, v in sd.items() if torch.is_floating_point(v)]     # (BatchNorm's int64 batch counter stays local)


def flat_size(network):
  return int(sum(v.numel() for _, v in _float_items(network.state_dict())))


def flatten_state(weights):
  """state_dict (or network) -> one float32 vector, state_dict order: parameters and float buffers (BatchNorm running
  statistics travel with the weights, as they do in the reference's pickled state_dict, networks.py:39-40)."""
  sd = weights.state_dict() if isinstance(weights, nn.Module) else weights
  return torch.cat([v.detach().reshape(-1).to(torch.float32).cpu() for _, v in _float_items(sd)]).contiguous()


def load_flat(network, flat):
  """inverse of flatten_state, device to device when `flat` lives where the network does (no host hop)"""
  off = 0
  with torch.no_grad():
    for _, v in _float_items(network.state_dict()):
      n = v.numel()
      v.copy_(flat[off:off + n].view_as(v))
      off += n
  if off != flat.numel():
    raise ValueError('flat weight buffer has %d floats, the network takes %d' % (flat.numel(), off))
