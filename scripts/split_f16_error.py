#!/usr/bin/env python3
"""How far the split-f16 evaluation of FCNetwork (csrc/mz_fused_h2.hip.h: x = xh + xl in float16, W X ~= Wh Xh + Wh Xl +
Wl Xh with float32 accumulation) sits from a float64 evaluation, next to the exact-float32 evaluation -- numpy emulation,
no GPU (the products of float16 values are exact in float32; the MFMA's internal accumulation order is not modelled).

  python scripts/split_f16_error.py        # prints max |error| of one recurrent inference over 4096 rows
"""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_based_rl_amd.networks import FCNetwork

O, A, B = 8, 4, 4096
torch.manual_seed(0)
sd = {k: v.numpy() for k, v in FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace()).state_dict().items()}


def rtz16(a):       # v_cvt_pkrtz_f16_f32 on values in the float16 normal range: keep 11 significant bits
  return (np.ascontiguousarray(a, np.float32).view(np.uint32) & np.uint32(0xFFFFE000)).view(np.float32)


def rn16(a):
  return a.astype(np.float16).astype(np.float32)


def split(a, f):
  h = f(np.ascontiguousarray(a, np.float32))
  return h, f(np.ascontiguousarray(a - h, np.float32))


def lin_split(x, w, b):       # activations truncated (device), weights rounded to nearest (pack kernel)
  xh, xl = split(x, rtz16); wh, wl = split(w, rn16)
  return (xh @ wh.T + xh @ wl.T + xl @ wh.T).astype(np.float32) + b


def lin_f32(x, w, b):
  return (x @ w.T + b).astype(np.float32)


def lin_f64(x, w, b):
  return x.astype(np.float64) @ w.T.astype(np.float64) + b


def ln(x, g, bb):
  m = x.mean(1, keepdims=True); v = ((x - m) ** 2).mean(1, keepdims=True)
  return (x - m) / np.sqrt(v + 1e-5) * g + bb


def forward(lin, hidden, act, dt):
  x = np.concatenate([hidden.astype(dt), np.eye(A, dtype=dt)[act]], 1)
  two = lambda head, out, inp: lin(np.maximum(lin(inp, sd[head + '.fc1.weight'], sd[head + '.fc1.bias']), 0).astype(dt),
                                   sd['%s.%s.weight' % (head, out)], sd['%s.%s.bias' % (head, out)])
  r = two('reward_head', 'reward', x)
  h = np.maximum(ln(two('transition_head', 'out', x).astype(dt), sd['LN.weight'], sd['LN.bias']), 0).astype(dt)
  return h, r, two('value_head', 'value', h), two('policy_head', 'policy', h)


rng = np.random.RandomState(0)
hid = np.maximum(rng.standard_normal((B, 50)), 0).astype(np.float32)
act = rng.randint(0, A, B)
ref = forward(lin_f64, hid, act, np.float64)
f32 = forward(lin_f32, hid, act, np.float32)
spl = forward(lin_split, hid, act, np.float32)
for name, i in (('next hidden state', 0), ('reward logits', 1), ('value logits', 2), ('policy logits', 3)):
  print('%-18s max |error| vs float64:  exact float32 %.2e   split float16 %.2e' %
        (name, np.abs(f32[i] - ref[i]).max(), np.abs(spl[i] - ref[i]).max()))
