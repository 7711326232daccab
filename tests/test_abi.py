"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950 without a GPU, loads,
and exports every symbol include/mz_engine.h declares; the ctypes struct matches the C struct size;
creating an engine without a GPU fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=('mz_engine.h', 'mz_engine_debug.h')):
  """the boundary (mz_engine.h) and the instrumentation / test hooks kept apart from it (mz_engine_debug.h)"""
  out = set()
  for h in headers:
    text = open(os.path.join(ROOT, 'include', h)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out |= set(re.findall(r'\b(mz_[a-z_0-9]+)\s*\(', text))
  return sorted(out)


def test_header_symbols_exported():
  from model_based_rl_amd import _abi
  _abi.build()
  lib = _abi.load()
  syms = declared_symbols()
  assert len(syms) >= 20
  for s in syms:
    assert hasattr(lib, s), 'libmz_hip.so does not export %s' % s
  assert set(_abi.SIGNATURES) == set(syms), set(_abi.SIGNATURES) ^ set(syms)
  assert lib.mz_version() == 1
  # the boundary header carries no profiling / test instrumentation (VERDICT r05 hygiene): those live in mz_engine_debug.h
  boundary, debug = set(declared_symbols(('mz_engine.h',))), set(declared_symbols(('mz_engine_debug.h',)))
  assert not (boundary & debug) and len(debug) >= 15
  assert not [s for s in boundary if re.search(r'profile|timed|_stats|read_tape|read_grad|sim_io|noise_log|phase_spread', s)], boundary


def test_replay_header_symbols_exported():
  """include/mz_replay.h (the host replay's C ABI) against libmz_replay.so and the ctypes table"""
  from model_based_rl_amd import _abi
  _abi.build_replay()
  lib = _abi.load_replay()
  text = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'mz_replay.h')).read(), flags=re.S)
  syms = sorted(set(re.findall(r'\b(mzr_[a-z_0-9]+)\s*\(', text)))
  assert len(syms) >= 15
  for s in syms:
    assert hasattr(lib, s), 'libmz_replay.so does not export %s' % s
  assert set(_abi.REPLAY_SIGNATURES) == set(syms), set(_abi.REPLAY_SIGNATURES) ^ set(syms)
  subprocess.check_call(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', os.path.join(ROOT, 'include', 'mz_replay.h')])
  src = '#include "mz_replay.h"\n#include <stdio.h>\nint main(){printf("%zu", sizeof(mzr_config));return 0;}\n'
  exe = '/tmp/mzr_sizeof_test'
  subprocess.run(['gcc', '-x', 'c', '-', '-I', os.path.join(ROOT, 'include'), '-o', exe], input=src.encode(), check=True)
  assert int(subprocess.check_output([exe])) == C.sizeof(_abi.MzrConfig)


def test_config_struct_size_matches_c():
  from model_based_rl_amd import _abi
  src = '#include "mz_engine.h"\n#include <stdio.h>\nint main(){printf("%zu", sizeof(mz_config));return 0;}\n'
  exe = '/tmp/mz_sizeof_test'
  subprocess.run(['gcc', '-x', 'c', '-', '-I', os.path.join(ROOT, 'include'), '-o', exe], input=src.encode(), check=True)
  assert int(subprocess.check_output([exe])) == C.sizeof(_abi.MzConfig)


def test_header_is_plain_c():
  subprocess.check_call(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', os.path.join(ROOT, 'include', 'mz_engine.h')])
  subprocess.check_call(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', os.path.join(ROOT, 'include', 'mz_engine_debug.h')])


def test_no_cpu_fallback():
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from model_based_rl_amd.engine import Engine
  with pytest.raises(RuntimeError, match='no CPU path'):
    Engine(16, 8, 4, 30)
  # and the C ABI itself refuses too
  from model_based_rl_amd import _abi
  lib = _abi.load()
  cfg = _abi.MzConfig(16, 8, 4, 30, 0, 0, 0, -15, 15, -15, 15, 0, 0.0, 0.0, 0.997, 19652.0, 1.25, 0.0, 0.25, 0.25, 0, 0, 0)
  h = C.c_void_p()
  assert lib.mz_create(C.byref(cfg), C.byref(h)) != 0
  assert b'no HIP device' in lib.mz_last_error()


def test_product_does_not_touch_oracle():
  """the product package must never import/link the oracle (parity would be void)."""
  pkg = os.path.join(ROOT, 'model-based-rl_amd')
  for d, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.h', '.inc')):
        text = open(os.path.join(d, f)).read()
        assert 'oracle' not in text.replace('no oracle', ''), os.path.join(d, f)


def cdef_text(header):
  """include/<header> as cffi.cdef() takes it: comments and preprocessor lines out, integer #defines kept"""
  text = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', header)).read(), flags=re.S)
  keep = []
  for line in text.splitlines():
    if line.startswith('#'):
      if re.match(r'#define\s+\w+\s+\d+\s*$', line):
        keep.append(line)
      continue
    if line.strip() in ('extern "C" {', '}'):
      continue
    keep.append(line)
  return '\n'.join(keep)


def test_headers_reduce_to_plain_declarations():
  """what the cffi binding would be given: after stripping, only typedefs, structs, integer #defines and prototypes remain"""
  for h, n in (('mz_engine.h', 40), ('mz_replay.h', 15)):
    t = cdef_text(h)
    assert '#include' not in t and '__cplusplus' not in t
    assert len(re.findall(r'\bmzr?_[a-z_0-9]+\s*\(', t)) >= n
    src = '#include <stdint.h>\n#include <stddef.h>\n' + t + '\nint main(void){return 0;}\n'
    subprocess.run(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', '-'], input=src.encode(), check=True)


def test_cffi_dlopen_fast_path():
  """north_star names cffi; it is not installed in this image (ctypes binds the same symbols), so this runs wherever cffi exists"""
  cffi = pytest.importorskip('cffi')
  from model_based_rl_amd import _abi
  _abi.build(); _abi.build_replay()
  ffi = cffi.FFI()
  ffi.cdef(cdef_text('mz_replay.h'))
  lib = ffi.dlopen(os.path.join(ROOT, 'model-based-rl_amd', 'csrc', 'libmz_replay.so'))
  cfg = ffi.new('mzr_config *', dict(window_size=64, window_step=64, obs_dim=3, action_space=2, num_unroll_steps=5, td_steps=10,
                                     max_history_length=500, batch_size=4, epsilon=0.01, alpha=1.0, beta=1.0,
                                     beta_increment_per_sampling=0.001, discount=0.997, seed=0))
  h = ffi.new('mz_replay **')
  assert lib.mzr_create(cfg, h) == 0
  pri = ffi.new('double[]', [1.0, 2.0, 3.0])
  assert lib.mzr_tree_add(h[0], pri, 3, ffi.NULL) == 0
  assert lib.mzr_total_priority(h[0]) == 6.0 and lib.mzr_tree_get_leaf(h[0], 2.5) == 64
  assert lib.mzr_destroy(h[0]) == 0
  ffi2 = cffi.FFI()
  ffi2.cdef(cdef_text('mz_engine.h'))
  import torch  # noqa: F401  (one HIP runtime per process: torch's first, model_based_rl_amd/_abi.py)
  eng = ffi2.dlopen(os.path.join(ROOT, 'model-based-rl_amd', 'csrc', 'libmz_hip.so'))
  assert eng.mz_version() == 1


def test_weights_scale_ok_is_host_arithmetic():
  """mz_weights_scale_ok (the host's side of the clamp-ReLU scale decision, mz_set_weights_async): no GPU involved; 1 for
  PyTorch-initialised weights, 0 for a weight set with an absurd or non-finite entry in the layers the bound covers, an
  error for a vector of the wrong length; and equal to a numpy restatement of the bound (k_relu_bound / k_relu_scale)."""
  import types
  import numpy as np
  import torch
  from model_based_rl_amd.engine import flatten_weights, weights_scale_ok
  from model_based_rl_amd.networks import FCNetwork
  torch.manual_seed(0)
  O, A, S = 8, 4, 31
  net = FCNetwork(O, A, torch.device('cpu'), types.SimpleNamespace())
  sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
  flat = flatten_weights(sd)
  assert weights_scale_ok(flat, O, A, S, S) == 1
  with pytest.raises(RuntimeError):
    weights_scale_ok(flat[:-1], O, A, S, S)
  for key, val in (('reward_head.fc1.weight', 1e30), ('transition_head.fc1.bias', float('nan')), ('LN.weight', float('inf')),
                   ('policy_head.policy.weight', 3e38), ('value_head.fc1.weight', float('nan'))):
    bad = {k: v.clone() for k, v in sd.items()}
    bad[key].view(-1)[3] = val
    assert weights_scale_ok(flatten_weights(bad), O, A, S, S) == 0, key
  # weights outside the bound's layers do not matter (the representation runs in the root kernel, unscaled)
  free = {k: v.clone() for k, v in sd.items()}
  free['representation_head.fc1.weight'].view(-1)[0] = 1e30
  assert weights_scale_ok(flatten_weights(free), O, A, S, S) == 1
  # a large-but-finite gain: the decision follows the bound
  for gain, want in ((1e3, 1), (1e9, 1), (1e13, 0)):
    g = {k: v.clone() for k, v in sd.items()}
    g['value_head.fc1.weight'] *= gain
    hb = 7.01 * g['LN.weight'].abs().double() + g['LN.bias'].abs().double()
    bound = 0.0
    for head, onehot in (('reward_head', A), ('transition_head', A), ('value_head', 0), ('policy_head', 0)):
      w, b = g[head + '.fc1.weight'].abs().double(), g[head + '.fc1.bias'].abs().double()
      s = (w[:, :50] * hb).sum(1) + (w[:, 50:].max(1).values if onehot else 0.0) + b
      bound = max(bound, float(s.max()))
    assert (bound * 1.02 < 2.0 ** 39) == bool(want), (gain, bound)
    assert weights_scale_ok(flatten_weights(g), O, A, S, S) == want, gain


def test_headline_kernel_keeps_its_register_budget():
  """k_search_fused<14,1,4,1,HEAD> -- the launch the headline is measured on -- is tuned to the last register: 256 + 256
  registers and 48 bytes of scratch per lane.  Two extra kernel-argument fields once cost it 16 more bytes and 1 % of the
  benchmark without any test noticing (profiles/r05_host_threads.txt); this compiles its translation unit with the
  compiler's resource remarks (gfx950 cross-compile, no GPU) and holds the line."""
  import re
  import shutil
  import subprocess
  import tempfile
  from model_based_rl_amd import _abi
  if shutil.which('hipcc') is None:
    pytest.skip('hipcc not on PATH')
  src = os.path.join(os.path.dirname(_abi.__file__), 'csrc')
  name, source, defs = [u for u in _abi.translation_units() if u[0] == 'mz_inst_f_14_1_4'][0]
  with tempfile.TemporaryDirectory() as tmp:
    r = subprocess.run(['hipcc'] + list(_abi.HIPCC_FLAGS) + defs + ['-Rpass-analysis=kernel-resource-usage', '-c', source, '-o',
                        os.path.join(tmp, 'unit.o')], cwd=src, capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-2000:]
  usage, cur = {}, None
  for line in r.stderr.splitlines():
    m = re.search(r'remark: +Function Name: (\S+)', line)
    if m:
      cur = usage.setdefault(m.group(1), {})
    m = re.search(r'remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)', line)
    if m and cur is not None:
      cur[m.group(1).split(' [')[0]] = int(m.group(2))
  # template arguments <KS1 14, JTP 1, G 4, LT 1 (trees in LDS), PROF false, SP true, HEAD true, GAME false>
  head = [v for k, v in usage.items() if k.startswith('_Z14k_search_fusedILi14ELi1ELi4ELi1ELb0ELb1ELb1ELb0EE')]
  assert len(head) == 1, sorted(usage)
  assert head[0]['VGPRs'] == 256 and head[0]['AGPRs'] == 256
  assert head[0]['ScratchSize'] <= 48, head[0]
