"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950 without a GPU, loads,
and exports every symbol include/mz_engine.h declares; the ctypes struct matches the C struct size;
creating an engine without a GPU fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'mz_engine.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(mz_[a-z_0-9]+)\s*\(', text)))


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


def test_config_struct_size_matches_c():
  from model_based_rl_amd import _abi
  src = '#include "mz_engine.h"\n#include <stdio.h>\nint main(){printf("%zu", sizeof(mz_config));return 0;}\n'
  exe = '/tmp/mz_sizeof_test'
  subprocess.run(['gcc', '-x', 'c', '-', '-I', os.path.join(ROOT, 'include'), '-o', exe], input=src.encode(), check=True)
  assert int(subprocess.check_output([exe])) == C.sizeof(_abi.MzConfig)


def test_header_is_plain_c():
  subprocess.check_call(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', os.path.join(ROOT, 'include', 'mz_engine.h')])


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
