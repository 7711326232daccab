"""ctypes binding of libmz_hip.so (include/mz_engine.h).  ABI-mode only: plain extern "C" symbols, so
cffi.dlopen() on the same header works too where cffi is installed (it is not in this image).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, loading/creating fails
loudly."""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
_SO = os.environ.get('MZ_HIP_LIB') or os.path.join(_CSRC, 'libmz_hip.so')      # (MZ_HIP_LIB: A/B runs of two builds on one box)
_SOURCES = ['mz_engine.hip', 'mz_comm.inc', 'mz_learner.hip.h', 'mz_fcl.hip.h', 'mz_fcl_abi.inc', 'mz_inst.hip', 'mz_kernels.inc', 'mz_common.h', 'mz_net.hip.h', 'mz_tree.hip.h', 'mz_rng.h',
            'mz_selfplay.hip.h', 'mz_selfplay_abi.inc', 'mz_fused.hip.h', 'mz_root.hip.h', 'mz_fused_h2.hip.h']
_ENGINE_ONLY = ('mz_engine.hip', 'mz_comm.inc', 'mz_learner.hip.h', 'mz_fcl.hip.h', 'mz_fcl_abi.inc', 'mz_selfplay_abi.inc')      # included by mz_engine.hip alone
_lib = None

HIPCC_FLAGS = ['-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-std=c++17', '-fPIC', '-Wno-unused-value',
               '-Wno-unused-result']
# translation units of libmz_hip.so: the host side + root / stepwise kernels, and one unit per shape of the two search
# kernels (mz_kernels.inc; the same list as launch_fused / launch_h2 in mz_engine.hip dispatch to)
FUSED_SHAPES = [(14, 1, 4), (14, 1, 8), (15, 1, 8), (15, 1, 16), (16, 1, 16), (18, 1, 16), (18, 2, 32), (21, 2, 32)]
H2_SHAPES = [4, 8, 16]
DEV_FUSED_SHAPES, DEV_H2_SHAPES = [(14, 1, 4), (14, 1, 8)], [4, 8]      # -DMZ_DEV_ONLY: the two bench shapes


def translation_units(extra=()):
  dev = '-DMZ_DEV_ONLY' in extra
  units = [('mz_engine', 'mz_engine.hip', [])]
  for ks1, jtp, g in (DEV_FUSED_SHAPES if dev else FUSED_SHAPES):
    units.append(('mz_inst_f_%d_%d_%d' % (ks1, jtp, g), 'mz_inst.hip', ['-DMZ_INST_F=%d,%d,%d' % (ks1, jtp, g)] +
                  (['-DMZ_INST_GAME=1'] if (ks1, jtp, g) == (15, 1, 16) else [])))
  for g in (DEV_H2_SHAPES if dev else H2_SHAPES):
    units.append(('mz_inst_h_%d' % g, 'mz_inst.hip', ['-DMZ_INST_H=%d' % g]))
  return units


class MzConfig(C.Structure):
  _fields_ = [('num_envs', C.c_int32), ('obs_dim', C.c_int32), ('action_space', C.c_int32),
              ('num_simulations', C.c_int32), ('two_players', C.c_int32), ('has_min_bound', C.c_int32),
              ('has_max_bound', C.c_int32), ('value_support_min', C.c_int32), ('value_support_max', C.c_int32),
              ('reward_support_min', C.c_int32), ('reward_support_max', C.c_int32),
              ('no_target_transform', C.c_int32), ('min_bound', C.c_double), ('max_bound', C.c_double),
              ('discount', C.c_double), ('pb_c_base', C.c_double), ('pb_c_init', C.c_double),
              ('init_value_score', C.c_double), ('root_dirichlet_alpha', C.c_double),
              ('root_exploration_fraction', C.c_double), ('seed', C.c_uint64), ('env_id_offset', C.c_int32),
              ('no_support', C.c_int32), ('split_f16', C.c_int32)]


class MzFclSource(C.Structure):
  """mz_fcl_source (include/mz_engine.h): the replay handle and the three entry points of libmz_replay.so mz_fcl_run calls"""
  _fields_ = [('replay', C.c_void_p), ('sample', C.c_void_p), ('refresh', C.c_void_p), ('last_error', C.c_void_p)]


def stale():
  if not os.path.exists(_SO):
    return True
  t = os.path.getmtime(_SO)
  inc = [os.path.join(_CSRC, '..', '..', 'include', h) for h in ('mz_engine.h', 'mz_engine_debug.h')]
  srcs = [os.path.join(_CSRC, s) for s in _SOURCES] + inc
  return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False, out=None, extra=None, jobs=None):
  """hipcc cross-compiles for gfx950 without a GPU; the .so stays in-tree (csrc/).  The translation units are compiled
  in parallel (objects under csrc/obj/, git-ignored), then linked.  out / extra: another library name and extra hipcc
  flags (kernel development: A/B builds under build/, -DMZ_DEV_ONLY)."""
  target = out or os.path.join(_CSRC, 'libmz_hip.so')
  if not (force or out or stale()):
    return _SO
  from concurrent.futures import ThreadPoolExecutor
  extra = list(extra) if extra is not None else os.environ.get('MZ_HIPCC_EXTRA', '').split()
  objdir = os.path.join(_CSRC, 'obj', os.path.splitext(os.path.basename(target))[0])
  os.makedirs(objdir, exist_ok=True)

  def compile_unit(unit):
    name, src, defs = unit
    obj = os.path.join(objdir, name + '.o')
    cmd = ['hipcc'] + HIPCC_FLAGS + extra + defs + ['-c', src, '-o', obj]
    # an object is kept when its command line is unchanged and none of the sources its unit includes is newer (the search
    # kernels' units do not include the learner / host-side files: a learner edit recompiles one unit, not thirteen)
    deps = [s for s in _SOURCES if src == 'mz_engine.hip' or s not in _ENGINE_ONLY]
    deps = deps + [os.path.join('..', '..', 'include', h) for h in (('mz_engine.h', 'mz_engine_debug.h') if src == 'mz_engine.hip' else ('mz_engine.h',))]
    stamp = obj + '.cmd'
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == ' '.join(cmd):
      t = os.path.getmtime(obj)
      if all(not os.path.exists(os.path.join(_CSRC, d)) or os.path.getmtime(os.path.join(_CSRC, d)) <= t for d in deps):
        return obj
    if verbose:
      print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=_CSRC)
    with open(stamp, 'w') as f:
      f.write(' '.join(cmd))
    return obj
  units = translation_units(extra)
  with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 1)) as pool:
    objs = list(pool.map(compile_unit, units))
  cmd = ['hipcc', '--offload-arch=gfx950', '-fPIC', '-shared'] + objs + ['-o', target]
  if verbose:
    print(' '.join(cmd), flush=True)
  subprocess.check_call(cmd, cwd=_CSRC)
  return target


_VP, _I, _D, _U64, _SZ = C.c_void_p, C.c_int, C.c_double, C.c_uint64, C.c_size_t

SIGNATURES = {
    'mz_last_error': (C.c_char_p, []),
    'mz_version': (_I, []),
    'mz_create': (_I, [C.POINTER(MzConfig), C.POINTER(_VP)]),
    'mz_destroy': (_I, [_VP]),
    'mz_num_weights': (_SZ, [_VP]),
    'mz_set_weights': (_I, [_VP, _VP, _SZ, _I, _VP]),
    'mz_set_weights_async': (_I, [_VP, _VP, _SZ, _I, _I, _VP]),
    'mz_weights_scale_ok': (_I, [_VP, _SZ, _I, _I, _I, _I]),
    'mz_comm_load': (_I, [C.c_char_p]),
    'mz_comm_unique_id': (_I, [_VP]),
    'mz_comm_create': (_I, [_I, _I, _VP, C.POINTER(_VP)]),
    'mz_comm_destroy': (_I, [_VP]),
    'mz_comm_count': (_I, [_VP, C.POINTER(_I)]),
    'mz_broadcast_weights': (_I, [_VP, _VP, _SZ, _I, _VP]),
    'mz_initial_inference': (_I, [_VP, _VP, _VP]),
    'mz_weight_scale': (_I, [_VP, _VP, _VP]),
    'mz_root_load': (_I, [_VP, _VP, _VP, _VP, _VP]),
    'mz_root_outputs': (_I, [_VP, _VP, _VP, _VP, _VP]),
    'mz_root_prepare': (_I, [_VP, _VP, _VP, _VP, _I, _U64, _VP]),
    'mz_root_set_priors': (_I, [_VP, _VP, _VP, _VP, _VP]),
    'mz_last_paths': (_I, [_VP, _VP, _VP, _VP]),
    'mz_search': (_I, [_VP, _I, _VP]),
    'mz_search_profiled': (_I, [_VP, _I, _VP, _VP]),
    'mz_search_timed': (_I, [_VP, _I, _VP, _VP]),
    'mz_search_phase_profile': (_I, [_VP, _I, _VP, _VP]),
    'mz_search_phase_spread': (_I, [_VP, _VP]),
    'mz_select': (_I, [_VP, _VP, _VP, _VP, _VP, _VP]),
    'mz_gather_hidden': (_I, [_VP, _VP, _VP]),
    'mz_tree_pair_timed': (_I, [_VP, _VP, _VP, _VP, _VP, _VP]),
    'mz_expand_backup': (_I, [_VP, _VP, _VP, _VP, _VP, _VP]),
    'mz_expand_backup_select': (_I, [_VP] * 10),
    'mz_recurrent_inference': (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    'mz_finalize': (_I, [_VP, _VP, _VP, _U64, _VP, _VP, _VP, _VP, _VP, _VP]),
    'mz_export_tree': (_I, [_VP] * 11),
    'mz_affine_relu': (_I, [_VP, _VP, _VP, _VP, _SZ, _I, _I, _VP]),
    'mz_nodes_per_tree': (_I, [_VP]),
    'mz_padded_envs': (_I, [_VP]),
    'mz_selfplay_reset': (_I, [_VP, _I, _D, _I, _VP]),
    'mz_selfplay_set_temperature': (_I, [_VP, _D, _VP]),
    'mz_selfplay_set_moves': (_I, [_VP, _U64]),
    'mz_selfplay_set_obs': (_I, [_VP, _I, _VP, _VP]),
    'mz_selfplay_set_env': (_I, [_VP, _I]),
    'mz_selfplay_set_draws': (_I, [_VP, _VP, _VP, _VP]),
    'mz_selfplay_export_trees': (_I, [_VP, _I]),
    'mz_selfplay_noise_log': (_I, [_VP, _I]),
    'mz_selfplay_read_noise': (_I, [_VP, _U64, _VP]),
    'mz_sim_io': (_I, [_VP, _I, _VP, _I]),
    'mz_search_kernel_info': (_I, [_VP, _VP]),
    'mz_learner_targets': (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP]),
    'mz_soft_ce_forward': (_I, [_VP, _VP, _I, _I, _I, C.c_int64, C.c_int64, _VP, _VP]),
    'mz_soft_ce_backward': (_I, [_VP, _VP, _VP, _I, _I, _I, C.c_int64, C.c_int64, _VP, _VP]),
    'mz_fcl_create': (_I, [_I] * 9 + [C.POINTER(_VP)]),
    'mz_fcl_destroy': (_I, [_VP]),
    'mz_fcl_num_params': (_SZ, [_VP]),
    'mz_fcl_bind': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _VP, _VP]),
    'mz_fcl_repack': (_I, [_VP, _VP]),
    'mz_fcl_step': (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _I, _D, _D, _D, _D, _D, _I, _I, _VP, _VP, _VP]),
    'mz_fcl_update': (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _I, _D, _D, _D, _D, _D, _I, _VP, _VP, C.POINTER(_I)]),
    'mz_fcl_errors': (_I, [_VP, _I, _VP]),
    'mz_fcl_run_stats': (_I, [_VP, _VP, _I]),
    'mz_fcl_slots': (_I, [_VP]),
    'mz_fcl_run': (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _D, _D, _D, _D, _D, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    'mz_fcl_read_grad': (_I, [_VP, _VP, _SZ]),
    'mz_fcl_read_tape': (C.c_longlong, [_VP, _I, _VP, _SZ]),
    'mz_fcl_heads_profile': (_I, [_VP, _I, _VP]),
    'mz_selfplay_steps': (_I, [_VP, _I, _VP]),
    'mz_selfplay_steps_into': (_I, [_VP, _I, _VP, _VP]),
    'mz_selfplay_steps_timed': (_I, [_VP, _I, _VP, _VP]),
    'mz_selfplay_phase_profile': (_I, [_VP, _I, _VP, _VP]),
    'mz_selfplay_moves_per_launch': (_I, [_VP]),
    'mz_selfplay_rec_floats': (_I, [_VP]),
    'mz_selfplay_ring_moves': (_I, [_VP]),
    'mz_selfplay_drain': (_I, [_VP, _VP, _I, C.POINTER(_I), _VP]),
    'mz_synth_obs': (_I, [_VP, _I, _I, _I, _VP, _VP]),
}


def load():
  global _lib
  if _lib is not None:
    return _lib
  # torch first: it brings its own copy of the HIP runtime, and a process must end up with ONE -- loaded the other way
  # round (this library, then torch), the library binds the system's runtime and sees no device once torch has
  # initialised its own
  import torch  # noqa: F401
  if not os.path.exists(_SO):
    raise RuntimeError('libmz_hip.so is not built (%s). Build it with `python -c "import __graft_entry__ as g; '
                       'g.build()"`; this engine has no CPU fallback.' % _SO)
  lib = C.CDLL(_SO)
  for name, (res, args) in SIGNATURES.items():
    try:
      fn = getattr(lib, name)    # AttributeError if the library does not export a declared symbol
    except AttributeError:
      if os.environ.get('MZ_HIP_LIB') and os.environ.get('MZ_HIP_LIB_OLD', '0')[:1] == '1':
        continue                 # (kernel development: an A/B build of an OLDER tree lacks entry points added since)
      raise
    fn.restype, fn.argtypes = res, args
  _lib = lib
  return lib


# ---------------------------------------------------------------- host replay library (include/mz_replay.h)
_RSO = os.environ.get('MZ_REPLAY_LIB') or os.path.join(_CSRC, 'libmz_replay.so')      # (MZ_REPLAY_LIB: a sanitizer build, tests/test_sanitizers.py)
_rlib = None
_I64 = C.c_int64


class MzrConfig(C.Structure):
  _fields_ = [('window_size', _I64), ('window_step', _I64), ('obs_dim', C.c_int32), ('action_space', C.c_int32),
              ('num_unroll_steps', C.c_int32), ('td_steps', C.c_int32), ('max_history_length', C.c_int32),
              ('batch_size', C.c_int32), ('epsilon', _D), ('alpha', _D), ('beta', _D),
              ('beta_increment_per_sampling', _D), ('discount', _D), ('seed', _U64), ('two_players', C.c_int32),
              ('episode_life', C.c_int32), ('ingest_threads', C.c_int32), ('obs_u8', C.c_int32)]


REPLAY_SIGNATURES = {
    'mzr_last_error': (C.c_char_p, []),
    'mzr_create': (_I, [C.POINTER(MzrConfig), C.POINTER(_VP)]),
    'mzr_destroy': (_I, [_VP]),
    'mzr_priorities': (_I, [_VP, _VP, _I64, _VP]),
    'mzr_tree_add': (_I, [_VP, _VP, _I64, _VP]),
    'mzr_tree_update': (_I, [_VP, _VP, _VP, _I64]),
    'mzr_tree_get_leaf': (_I64, [_VP, _D]),
    'mzr_leaf_info': (_I, [_VP, _I64, _VP, _VP, _VP, _VP]),
    'mzr_leaf_history': (_I, [_VP, _I64, _VP, _I64]),
    'mzr_total_priority': (_D, [_VP]),
    'mzr_size': (_I64, [_VP]),
    'mzr_tree_leaves': (_I, [_VP, _I64, _VP]),
    'mzr_save_history': (_I, [_VP, _I64, _VP, _I64, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'mzr_ingest_records': (_I, [_VP, _VP, _I, _I, _I]),
    'mzr_ingest_records_from': (_I, [_VP, _VP, _I, _I, _I, _I]),
    'mzr_ingest_records_packed': (_I, [_VP, _VP, _I, _I, _I, _I]),
    'mzr_pack_env_major': (None, [_VP, _VP, _I, _I, _I]),
    'mzr_asm_create': (_I, [C.POINTER(MzrConfig), _I, C.POINTER(_VP)]),
    'mzr_asm_destroy': (_I, [_VP]),
    'mzr_asm_feed': (_I, [_VP, _VP, _I, _I, _I]),
    'mzr_asm_pending': (_I64, [_VP]),
    'mzr_asm_take': (_I64, [_VP, _VP, _I64]),
    'mzr_ingest_slices': (_I, [_VP, _VP, _I64, _I]),
    'mzr_sample_batch': (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'mzr_sample_batch_words': (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'mzr_sample_batches_words': (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'mzr_set_ingest_threads': (_I, [_VP, _I]),
    'mzr_ingest_threads': (_I, [_VP]),
    'mzr_frames': (_I64, [_VP]),
    'mzr_games': (_I64, [_VP]),
    'mzr_add_initial_throughput': (_I, [_VP, _I64, _I64]),
    'mzr_priorities_f32': (_I, [_VP, _VP, _I64, _VP]),
    'mzr_update_errors_f32': (_I, [_VP, _VP, _VP, _I64]),
    'mzr_sample_batches_full': (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'mzr_store_release_i64': (None, [_VP, _I64]),
    'mzr_load_acquire_i64': (_I64, [_VP]),
}


REPLAY_FLAGS = ['-mavx2', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-pthread']


def build_replay(force=False, verbose=False, out=None, extra=None):
  """out / extra: another library file and other optimisation / instrumentation flags than -O3 (the sanitizer builds of
  tests/test_sanitizers.py)"""
  src = os.path.join(_CSRC, 'mz_replay.cpp')
  hdr = os.path.join(_CSRC, '..', '..', 'include', 'mz_replay.h')
  target = out or os.path.join(_CSRC, 'libmz_replay.so')
  if force or out or not os.path.exists(target) or os.path.getmtime(target) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
    # -mavx2: the sum tree's per-level running sums are 4-wide double vectors (every x86-64 host of an MI355X box has
    # AVX2; no FMA is enabled and contraction stays off: the sums are the reference's, term by term)
    cmd = ['g++'] + (list(extra) if extra is not None else ['-O3']) + REPLAY_FLAGS + ['mz_replay.cpp', '-o', target]
    if verbose:
      print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
  return target


def load_replay():
  global _rlib
  if _rlib is not None:
    return _rlib
  if not os.path.exists(_RSO):
    raise RuntimeError('libmz_replay.so is not built (%s); run __graft_entry__.build()' % _RSO)
  lib = C.CDLL(_RSO)
  for name, (res, args) in REPLAY_SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
  _rlib = lib
  return lib


def check_replay(rc, what=''):
  if rc != 0:
    msg = load_replay().mzr_last_error()
    raise RuntimeError('%s failed: %s' % (what or 'mzr call', msg.decode() if msg else 'unknown error'))


def check(rc, what=''):
  if rc != 0:
    msg = load().mz_last_error()
    raise RuntimeError('%s failed: %s' % (what or 'mz call', msg.decode() if msg else 'unknown error'))
