"""MI355X-native MuZero self-play / MCTS engine behind the reference's actors/mcts/replay surface.

Layout: csrc/ (HIP kernels + the C ABI of include/mz_engine.h), _abi.py (ctypes binding),
engine.py (batched device engine), and the host-side mirror of the reference interface
(mcts.py, actors.py, replay_buffer.py, shared_storage.py, learners.py, ...).
"""
__version__ = '0.1.0'
