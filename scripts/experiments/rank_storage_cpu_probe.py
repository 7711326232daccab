"""Where do the host cores go in the native RCCL weight path (distributed.RankStorage at world size 1)?  CPU seconds of the whole
process per wall second while (a) idle after torch's process group is up, (b) idle after the gloo metadata group exists, (c) idle
after mz_comm_create, (d) calling get_weights in a loop.  usage: python -m torch.distributed.run --nproc-per-node 1 <this>"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import model_based_rl_amd
from model_based_rl_amd import distributed as D
from model_based_rl_amd.engine import flatten_weights, config_scale_check
from model_based_rl_amd.networks import FCNetwork
from model_based_rl_amd.shared_storage import SharedStorage
from model_based_rl_amd.actors import _call

def busy(seconds, fn=None):
  t0, c0 = time.perf_counter(), time.process_time()
  n = 0
  while time.perf_counter() - t0 < seconds:
    if fn is None: time.sleep(0.01)
    else: fn(); n += 1
  return (time.process_time() - c0) / (time.perf_counter() - t0), n

rank, world, device, backend = D.init_process_group()
print('idle, torch nccl group up:', busy(1.0))
g = dist.new_group(backend='gloo')
print('idle, + gloo group:', busy(1.0))
torch.manual_seed(0)
net = FCNetwork(8, 4, torch.device('cpu'), types.SimpleNamespace())
cfg = types.SimpleNamespace(num_actors=1, architecture='FCNetwork', obs_space=(8,), action_space=4, value_support=(-15, 15), reward_support=(-15, 15))
storage = SharedStorage(cfg)
storage.store_weights(net.get_weights(), 1)
n_flat = int(flatten_weights(net.get_weights()).numel())
rs = D.RankStorage(rank, world, device, n_flat, storage=storage, storage_call=_call, backend=backend, flatten=flatten_weights,
                   scale_check=config_scale_check(cfg))
print('idle, + mz_comm + RankStorage:', busy(1.0))
print('get_weights loop:', busy(2.0, lambda: rs.get_weights(0, 0)))
torch.cuda.synchronize()
print('idle after the loop:', busy(1.0))
os.environ['MZ_TORCH_COLLECTIVES'] = '1'
rs2 = D.RankStorage(rank, world, device, n_flat, storage=storage, storage_call=_call, backend=backend, flatten=flatten_weights)
print('torch collectives get_weights loop:', busy(2.0, lambda: rs2.get_weights(0, 0)))
rs.close()
dist.destroy_process_group()
