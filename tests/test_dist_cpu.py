"""N > 1 host logic on CPU (gloo, world_size 2): the single exchange of the path -- the flattened weight
broadcast from the learner/storage rank (shared_storage.broadcast_flat; RCCL on the GPU box) -- and the
rank-wise reductions bench.py reports (max time, summed frames); plus the shard map env_id_offset."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from model_based_rl_amd.shared_storage import broadcast_flat
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
flat = torch.arange(198410, dtype=torch.float32) * 0.5 if rank == 0 else torch.zeros(198410)
broadcast_flat(flat, src=0)
assert torch.equal(flat, torch.arange(198410, dtype=torch.float32) * 0.5)
t = torch.tensor([1.0 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
f = torch.tensor([100.0 * (rank + 1)], dtype=torch.float64); dist.all_reduce(f, op=dist.ReduceOp.SUM)
assert t.item() == world and f.item() == 100.0 * world * (world + 1) / 2
# shard map: rank r owns global env ids [r*B, (r+1)*B)
B = 16
ids = torch.arange(rank * B, (rank + 1) * B)
allids = [torch.zeros(B, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allids, ids)
assert torch.equal(torch.cat(allids), torch.arange(world * B))
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
''' % ROOT


def test_weight_broadcast_and_reductions_world2(tmp_path):
  script = tmp_path / 'w.py'
  script.write_text(WORKER)
  env = dict(os.environ, MASTER_ADDR='127.0.0.1')
  out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)], env=env,
                       capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  assert out.stdout.count('ok') == 2
