"""MuZeroNetwork recurrent_inference throughput on the MI355X under PyTorch-ROCm / MIOpen settings (exploration for the
--workload breakout secondary line): default, MIOpen find mode (cudnn.benchmark), channels_last."""
import sys, time, types, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_based_rl_amd.networks import MuZeroNetwork
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for name, bench, cl in (('default', False, False), ('benchmark', True, False), ('benchmark+channels_last', True, True)):
  torch.backends.cudnn.benchmark = bench
  torch.manual_seed(0)
  net = MuZeroNetwork(4, 4, torch.device('cuda'), types.SimpleNamespace()).eval()
  h = torch.rand(B, 128, 6, 6, device='cuda')
  if cl:
    net = net.to(memory_format=torch.channels_last); h = h.contiguous(memory_format=torch.channels_last)
  a = torch.randint(0, 4, (B,), device='cuda', dtype=torch.int32)
  with torch.inference_mode():
    for _ in range(3):
      out = net.recurrent_inference(h, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
      out = net.recurrent_inference(h, a)
    torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 20
  print('%-26s %.2f ms per recurrent_inference of %d rows -> %.1f TFLOP/s' % (name, dt * 1e3, B, 0.7044 * B / dt / 1e3))
