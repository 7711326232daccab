"""Which kernels does MIOpen pick for the MuZeroNetwork's representation conv1 ([B, 4, 96, 96] -> [B, 64, 48, 48], 3x3
stride 2; reference networks.py:413-446) and for the dynamics' one-channel action-plane response, and what do they cost?
usage: conv1_probe.py nchw|pad8|channels_last|response [B]     (run it under rocprofv3 --kernel-trace --stats for the names)"""
import sys, time, torch
mode, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4096
torch.manual_seed(0)
dev = torch.device('cuda')
ev = lambda: torch.cuda.Event(enable_timing=True)
with torch.inference_mode():
  if mode == 'response':
    w = torch.randn(128, 129, 3, 3, device=dev)
    ones = torch.ones(1, 1, 6, 6, device=dev)
    f = lambda: torch.nn.functional.conv2d(ones, w[:, 128:129].contiguous(), None, 1, 1)
  else:
    conv = torch.nn.Conv2d(4, 64, 3, 2, 1).to(dev)
    x = torch.rand(B, 4, 96, 96, device=dev)
    if mode == 'pad8':
      w8 = torch.zeros(64, 8, 3, 3, device=dev); w8[:, :4] = conv.weight
      x8 = torch.zeros(B, 8, 96, 96, device=dev); x8[:, :4] = x
      f = lambda: torch.nn.functional.conv2d(x8, w8, conv.bias, 2, 1)
    elif mode == 'channels_last':
      conv = conv.to(memory_format=torch.channels_last); xc = x.contiguous(memory_format=torch.channels_last)
      f = lambda: conv(xc)
    else:
      f = lambda: conv(x)
  ref = torch.nn.functional.conv2d(x, conv.weight, conv.bias, 2, 1) if mode != 'response' else None
  for _ in range(3):
    y = f()
  torch.cuda.synchronize()
  a, b = ev(), ev()
  a.record()
  for _ in range(10):
    y = f()
  b.record(); torch.cuda.synchronize()
  print(mode, 'B', B, '%.3f ms per call' % (a.elapsed_time(b) / 10), 'max |diff| vs nchw %.2e' % ((y.contiguous() - ref).abs().max().item() if ref is not None else 0.0))
