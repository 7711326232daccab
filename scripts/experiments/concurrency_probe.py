"""Do small kernels on one stream run beside the whole-moves launch on another?  Times a chain of tiny torch kernels on the
default stream while the device loop runs on its own stream, for several pool sizes."""
import os, sys, time, threading, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
torch.manual_seed(0)
w = flatten_weights(FCNetwork(8, 4, torch.device('cpu'), types.SimpleNamespace()).state_dict())
x = torch.randn(256, 512, device='cuda')
def chain(n=200):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  y = x
  for _ in range(n):
    y = y * 1.0001 + 0.5
  torch.cuda.current_stream().synchronize()
  return (time.perf_counter() - t0) / n * 1e6
print('tiny-kernel chain alone: %.1f us per kernel' % chain())
for B in (16, 1024, 3584, 4096):
  eng = Engine(B, 8, 4, 30, seed=1)
  eng.set_weights(w); eng.selfplay_reset(256, 1.0, stagger=True)
  stop = False
  def actor():
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
      while not stop:
        eng.selfplay_steps(8); eng.selfplay_drain(); st.synchronize()
  th = threading.Thread(target=actor); th.start(); time.sleep(0.3)
  print('beside the device loop with %4d envs (%3d workgroups): %.1f us per kernel' % (B, (B + 15) // 16, chain()))
  stop = True; th.join(); eng.close()
