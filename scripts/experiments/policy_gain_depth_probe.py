"""mean / max leaf depth of one LunarLander-shape search (4096 trees, 30 simulations, random-init weights) against the policy gain
(bench.sharpened: policy-head output layer x gain): which gains stand for the tree depths of a trained policy (bench.py depth_sensitivity)"""
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from model_based_rl_amd.engine import Engine
from model_based_rl_amd.networks import FCNetwork
torch.manual_seed(0)
net = FCNetwork(8, 4, torch.device('cpu'), types.SimpleNamespace()).eval()
w = net.get_weights()
for g in (1, 8, 16, 24, 32, 48, 64, 128, 256):
  eng = Engine(4096, 8, 4, 30, seed=1, device='cuda:0')
  eng.set_weights(bench.sharpened(w, g))
  print(g, bench.leaf_depths(eng, 8, 4096))
  eng.close()
