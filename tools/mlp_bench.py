#!/usr/bin/env python3
"""sdirt_psfnet_mlp alone at the config-5 shape: 2 x 512 x 768 rows through 3 -> 128 -> 512 x 9 -> 441."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet_arch import MLP, initialize_weights

dev = "cuda:0"
torch.manual_seed(0)
net = MLP(3, 441, hidden_features=512, hidden_layers=8)
net.apply(initialize_weights)
net = net.to(dev)
n = 512 * 768
x = torch.rand(n, 3, device=dev) * 2 - 1
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    net.forward_fused(x, mirror=True)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net.forward_fused(x, mirror=True); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
macs = 3 * 128 + 128 * 512 + 8 * 512 * 512 + 512 * 441
padded = 16 * 128 + 128 * 512 + 8 * 512 * 512 + 512 * 512
t = float(np.median(ts))
print(f"fused MLP, {2 * n} rows: median {t:.3f} ms  min {min(ts):.3f}  "
      f"{2 * 2 * n * macs / t / 1e9:.0f} TFLOP/s useful, {2 * 2 * n * padded / t / 1e9:.0f} TFLOP/s issued")
