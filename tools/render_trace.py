#!/usr/bin/env python3
"""Kernel-level breakdown of one PSFNet.render call (torch profiler)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet

dev = "cuda:0"
H, W, ks = 512, 768, 21
torch.manual_seed(0)
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
           sensor_res=(H, W), kernel_size=ks, device=dev, post_computation=False)
m.d_sensor = 62.25
img = torch.rand(1, 3, H, W, device=dev)
depth = -(500 + 4500 * torch.rand(1, 1, H, W, device=dev))
foc = torch.tensor([-1000.0], device=dev)
for _ in range(3):
    m.render(img, depth, foc)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        m.render(img, depth, foc)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
