#!/usr/bin/env python3
"""Micro-benchmark of sdirt_local_psf_render at BASELINE config 5's shape (512x768, ks 21)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd import local_psf_render_fast

dev = "cuda:0"
ks, H, W = 21, 512, 768
g = torch.Generator(device=dev).manual_seed(0)
psf = torch.rand(1, H, W, 2, ks, ks, device=dev, generator=g)
img = torch.rand(1, 3, H, W, device=dev, generator=g)
for _ in range(3):
    local_psf_render_fast(img, psf, kernel_size=ks)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); local_psf_render_fast(img, psf, kernel_size=ks); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
byt = psf.numel() * 4 + img.numel() * 4 * 3
print(f"local_psf_render_fast 512x768 ks21: median {ts[10]:.3f} ms  {byt / ts[10] / 1e6:.0f} GB/s")
