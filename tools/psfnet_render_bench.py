#!/usr/bin/env python3
"""sdirt_psfnet_render alone at 512x768, ks 21 (694 MB of raw fp16 network outputs)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.render_psf import psfnet_render

dev = "cuda:0"; H, W, ks = 512, 768, 21
g = torch.Generator(device=dev).manual_seed(0)
raw = torch.rand(2, 1, H, W, ks, ks, device=dev, generator=g).half()
img = torch.rand(1, 3, H, W, device=dev, generator=g)
for _ in range(3):
    psfnet_render(img, raw[0], raw[1], ks)
torch.cuda.synchronize(); ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = psfnet_render(img, raw[0], raw[1], ks); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = float(np.median(ts))
print(f"psfnet_render 512x768 ks21: median {t:.3f} ms  {raw.numel() * 2 / t / 1e6:.0f} GB/s  "
      f"checksum {out[0].double().sum().item():.6f} {out[1].double().sum().item():.6f}")
