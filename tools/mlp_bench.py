#!/usr/bin/env python3
"""The fused PSF network alone (sdirt_psfnet_mlp) on a 512x768 frame, both passes: time and
a checksum of the fp16 outputs.  `--once` runs a single call (for rocprofv3 --pmc)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet

dev = "cuda:0"
H, W, ks = 512, 768, 21
torch.manual_seed(0)
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
           sensor_res=(H, W), kernel_size=ks, device=dev)
with torch.no_grad():
    m.psfnet.net[-2].bias.add_(0.02)
g = torch.Generator(device=dev).manual_seed(0)
o = torch.rand(1, H, W, 3, device=dev, generator=g) * 2 - 1
with torch.no_grad():
    out = m.psfnet.forward_fused(o, mirror=True)
    torch.cuda.synchronize()
    if "--once" not in sys.argv:
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = m.psfnet.forward_fused(o, mirror=True); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts))
        fl = 2 * (3 * 128 + 128 * 512 + 8 * 512 * 512 + 512 * ks * ks) * H * W * 2
        print(f"fused MLP, 2 x {H * W} rows: median {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s  "
              f"checksum {out[0].float().sum().item():.4f} {out[1].float().abs().sum().item():.4f}")
