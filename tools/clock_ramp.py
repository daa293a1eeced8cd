#!/usr/bin/env python3
"""Kernel time of the first N config-2 steps of a process that finds the GPU idle (HIP events around every launch):
how long the chip takes to reach the clock it then holds.  python tools/clock_ramp.py [steps, default 150]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
lens = bench.build_lens(dev, "rf50mm", 62.25)
pts = bench.volume_points(1).to(dev)
outs = [tuple(torch.empty((pts.shape[0], 65, 65), device=dev) for _ in range(2)) for _ in range(3)]
lens.kernel_events = {}
pend = []
for i in range(n):
    pend.append(lens.psf_lr(pts, ks=65, spp=4096, dp=(0.78, 1.44, 0.3, 0.5), out=outs[i % 3], defer=True))
    if len(pend) > 2:
        pend.pop(0).wait()
for p in pend:
    p.wait()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in lens.kernel_events["psf_lr_centered"]]
print(f"{len(ms)} launches (the first two discover the Newton trip tables)")
for i in range(0, len(ms), 10):
    print(f"launch {i:4d}-{min(i + 9, len(ms) - 1):4d}: " + " ".join(f"{v:6.3f}" for v in ms[i:i + 10]))
steady = sorted(ms[len(ms) // 2:])[len(ms) // 4]
first = next((i for i, v in enumerate(ms) if i >= 2 and v <= 1.005 * steady), None)
print(f"steady {steady:.3f} ms; first launch within 0.5 % of it: #{first} "
      f"({sum(ms[:first]) if first is not None else float('nan'):.0f} ms of kernel time after the start)")
