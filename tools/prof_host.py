#!/usr/bin/env python3
"""Host-side cost of one deferred psf_lr call at the PSFNet training shape (cProfile)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd import Lensgroup

lens = Lensgroup(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
                 sensor_res=(512, 768), device="cuda:0")
pts = torch.rand(64, 3)
pts[:, :2] = pts[:, :2] * 2 - 1
pts[:, 2] = -200 - 19800 * pts[:, 2]
for _ in range(20):
    lens.psf_lr(pts, ks=21, spp=20000)
q = []
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(300):
    q.append(lens.psf_lr(pts, ks=21, spp=20000, defer=True))
    if len(q) > 2:
        q.pop(0).wait()
while q:
    q.pop(0).wait()
torch.cuda.synchronize(); print("deferred psf_lr 64 x 20000:", (time.perf_counter() - t) / 300 * 1e3, "ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    q.append(lens.psf_lr(pts, ks=21, spp=20000, defer=True))
    if len(q) > 2:
        q.pop(0).wait()
while q:
    q.pop(0).wait()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
