#!/usr/bin/env python3
"""Host- and device-side cost of the SYNCHRONOUS psf call at the PSFNet training shape
(N = 64, spp 20000, ks 21: what 1_fit_psfnet.py calls 90 000 times, psfnet.py:101-167):
ms per call on a fixed batch and on random batches (whose trip tables flip), relaunch counts,
and a cProfile of the host side.   python tools/prof_sync_call.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet

torch.manual_seed(0); np.random.seed(0)
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
           sensor_res=(512, 768), kernel_size=21, device="cuda:0")
m.refocus(-1000 + m.d_sensor)
pts = torch.rand(64, 3)
pts[:, :2] = pts[:, :2] * 2 - 1
pts[:, 2] = -200 - 19800 * pts[:, 2]
ptd = pts.cuda()
for _ in range(50):
    m.psf(ptd, ks=21, spp=20000)
if "--no-freeze" not in sys.argv:
    # a full garbage collection of torch's heap takes 40-55 ms: keep it out of the 0.3-second loops timed below
    import gc
    gc.collect()
    gc.freeze()


def timed(fn, n):
    torch.cuda.synchronize(); r0 = m.trips.relaunches; l0 = m.trips.launches; t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    return dt * 1e3, m.trips.relaunches - r0, m.trips.launches - l0


n = 1000
ms, rl, la = timed(lambda: m.psf(ptd, ks=21, spp=20000), n)
print(f"fixed batch, device points : {ms:.3f} ms/call, host relaunches {rl} in {la} rounds")
ms, rl, la = timed(lambda: m.psf_lr(ptd, ks=21, spp=20000), n)
print(f"fixed batch, L and R       : {ms:.3f} ms/call, host relaunches {rl} in {la} rounds")
ms, rl, la = timed(lambda: m.get_training_data(bs=64, spp=20000), n)
print(f"random batches (get_training_data): {ms:.3f} ms/call, host relaunches {rl} in {la} rounds")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500):
        m.psf(ptd, ks=21, spp=20000)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
