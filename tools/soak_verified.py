#!/usr/bin/env python3
"""Soak of the device-verified PSF call (sdirt_psf_call / sdirt_psf_lr_verified) against the host-verified route:
N random PSFNet training batches (64 points, 20000 spp, ks 21; the point distribution of PSFNet.get_training_data),
each rendered twice from the same seed -- once as Lensgroup.psf_lr takes it by default (trip tables verified and, if
wrong, corrected on the device), once with the host-driven speculate / verify / re-launch loop (a pass-through
`mask_reduce` hook switches the device route off) -- and compared: the verified trip tables must be equal, the PSFs
equal up to the summation order of their atomics.

  python tools/soak_verified.py [batches, default 2000]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
           sensor_res=(512, 768), kernel_size=21, device="cuda:0")
m.refocus(-1000 + m.d_sensor)
DP = (0.78, 1.44, 0.3, 0.5)
from sdirt_amd.newton import TripPlanner
planners = (m.trips, TripPlanner())          # each route keeps its own history of verified tables
worst, tables_differ, distinct = 0.0, 0, set()
for i in range(n_batches):
    torch.manual_seed(10_000 + i); np.random.seed(10_000 + i)
    # the reference's point distribution (psfnet.py:181-196), without rendering anything
    foc_z = np.random.choice(m.foc_z_arr)
    x, y = (torch.rand(64) - 0.5) * 2, (torch.rand(64) - 0.5) * 2
    z = m._warp_z(torch.clamp(torch.randn(64), min=-3, max=3), foc_z)
    pts = torch.stack((x, y, m.z2depth(z)), dim=-1)
    res = []
    for hook, planner in zip((None, lambda mask: mask), planners):
        m.mask_reduce, m.trips = hook, planner
        torch.manual_seed(i)
        L, R = m.psf_lr(pts, ks=21, spp=20000, dp=DP)
        tab = tuple(tuple(int(v) for v in m.trips.cache[k]) for k in (("psf", 0.589, "lean"), ("center", "lean")))
        res.append((L, R, tab, torch.rand(1).item()))
    m.mask_reduce, m.trips = None, planners[0]
    (La, Ra, ta, ea), (Lb, Rb, tb, eb) = res
    assert ea == eb, "the two routes drew different numbers of uniforms"
    d = max(float((La - Lb).abs().max()), float((Ra - Rb).abs().max()))
    worst = max(worst, d)
    tables_differ += ta != tb
    distinct.add(ta)
    assert d <= 5e-6, (i, d)
print(f"{n_batches} random batches: device-verified vs host-verified route: max |dPSF| {worst:.2e} of peak, "
      f"trip tables differing {tables_differ}, distinct verified table pairs {len(distinct)}; "
      f"device route: {planners[0].device_relaunches} corrections on the device, {planners[0].relaunches} host re-launches; "
      f"host route: {planners[1].relaunches} host re-launches")
assert tables_differ == 0
