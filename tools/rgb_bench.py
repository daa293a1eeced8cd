#!/usr/bin/env python3
"""psf_rgb: three wavelengths in ONE launch (sdirt_psf_rgb_centered) against three psf_diff calls."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
lens = bench.build_lens(dev)
pts = bench.volume_points(1)[::4].contiguous().to(dev)          # 4096 points
ks, spp = 65, 4096


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


from sdirt_amd.basics import WAVE_RGB
t_end = time.perf_counter() + 1.5                 # clocks, trip tables, allocator pools
while time.perf_counter() < t_end:
    lens.psf_rgb(pts, ks=ks, spp=spp)
    [lens.psf_diff(pts, ks=ks, wvln=w, spp=spp) for w in WAVE_RGB]
for n in (4096, 64):
    p_ = pts[:: pts.shape[0] // n].contiguous()
    for _ in range(3):
        lens.psf_rgb(p_, ks=ks, spp=spp)
        [lens.psf_diff(p_, ks=ks, wvln=w, spp=spp) for w in WAVE_RGB]
    r0 = lens.trips.relaunches
    res = []
    for rep in range(2):                          # A B A B
        res.append((timed(lambda: lens.psf_rgb(p_, ks=ks, spp=spp)),
                    timed(lambda: [lens.psf_diff(p_, ks=ks, wvln=w, spp=spp) for w in WAVE_RGB])))
    print(f"{n} points, {spp} spp, ks {ks}: psf_rgb (one launch, one readback) {res[0][0]:.3f} / {res[1][0]:.3f} ms; "
          f"three psf_diff calls {res[0][1]:.3f} / {res[1][1]:.3f} ms; re-launches {lens.trips.relaunches - r0}")
