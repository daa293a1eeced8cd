#!/usr/bin/env python3
"""psf_lr over batch sizes, sample counts and tile sizes: primary rays/s of the synchronous call
(launch + trip check + readback), to spot cliffs in the launch geometry (spp split, tile in LDS)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
lens = bench.build_lens(dev)
allp = bench.volume_points(1).to(dev)
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    lens.psf_lr(allp[::4], ks=21, spp=4096)
print(f"{'N':>6} {'spp':>6} {'ks':>4} {'ms/call':>9} {'Grays/s':>8}")
for n in (1, 8, 64, 512, 4096, 16384):
    pts = allp[:: allp.shape[0] // n][:n].contiguous()
    for spp in (256, 4096, 20000):
        for ks in (17, 21, 65, 101):
            if n * ks * ks * 8 > 2e9:
                continue
            out = tuple(torch.empty((n, ks, ks), device=dev) for _ in range(2))
            for _ in range(3):
                lens.psf_lr(pts, ks=ks, spp=spp, out=out)
            torch.cuda.synchronize()
            reps = 5 if n * spp > 1e7 else 20
            t0 = time.perf_counter()
            for _ in range(reps):
                lens.psf_lr(pts, ks=ks, spp=spp, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{n:>6} {spp:>6} {ks:>4} {dt * 1e3:>9.3f} {n * spp / dt / 1e9:>8.3f}")
