#!/usr/bin/env python3
"""Where the ray-tracing span of the reference's timing harness (PSFNet.time_compare_psf, psfnet.py:570-586) goes:
24576 random points x 4096 spp, ks 21, L only, the PSFs copied to the host inside the span.  Wall-clock pieces."""
import os
import sys
import time

# idle OpenMP threads of torch's CPU ops must sleep, not spin: under the container's CPU quota a few hundred spinning
# worker threads exhaust the cgroup's budget and the calling thread is throttled for tens of ms (every third span, round 5)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdirt_amd.psfnet import PSFNet  # noqa: E402

import gc
gc.disable()          # a full collection over torch's heap takes 40-55 ms and lands inside every third span otherwise
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = PSFNet(os.path.join(ROOT, "sdirt_amd", "data", "rf50mm.json"), sensor_res=(512, 768), kernel_size=21, device=dev)
m.refocus(-1000 + m.d_sensor)
n, spp, ks = 24576, 4096, 21


def wall(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return f"{np.median(ts) * 1e3:7.3f} (min {np.min(ts) * 1e3:.3f})"


inp = torch.rand(n, 3)
inp[:, 2] = m.z2depth(inp[:, 2])
ind = inp.to(dev)
out = m.psf(points=ind, ks=ks, spp=spp)
print(f"psf(device points)            {wall(lambda: m.psf(points=ind, ks=ks, spp=spp))} ms")
print(f"psf(host points)              {wall(lambda: m.psf(points=inp, ks=ks, spp=spp))} ms")
print(f"to_host (page-locked, cached) {wall(lambda: m.to_host(out))} ms  ({out.numel() * 4 / 1e6:.1f} MB)")
print(f".to('cpu') (pageable)         {wall(lambda: out.to('cpu'))} ms")
print(f"psf(host points) + to_host    {wall(lambda: m.to_host(m.psf(points=inp, ks=ks, spp=spp)))} ms")
tt = [m.time_compare_psf(verbose=False)[0] for _ in range(10)]
print(f"time_compare_psf span         {np.median(tt) * 1e3:7.3f} ms -> {n * spp / np.median(tt) / 1e9:.2f} G rays/s PCIe-inclusive")
r0 = m.trips.relaunches
tt = [m.time_compare_psf(verbose=False)[0] * 1e3 for _ in range(10)]
print("time_compare_psf spans (ms):", " ".join(f"{t:.2f}" for t in tt), " host re-launches:", m.trips.relaunches - r0)
# the same batch again and again through the harness' own statements
spans = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.time()
    psfl = m.to_host(m.psf(points=inp, ks=ks, center=True, spp=spp))
    spans.append((time.time() - t0) * 1e3)
print("same statements, one fixed batch (ms):", " ".join(f"{t:.2f}" for t in spans))
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    m.psf(points=inp, ks=ks, spp=spp)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
