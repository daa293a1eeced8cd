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
import time


def burst(label):
    lens.kernel_events = {}
    pend, host = [], []
    for i in range(n):
        t0 = time.perf_counter()
        pend.append(lens.psf_lr(pts, ks=65, spp=4096, dp=(0.78, 1.44, 0.3, 0.5), out=outs[i % 3], defer=True))
        t1 = time.perf_counter()
        if len(pend) > 2:
            pend.pop(0).wait()
        host.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
    for p in pend:
        p.wait()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in lens.kernel_events["psf_lr_centered"]]
    print(f"{label}: {len(ms)} launches")
    for i in range(0, min(len(ms), 60), 10):
        print(f"launch {i:4d}-{min(i + 9, len(ms) - 1):4d}: " + " ".join(f"{v:6.3f}" for v in ms[i:i + 10]))
    steady = sorted(ms[len(ms) // 2:])[len(ms) // 4]
    slow = [(i, round(v, 2)) for i, v in enumerate(ms) if i >= 3 and v > 1.02 * steady]
    print(f"  steady {steady:.3f} ms; launches more than 2 % above it (after the first three): {slow}")
    print("  host, iterations 8-24 (enqueue ms / wait ms): " + " ".join(f"{a:.1f}/{b:.1f}" for a, b in host[8:25]))


burst("from idle, new process (the first two launches discover the Newton trip tables)")
time.sleep(6.0)                       # let the chip fall back to its idle state
burst("after 6 s of idling, same process")
burst("at once again")
