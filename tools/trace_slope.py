#!/usr/bin/env python3
"""Marginal cost of one Newton trip inside the STAGED trace kernel (k_trace: rays from / to HBM, no
LDS tiles, no barriers, 256-thread workgroups), to set beside the fused kernel's (tools/kbench.py
--trips) and the isolated loop's (tools/newton_bench.hip):  python tools/trace_slope.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import load_state, make_lens
    from sdirt_amd import _lib
    from sdirt_amd.basics import stream_ptr
    import bench
    dev = torch.device("cuda:0")
    st = load_state("rf50mm")
    lens = make_lens("rf50mm", "cuda:0", st)
    pts = bench.volume_points(1).to(dev)
    S = 1024
    ray0 = lens.sample_from_points(lens._points_to_object(pts), spp=S)
    K = len(lens.surfaces)
    h, hl, sp = _lib.lib(), lens.dev_lens(0.589), stream_ptr(dev)
    work = ray0.clone()
    n_waves = ray0.numel / 64
    res = {}
    for t in (2, 4, 8):
        trips = (C.c_int32 * K)(*[t if s.kind != 0 else 0 for s in lens.surfaces])
        ts = []
        for rep in range(6):
            work.soa.copy_(ray0.soa)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(h.sdirt_trace(hl, 0, K, 0, trips, 0, work.c_rays(), work.numel, None, sp))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[t] = float(np.median(ts[1:]))
        print(f"k_trace, {ray0.numel / 1e6:.1f} M rays, {t} trips on each of 11 curved surfaces: {res[t]:.3f} ms")
    clock = 2.38e9
    for a, b in ((2, 4), (4, 8)):
        cyc = (res[b] - res[a]) * 1e-3 * clock / ((b - a) * 11 * n_waves / 1024)
        print(f"  marginal cost per trip, {a}->{b}: {cyc:.0f} cycles per SIMD (at {clock / 1e9:.2f} GHz)")
    base = res[2] - 2 * (res[4] - res[2]) / 2
    print(f"  everything but the loop trips: {base:.3f} ms = {base * 1e-3 * clock / (n_waves / 1024):.0f} cycles per ray-wave per SIMD")


if __name__ == "__main__":
    main()
