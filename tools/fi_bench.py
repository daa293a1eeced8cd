#!/usr/bin/env python3
"""Times sdirt_forward_integral alone (HIP events) on traced rays of N points of the config-2 volume: L + R in one
launch (the product), and L alone (r_grid = NULL: one float64 tile per workgroup -- at ks 65 33.8 KB, four workgroups per
CU).  Twice the L-only time is what a launch that gives L and R to two workgroups of the same point would take (each reads
the point's rays once and computes its own side's weights): the experiment VERDICT r04 item 8 asked for.
  SDIRT_AMD_LIB=build/libsdirt_dp_<tag>.so python tools/fi_bench.py [N] [S] [ks ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sdirt_amd import _lib  # noqa: E402
from sdirt_amd.basics import Ray, dptr, stream_ptr  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
KS = [int(v) for v in sys.argv[3:]] or [65, 21]
dev = torch.device("cuda", 0)
lens = bench.build_lens(dev)
h, st = _lib.lib(), stream_ptr(dev)
pts_all = bench.volume_points(1, "c2")
pts = pts_all[:: len(pts_all) // N][:N].contiguous()
po = lens._points_to_object(pts)
torch.manual_seed(0)
ray = lens.sample_from_points(po, spp=S)
lens.trace2sensor(ray)
cen = torch.empty((N, 2), device=dev)
_lib.check(h.sdirt_center_from_rays(ray.c_rays(), S, N, dptr(cen), None, st))
dp = _lib.DpParams(*bench.DP)
for ks in KS:
    L = torch.empty((N, ks, ks), device=dev)
    R = torch.empty_like(L)
    fn = lambda: _lib.check(h.sdirt_forward_integral(ray.c_rays(), S, N, float(lens.pixel_size), ks, dptr(cen),
                                                     C.byref(dp), 0, dptr(L), dptr(R), st))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    fn1 = lambda: _lib.check(h.sdirt_forward_integral(ray.c_rays(), S, N, float(lens.pixel_size), ks, dptr(cen),
                                                      C.byref(dp), 0, dptr(L), None, st))
    for _ in range(3):
        fn1()
    torch.cuda.synchronize()
    ev = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn1(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ms1 = np.array([a.elapsed_time(b) for a, b in ev])
    alg = 20 * N * S + 8 * N + 2 * N * ks * ks * 4
    print(f"   L only: {np.median(ms1) * 1e3:8.1f} us -> split L | R over two workgroups = 2 x = {2 * np.median(ms1) * 1e3:8.1f} us "
          f"({(alg + 20 * N * S) / (2 * np.median(ms1) * 1e-3) / 1e9:.0f} GB/s of its {(alg + 20 * N * S) / 1e6:.0f} MB; "
          f"L + R in one launch: {alg / (np.median(ms) * 1e-3) / 1e9:.0f} GB/s of {alg / 1e6:.0f} MB)")
    print(f"{os.environ.get('SDIRT_AMD_LIB', 'product'):40s} P={os.environ.get('SDIRT_FI_P', 'auto'):4s} N={N} S={S} ks={ks}: "
          f"{np.median(ms) * 1e3:8.1f} us (min {ms.min() * 1e3:.1f})  {N * S / np.median(ms) / 1e6:.1f} Grays/s  "
          f"sumL={float(L.double().sum()):.6e}")
