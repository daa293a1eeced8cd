#!/usr/bin/env python3
"""Times sdirt_forward_integral alone (HIP events) on traced rays of N points of the config-2 volume.
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
    print(f"{os.environ.get('SDIRT_AMD_LIB', 'product'):40s} P={os.environ.get('SDIRT_FI_P', 'auto'):4s} N={N} S={S} ks={ks}: "
          f"{np.median(ms) * 1e3:8.1f} us (min {ms.min() * 1e3:.1f})  {N * S / np.median(ms) / 1e6:.1f} Grays/s  "
          f"sumL={float(L.double().sum()):.6e}")
