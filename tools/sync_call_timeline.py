#!/usr/bin/env python3
"""Where the time of ONE synchronous psf call at the PSFNet fitting shape (64 points x 20000 spp, ks 21) goes:
host before the random draw | the draw | host up to the library call | the library call | wait for the GPU |
checking what the device did.  Medians over 500 calls."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet
from sdirt_amd import _lib
torch.manual_seed(0); np.random.seed(0)
m = PSFNet(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdirt_amd", "data", "rf50mm.json"), sensor_res=(512, 768), kernel_size=21, device="cuda:0")
m.refocus(-1000 + m.d_sensor)
pts = torch.rand(64, 3); pts[:, :2] = pts[:, :2] * 2 - 1; pts[:, 2] = -200 - 19800 * pts[:, 2]
ptd = pts.cuda()
for _ in range(50): m.psf(ptd, ks=21, spp=20000)
T = {}
def stamp(k): T.setdefault(k, []).append(time.perf_counter())
h = _lib.lib()
from sdirt_amd import _hostrng
real_v, real_rand = h.sdirt_psf_call, _hostrng.rand_into
def v(*a):
    stamp("launch0"); r = real_v(*a); stamp("launch1"); return r
def rnd(*a, **k):
    stamp("rand0"); r = real_rand(*a, **k); stamp("rand1"); return r
h.sdirt_psf_call = v; _hostrng.rand_into = rnd
real_sync = torch.cuda.Stream.synchronize
def sync(self):
    stamp("sync0"); real_sync(self); stamp("sync1")
torch.cuda.Stream.synchronize = sync
n = 500
for _ in range(n):
    stamp("t0"); m.psf(ptd, ks=21, spp=20000); stamp("t1")
A = {k: np.array(v) for k, v in T.items()}
us = lambda a, b: float(np.median(A[b] - A[a]) * 1e6)
print(f"call {us('t0','t1'):.0f} us = pre-rand {us('t0','rand0'):.0f} + rand {us('rand0','rand1'):.0f} + to-launch {us('rand1','launch0'):.0f} "
      f"+ C call {us('launch0','launch1'):.0f} + to-sync {us('launch1','sync0'):.0f} + wait {us('sync0','sync1'):.0f} + settle {us('sync1','t1'):.0f}")
