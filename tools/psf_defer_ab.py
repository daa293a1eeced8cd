#!/usr/bin/env python3
"""psf_lr(defer=True) of the fitting shape (64 points x 20000 spp, ks 21): the general path against ONE library call
(Lensgroup.defer_one_call) -- GPU span of a call, wall time per call with two in flight, host time of the enqueue."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from sdirt_amd.psfnet import PSFNet
from sdirt_amd.optics import Lensgroup
dev = "cuda:0"
torch.manual_seed(0); np.random.seed(0)
m = PSFNet("sdirt_amd/data/rf50mm.json", sensor_res=(512, 768), kernel_size=21, device=dev)
m.refocus(-1000 + m.d_sensor)
for flag in (False, True, False, True):
    Lensgroup.defer_one_call = flag
    for _ in range(20):
        m.get_training_data(bs=64, spp=20000, _defer=True)[1].wait()
    torch.cuda.synchronize()
    # (1) GPU time of one deferred call alone (events around it)
    ts = []
    for _ in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); inp, pend = m.get_training_data(bs=64, spp=20000, _defer=True); e1.record()
        pend.wait(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    # (2) back-to-back, two in flight: wall per call, host time of the enqueue
    q = []; t_enq = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(500):
        a = time.perf_counter(); q.append(m.get_training_data(bs=64, spp=20000, _defer=True)); t_enq += time.perf_counter() - a
        if len(q) > 2:
            q.pop(0)[1].wait()
    for x in q: x[1].wait()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"defer_one_call={flag}: GPU span of one call {np.median(ts):.3f} ms; pipelined {dt / 500 * 1e3:.3f} ms per call, enqueue host {t_enq / 500 * 1e3:.3f} ms; "
          f"relaunches host {m.trips.relaunches} device {m.trips.device_relaunches}")
