#!/usr/bin/env python3
"""The fitting loop (train_psfnet: bs 64, spp 20000), iterations per second over 1000 iterations, six fresh calls per setting
(every call makes its own side stream: HIP hands streams to hardware queues in order of creation, and a side stream that shares
its queue with the step's stream serialises the PSF batches with the step):
  priority   of the side stream the PSF batches are ray-traced on (0 = the default, -1 = high: a queue of its own class)
  one_call   the deferred PSF call as ONE library call (Lensgroup.defer_one_call) or through the general path
(A third knob was tried with this tool and dropped: the library's launches sized for 3/4 of the chip's compute units, so that a
batch's workgroups leave a slot per CU free -- no gain, profiles/r06/fit_ab.txt.)"""
import os, sys, time, tempfile, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdirt_amd.psfnet as P
from sdirt_amd.psfnet import PSFNet
from sdirt_amd.optics import Lensgroup
dev = "cuda:0"
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"), sensor_res=(512, 768), kernel_size=21, device=dev)
m.refocus(-1000 + m.d_sensor)
for prio in (0, -1):
    for one_call in (False, True):
        P._SIDE_PRIORITY = prio
        Lensgroup.defer_one_call = one_call
        res = []
        for rep in range(6):
            torch.manual_seed(0); np.random.seed(0)
            with tempfile.TemporaryDirectory() as tmp:
                kw = dict(bs=64, lr=1e-4, spp=20000, evaluate_every=10 ** 9, result_dir=tmp, figures=False)
                m.train_psfnet(iters=20, **kw)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                m.train_psfnet(iters=999, **kw)
                torch.cuda.synchronize(); res.append(time.perf_counter() - t0)
        if True:
            print(f"side-stream priority {prio:2d}, one library call {str(one_call):5}: "
                  + " ".join(f"{r:.3f}" for r in res) + f" ms per iteration (median {np.median(res):.3f})", flush=True)
P._SIDE_PRIORITY = 0
Lensgroup.defer_one_call = False
