#!/usr/bin/env python3
"""1_fit_psfnet.py's loop at its own settings (bs 64, spp 20000, ks 21, full MLP): iterations/s
and the split between PSF generation (HIP ray tracer) and the network step."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet

dev = "cuda:0"
torch.manual_seed(0); np.random.seed(0)
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
           sensor_res=(512, 768), kernel_size=21, device=dev)
m.refocus(-1000 + m.d_sensor)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kw = {}
if len(sys.argv) > 2:
    kw["pipelined"] = sys.argv[2] == "pipelined"
with tempfile.TemporaryDirectory() as tmp:
    m.train_psfnet(iters=20, bs=64, lr=1e-4, spp=20000, evaluate_every=10 ** 6, result_dir=tmp, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    losses = m.train_psfnet(iters=iters, bs=64, lr=1e-4, spp=20000, evaluate_every=10 ** 6,
                            result_dir=tmp, **kw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"train_psfnet: {(iters + 1) / (t1 - t0):.1f} it/s ({(t1 - t0) / (iters + 1) * 1e3:.2f} ms/it), "
      f"loss {losses[0]:.4f} -> {losses[-1]:.4f}")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    inp, psf = m.get_training_data(bs=64, spp=20000)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"get_training_data alone: {(t1 - t0) * 10:.2f} ms/batch; trip relaunches "
      f"{m.trips.relaunches}")
